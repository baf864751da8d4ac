"""ctypes access to the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from decentralized_ekf_mhe_amd.params import DekfParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    srcs.append(os.path.join(ROOT, "include", "dekf.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    if a.dtype == np.float64:
        return a.ctypes.data_as(_dp)
    if a.dtype == np.int32:
        return a.ctypes.data_as(_ip)
    raise TypeError(a.dtype)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        pp = C.POINTER(DekfParams)
        for name in ("orc_ekf_create", "orc_est_create", "orc_pipe_create"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [pp]
        for name in ("orc_pipe_est", "orc_pipe_ekf"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        for name in ("orc_ekf_destroy", "orc_est_destroy", "orc_pipe_destroy", "orc_ekf_step",
                     "orc_est_initialize"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp]
        L.orc_bezier.restype = C.c_int
        L.orc_bezier.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_double, _dp, _dp]
        L.orc_ekf_set_imu.argtypes = [vp, C.c_double, _dp, _dp]
        L.orc_ekf_set_vo.argtypes = [vp, C.c_double, _dp]
        L.orc_ekf_last_replay.argtypes = [vp]
        L.orc_ekf_last_replay.restype = C.c_int
        L.orc_ekf_get.argtypes = [vp, _dp, _dp]
        for name in ("orc_ekf_predict", "orc_ekf_correct", "orc_ekf_vo_correct"):
            getattr(L, name).argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
            getattr(L, name).restype = None
        L.orc_est_set_imu.argtypes = [vp, C.c_double, _dp, _dp]
        L.orc_est_set_quat.argtypes = [vp, _dp]
        L.orc_est_set_leg.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.orc_est_set_vo.argtypes = [vp, C.c_double, C.c_double, _dp]
        L.orc_est_update.argtypes = [vp, C.c_int]
        L.orc_est_get.argtypes = [vp, _dp, _dp, _dp]
        L.orc_est_qp_dims.argtypes = [vp, _ip, _ip]
        L.orc_est_qp_copy.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
        L.orc_est_solution.argtypes = [vp, _dp]
        L.orc_est_solver_info.argtypes = [vp, _ip, _ip, _ip, _ip, _dp, _dp, _ip, _dp]
        L.orc_est_polish_info.argtypes = [vp, _ip, _dp, _dp]
        L.orc_est_scaling.argtypes = [vp, _dp, _dp, _dp]
        L.orc_est_arrival.argtypes = [vp, _dp, _dp]
        L.orc_est_kf_cov.argtypes = [vp, _dp]
        L.orc_est_rotation.argtypes = [vp, _dp]
        L.orc_pipe_set_imu.argtypes = [vp, C.c_double, _dp, _dp]
        L.orc_pipe_set_leg.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.orc_pipe_set_vo.argtypes = [vp, C.c_double, C.c_double, _dp, C.c_double, _dp]
        L.orc_pipe_step.argtypes = [vp, C.c_int]
        L.orc_pipe_run.restype = C.c_double
        L.orc_pipe_run.argtypes = [pp, C.c_int, C.c_int, C.c_int] + [_dp] * 7 + [_ip] + [_dp] * 5 + [_dp, _dp, _dp, _ip]
        _lib = L
    return _lib


def bezier(points, times, t_start, num, dt):
    """BezierOracle: add the way points in order, interpolate `num` nodes from t_start on; (nodes, distances)"""
    P, t = np.ascontiguousarray(points, float), np.ascontiguousarray(times, float)
    nodes, dist = np.zeros((num, 3)), np.zeros((num, 3))
    n = lib().orc_bezier(len(t), _ptr(P), _ptr(t), float(t_start), int(num), float(dt), _ptr(nodes), _ptr(dist))
    return nodes[:n], dist[:n]


class Ekf:
    def __init__(self, params):
        self.p = params
        self.h = lib().orc_ekf_create(C.byref(params))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ekf_destroy(self.h)
            self.h = None

    def set_imu(self, t, accel, gyro):
        a, g = np.ascontiguousarray(accel, float), np.ascontiguousarray(gyro, float)
        lib().orc_ekf_set_imu(self.h, float(t), _ptr(a), _ptr(g))

    def set_vo(self, t, q):
        q = np.ascontiguousarray(q, float)
        lib().orc_ekf_set_vo(self.h, float(t), _ptr(q))

    def step(self):
        lib().orc_ekf_step(self.h)

    def get(self):
        q, P = np.zeros(4), np.zeros((4, 4))
        lib().orc_ekf_get(self.h, _ptr(q), _ptr(P))
        return q, P

    def last_replay(self):
        return lib().orc_ekf_last_replay(self.h)

    def _fn(self, name, q, v, cov):
        q, v, cov = (np.ascontiguousarray(x, float) for x in (q, v, cov))
        qo, co = np.zeros(4), np.zeros((4, 4))
        getattr(lib(), name)(self.h, _ptr(q), _ptr(v), _ptr(cov), _ptr(qo), _ptr(co))
        return qo, co

    def predict(self, q, gyro, cov):
        return self._fn("orc_ekf_predict", q, gyro, cov)

    def correct(self, q, accel, cov):
        return self._fn("orc_ekf_correct", q, accel, cov)

    def vo_correct(self, q, q_vo, cov):
        return self._fn("orc_ekf_vo_correct", q, q_vo, cov)


class Est:
    """Single-instance estimator oracle (DecentralizedEstimation restatement)."""

    def __init__(self, params, handle=None, owner=None):
        self.p = params
        self.ns = params.dim_state
        self._owner = owner
        self.h = handle if handle is not None else lib().orc_est_create(C.byref(params))
        self._own = handle is None

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().orc_est_destroy(self.h)
            self.h = None

    def set_imu(self, t, accel, gyro):
        a, g = np.ascontiguousarray(accel, float), np.ascontiguousarray(gyro, float)
        lib().orc_est_set_imu(self.h, float(t), _ptr(a), _ptr(g))

    def set_quat(self, q):
        q = np.ascontiguousarray(q, float)
        lib().orc_est_set_quat(self.h, _ptr(q))

    def set_leg(self, p_foot, J, qdot, contact):
        a = [np.ascontiguousarray(x, float) for x in (p_foot, J, qdot, contact)]
        lib().orc_est_set_leg(self.h, *[_ptr(x) for x in a])

    def set_vo(self, t_pre, t_now, dp):
        dp = np.ascontiguousarray(dp, float)
        lib().orc_est_set_vo(self.h, float(t_pre), float(t_now), _ptr(dp))

    def initialize(self):
        lib().orc_est_initialize(self.h)

    def update(self, T):
        lib().orc_est_update(self.h, int(T))

    def get(self):
        x, v, pv = np.zeros(self.ns), np.zeros(3), np.zeros(3)
        lib().orc_est_get(self.h, _ptr(x), _ptr(v), _ptr(pv))
        return x, v, pv

    def qp(self):
        n, m = C.c_int(), C.c_int()
        lib().orc_est_qp_dims(self.h, C.byref(n), C.byref(m))
        n, m = n.value, m.value
        H, g, A, l, u = np.zeros((n, n)), np.zeros(n), np.zeros((m, n)), np.zeros(m), np.zeros(m)
        lib().orc_est_qp_copy(self.h, _ptr(H), _ptr(g), _ptr(A), _ptr(l), _ptr(u))
        return H, g, A, l, u

    def solution(self):
        n, m = C.c_int(), C.c_int()
        lib().orc_est_qp_dims(self.h, C.byref(n), C.byref(m))
        x = np.zeros(n.value)
        lib().orc_est_solution(self.h, _ptr(x))
        return x

    def solver_info(self):
        it, st, ru, fa, nz = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        pr, du, rho = C.c_double(), C.c_double(), C.c_double()
        lib().orc_est_solver_info(self.h, C.byref(it), C.byref(st), C.byref(ru), C.byref(fa), C.byref(pr),
                                  C.byref(du), C.byref(nz), C.byref(rho))
        return dict(iters=it.value, status=st.value, rho_updates=ru.value, factorizations=fa.value,
                    pri_res=pr.value, dua_res=du.value, nnzL=nz.value, rho=rho.value)

    def polish_info(self):
        st, pr, du = C.c_int(), C.c_double(), C.c_double()
        lib().orc_est_polish_info(self.h, C.byref(st), C.byref(pr), C.byref(du))
        return dict(status=st.value, pri_res=pr.value, dua_res=du.value)

    def scaling(self):
        n, m = C.c_int(), C.c_int()
        lib().orc_est_qp_dims(self.h, C.byref(n), C.byref(m))
        D, E, c = np.zeros(n.value), np.zeros(m.value), C.c_double()
        lib().orc_est_scaling(self.h, _ptr(D), _ptr(E), C.byref(c))
        return D, E, c.value

    def arrival(self):
        M, n = np.zeros((self.ns, self.ns)), np.zeros(self.ns)
        lib().orc_est_arrival(self.h, _ptr(M), _ptr(n))
        return M, n

    def kf_cov(self):
        Cm = np.zeros((self.ns, self.ns))
        lib().orc_est_kf_cov(self.h, _ptr(Cm))
        return Cm

    def rotation(self):
        R = np.zeros((3, 3))
        lib().orc_est_rotation(self.h, _ptr(R))
        return R


class Pipe:
    """EKF -> estimator chain for one instance (what dekf_step does for a batch)."""

    def __init__(self, params):
        self.p = params
        self.h = lib().orc_pipe_create(C.byref(params))
        self.est = Est(params, handle=lib().orc_pipe_est(self.h), owner=self)
        self.ekf_h = lib().orc_pipe_ekf(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pipe_destroy(self.h)
            self.h = None

    def feed(self, s, k, b):
        """latch step k of instance b of a streams dict"""
        lib().orc_pipe_set_imu(self.h, float(s["imu_t"][k, b]), _ptr(s["accel"][k, b]), _ptr(s["gyro"][k, b]))
        lib().orc_pipe_set_leg(self.h, _ptr(s["p_foot"][k, b]), _ptr(s["J"][k, b]), _ptr(s["qdot"][k, b]),
                               _ptr(s["contact"][k, b]))
        if s["vo_mask"][k, b]:
            lib().orc_pipe_set_vo(self.h, float(s["vo_t_pre"][k, b]), float(s["vo_t_now"][k, b]),
                                  _ptr(s["vo_dp"][k, b]), float(s["vo_t_pose"][k, b]), _ptr(s["vo_q"][k, b]))

    def step(self, T):
        lib().orc_pipe_step(self.h, int(T))

    def quat(self):
        q = np.zeros(4)
        lib().orc_ekf_get(self.ekf_h, _ptr(q), None)
        return q


class marg_inverse_variant:
    """TEST KNOB of the oracle (oracle/densemat.hpp: marg_inverse_variant), as a context manager: how the saddle matrix of
    marginalizeQP (MheSrb.cpp:588,640) is inverted — 0 the reference's evaluation (what every parity test compares with),
    1 long double, 2 reversed pivot order, 5 on S moved by one unit in the last place.  Process-wide: no oracle threads of
    another variant may run meanwhile."""

    def __init__(self, v):
        self.v = int(v)

    def __enter__(self):
        L = lib()
        L.orc_set_marg_inverse_variant.argtypes = [C.c_int]
        L.orc_set_marg_inverse_variant.restype = None
        L.orc_set_marg_inverse_variant(self.v)

    def __exit__(self, *a):
        lib().orc_set_marg_inverse_variant(0)


def run_streams(params, s, nthreads=1, want_iters=False):
    """Run the whole log through the oracle; returns x[K,B,ns], v_b[K,B,3], quat[K,B,4], secs."""
    K, B = s["imu_t"].shape
    ns = params.dim_state
    x = np.zeros((K, B, ns))
    vb = np.zeros((K, B, 3))
    qt = np.zeros((K, B, 4))
    it = np.zeros((K, B), dtype=np.int32)
    secs = lib().orc_pipe_run(C.byref(params), B, K, int(nthreads), _ptr(s["imu_t"]), _ptr(s["accel"]),
                              _ptr(s["gyro"]), _ptr(s["p_foot"]), _ptr(s["J"]), _ptr(s["qdot"]),
                              _ptr(s["contact"]), _ptr(s["vo_mask"]), _ptr(s["vo_t_pre"]), _ptr(s["vo_t_now"]),
                              _ptr(s["vo_dp"]), _ptr(s["vo_t_pose"]), _ptr(s["vo_q"]), _ptr(x), _ptr(vb),
                              _ptr(qt), _ptr(it))
    if want_iters:
        return x, vb, qt, secs, it
    return x, vb, qt, secs
