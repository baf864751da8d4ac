"""ctypes access to tests/hostsim/libhostsim.so — the lane-sequential CPU build of the device
cores.  TEST INFRASTRUCTURE ONLY (see tests/hostsim/hostsim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from decentralized_ekf_mhe_amd.params import DekfParams

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
LIB = os.path.join(HERE, "libhostsim.so")
CSRC = os.path.join(os.path.dirname(HERE), "..", "decentralized_ekf_mhe_amd", "csrc")
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def _cpu_tag():
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        model = [ln for ln in txt.splitlines() if ln.startswith("model name")][:1]
        flags = [ln for ln in txt.splitlines() if ln.startswith("flags")][:1]
        return hashlib.md5("".join(model + flags).encode()).hexdigest()[:8]
    except Exception:  # noqa: BLE001
        return "unknown"


def build(force=False, sanitize=False, fast=False):
    """fast: -O3 -march=native under another name (bench.py's cpu_baseline.structured leg times THAT build on the GPU box's host)"""
    srcs = [os.path.join(HERE, "hostsim.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    out = LIB if not sanitize else LIB.replace(".so", "_asan.so")
    if fast:
        out = LIB.replace(".so", f"_o3_{_cpu_tag()}.so")   # -march=native: a build from ANOTHER CPU (the container's, shipped with the snapshot) must never be loaded
    stale = (not os.path.exists(out)) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)
    if force or stale:
        flags = ["-O2"] if not sanitize else ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
        if fast:
            flags = ["-O3", "-march=native"]
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-DDEKF_HOSTSIM", "-w"] + flags +
                              ["-o", out, os.path.join(HERE, "hostsim.cpp")])
    return out


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp if a.dtype == np.float64 else _ip)


_lib = None
_libs = {}


def lib(fast=False):
    global _lib
    if fast:
        if "fast" not in _libs:
            _libs["fast"] = _bind(C.CDLL(build(fast=True)))
        return _libs["fast"]
    if _lib is None:
        _lib = _bind(C.CDLL(build()))
    return _lib


def _bind(L):
    L.hs_create.restype = C.c_void_p
    L.hs_create.argtypes = [C.POINTER(DekfParams), C.c_int]
    vp = C.c_void_p
    L.hs_destroy.argtypes = [vp]
    L.hs_push_imu.argtypes = [vp, _dp, _dp, _dp]
    L.hs_push_leg.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.hs_push_vo.argtypes = [vp, _ip, _dp, _dp, _dp, _dp, _dp]
    L.hs_push_quat.argtypes = [vp, _dp]
    L.hs_ekf_step.argtypes = [vp]
    L.hs_initialize.argtypes = [vp]
    L.hs_update.argtypes = [vp, C.c_int]
    L.hs_get.argtypes = [vp, _dp, _dp, _dp, _dp, _ip, _ip, _ip]
    L.hs_get_ekf_cov.argtypes = [vp, _dp]
    L.hs_get_polish_status.argtypes = [vp, _ip]
    L.hs_get_residuals.argtypes = [vp, _dp, _dp]
    L.hs_get_arrival.argtypes = [vp, _dp, _dp]
    L.hs_get_scaling.argtypes = [vp, C.c_int, C.c_int, _dp, _dp]
    return L


class HostSim:
    def __init__(self, params, batch, fast=False):
        self.p, self.B, self.L = params, batch, lib(fast)
        self.h = self.L.hs_create(C.byref(params), batch)
        assert self.h, "hs_create rejected the parameters"

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hs_destroy(self.h)
            self.h = None

    def feed(self, s, k):
        L = self.L
        L.hs_push_imu(self.h, _p(s["imu_t"][k]), _p(s["accel"][k]), _p(s["gyro"][k]))
        L.hs_push_leg(self.h, _p(s["p_foot"][k]), _p(s["J"][k]), _p(s["qdot"][k]), _p(s["contact"][k]))
        if s["vo_mask"][k].any():
            L.hs_push_vo(self.h, _p(s["vo_mask"][k]), _p(s["vo_t_pre"][k]), _p(s["vo_t_now"][k]), _p(s["vo_dp"][k]),
                         _p(s["vo_t_pose"][k]), _p(s["vo_q"][k]))

    def step(self, T):
        L = self.L
        L.hs_ekf_step(self.h)
        if T == 0:
            L.hs_initialize(self.h)
        else:
            L.hs_update(self.h, T)

    def get(self):
        B = self.B
        x, vb, q, pv = np.zeros((B, self.p.dim_state)), np.zeros((B, 3)), np.zeros((B, 4)), np.zeros((B, 3))
        st, it, ru = (np.zeros(B, np.int32) for _ in range(3))
        self.L.hs_get(self.h, _p(x), _p(vb), _p(q), _p(pv), _p(st), _p(it), _p(ru))
        ps = np.zeros(B, np.int32)
        self.L.hs_get_polish_status(self.h, _p(ps))
        pr, du = np.zeros(B), np.zeros(B)
        self.L.hs_get_residuals(self.h, _p(pr), _p(du))
        return dict(x=x, v_b=vb, quat=q, p_vo=pv, status=st, iters=it, rho_updates=ru, polish_status=ps, pri_res=pr, dua_res=du)

    def ekf_cov(self):
        P = np.zeros((self.B, 4, 4))
        self.L.hs_get_ekf_cov(self.h, _p(P))
        return P

    def arrival(self):
        ns = self.p.dim_state
        M, n = np.zeros((self.B, ns, ns)), np.zeros((self.B, ns))
        self.L.hs_get_arrival(self.h, _p(M), _p(n))
        return M, n

    def scaling(self, n, m):
        D, E = np.zeros(n), np.zeros(m)
        self.L.hs_get_scaling(self.h, n, m, _p(D), _p(E))
        return D, E


def run_streams_timed(params, s, nthreads=1, fast=True):
    """Every instance of the streams dict through the lane-sequential build, `nthreads` Python threads each driving its own
    simulator over a contiguous share of the instances (ctypes releases the GIL inside a call).  Returns (x_final [B][ns], seconds)."""
    import threading
    import time
    B, K = s["imu_t"].shape[1], s["imu_t"].shape[0]
    nthreads = max(1, min(nthreads, B))
    cuts = [B * i // nthreads for i in range(nthreads + 1)]
    shares = []
    for i in range(nthreads):
        a, b = cuts[i], cuts[i + 1]
        shares.append({k: (np.ascontiguousarray(v[:, a:b]) if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[1] == B else v) for k, v in s.items()})
    sims = [HostSim(params, cuts[i + 1] - cuts[i], fast=fast) for i in range(nthreads)]

    def work(sim, sh):
        for k in range(K):
            sim.feed(sh, k)
            sim.step(k)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(sims[i], shares[i])) for i in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    secs = time.perf_counter() - t0
    x = np.concatenate([sim.get()["x"] for sim in sims], axis=0)
    return x, secs
