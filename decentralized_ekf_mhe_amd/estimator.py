"""Host-side mirror of the reference's estimator interface for a batch of robots.

``BatchedEstimator`` keeps the method names of ``DecentralizedEstimation``
(initialize / update / reset, DecentralEst.hpp:96-103) and exposes what the reference keeps
in public members (x_MHE_, v_MHE_b_, p_vo_accmulate_, ...) as ``get()``.  All arithmetic
happens in csrc/libdekf.so on the GPU; arguments are numpy arrays (host) or torch CUDA
tensors (HBM, zero-copy)."""
import ctypes as C
import os
import sys

import numpy as np

from . import capi
from .params import DekfParams


def _ptr_where(a, dtype):
    """(void*, DEKF_HOST|DEKF_DEVICE) of a contiguous numpy array or torch tensor"""
    if a is None:
        return None, None
    if isinstance(a, np.ndarray):
        assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
        return C.c_void_p(a.ctypes.data), capi.DEKF_HOST
    import torch
    assert isinstance(a, torch.Tensor) and a.is_contiguous()
    want = torch.float64 if dtype == np.float64 else torch.int32
    assert a.dtype == want, (a.dtype, want)
    return C.c_void_p(a.data_ptr()), (capi.DEKF_DEVICE if a.is_cuda else capi.DEKF_HOST)


def _torch_runtime_first():
    """PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so).  If libdekf.so has already brought up the
    system's copy when torch initialises its own, torch finds no device ("No HIP GPUs are available": seen on the GPU box when a tool
    created an estimator first and uploaded tensors afterwards).  Loaded the other way round, libdekf.so binds to the runtime that is
    already in the process.  So a process that HAS imported torch — i.e. one that may hand tensors to this class — gets torch's
    runtime initialised before libdekf.so touches a device.  A process that has not imported torch (numpy arguments, the C / C++
    callers) is left alone: no torch import, no second HIP runtime in the address space, no seconds of start-up.  A caller that
    imports torch only AFTER creating an estimator sets DEKF_TORCH_RUNTIME_FIRST=1 (or imports torch first)."""
    if "torch" not in sys.modules and os.environ.get("DEKF_TORCH_RUNTIME_FIRST", "0") != "1":
        return
    try:
        import torch
        torch.cuda.is_available()
    except Exception as e:  # noqa: BLE001
        print(f"decentralized_ekf_mhe_amd: torch is imported but its HIP runtime did not come up ({e}); continuing with the system runtime",
              file=sys.stderr)


class BatchedEstimator:
    def __init__(self, params: DekfParams, batch: int, device: int = 0, stream=None):
        _torch_runtime_first()
        self.lib = capi.load()
        self.params = params.copy()
        self.batch = batch
        self.device = device
        h = C.c_void_p()
        capi.check(self.lib.dekf_create(C.byref(self.params), batch, device, C.c_void_p(stream or 0), C.byref(h)))
        self.h = h
        self.T = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.dekf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- sensor latches --------------------------------------------------------------
    def _where(self, *arrays, dtypes=None):
        ptrs, wheres = [], set()
        for i, a in enumerate(arrays):
            p, w = _ptr_where(a, (dtypes[i] if dtypes else np.float64))
            ptrs.append(p)
            if w is not None:
                wheres.add(w)
        assert len(wheres) == 1, "all arguments of one call must live on the same side"
        return ptrs, wheres.pop()

    def push_imu(self, imu_t, accel, gyro):
        p, w = self._where(imu_t, accel, gyro)
        capi.check(self.lib.dekf_push_imu(self.h, *p, w))

    def push_leg(self, p_foot, J, qdot, contact):
        p, w = self._where(p_foot, J, qdot, contact)
        capi.check(self.lib.dekf_push_leg(self.h, *p, w))

    def push_go1_joints(self, joint_position, joint_velocity, foot_force):
        p, w = self._where(joint_position, joint_velocity, foot_force)
        capi.check(self.lib.dekf_push_go1_joints(self.h, *p, w))

    def push_vo(self, mask, t_pre, t_now, dp, t_pose=None, q_vo=None):
        p, w = self._where(mask, t_pre, t_now, dp, t_pose, q_vo,
                           dtypes=[np.int32] + [np.float64] * 5)
        capi.check(self.lib.dekf_push_vo(self.h, *p, w))

    def push_quaternion(self, quat):
        p, w = self._where(quat)
        capi.check(self.lib.dekf_push_quaternion(self.h, *p, w))

    def push_stream_step(self, s, k):
        """latch step k of a streams dict (numpy or torch-on-device, [K][B][...])"""
        self.push_imu(s["imu_t"][k], s["accel"][k], s["gyro"][k])
        self.push_leg(s["p_foot"][k], s["J"][k], s["qdot"][k], s["contact"][k])
        if s["vo_any"][k]:
            self.push_vo(s["vo_mask"][k], s["vo_t_pre"][k], s["vo_t_now"][k], s["vo_dp"][k],
                         s["vo_t_pose"][k], s["vo_q"][k])

    # ---- the hot path ----------------------------------------------------------------
    def ekf_step(self):
        capi.check(self.lib.dekf_ekf_step(self.h))

    def initialize(self):
        capi.check(self.lib.dekf_initialize(self.h))
        self.T = 1

    def update(self, T):
        capi.check(self.lib.dekf_update(self.h, int(T)))
        self.T = T + 1

    def step(self, T):
        capi.check(self.lib.dekf_step(self.h, int(T)))
        self.T = T + 1

    def reset(self):
        capi.check(self.lib.dekf_reset(self.h))
        self.T = 0

    def sync(self):
        capi.check(self.lib.dekf_sync(self.h))

    # ---- results ---------------------------------------------------------------------
    def get(self):
        B = self.batch
        out = dict(x=np.zeros((B, self.params.dim_state)), v_b=np.zeros((B, 3)), quat=np.zeros((B, 4)), p_vo=np.zeros((B, 3)),
                   status=np.zeros(B, np.int32))
        capi.check(self.lib.dekf_get(self.h, *[C.c_void_p(out[k].ctypes.data) for k in ("x", "v_b", "quat", "p_vo", "status")],
                                     capi.DEKF_HOST))
        return out

    def get_into(self, x=None, v_b=None, quat=None, p_vo=None, status=None):
        """device-to-device copies into caller-owned torch CUDA tensors (asynchronous)"""
        ptrs = []
        for a, dt in ((x, np.float64), (v_b, np.float64), (quat, np.float64), (p_vo, np.float64), (status, np.int32)):
            p, w = _ptr_where(a, dt)
            assert a is None or w == capi.DEKF_DEVICE
            ptrs.append(p)
        capi.check(self.lib.dekf_get(self.h, *ptrs, capi.DEKF_DEVICE))

    def solver_info(self):
        B = self.batch
        out = dict(iters=np.zeros(B, np.int32), rho_updates=np.zeros(B, np.int32), pri_res=np.zeros(B), dua_res=np.zeros(B))
        capi.check(self.lib.dekf_get_solver_info(self.h, *[C.c_void_p(out[k].ctypes.data) for k in ("iters", "rho_updates", "pri_res", "dua_res")],
                                                 capi.DEKF_HOST))
        out["polish_status"] = np.zeros(B, np.int32)
        capi.check(self.lib.dekf_get_polish_status(self.h, C.c_void_p(out["polish_status"].ctypes.data), capi.DEKF_HOST))
        return out

    def ekf_cov(self):
        P = np.zeros((self.batch, 4, 4))
        capi.check(self.lib.dekf_get_ekf_cov(self.h, C.c_void_p(P.ctypes.data), capi.DEKF_HOST))
        return P

    def kf_cov(self):
        ns = self.params.dim_state
        Cm = np.zeros((self.batch, ns, ns))
        capi.check(self.lib.dekf_get_kf_cov(self.h, C.c_void_p(Cm.ctypes.data), capi.DEKF_HOST))
        return Cm

    # ---- measurement ------------------------------------------------------------------
    def timing_enable(self, on=True):
        """True / 1: HIP events around every kernel launch; 2: around the MHE solve launches only (an event pair costs the stream ~7 us)"""
        capi.check(self.lib.dekf_timing_enable(self.h, int(on)))

    def timing_read(self):
        """{'ekf'|'assemble'|'solve'|'allgather': (device ms summed, launches)} since the last read"""
        ms = (C.c_double * 4)()
        cnt = (C.c_int * 4)()
        capi.check(self.lib.dekf_timing_read(self.h, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(("ekf", "assemble", "solve", "allgather"))}

    def launch_info(self):
        """{'solve_workgroups', 'compute_units', 'clock_hz'} of the solve kernel's launch on this device"""
        wg, cu, hz = C.c_int(), C.c_int(), C.c_double()
        capi.check(self.lib.dekf_launch_info(self.h, C.byref(wg), C.byref(cu), C.byref(hz)))
        return dict(solve_workgroups=wg.value, compute_units=cu.value, clock_hz=hz.value)

    def solve_kernel_name(self, full_window=True):
        """symbol of the solve kernel this handle launches (full windows / window-fill ticks); None for a KF handle"""
        n = self.lib.dekf_solve_kernel_name(self.h, int(bool(full_window)))
        return n.decode() if n else None

    # ---- multi-GPU --------------------------------------------------------------------
    def comm_init(self, world, rank, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, capi.DEKF_UNIQUE_ID_BYTES)
        capi.check(self.lib.dekf_comm_init(self.h, world, rank, buf))

    def comm_info(self):
        """(world, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)"""
        w, r = C.c_int(0), C.c_int(-1)
        capi.check(self.lib.dekf_comm_info(self.h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def comm_ranks_seen(self):
        """collective: all-gathers every rank's number through the handle's communicator and stream; distinct rank numbers that arrived"""
        n = C.c_int(0)
        capi.check(self.lib.dekf_comm_ranks_seen(self.h, C.byref(n)))
        return n.value

    def allgather_vb(self, out_tensor):
        """asynchronous (own stream, overlaps the next step); complete after sync() or allgather_wait()"""
        capi.check(self.lib.dekf_allgather_vb(self.h, C.c_void_p(out_tensor.data_ptr())))

    def allgather_wait(self):
        """the estimator's stream waits for the last all-gather (no host block)"""
        capi.check(self.lib.dekf_allgather_wait(self.h))


def new_unique_id() -> bytes:
    _torch_runtime_first()
    buf = C.create_string_buffer(capi.DEKF_UNIQUE_ID_BYTES)
    capi.check(capi.load().dekf_comm_unique_id(buf))
    return buf.raw


def streams_to_device(s, device="cuda"):
    """upload a streams dict once so the timed loop only hands device pointers over"""
    import torch
    out = {}
    for k, v in s.items():
        if isinstance(v, np.ndarray) and not k.startswith("gt_"):
            out[k] = torch.from_numpy(v).to(device)
    out["vo_any"] = [bool(m.any()) for m in s["vo_mask"]]
    return out


def streams_host(s):
    out = dict(s)
    out["vo_any"] = [bool(m.any()) for m in s["vo_mask"]]
    return out
