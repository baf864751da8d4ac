"""ctypes binding of the C ABI (include/dekf.h -> csrc/libdekf.so).

The shared library is the product; this module only declares prototypes.  It raises
ImportError-like RuntimeError when the library has not been built — there is no Python or
CPU implementation to fall back to."""
import ctypes as C
import os

from .params import DekfParams

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEKF_LIB lets tools/profile_sections.py point at the diagnostic build (libdekf_prof.so)
LIB_PATH = os.environ.get("DEKF_LIB", os.path.join(_HERE, "csrc", "libdekf.so"))

DEKF_OK, DEKF_ERR_INVALID, DEKF_ERR_NO_DEVICE, DEKF_ERR_HIP, DEKF_ERR_ORDER, DEKF_ERR_COMM = range(6)
DEKF_HOST, DEKF_DEVICE = 0, 1
DEKF_SOLVE_NONE, DEKF_SOLVE_OK, DEKF_SOLVE_MAX_ITER, DEKF_SOLVE_NUMERIC = 0, 1, 2, -1
DEKF_UNIQUE_ID_BYTES = 128
DEKF_ABI_VERSION = 4  # include/dekf.h; the ctypes mirror of dekf_params (params.py) is laid out for exactly this version

_dp, _ip, _vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p

PROTOTYPES = {
    "dekf_default_params": (None, [C.POINTER(DekfParams)]),
    "dekf_abi_version": (C.c_int, []),
    "dekf_last_error": (C.c_char_p, []),
    "dekf_hip_runtime_version": (C.c_int, []),
    "dekf_create": (C.c_int, [C.POINTER(DekfParams), C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "dekf_destroy": (C.c_int, [_vp]),
    "dekf_reset": (C.c_int, [_vp]),
    "dekf_sync": (C.c_int, [_vp]),
    "dekf_batch": (C.c_int, [_vp]),
    "dekf_stream": (_vp, [_vp]),
    "dekf_push_imu": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "dekf_push_leg": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "dekf_push_go1_joints": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "dekf_push_vo": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "dekf_push_quaternion": (C.c_int, [_vp, _vp, C.c_int]),
    "dekf_ekf_step": (C.c_int, [_vp]),
    "dekf_initialize": (C.c_int, [_vp]),
    "dekf_update": (C.c_int, [_vp, C.c_int]),
    "dekf_step": (C.c_int, [_vp, C.c_int]),
    "dekf_get": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "dekf_get_ekf_cov": (C.c_int, [_vp, _vp, C.c_int]),
    "dekf_get_solver_info": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "dekf_get_polish_status": (C.c_int, [_vp, _vp, C.c_int]),
    "dekf_get_kf_cov": (C.c_int, [_vp, _vp, C.c_int]),
    "dekf_timing_enable": (C.c_int, [_vp, C.c_int]),
    "dekf_timing_read": (C.c_int, [_vp, _dp, _ip]),
    "dekf_launch_info": (C.c_int, [_vp, _ip, _ip, _dp]),
    "dekf_solve_kernel_name": (C.c_char_p, [_vp, C.c_int]),
    "dekf_comm_unique_id": (C.c_int, [_vp]),
    "dekf_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "dekf_allgather_vb": (C.c_int, [_vp, _vp]),
    "dekf_allgather_wait": (C.c_int, [_vp]),
    "dekf_comm_info": (C.c_int, [_vp, _ip, _ip]),
    "dekf_comm_ranks_seen": (C.c_int, [_vp, _ip]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with decentralized_ekf_mhe_amd/csrc/build.sh "
                               "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        lib = C.CDLL(LIB_PATH)
        # a stale prebuilt library (the .so is git-ignored) would be handed a parameter block of the wrong layout
        lib.dekf_abi_version.restype = C.c_int
        have = lib.dekf_abi_version()
        if have != DEKF_ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} implements ABI version {have}, this package binds version {DEKF_ABI_VERSION}: "
                               "rebuild it with decentralized_ekf_mhe_amd/csrc/build.sh")
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                raise RuntimeError(f"{LIB_PATH} does not export {name}: rebuild it with decentralized_ekf_mhe_amd/csrc/build.sh")
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class DekfError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"dekf status {status}: {msg}")
        self.status = status


def check(status):
    if status != DEKF_OK:
        raise DekfError(status, load().dekf_last_error().decode())
