"""The C-ABI library loads on a box without a GPU, exports every symbol include/dekf.h declares,
agrees with the Python parameter defaults, and refuses to run without a device (no fallback)."""
import ctypes as C
import os
import re

import pytest

from decentralized_ekf_mhe_amd import capi, go1_params
from decentralized_ekf_mhe_amd.params import DekfParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "dekf.h")).read()
    names = sorted(set(re.findall(r"\b(dekf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 24
    lib = capi.load()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(capi.PROTOTYPES), set(names) ^ set(capi.PROTOTYPES)
    assert lib.dekf_abi_version() == capi.DEKF_ABI_VERSION == 3


def test_default_params_match_python_and_yaml_values():
    lib = capi.load()
    p = DekfParams()
    lib.dekf_default_params(C.byref(p))
    assert bytes(p) == bytes(go1_params())
    assert (p.rate, p.N, p.num_legs, p.max_qp_iter) == (200, 20, 4, 4000)
    assert p.vo_p_std[0] == 0.000015 and p.foot_swing_std[0] == 1.0e7 and p.sigma == 1e-5 and p.alpha == 1.6


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = capi.load()
    h = C.c_void_p()
    st = lib.dekf_create(C.byref(go1_params()), 4, 0, None, C.byref(h))
    assert st == capi.DEKF_ERR_NO_DEVICE and not h.value
    assert b"no CPU path" in lib.dekf_last_error()


def test_product_does_not_reference_oracle():
    """nothing under the package may import / link / load oracle/ or tests/hostsim"""
    pkg = os.path.join(ROOT, "decentralized_ekf_mhe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".sh", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "libhostsim" not in txt, f
