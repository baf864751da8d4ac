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
    assert lib.dekf_abi_version() == capi.DEKF_ABI_VERSION == 4


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


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """the boundary is a C ABI: include/dekf.h must compile as C99 (no C++ in the signatures), and a program written in C must link
    against libdekf.so, read the defaults and be refused a handle on a box without a device — through the same entry points the
    reference-side binding of INTEGRATION.md uses"""
    import subprocess
    src = tmp_path / "c_client.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "dekf.h"\n'
        "int main(void) {\n"
        "    dekf_params p;\n"
        "    dekf_handle h = 0;\n"
        "    dekf_default_params(&p);\n"
        '    printf("abi %d N %d legs %d rate %d ring %d\\n", dekf_abi_version(), p.N, p.num_legs, p.rate, p.ekf_history);\n'
        "    int st = dekf_create(&p, 4, 0, 0, &h);\n"
        '    printf("create %d handle %s\\n", st, h ? "set" : "null");\n'
        "    if (st == 0) dekf_destroy(h);\n"
        "    return 0;\n}\n")
    lib_dir = os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc")
    exe = tmp_path / "c_client"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-ldekf", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "abi 4 N 20 legs 4 rate 200 ring 256" in out.stdout, out.stdout
    import torch
    if not torch.cuda.is_available():
        assert f"create {capi.DEKF_ERR_NO_DEVICE} handle null" in out.stdout, out.stdout
