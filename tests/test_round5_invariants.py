"""Small CPU-side guards for two round-5 decisions that a later edit could undo silently (the properties themselves are held on the
GPU: tests/test_gpu_configs.py::test_three_workgroup_kernel_* / ::test_one_legged_long_window_kernels_agree_bit_for_bit)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc")


def test_admm_core_is_compiled_with_contraction_off_and_hands_it_back():
    """One result per robot whatever the batch: the iteration phases exist in several code shapes, and only with floating-point
    contraction OFF (plus explicit fma in the same form everywhere) do the shapes produce the same bits.  The header switches
    contraction off for itself and back to `fast` for what mhe_solve_core.h compiles after the include."""
    src = open(os.path.join(CSRC, "mhe_admm_core.h")).read()
    off = [m.start() for m in re.finditer(r"^#pragma clang fp contract\(off\)", src, re.M)]
    fast = [m.start() for m in re.finditer(r"^#pragma clang fp contract\(fast\)", src, re.M)]
    assert len(off) == 1 and len(fast) == 1 and off[0] < fast[0]
    body = src[off[0]:fast[0]]
    # every device function of the header sits between the two pragmas
    assert "DEKF_FN void row_regs_iter" in body and "DEKF_FN void row_block_compute" in body and "DEKF_FN void residual_norms" in body
    assert "DEKF_FN" not in src[fast[0]:]
    first_fn = src.index("DEKF_FN")
    assert off[0] < first_fn
    # no build flag overrides the pragma for a kernel set
    assert "-ffp-contract" not in open(os.path.join(CSRC, "build.sh")).read()
    # the 3-term sums of every row-phase shape go through ONE helper (wave.h: dot3); a plain `a * b + c * d + e * f` on the slack-block
    # applies would be rounded as three products and two sums under contraction off — correct, identical in every shape, but slower
    assert body.count("dot3(") >= 8
    assert "dot3(double a0, double b0, double a1, double b1, double a2, double b2) { return fma(a2, b2, fma(a1, b1, a0 * b0)); }" in \
        open(os.path.join(CSRC, "wave.h")).read()


def test_algorithmic_bytes_formula_matches_the_survey():
    """B_alg(Go1, N = 20) = 5 736, Cassie 4 160, PogoX 11 596 bytes per estimator-step"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import bench_shapes
    from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params
    assert bench_shapes.alg_bytes_per_step(go1_params()) == 5736 == bench.B_ALG_GO1
    assert bench_shapes.alg_bytes_per_step(cassie_params()) == 4160
    assert bench_shapes.alg_bytes_per_step(pogox_params()) == 11596
    assert set(bench_shapes.SHAPES) >= {"go1", "cassie", "pogox", "go1foot", "go1_8192"}
