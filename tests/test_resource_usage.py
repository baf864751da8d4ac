"""The register budget of the benchmark kernels is part of the design: three workgroups per CU need
<= 168 VGPRs, and what does not fit is spilled to scratch, i.e. to traffic beyond L2.  Those numbers hang on one hidden
compiler flag (-mllvm -disable-machine-licm, csrc/build.sh).  This test reads the compiler's own account of every kernel
(-Rpass-analysis=kernel-resource-usage, written by build.sh to csrc/libdekf_resource_usage.txt) and fails when a kernel has
left its design point — so that a toolchain update shows up here, on the CPU, before it shows up in the bench.

No GPU needed: hipcc cross-compiles.  If the product build has not been run (or is older than the sources) the two
benchmark kernel sets are compiled here, in parallel (about a minute)."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc")
USAGE = os.path.join(CSRC, "libdekf_resource_usage.txt")

# design points: kernel -> (max VGPRs, min waves per SIMD, max spilled VGPRs)
DESIGN_POINTS = {
    # three workgroups per CU: 7 spilled VGPRs at the end of round 4, none of them in an iteration loop
    "k_mhe_solve_r3_4_n20": (168, 3, 16),
    "k_mhe_solve_r3_2_n20": (168, 3, 16),
    # four workgroups of three wavefronts per CU (round 6, opt-in: solve_workgroups_per_cu = 4): the same 168-VGPR budget; the worker
    # that carries two row tiles spills a few more (16 / 14; the polishing twin 68)
    "k_mhe_solve_r4_4_n20": (168, 3, 24),
    "k_mhe_solve_r4_2_n20": (168, 3, 24),
    "k_mhe_solve_r4_4_n20_pol": (168, 3, 96),
    # rows in registers at a run-time horizon, two workgroups per CU: spill-free (40 spilled VGPRs inside its loops with machine LICM on)
    "k_mhe_solve_rr_1": (256, 2, 0),
    # window-fill ticks of the benchmark shapes
    "k_mhe_solve_ll_4_n20": (256, 2, 0),
    "k_mhe_solve_lg_2_n20": (256, 2, 0),
    # polishing twins of the benchmark kernels (45 / 4 spilled at the end of round 4)
    "k_mhe_solve_r3_4_n20_pol": (168, 3, 64),
    "k_mhe_solve_rr_1_pol": (256, 2, 16),
    # term construction: three wavefronts per SIMD by launch bound
    "k_mhe_assemble": (168, 3, 40),
}


def parse_usage(text):
    out = {}
    for blk in text.split("Function Name: ")[1:]:
        name = blk.split("\n")[0].strip()

        def g(key):
            m = re.search(key + r": (\S+)", blk)
            return int(m.group(1)) if m else None
        out[name] = dict(vgprs=g("VGPRs"), agprs=g("AGPRs"), spill=g("VGPRs Spill"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         occupancy=g(r"Occupancy \[waves/SIMD\]"), sgpr_spill=g("SGPRs Spill"))
    return out


def _sources_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".sh")))


def _compile_sets(masks):
    """the product's compile line for the given kernel sets (build.sh: unit), remarks only"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    with tempfile.TemporaryDirectory() as tmp:
        for m in masks:
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I/opt/rocm/include", f"-I{CSRC}"]
            if m != 1024:
                cmd += ["-mllvm", "-disable-machine-licm"]
            cmd += ["-Rpass-analysis=kernel-resource-usage", "-c", f"-DDEKF_KSET={m}", "-DDEKF_KSET_ONLY", "-o", os.path.join(tmp, f"k_{m}.o"),
                    os.path.join(CSRC, "kernels.hip")]
            procs.append(subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True))
        text = ""
        for p in procs:
            _, err = p.communicate()
            assert p.returncode == 0, err[-2000:]
            text += "\n".join(re.sub(r"^.*remark: *", "", ln).replace(" [-Rpass-analysis=kernel-resource-usage]", "")
                              for ln in err.splitlines() if "remark:" in ln) + "\n"
    return text


@pytest.fixture(scope="module")
def usage():
    if os.path.exists(USAGE) and os.path.getmtime(USAGE) >= _sources_mtime():
        return parse_usage(open(USAGE).read()), open(USAGE).read()
    text = _compile_sets([1, 2, 4, 1024])
    return parse_usage(text), text


def test_benchmark_kernels_are_at_their_design_points(usage):
    table, _ = usage
    bad = []
    for name, (max_vgprs, min_waves, max_spill) in DESIGN_POINTS.items():
        assert name in table, f"{name} is not in the compiler's resource remarks"
        u = table[name]
        if u["vgprs"] + (u["agprs"] or 0) > max_vgprs or u["occupancy"] < min_waves or u["spill"] > max_spill:
            bad.append(f"{name}: {u['vgprs']} VGPRs (+{u['agprs']} AGPRs), {u['occupancy']} waves/SIMD, {u['spill']} spilled VGPRs, "
                       f"{u['scratch']} B scratch per lane — design point: <= {max_vgprs} VGPRs, >= {min_waves} waves/SIMD, <= {max_spill} spilled")
    assert not bad, "kernels off their design point (toolchain change? -disable-machine-licm dropped?):\n" + "\n".join(bad)


def test_every_solve_kernel_keeps_two_workgroups_per_cu(usage):
    """every k_mhe_solve_* instantiation is launched with two (or three) 4-wavefront workgroups per CU: below 2 waves per SIMD the
    hardware silently places one and the shape runs at half speed (kernels.hip: DEKF_SOLVE_MIN_WAVES)"""
    table, _ = usage
    solves = {k: v for k, v in table.items() if k.startswith("k_mhe_solve_")}
    assert solves
    low = {k: v["occupancy"] for k, v in solves.items() if v["occupancy"] < 2}
    assert not low, low


def test_no_kernel_set_fell_back_to_the_retry_path(usage):
    """build.sh retries a kernel set without -disable-machine-licm when the compiler dies on it, and says so in the file"""
    _, text = usage
    assert "compiled WITHOUT -disable-machine-licm" not in text, [ln for ln in text.splitlines() if "WITHOUT" in ln]
