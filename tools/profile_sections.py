#!/usr/bin/env python3
"""Where does one MHE solve spend its cycles?  Runs the DIAGNOSTIC build of the library
(csrc/libdekf_prof.so, -DDEKF_PROFILE: s_memtime stamps at every phase boundary, written to a
debug buffer only) on the bench workload and prints the share of each section.  Never quote the
absolute time of this build; read the shares.

    DEKF_LIB=decentralized_ekf_mhe_amd/csrc/libdekf_prof.so python tools/profile_sections.py [batch [ticks [go1|cassie|pogox|go1foot]]]
    DEKF_TIMELINE=1 DEKF_LIB=.../libdekf_tl.so python tools/profile_sections.py 4096 70   # -DDEKF_PROFILE -DDEKF_PROFILE_TL
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEKF_LIB", os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc", "libdekf_prof.so"))

import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import capi, cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

NAMES = ["0 copy R + Ruiz scaling", "1 bounds + first factorisation + restart", "2 X: reduced rhs on x columns",
         "3 S1: forward legs (two-wavefront form only)", "4 S2: meeting block | g_k (two-wavefront form only)", "5 S: block-tridiagonal solve on one wavefront (or S3: outward legs)", "6 factor 3a: slack blocks (both factorisations)", "7 factor 3b-3c: PA, T_kk, C_k", "8 factor 3d: block LDL'",
         "9 R: fused row blocks", "10 residuals + termination", "11 rho update + refactor", "12 epilogue", "13 total",
         "14 R3: chunk prologue (x blocks in, row blocks into registers = restart)", "15 R3: chunk epilogue (row state and x blocks back to the slab)",
         "16 R copy, P staging, first column norms", "17 -", "18 -", "19 scaled bounds + cold start", "20 residuals: D x", "21 residuals: tile phase (state + window records)",
         "22 block LDL': the two legs", "23 -",
         "24 Ruiz (-DDEKF_PROFILE_RUIZ): tile phase, wavefront 0", "25 Ruiz tile phase, wavefront 1", "26 Ruiz tile phase, wavefront 2", "27 Ruiz tile phase, wavefront 3",
         "28 Ruiz: wait at the barrier behind the tiles, wavefront 0", "29 Ruiz barrier wait, wavefront 1", "30 Ruiz barrier wait, wavefront 2", "31 Ruiz barrier wait, wavefront 3"]
NAMES[17] = "17 Ruiz (-DDEKF_PROFILE_RUIZ): x_0 norms + sum, mean of wavefronts 0, 2"
NAMES[18] = "18 Ruiz (-DDEKF_PROFILE_RUIZ): x_0 norms + sum, mean of wavefronts 1, 3"


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    shape = sys.argv[3] if len(sys.argv) > 3 else "go1"
    p = {"go1": go1_params, "cassie": cassie_params, "pogox": pogox_params, "go1foot": go1_params}[shape]()
    p.ekf_rate = p.rate
    if os.environ.get("DEKF_WGS"):  # solve_workgroups_per_cu (4: the four-per-CU kernels of round 6)
        p.solve_workgroups_per_cu = int(os.environ["DEKF_WGS"])
    if shape == "go1foot":
        p.leg_odom_type = 1  # foot positions as states: 21-dim blocks, factor streamed from the HBM slab
    s = make_streams(p, B, K)
    sd = streams_to_device(s)
    est = BatchedEstimator(p, B)
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    lib = capi.load()
    out = np.zeros((B, 32))
    lib.dekf_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
    capi.check(lib.dekf_debug_sections(est.h, C.c_void_p(out.ctypes.data)))
    info = est.solver_info()
    mean = out.mean(axis=0)
    if os.environ.get("DEKF_TIMELINE"):
        # library built with -DDEKF_PROFILE -DDEKF_PROFILE_TL (csrc/libdekf_tl.so): per-wavefront intervals of the
        # fixed-horizon iteration, summed over the iterations of the last solve
        it = float(info["iters"].mean())
        print(f"iterations {it:.1f}; cycles per iteration and wavefront (w0 runs the solve)")
        for name, base in (("solve | tile prefetch, up to the barrier", 0), ("row-tile work behind the barrier", 4),
                           ("wait at the barrier that ends the row phase", 8)):
            print(f"  {name:46s}" + "".join(f"  w{w}: {mean[base + w] / it:7.0f}" for w in range(4)))
        print(json.dumps({"batch": B, "T": K - 1, "mean_iters": it, "per_iteration_cycles": {
            "to_barrier": [mean[w] / it for w in range(4)], "row_work": [mean[4 + w] / it for w in range(4)],
            "row_barrier_wait": [mean[8 + w] / it for w in range(4)]}}))
        est.close()
        return
    tot = mean[13]
    res = {"batch": B, "T": K - 1, "mean_iters": float(info["iters"].mean()), "mean_rho_updates": float(info["rho_updates"].mean()),
           "total_cycles_mean": tot, "sections": {}}
    for i, nme in enumerate(NAMES):
        if i == 13 or nme.endswith(" -") or (i > 13 and mean[i] == 0.0):
            continue
        res["sections"][nme] = {"cycles": mean[i], "share": mean[i] / tot if tot else 0.0}
        print(f"{nme:38s} {mean[i]:12.0f} cyc  {100 * mean[i] / max(tot, 1):5.1f} %")
    print(f"{'total':38s} {tot:12.0f} cyc   iters {res['mean_iters']:.1f}  rho updates {res['mean_rho_updates']:.2f}")
    print(json.dumps(res))
    est.close()


if __name__ == "__main__":
    main()
