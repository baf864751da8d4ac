#!/usr/bin/env python3
"""Runs the benchmark shapes through the bounds-checked diagnostic build (csrc/libdekf_bounds.so, -DDEKF_BOUNDS: every pointer of
the solve cores carries the extent of the array it was carved from; wave.h) and prints the device-side violation counter.
    tools/build_bounds.sh                      # one library per kernel set, ~10 minutes in parallel
    python tools/bounds_check.py [set ...]     # every case (or the cases of the named kernel sets), one process per case"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decentralized_ekf_mhe_amd import capi, cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402


def hits():
    lib = capi.load()
    out = (C.c_ulonglong * 4)()
    capi.check(lib.dekf_debug_bounds(out))
    return list(out)


def run(name, p, B, K, **kw):
    p.ekf_rate = p.rate
    for k, v in kw.items():
        setattr(p, k, v)
    capi.check(capi.load().dekf_debug_bounds_reset())
    before = hits()[0]
    sd = streams_to_device(make_streams(p, B, K))
    est = BatchedEstimator(p, B)
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
    o, info = est.get(), est.solver_info()
    kern = (est.solve_kernel_name(False), est.solve_kernel_name(True))
    est.close()
    h = hits()
    print(json.dumps({"case": name, "kernels": kern, "batch": B, "ticks": K, "solved": float((o["status"] == 1).mean()), "mean_iters": float(info["iters"].mean()),
                      "violations": h[0] - before, "first_violation": None if h[0] == 0 else {"offset": int(np.int64(np.uint64(h[1]))), "extent": h[2]}}), flush=True)
    return h[0] - before


CASES = [  # (kernel set = library suffix, name, params maker, batch, ticks, overrides)
    ("go1", "go1 N=20: window fill (k_mhe_solve_ll_4_n20) and full windows (k_mhe_solve_r3_4_n20), VO, two refactorisations", "go1", 1000, 70, {}),
    ("go1", "go1 N=20, two workgroups per CU for full windows too", "go1", 600, 45, dict(solve_workgroups_per_cu=2)),
    ("cassie", "cassie N=20 (k_mhe_solve_lg_2_n20, k_mhe_solve_r3_2_n20)", "cassie", 900, 60, {}),
    # round 6: four workgroups of three wavefronts per CU (LDS carved to 4928 doubles, D / E / PA / stash in the slab, Dxb in the pad +
    # Gauss-Jordan scratch), the instance queue with more instances than workgroups, and the three-workgroup kernel at a small batch
    ("go1", "go1 N=20, four per CU (k_mhe_solve_r4_4_n20), 1100 instances on 1024 workgroups", "go1", 1100, 60, dict(solve_workgroups_per_cu=4)),
    ("go1", "go1 N=20, four per CU with osqp.polish (k_mhe_solve_r4_4_n20_pol)", "go1", 1050, 40, dict(solve_workgroups_per_cu=4, polish=1)),
    ("cassie", "cassie N=20, four per CU (k_mhe_solve_r4_2_n20)", "cassie", 1100, 50, dict(solve_workgroups_per_cu=4)),
    ("go1", "go1 N=20, 48 instances: full windows on k_mhe_solve_r3_4_n20 (every batch since round 6)", "go1", 48, 50, {}),
    ("legs1", "pogox N=100 (factor streamed from the slab)", "pogox", 64, 130, {}),
    ("legs1", "pogox N=100, 320 instances: full windows on k_mhe_solve_rr_1 (row state in registers, two workgroups per CU)", "pogox", 320, 128, {}),
    ("legs4", "go1 N=5", "go1", 64, 30, dict(N=5)),
    ("legs4", "go1 N=7 (odd horizon: two-wavefront solve form)", "go1", 64, 30, dict(N=7)),
    ("legs3", "three legs, six joints, N=12", "go1", 64, 40, dict(num_legs=3, joints_per_leg=6, N=12)),
    ("legs2", "two legs, N=30", "cassie", 64, 60, dict(N=30)),
    ("foot4", "go1 leg_odom_type 1 (factor in the slab)", "go1", 256, 45, dict(leg_odom_type=1)),
    ("foot2", "cassie leg_odom_type 1, N=8 (factor in LDS)", "cassie", 64, 30, dict(leg_odom_type=1, N=8)),
    ("foot1", "one leg, leg_odom_type 1, N=30", "go1", 64, 50, dict(leg_odom_type=1, num_legs=1, N=30)),
    ("foot3", "three legs, leg_odom_type 1, N=5", "go1", 64, 20, dict(leg_odom_type=1, num_legs=3, N=5)),
    # osqp.polish: the kernels' twins with the polishing step (a second factorisation and 1 + polish_refine_iter iterations per solve)
    ("go1", "go1 N=20 with osqp.polish (k_mhe_solve_ll_4_n20_pol, k_mhe_solve_r3_4_n20_pol)", "go1", 800, 50, dict(polish=1)),
    ("cassie", "cassie N=20 with osqp.polish (k_mhe_solve_lg_2_n20_pol, k_mhe_solve_r3_2_n20_pol)", "cassie", 800, 45, dict(polish=1)),
    ("legs1", "pogox N=100 with osqp.polish (factor streamed from the slab)", "pogox", 64, 110, dict(polish=1)),
    ("legs1", "pogox N=100 with osqp.polish, 320 instances (k_mhe_solve_rr_1_pol)", "pogox", 320, 112, dict(polish=1)),
    ("foot4", "go1 leg_odom_type 1 with osqp.polish", "go1", 128, 30, dict(leg_odom_type=1, polish=1)),
    # solve_pipeline = 1: N + 2 window records, three copies of the solve's input snapshot, two sets of outputs and slabs
    ("go1", "go1 N=20, pipelined steps (1000 instances: two rounds per launch, the handle's stream two steps ahead)", "go1", 1000, 70, dict(solve_pipeline=1)),
    ("legs1", "pogox N=100, pipelined steps, 320 instances (k_mhe_solve_rr_1)", "pogox", 320, 128, dict(solve_pipeline=1)),
    ("foot4", "go1 leg_odom_type 1, pipelined steps", "go1", 256, 45, dict(leg_odom_type=1, solve_pipeline=1)),
]
MAKERS = {"go1": go1_params, "cassie": cassie_params, "pogox": pogox_params}


def child(idx):
    import torch
    torch.cuda.init()  # (before the library's first HIP call: the other order leaves torch without a device on this image)
    lib = capi.load()
    lib.dekf_debug_bounds.argtypes = [C.POINTER(C.c_ulonglong)]
    kset, name, maker, B, K, kw = CASES[idx]
    if idx == 0:
        lib.dekf_debug_bounds_selftest.restype = C.c_int
        capi.check(lib.dekf_debug_bounds_selftest())
        st = hits()
        print(json.dumps({"selftest": "one deliberate out-of-range read and write in a 16-element array", "violations": st[0], "first": {"offset": st[1], "extent": st[2]}}), flush=True)
        assert st[0] == 2 and st[2] == 16, st
    return run(name, MAKERS[maker](), B, K, **kw)


def main():
    """one child process per case: each kernel set is its own library (tools/build_bounds.sh) and a process binds one library"""
    import subprocess
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        sys.exit(1 if child(int(sys.argv[2])) else 0)
    only = sys.argv[1:] or None
    bad = 0
    for idx, (kset, name, *_rest) in enumerate(CASES):
        if only and kset not in only:
            continue
        libp = os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc", f"libdekf_bounds_{kset}.so")
        if not os.path.exists(libp):
            print(json.dumps({"case": name, "skipped": f"{os.path.basename(libp)} not built"}), flush=True)
            bad += 1
            continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--case", str(idx)], env=dict(os.environ, DEKF_LIB=libp), timeout=600)
        bad += r.returncode != 0
    print(json.dumps({"cases_with_violations_or_errors": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
