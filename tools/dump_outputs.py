#!/usr/bin/env python3
"""Runs one shape for K ticks and saves every tick's outputs (state blocks, v_b, quaternion, iteration counts, arrival cost) to an
.npz: two libraries (DEKF_LIB=...) can then be compared BIT for bit — the check behind "same arithmetic, another place for the
operands" changes.    DEKF_LIB=csrc/libdekf_old.so python tools/dump_outputs.py go1foot out_old.npz ;  ... ; python tools/dump_outputs.py --diff a.npz b.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def diff(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k].view(np.uint8), B[k].view(np.uint8))
        n = int((A[k] != B[k]).sum()) if not same else 0
        print(f"{k}: {'bit-identical' if same else f'{n} of {A[k].size} entries differ, max |diff| {np.nanmax(np.abs(A[k] - B[k])):.3e}'}")
        bad += not same
    return bad


if sys.argv[1] == "--diff":
    sys.exit(diff(sys.argv[2], sys.argv[3]))

import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

shape, out = sys.argv[1], sys.argv[2]
cases = {"go1foot": (go1_params, 96, 260, dict(leg_odom_type=1)), "cassiefoot": (cassie_params, 64, 200, dict(leg_odom_type=1)),
         "pogoxfoot": (pogox_params, 32, 160, dict(leg_odom_type=1, N=30)), "go1": (go1_params, 64, 120, {}),
         "pogox": (pogox_params, 320, 128, {}), "pogox64": (pogox_params, 64, 128, {}), "cassie30": (cassie_params, 64, 70, dict(N=30)),
         "go1n13": (go1_params, 64, 50, dict(N=13)), "go1n12": (go1_params, 64, 50, dict(N=12)),
         "go1r3": (go1_params, 832, 70, {}), "cassier3": (cassie_params, 832, 70, {})}  # batches that launch the three-workgroup kernels
maker, B, K, kw = cases[shape]
p = maker()
p.ekf_rate = p.rate
for k, v in kw.items():
    setattr(p, k, v)
s = make_streams(p, B, K, gait_hz=5.0)  # many swing phases: the foot blocks of the arrival cost lose and regain their information
sd = streams_to_device(s)
est = BatchedEstimator(p, B)
rows = {"x": [], "vb": [], "quat": [], "iters": [], "status": []}
for k in range(K):
    est.push_stream_step(sd, k)
    est.step(k)
    o, info = est.get(), est.solver_info()
    rows["x"].append(o["x"]); rows["vb"].append(o["v_b"]); rows["quat"].append(o["quat"]); rows["iters"].append(info["iters"]); rows["status"].append(o["status"])
print(shape, os.environ.get("DEKF_LIB", "product"), "kernel", est.solve_kernel_name(True), "mean iters", float(np.mean(rows["iters"][-1])), "solved", float((rows["status"][-1] == 1).mean()))
est.close()
np.savez(out, **{k: np.stack(v) for k, v in rows.items()})
