#!/usr/bin/env python3
"""The three-workgroup solve kernel against the two-workgroup one on the same logs (bit identity expected): prints the first
tick and instance where states, residuals or iteration counts differ.  usage: DEKF_LIB=... python tools/r3_identity_check.py"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import go1_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
p = go1_params(); p.ekf_rate = p.rate
B, K = 1000, p.N + 30
s = make_streams(p, B, K); sd = streams_to_device(s)
ests = []
for cap in (0, 2):
    q = p.copy(); q.solve_workgroups_per_cu = cap
    ests.append(BatchedEstimator(q, B))
print(ests[0].solve_kernel_name(True), ests[1].solve_kernel_name(True))
bad = False
for k in range(K):
    outs = []
    for e in ests:
        e.push_stream_step(sd, k); e.step(k)
        outs.append((e.get(), e.solver_info()))
    (o3, i3), (o2, i2) = outs
    same = all(np.array_equal(o3[key], o2[key]) for key in ("x", "v_b", "quat")) and all(np.array_equal(i3[key], i2[key]) for key in ("iters", "pri_res", "dua_res"))
    if not same and not bad:
        bad = True
        d = np.abs(o3["x"] - o2["x"]).max(axis=1); b = int(d.argmax())
        print("first difference at tick", k, "instance", b, "dx", d[b], "iters", i3["iters"][b], i2["iters"][b],
              "pri", i3["pri_res"][b], i2["pri_res"][b], "dua", i3["dua_res"][b], i2["dua_res"][b], "rho updates", i3["rho_updates"][b], i2["rho_updates"][b],
              "instances differing", int((d > 0).sum()))
print("identical" if not bad else "NOT identical", "final max dx", np.abs(o3["x"] - o2["x"]).max(), "mean iters", i3["iters"].mean())
sys.exit(1 if bad else 0)
