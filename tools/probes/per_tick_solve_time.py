#!/usr/bin/env python3
"""Solve-launch duration and iteration count tick by tick over the bench's streams (Go1 at 4096): is a short timed region (the driver's
--steps 20 covers ticks 50-69) the same workload as a long one (ticks 50-249)?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

p = go1_params()
p.ekf_rate = p.rate
B, K = 4096, 250
sd = streams_to_device(make_streams(p, B, K))
est = BatchedEstimator(p, B)
est.timing_enable(2)
rows = []
for k in range(K):
    est.push_stream_step(sd, k)
    est.step(k)
    if k >= 40:
        est.sync()
        t = est.timing_read()["solve"]
        info = est.solver_info()
        rows.append((k, t[0] / max(t[1], 1), float(info["iters"].mean()), float(info["rho_updates"].mean())))
    elif k == 39:
        est.sync(); est.timing_read()
est.close()
a = np.array(rows)
for lo, hi in ((50, 70), (70, 100), (100, 150), (150, 250), (50, 250)):
    m = (a[:, 0] >= lo) & (a[:, 0] < hi)
    print(json.dumps({"ticks": [lo, hi], "solve_ms_mean": round(float(a[m, 1].mean()), 4), "iters_mean": round(float(a[m, 2].mean()), 2),
                      "rho_updates_mean": round(float(a[m, 3].mean()), 3)}))
print("per tick 50..69:", [(int(r[0]), round(r[1], 3), r[2]) for r in rows if 50 <= r[0] < 70])
