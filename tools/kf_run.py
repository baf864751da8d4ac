#!/usr/bin/env python3
"""Runs the Kalman-filter alternative (est_type 1) on synthetic Go1 logs: the workload of tools/collect_mfma.sh.
    python tools/kf_run.py [batch [steps]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
p = go1_params()
p.ekf_rate = p.rate
p.est_type = 1
sd = streams_to_device(make_streams(p, B, K))
est = BatchedEstimator(p, B)
est.sync()
t0 = time.perf_counter()
for k in range(K):
    est.push_stream_step(sd, k)
    est.step(k)
est.sync()
torch.cuda.synchronize()
print(json.dumps({"mode": "KF", "batch": B, "steps": K, "estimator_steps_per_s": B * K / (time.perf_counter() - t0)}))
est.close()
