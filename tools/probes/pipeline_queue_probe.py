#!/usr/bin/env python3
"""solve_pipeline = 1 with OTHER streams alive in the process: does the overlap survive the round-robin mapping of HIP streams onto
hardware queues (GPU_MAX_HW_QUEUES, default 4)?
    [GPU_MAX_HW_QUEUES=8] python tools/probes/pipeline_queue_probe.py <idle handles alive beside the pipelined one>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

idle = int(sys.argv[1]) if len(sys.argv) > 1 else 0
p = go1_params()
p.ekf_rate = p.rate
B, W, K = 4096, 50, 100
sd = streams_to_device(make_streams(p, B, W + K))
others = [BatchedEstimator(p, 64) for _ in range(idle)]   # each owns one stream; never stepped
p.solve_pipeline = 1
est = BatchedEstimator(p, B)
for k in range(W):
    est.push_stream_step(sd, k); est.step(k)
est.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(W, W + K):
    est.push_stream_step(sd, k); est.step(k)
est.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"idle_handles_alive": idle, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"),
                  "steps_per_s": round(B * K / dt), "ms_per_step": round(1e3 * dt / K, 4)}))
est.close()
for o in others:
    o.close()
