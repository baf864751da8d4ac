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
gather = len(sys.argv) > 2 and sys.argv[2] == "gather"   # + the single-rank RCCL all-gather of v_b after every step
p = go1_params()
p.ekf_rate = p.rate
B, W, K = 4096, 50, int(os.environ.get("PROBE_STEPS", "100"))
sd = streams_to_device(make_streams(p, B, W + K))
others = [BatchedEstimator(p, 64) for _ in range(idle)]   # each owns one stream; never stepped
p.solve_pipeline = 1
est = BatchedEstimator(p, B)
if gather:
    from decentralized_ekf_mhe_amd.estimator import new_unique_id
    est.comm_init(1, 0, new_unique_id())
    vb_all = torch.empty((1, B, 3), dtype=torch.float64, device="cuda")


def run(k0, k1):
    for k in range(k0, k1):
        est.push_stream_step(sd, k); est.step(k)
        if gather:
            est.allgather_vb(vb_all)


run(0, W)
est.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
run(W, W + K)
est.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"idle_handles_alive": idle, "all_gather": gather, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"),
                  "steps_per_s": round(B * K / dt), "ms_per_step": round(1e3 * dt / K, 4)}))
est.close()
for o in others:
    o.close()
