#!/usr/bin/env python3
"""What the per-step RCCL all-gather of v_b (dekf_allgather_vb, second stream) costs the step on ONE GPU (world 1: the exchange is a
copy kernel, but the stream choreography — snapshot copy, event, stream-wait, launch on the communication stream — is the real one).
    python tools/probes/allgather_cost_probe.py [batch] [pipeline 0|1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, new_unique_id, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = go1_params()
p.ekf_rate = p.rate
p.solve_pipeline = pl
W, K = 50, 100
sd = streams_to_device(make_streams(p, B, W + K))
res = {}
for gather in (False, True, False, True):
    est = BatchedEstimator(p, B)
    if gather:
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
    res.setdefault("with all-gather" if gather else "without", []).append(round(1e3 * dt / K, 4))
    if gather:
        assert torch.equal(vb_all[0], torch.from_numpy(est.get()["v_b"]).cuda())
    est.close()
print(json.dumps({"batch": B, "solve_pipeline": pl, "ms_per_step": res}))
