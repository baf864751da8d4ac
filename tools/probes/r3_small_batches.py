import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from decentralized_ekf_mhe_amd import go1_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
p = go1_params(); p.ekf_rate = p.rate
for B in (64, 128, 256, 384, 512):
    s = make_streams(p, B, 64 + 60 * 3); sd = streams_to_device(s)
    out = {}
    for name, env in (("ll", None), ("r3", "1")):
        if env: os.environ["DEKF_DEBUG_R3_ALWAYS"] = env
        else: os.environ.pop("DEKF_DEBUG_R3_ALWAYS", None)
        est = BatchedEstimator(p, B)
        for k in range(64):
            est.push_stream_step(sd, k); est.step(k)
        est.sync(); torch.cuda.synchronize()
        est.timing_enable(2); est.timing_read()
        rates = []
        for r in range(3):
            t0 = time.perf_counter()
            for k in range(64 + 60 * r, 64 + 60 * (r + 1)):
                est.push_stream_step(sd, k); est.step(k)
            est.sync(); torch.cuda.synchronize()
            rates.append(B * 60 / (time.perf_counter() - t0))
        tim = est.timing_read()
        out[name] = dict(kernel=est.solve_kernel_name(True), grid=est.launch_info()["solve_workgroups"], steps_per_s=round(float(np.median(rates))), solve_ms=round(tim["solve"][0] / tim["solve"][1], 4))
        est.close()
    print(json.dumps(dict(batch=B, ll=out["ll"], r3=out["r3"], r3_over_ll_solve_time=round(out["r3"]["solve_ms"] / out["ll"]["solve_ms"], 4))), flush=True)
