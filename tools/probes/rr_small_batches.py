#!/usr/bin/env python3
"""PogoX full windows at small batches: k_mhe_solve_rr_1 (rows in registers, two workgroups per CU) against the generic k_mhe_solve_gg_1
(A/B build with -DDEKF_AB_KNOBS: DEKF_DEBUG_RR_ALWAYS selects rr below 256 instances too)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from decentralized_ekf_mhe_amd import pogox_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
p = pogox_params(); p.ekf_rate = p.rate
W = p.N + 10
for B in (32, 128, 256):
    s = make_streams(p, B, W + 20 * 2); sd = streams_to_device(s)
    out = {}
    for name, env in (("gg", None), ("rr", "1")):
        if env: os.environ["DEKF_DEBUG_RR_ALWAYS"] = env
        else: os.environ.pop("DEKF_DEBUG_RR_ALWAYS", None)
        est = BatchedEstimator(p, B)
        for k in range(W):
            est.push_stream_step(sd, k); est.step(k)
        est.sync(); torch.cuda.synchronize()
        est.timing_enable(2); est.timing_read()
        rates = []
        for r in range(2):
            t0 = time.perf_counter()
            for k in range(W + 20 * r, W + 20 * (r + 1)):
                est.push_stream_step(sd, k); est.step(k)
            est.sync(); torch.cuda.synchronize()
            rates.append(B * 20 / (time.perf_counter() - t0))
        tim = est.timing_read()
        out[name] = dict(kernel=est.solve_kernel_name(True), grid=est.launch_info()["solve_workgroups"], steps_per_s=round(float(np.median(rates))), solve_ms=round(tim["solve"][0] / tim["solve"][1], 4))
        est.close()
    print(json.dumps(dict(batch=B, gg=out["gg"], rr=out["rr"], rr_over_gg_solve_time=round(out["rr"]["solve_ms"] / out["gg"]["solve_ms"], 4))), flush=True)
