#!/usr/bin/env python3
"""The arrival cost computed one step ahead against the diagnostic switch that turns it off, over vision schedules that hit the
look-ahead's one dependency from every side (a vision interval rewriting the bound of the very record the step folds away): rates of
5 … 100 Hz, latencies of 0 … 0.2 s, no vision at all; 2048 instances (several rounds per launch, so the background kernel really runs
beside the solve), one read at the end and one in the middle.  Exit 1 on the first differing bit."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from decentralized_ekf_mhe_amd import cassie_params, go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402


def tile(s, reps):
    return {k: (np.ascontiguousarray(np.tile(v, (1, reps) + (1,) * (v.ndim - 2))) if isinstance(v, np.ndarray) else v) for k, v in s.items()}


def run(p, sd, B, K, early):
    if early:
        os.environ.pop("DEKF_DEBUG_NO_EARLY_MARGINALIZE", None)
    else:
        os.environ["DEKF_DEBUG_NO_EARLY_MARGINALIZE"] = "1"
    est = BatchedEstimator(p, B)
    mid = None
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
        if k == K // 2:
            mid = est.get()["x"].copy()
    o, info = est.get(), est.solver_info()
    est.close()
    return o, info, mid


bad = 0
for maker, name in ((go1_params, "go1"), (cassie_params, "cassie")):
    for vo_rate, lat in ((5.0, 0.0), (5.0, 0.2), (20.0, 0.09), (30.0, 0.03), (50.0, 0.0), (50.0, 0.12), (100.0, 0.01), (None, 0.0)):
        p = maker()
        p.ekf_rate = p.rate
        B, K = 2048, 90
        s = make_streams(p, 32, K, vo=vo_rate is not None, vo_rate=vo_rate or 30.0, vo_latency=lat, gait_hz=3.0)
        sd = streams_to_device(tile(s, B // 32))
        a, b = run(p, sd, B, K, False), run(p, sd, B, K, True)
        same = all(np.array_equal(a[0][k], b[0][k]) for k in ("x", "v_b", "quat", "p_vo", "status")) and \
            all(np.array_equal(a[1][k], b[1][k]) for k in ("iters", "rho_updates", "pri_res", "dua_res")) and np.array_equal(a[2], b[2])
        bad += not same
        print(json.dumps({"robot": name, "vo_rate_hz": vo_rate, "vo_latency_s": lat, "instances": B, "ticks": K, "bit_identical": bool(same),
                          "solved_frac": float((b[0]["status"] == 1).mean()), "mean_iters": float(b[1]["iters"].mean())}), flush=True)
sys.exit(1 if bad else 0)
