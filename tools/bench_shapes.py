#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (they are parity cases, not the bench line):
Cassie (2 legs, 5 joints) B=4096 N=20, PogoX (1 leg) B=1024 N=100, Go1 at the 8-GPU per-rank batch 8192,
and the Kalman-filter alternative (est_type 1).
Same loop as bench.py (device-resident synthetic logs, steady state), one JSON line per shape."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402


def run(name, maker, B, steps, **kw):
    p = maker()
    p.ekf_rate = p.rate
    for k, v in kw.items():
        setattr(p, k, v)
    W = p.N + 10
    s = make_streams(p, B, W + steps)
    sd = streams_to_device(s)
    est = BatchedEstimator(p, B)
    for k in range(W):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + steps):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    info = est.solver_info()
    o = est.get()
    est.close()
    print(json.dumps({"shape": name, "legs": p.num_legs, "N": p.N, "batch": B, "steps": steps,
                      "estimator_steps_per_s": B * steps / dt, "ms_per_step": 1e3 * dt / steps,
                      "mean_iters": float(info["iters"].mean()), "solved_frac": float((o["status"] == 1).mean()),
                      "polish_accepted_frac": float((info["polish_status"] == 1).mean())}), flush=True)


if __name__ == "__main__":
    run("go1 N=20 (bench line shape)", go1_params, 4096, 100)
    run("go1 N=20, per-rank batch of the 8-GPU config", go1_params, 8192, 60)
    run("cassie N=20", cassie_params, 4096, 100)
    run("pogox N=100", pogox_params, 1024, 60)
    run("go1 N=20 with osqp.polish (the node's declared default; k_mhe_solve_r3_4_n20_pol)", go1_params, 4096, 100, polish=1)
    run("pogox N=100 with osqp.polish", pogox_params, 1024, 40, polish=1)
    run("go1 KF mode (est_type 1: recursion instead of the QP)", go1_params, 4096, 200, est_type=1)
    run("go1 KF mode, batch 65536", go1_params, 65536, 100, est_type=1)
    run("go1 with foot-position states (leg_odom_type 1, 21-dim blocks)", go1_params, 4096, 40, leg_odom_type=1)
    run("go1 foot-position states, KF mode", go1_params, 4096, 100, leg_odom_type=1, est_type=1)
