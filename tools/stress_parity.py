#!/usr/bin/env python3
"""Long parity run on the GPU box: many instances x many ticks of every robot shape through the C ABI
against the CPU oracle (tests/oracle_lib.py — test infrastructure), every tick compared.  Prints one
JSON line per case; exits non-zero if any block exceeds the tolerance of the parity tests
(|gpu - oracle|_inf <= 1e-4 |oracle|_inf + 1e-6 per 3-vector block, quaternion 1e-9).

    python tools/stress_parity.py            # about a minute on an MI355X box
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_host  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

RTOL, ATOL = 1e-4, 1e-6


def block_err(x, ref):
    worst = 0.0
    for blk in (slice(0, 3), slice(3, 6), slice(6, 9)):
        num = np.abs(x[..., blk] - ref[..., blk]).max(axis=-1)  # (foot-position blocks beyond the ninth state: not part of the tolerance contract)
        den = RTOL * np.abs(ref[..., blk]).max(axis=-1) + ATOL
        worst = max(worst, float((num / den).max()))
    return worst


def case(name, maker, B, K, threads, **kw):
    p = maker()
    p.ekf_rate = p.rate
    for k, v in kw.items():
        setattr(p, k, v)
    s = make_streams(p, B, K)
    t0 = time.time()
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=threads, want_iters=True)
    t_cpu = time.time() - t0
    est = BatchedEstimator(p, B)
    sd = streams_host(s)
    xs, qs, its, sts = [], [], [], []
    t0 = time.time()
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
        o = est.get()
        xs.append(o["x"]); qs.append(o["quat"]); sts.append(o["status"]); its.append(est.solver_info()["iters"])
    t_gpu = time.time() - t0
    est.close()
    x, q, it, st = np.array(xs), np.array(qs), np.array(its), np.array(sts)
    res = {"case": name, "instances": B, "ticks": K, "worst_block_error_over_tolerance": block_err(x[1:], x_ref[1:]),
           "max_abs_dx": float(np.abs(x[1:] - x_ref[1:]).max()), "max_abs_dquat": float(np.abs(q - q_ref).max()),
           "all_solved": bool((st[1:] == 1).all()) if p.est_type == 0 else None,  # the KF mode has no solver status
           "iteration_counts_equal_frac": float((it[1:] == it_ref[1:]).mean()),
           "mean_iters": float(it[1:].mean()), "oracle_s": round(t_cpu, 1), "gpu_s_incl_host_copies": round(t_gpu, 1)}
    print(json.dumps(res), flush=True)
    return res["worst_block_error_over_tolerance"] <= 1.0 and res["max_abs_dquat"] < 1e-9 and res["all_solved"] is not False


def main():
    th = min(16, os.cpu_count() or 1)
    ok = True
    ok &= case("go1 N=20 (BASELINE configs[1] shape)", go1_params, 256, 400, th)
    ok &= case("go1 N=20, KF mode", go1_params, 64, 200, th, est_type=1)
    ok &= case("cassie N=20", cassie_params, 128, 200, th)
    ok &= case("pogox N=100", pogox_params, 32, 260, th)
    ok &= case("go1 N=5", go1_params, 64, 120, th, N=5)
    ok &= case("go1 N=20, foot positions as states (leg_odom_type 1)", go1_params, 32, 120, th, leg_odom_type=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
