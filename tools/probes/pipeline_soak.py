#!/usr/bin/env python3
"""Long run of the pipelined mode against the in-order mode: Go1 at 4096 (64 distinct logs tiled), `ticks` steps, outputs compared bit for bit
every `every` ticks (a read waits for the newest solve only) and at the end; then PogoX at 1024 and the foot-state shape at 1024.
    python tools/probes/pipeline_soak.py [ticks] [every]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
every = int(sys.argv[2]) if len(sys.argv) > 2 else 97


def tile(s, reps):
    return {k: (np.ascontiguousarray(np.tile(v, (1, reps) + (1,) * (v.ndim - 2))) if isinstance(v, np.ndarray) else v) for k, v in s.items()}


def case(name, p, B, distinct, K):
    p.ekf_rate = p.rate
    sd = streams_to_device(tile(make_streams(p, distinct, K), B // distinct))
    outs = []
    for pipe in (0, 1):
        q = p.copy()
        q.solve_pipeline = pipe
        est = BatchedEstimator(q, B)
        kern = est.solve_kernel_name(True)
        mids = []
        t0 = time.perf_counter()
        for k in range(K):
            est.push_stream_step(sd, k)
            est.step(k)
            if k % every == every - 1:
                o = est.get()
                mids.append((o["x"].copy(), o["v_b"].copy(), o["status"].copy()))
        o, info = est.get(), est.solver_info()
        dt = time.perf_counter() - t0
        est.close()
        outs.append((o, info, mids, dt))
    (a, ia, ma, ta), (b, ib, mb, tb) = outs
    same = all(np.array_equal(a[k], b[k]) for k in ("x", "v_b", "quat", "p_vo", "status"))
    same = same and all(np.array_equal(ia[k], ib[k]) for k in ("iters", "rho_updates", "pri_res", "dua_res"))
    same_mid = all(np.array_equal(x[i], y[i]) for x, y in zip(ma, mb) for i in range(3))
    print(json.dumps({"case": name, "kernel": kern, "instances": B, "ticks": K, "reads": len(ma), "final_bit_identical": bool(same),
                      "every_read_bit_identical": bool(same_mid), "solved_frac": float((a["status"] == 1).mean()),
                      "in_order_s": round(ta, 2), "pipelined_s": round(tb, 2)}), flush=True)
    return same and same_mid


ok = case("go1", go1_params(), 4096, 64, ticks)
pp = pogox_params()
ok = case("pogox", pp, 1024, 32, max(200, ticks // 5)) and ok
pf = go1_params()
pf.leg_odom_type = 1
ok = case("go1 foot states", pf, 1024, 32, max(200, ticks // 5)) and ok
sys.exit(0 if ok else 1)
