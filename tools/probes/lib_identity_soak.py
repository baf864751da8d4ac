#!/usr/bin/env python3
"""Outputs of a long run with FEW reads (so that whatever the library overlaps really runs concurrently), saved for a bit-for-bit comparison
between two builds of the library:   DEKF_LIB=.../libA.so python tools/probes/lib_identity_soak.py a.npz [ticks] [every]; the same with libB;
python tools/dump_outputs.py --diff a.npz b.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

out = sys.argv[1]
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
every = int(sys.argv[3]) if len(sys.argv) > 3 else 37


def tile(s, reps):
    return {k: (np.ascontiguousarray(np.tile(v, (1, reps) + (1,) * (v.ndim - 2))) if isinstance(v, np.ndarray) else v) for k, v in s.items()}


CASES = [("go1", go1_params, 4096, 64, ticks, {}), ("go1_polish", go1_params, 4096, 64, ticks // 2, dict(polish=1)),
         ("cassie", cassie_params, 4096, 64, ticks // 2, {}), ("pogox", pogox_params, 1024, 32, max(160, ticks // 3), {}),
         ("go1foot", go1_params, 1024, 32, max(120, ticks // 3), dict(leg_odom_type=1)),
         ("go1_small", go1_params, 96, 96, ticks // 2, {}), ("go1_n7", go1_params, 256, 64, ticks // 3, dict(N=7)),
         ("go1_pipelined", go1_params, 4096, 64, ticks // 2, dict(solve_pipeline=1))]
res = {}
for name, maker, B, distinct, K, kw in CASES:
    p = maker()
    p.ekf_rate = p.rate
    for k, v in kw.items():
        setattr(p, k, v)
    sd = streams_to_device(tile(make_streams(p, distinct, K, gait_hz=3.0), B // distinct))
    est = BatchedEstimator(p, B)
    xs, it = [], []
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
        if k % every == every - 1 or k == K - 1:
            o, info = est.get(), est.solver_info()
            xs.append(np.concatenate([o["x"], o["v_b"], o["quat"], o["p_vo"]], axis=1))
            it.append(np.stack([info["iters"], o["status"]], axis=1))
    print(name, os.path.basename(os.environ.get("DEKF_LIB", "product")), est.solve_kernel_name(True), "reads", len(xs), "solved", float((it[-1][:, 1] == 1).mean()), flush=True)
    est.close()
    res[name + "_x"] = np.stack(xs)
    res[name + "_iters_status"] = np.stack(it)
np.savez(out, **res)
