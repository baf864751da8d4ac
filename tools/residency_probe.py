#!/usr/bin/env python3
"""How does throughput scale with the number of resident workgroups per CU?  Go1 with a short window (N = 12: the
iterates and the factor need < 53 KiB of LDS, so THREE workgroups fit a CU) on the generic solve kernel, with the
resident count forced down through the diagnostic build's LDS padding (DEKF_DEBUG_LDS_PAD, -DDEKF_PROFILE builds only).
usage: DEKF_LIB=.../libdekf_w3.so python tools/residency_probe.py   (w3: -DDEKF_PROFILE -DDEKF_SOLVE_MIN_WAVES=3)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402


def run(N, pad, B=3072, W=24, K=40):
    if pad:
        os.environ["DEKF_DEBUG_LDS_PAD"] = str(pad)
    else:
        os.environ.pop("DEKF_DEBUG_LDS_PAD", None)
    p = go1_params()
    p.ekf_rate = p.rate
    p.N = N
    s = make_streams(p, B, W + K)
    sd = streams_to_device(s)
    est = BatchedEstimator(p, B)
    for k in range(W):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    est.timing_enable(True)
    for k in range(W, W + K):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    t = est.timing_read()
    info = est.solver_info()
    est.close()
    ms = t["solve"][0] / t["solve"][1]
    print(json.dumps({"N": N, "lds_pad_bytes": pad, "solve_ms": ms, "instances_per_s": B / ms * 1e3, "iters": float(info["iters"].mean())}), flush=True)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    for pad in (0, 8 * 1024, 30 * 1024, 60 * 1024):   # 3, 2(?), 2/1, 1 workgroups per CU depending on the base size
        run(N, pad)
