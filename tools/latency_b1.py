#!/usr/bin/env python3
"""Latency of ONE robot through the host-pointer path (what the C++ shim does per timer tick: latch from host
memory, step, read the estimate back), i.e. the reference's own deployment shape: 200 Hz, 5 ms per tick, OSQP capped
at 2.8 ms.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: F401,E402  (torch's HIP runtime first: estimator._torch_runtime_first acts only when torch is already imported)
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_host  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    p = go1_params()
    p.ekf_rate = p.rate
    W, K = 40, 400
    s = streams_host(make_streams(p, B, W + K))
    est = BatchedEstimator(p, B)
    for k in range(W):
        est.push_stream_step(s, k)
        est.step(k)
    est.get()
    lat = []
    for k in range(W, W + K):
        t0 = time.perf_counter()
        est.push_stream_step(s, k)
        est.step(k)
        est.get()  # blocks until the estimate is on the host
        lat.append(time.perf_counter() - t0)
    info = est.solver_info()
    est.close()
    lat = np.array(lat) * 1e3
    print(json.dumps({"batch": B, "ticks": K, "ms_per_tick_median": float(np.median(lat)), "ms_p99": float(np.percentile(lat, 99)),
                      "ms_max": float(lat.max()), "iters_last": int(info["iters"].max()),
                      "note": "host pointers in, host arrays out, one blocking read per tick (python ctypes overhead included)"}))


if __name__ == "__main__":
    main()
