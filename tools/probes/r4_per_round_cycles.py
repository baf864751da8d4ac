#!/usr/bin/env python3
"""Round 6: per-instance solve cycles (diagnostic build, -DDEKF_PROFILE: slot 13 = total) by instance index, in blocks of 512 — does the
cost of a solve depend on WHICH instance it is (data, addresses) or on WHEN it runs inside the launch?
    DEKF_WGS=4 DEKF_LIB=.../libdekf_prof.so python tools/probes/r4_per_round_cycles.py [batch]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from decentralized_ekf_mhe_amd import capi, go1_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = go1_params(); p.ekf_rate = p.rate
p.solve_workgroups_per_cu = int(os.environ.get("DEKF_WGS", 0))
K = 70
s = make_streams(p, B, K); sd = streams_to_device(s)
est = BatchedEstimator(p, B)
for k in range(K):
    est.push_stream_step(sd, k); est.step(k)
est.sync()
lib = capi.load()
out = np.zeros((B, 32))
lib.dekf_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
capi.check(lib.dekf_debug_sections(est.h, C.c_void_p(out.ctypes.data)))
tot = out[:, 13]
print("kernel", est.solve_kernel_name(True), "batch", B, "mean total cycles", round(tot.mean()))
for i in range(0, B, 512):
    blk = out[i:i + 512]
    print(f"  instances {i:5d}..{min(i + 512, B) - 1:5d}: total {blk[:, 13].mean():9.0f}  Ruiz {blk[:, 0].mean():8.0f}  iterations {blk[:, 9].mean():8.0f}  LDL legs {blk[:, 22].mean():7.0f}  residual tiles {blk[:, 21].mean():7.0f}  wall {blk[:, 23].mean() / 100:7.1f} us  shader clock {blk[:, 13].mean() / max(blk[:, 23].mean(), 1) * 100:6.0f} MHz")
est.close()
