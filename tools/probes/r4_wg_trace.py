#!/usr/bin/env python3
"""Round 6: what the persistent workgroups of the last solve launch did — start delay after the first one, instances solved, time resident
(A/B build with -DDEKF_AB_KNOBS: kernels.hip writes them into the section-stamp buffer).
    DEKF_WGS=4 DEKF_LIB=.../libdekf_q0.so python tools/probes/r4_wg_trace.py [batch]"""
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
K = 80
s = make_streams(p, B, K); sd = streams_to_device(s)
est = BatchedEstimator(p, B)
for k in range(K):
    est.push_stream_step(sd, k); est.step(k)
est.sync()
lib = capi.load()
out = np.zeros((B, 32))
lib.dekf_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
capi.check(lib.dekf_debug_sections(est.h, C.c_void_p(out.ctypes.data)))
G = est.launch_info()["solve_workgroups"]
w = out[:G]
t0 = w[:, 0].min()
start = (w[:, 0] - t0) / 100.0; end = (w[:, 1] - t0) / 100.0; n = w[:, 2].astype(int)
print("kernel", est.solve_kernel_name(True), "batch", B, "grid", G, "launch span %.1f us" % end.max())
print("  start delay (us): median %.1f  p90 %.1f  p99 %.1f  max %.1f;  workgroups starting later than 100 us: %d" %
      (np.median(start), np.percentile(start, 90), np.percentile(start, 99), start.max(), int((start > 100).sum())))
vals, cnt = np.unique(n, return_counts=True)
print("  instances per workgroup:", dict(zip(vals.tolist(), cnt.tolist())))
late = np.where(start > 100)[0]
if len(late):
    print("  late workgroups: start %s us, instances %s" % (np.round(start[late][:12], 1).tolist(), n[late][:12].tolist()))
hw = w[:, 3].astype(np.int64)
xcc = hw >> 16; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 4096 + se * 256 + sh * 16 + cu
early = start <= 100
ke, ce = np.unique(key[early], return_counts=True)
print("  CUs seen by the early workgroups: %d; workgroups per CU: %s" % (len(ke), dict(zip(*[a.tolist() for a in np.unique(ce, return_counts=True)]))))
if len(late):
    print("  late workgroups by XCD:", dict(zip(*[a.tolist() for a in np.unique(xcc[late], return_counts=True)])), " blockIdx of the first late ones:", late[:16].tolist())
    short = set(ke[ce < 4].tolist())
    print("  late workgroups that ran on a CU which held fewer than four early ones: %d of %d" % (sum(int(k) in short for k in key[late]), len(late)))
print("  residence per instance (us): mean %.1f  min %.1f  max %.1f" % (((end - start) / np.maximum(n, 1)).mean(), ((end - start) / np.maximum(n, 1))[n > 0].min(), ((end - start) / np.maximum(n, 1)).max()))
est.close()
