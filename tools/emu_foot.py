#!/usr/bin/env python3
"""Go1 with foot-position states (leg_odom_type 1) with a FIXED number of ADMM iterations (no termination check before the cap, no rho adaptation): timing harness for
emulation builds whose numerics are garbage on purpose (e.g. -DDEKF_EMU_WLDS: every read of the W factor redirected to LDS).
    DEKF_LIB=.../libdekf_x.so python tools/emu_foot.py [iters]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 75
p = go1_params()
p.ekf_rate = p.rate; p.leg_odom_type = 1
p.max_qp_iter, p.adapt_rho, p.check_termination = iters, 0, 100000
B, steps = 4096, 16
W = p.N + 6
s = make_streams(p, B, W + steps)
sd = streams_to_device(s)
est = BatchedEstimator(p, B)
for k in range(W):
    est.push_stream_step(sd, k); est.step(k)
est.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(W, W + steps):
    est.push_stream_step(sd, k); est.step(k)
est.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"lib": os.environ.get("DEKF_LIB", "product"), "fixed_iters": iters, "ms_per_step": 1e3 * dt / steps,
                  "steps_per_s_at_this_iteration_count": B * steps / dt, "mean_iters": float(est.solver_info()["iters"].mean()),
                  "solve_workgroups": est.launch_info()["solve_workgroups"]}))
est.close()
