#!/usr/bin/env python3
"""Round 6: the four-per-CU solve kernel (workgroups of three wavefronts, solve_workgroups_per_cu = 4) against the three-per-CU one.
  1. bit identity on the same logs (states, v_b, quaternion, residuals, iteration and rho-update counts) over window fill + 40 ticks
  2. solve-kernel time and steps/s of both at B = 4096, alternating, on one box
usage: [DEKF_LIB=...] python tools/r4_check.py [identity_batch [timing_batch [rounds]]] [--shape go1|cassie]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from decentralized_ekf_mhe_amd import go1_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = "go1"
if "--shape" in sys.argv:
    shape = sys.argv[sys.argv.index("--shape") + 1]
    args = [a for a in args if a != shape]
BI = int(args[0]) if len(args) > 0 else 1100
BT = int(args[1]) if len(args) > 1 else 4096
ROUNDS = int(args[2]) if len(args) > 2 else 3
p = go1_params()
if shape == "cassie":
    from decentralized_ekf_mhe_amd import cassie_params
    p = cassie_params()
p.ekf_rate = p.rate
CAPS = (int(os.environ.get("R4_CAP_A", 0)), int(os.environ.get("R4_CAP_B", 4)))

# ---- 1. identity
if BI > 0:
    K = p.N + 40
    s = make_streams(p, BI, K); sd = streams_to_device(s)
    ests = []
    for cap in CAPS:
        q = p.copy(); q.solve_workgroups_per_cu = cap
        ests.append(BatchedEstimator(q, BI))
    print("kernels:", ests[0].solve_kernel_name(True), ests[1].solve_kernel_name(True), flush=True)
    bad = False
    for k in range(K):
        outs = []
        for e in ests:
            e.push_stream_step(sd, k); e.step(k)
            outs.append((e.get(), e.solver_info()))
        (oa, ia), (ob, ib) = outs
        same = all(np.array_equal(oa[key], ob[key]) for key in ("x", "v_b", "quat")) and \
            all(np.array_equal(ia[key], ib[key]) for key in ("iters", "pri_res", "dua_res", "rho_updates"))
        if not same and not bad:
            bad = True
            d = np.abs(oa["x"] - ob["x"]).max(axis=1); b = int(d.argmax())
            print("first difference at tick", k, "instance", b, "dx", d[b], "iters", ia["iters"][b], ib["iters"][b], "pri", ia["pri_res"][b],
                  ib["pri_res"][b], "dua", ia["dua_res"][b], ib["dua_res"][b], "instances differing", int((d > 0).sum()),
                  "status", np.unique(ib["status"], return_counts=True), flush=True)
    print("identity:", "identical" if not bad else "NOT identical", "final max dx", float(np.abs(oa["x"] - ob["x"]).max()), "mean iters", float(ia["iters"].mean()),
          float(ib["iters"].mean()), "finite", bool(np.isfinite(ob["x"]).all()), flush=True)
    for e in ests:
        e.close()

# ---- 2. timing
if BT > 0:
    W, STEPS = 64, 60
    s = make_streams(p, BT, W + STEPS * ROUNDS); sd = streams_to_device(s)
    res = {}
    for cap in sorted(set(CAPS), key=CAPS.index):
        q = p.copy(); q.solve_workgroups_per_cu = cap
        if os.environ.get("R4_PIPELINE"):
            q.solve_pipeline = int(os.environ["R4_PIPELINE"])
        if os.environ.get("R4_EKF_HISTORY"):
            q.ekf_history = int(os.environ["R4_EKF_HISTORY"])
        est = BatchedEstimator(q, BT)
        for k in range(W):
            est.push_stream_step(sd, k); est.step(k)
        est.sync(); torch.cuda.synchronize()
        est.timing_enable(2); est.timing_read()
        rates = []
        for r in range(ROUNDS):
            t0 = time.perf_counter()
            for k in range(W + r * STEPS, W + (r + 1) * STEPS):
                est.push_stream_step(sd, k); est.step(k)
            est.sync(); torch.cuda.synchronize()
            rates.append(BT * STEPS / (time.perf_counter() - t0))
        tim = est.timing_read()
        info = est.solver_info()
        res[cap] = dict(kernel=est.solve_kernel_name(True), workgroups=est.launch_info()["solve_workgroups"], steps_per_s=[round(x) for x in rates],
                        mean_iters=float(info["iters"].mean()),
                        solve_ms=round(tim["solve"][0] / max(tim["solve"][1], 1), 4))
        est.close()
    a, b = res[CAPS[0]], res[CAPS[1]]
    print(json.dumps(dict(batch=BT, shape=shape, a=a, b=b, ratio=round(float(np.median(b["steps_per_s"]) / np.median(a["steps_per_s"])), 4))), flush=True)
