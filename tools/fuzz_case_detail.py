#!/usr/bin/env python3
"""One fuzz case again, tick by tick: where do device and oracle differ, and do they differ only where their ADMM iteration counts do?
    python tools/fuzz_case_detail.py '<shape>' B K reps '<json of params and stream.* keys>'      (the line tools/fuzz_parity.py printed)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import oracle_lib as O  # noqa: E402
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_host  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402

shape, B, K, reps, cfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), json.loads(sys.argv[5])
maker = {"go1": go1_params, "go1_short": go1_params, "go1_odd": go1_params, "go1foot": go1_params, "cassie": cassie_params,
         "cassie_long": cassie_params, "pogox": pogox_params}[shape]
p = maker()
p.ekf_rate = p.rate
skw = {}
for k, v in cfg.items():
    if k.startswith("stream."):
        skw[k[7:]] = v
    else:
        setattr(p, k, v)
s = make_streams(p, B, K, **skw)
x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
tile = {k: (np.ascontiguousarray(np.tile(v, (1, reps) + (1,) * (v.ndim - 2))) if isinstance(v, np.ndarray) else v) for k, v in s.items()}
est = BatchedEstimator(p, B * reps)
print("kernels", est.solve_kernel_name(True), est.solve_kernel_name(False))
sh = streams_host(tile)
xs, its, sts, pri, dua = [], [], [], [], []
for k in range(K):
    est.push_stream_step(sh, k)
    est.step(k)
    o, info = est.get(), est.solver_info()
    xs.append(o["x"][:B]); its.append(info["iters"][:B]); sts.append(o["status"][:B]); pri.append(info["pri_res"][:B]); dua.append(info["dua_res"][:B])
x, it, st = np.array(xs), np.array(its), np.array(sts)
ns = x.shape[-1]


def blk_err(a, b):
    e = np.zeros(a.shape[:-1])
    for j in range(0, ns, 3):
        num = np.abs(a[..., j:j + 3] - b[..., j:j + 3]).max(axis=-1)
        den = 1e-4 * np.abs(b[..., j:j + 3]).max(axis=-1) + 1e-6
        e = np.maximum(e, num / den)
    return e


err = blk_err(x[1:], x_ref[1:K])
same = it[1:] == it_ref[1:K]
print("ticks x instances", err.shape, "iteration counts equal", same.mean(), "status != 1:", int((st[1:] != 1).sum()))
print("worst error over tolerance where counts are EQUAL  :", err[same].max() if same.any() else None)
print("worst error over tolerance where counts DIFFER     :", err[~same].max() if (~same).any() else None)
bad = np.argwhere(err > 1.0)
print("entries above the tolerance:", len(bad))
for (t, b) in bad[:12]:
    print(f"  tick {t + 1} instance {b}: err {err[t, b]:.2f} x tol, iters device {it[t + 1, b]} oracle {it_ref[t + 1, b]}, status {st[t + 1, b]}, "
          f"pri {pri[t + 1][b]:.2e} dua {dua[t + 1][b]:.2e}")
# Who is right where they differ?  The oracle's own QP of that tick (its H, g, A, l, u as the reference would hand them to OSQP), solved
# EXACTLY (dense KKT on the active equalities, tests/ref_numpy.py: kkt_exact): distance of the oracle's and of the device's x_T from it.
if len(bad):
    import ref_numpy as RN
    nm = 3 * p.num_legs
    seen = set()
    for (t, b) in bad:
        if b in seen or len(seen) >= 4:
            continue
        seen.add(b)
        pipe = O.Pipe(p)
        for k in range(t + 2):
            pipe.feed(s, k, int(b))
            pipe.step(k)
        H, g, A, l, u = pipe.est.qp()
        sol = RN.kkt_exact(H, g, A, l, u)[0]
        xt = sol[len(sol) - ns - nm:len(sol) - nm]
        inf = pipe.est.solver_info()
        print(f"  tick {t + 1} instance {b}: |oracle - exact| {np.abs(x_ref[t + 1][b] - xt).max():.2e}   |device - exact| {np.abs(x[t + 1][b] - xt).max():.2e}   "
              f"(oracle: {inf['iters']} iterations, rho {inf['rho']:.3g}, {inf['rho_updates']} rho updates; device: {it[t + 1, b]} iterations)")
dif = np.argwhere(~same)
for (t, b) in dif[:8]:
    print(f"  count differs: tick {t + 1} instance {b}: device {it[t + 1, b]} oracle {it_ref[t + 1, b]} err {err[t, b]:.3f} x tol")
est.close()
