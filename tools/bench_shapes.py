#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (they are parity cases, not the bench line):
Cassie (2 legs, 5 joints) B=4096 N=20, PogoX (1 leg) B=1024 N=100, Go1 at the 8-GPU per-rank batch 8192,
and the Kalman-filter alternative (est_type 1).
Same loop as bench.py (device-resident synthetic logs, steady state), one JSON line per shape; an MHE shape's line carries the
same `roofline` block as bench.py's: SURVEY.md section 8(d)'s algorithmic bytes per estimator-step x the instances one launch of
the solve kernel processes / that kernel's average launch time (HIP events on the handle's stream, dekf_timing_read), against the
8 TB/s of HBM; `traffic` is quoted from profiles/traffic_<kernel>.json (tools/collect_traffic.sh <shape>) when that file was
collected for the same kernel, batch and revision of csrc/.
    python tools/bench_shapes.py                 every shape
    python tools/bench_shapes.py pogox [steps]   one shape (the form tools/collect_traffic.sh and collect_sq.sh put behind rocprofv3)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params  # noqa: E402
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device  # noqa: E402
from decentralized_ekf_mhe_amd.streams import make_streams  # noqa: E402


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


def alg_bytes_per_step(p):
    """inputs + outputs + EKF state and arrival cost read and written + the window ring (fp32 I/O model);
    Go1 N = 20: 5 736, Cassie: 4 160, PogoX: 11 596"""
    L, nj, N = int(p.num_legs), int(p.joints_per_leg), int(p.N)
    return (80 + 16 * L * (1 + nj)) + 68 + 880 + (N + 1) * 4 * (17 + 9 * L)


def measured_traffic(kernel, batch):
    import bench
    f = os.path.join(ROOT, "profiles", f"traffic_{kernel}.json")
    if not os.path.exists(f):  # (the bench line's kernel: tools/collect_traffic.sh without a shape writes traffic_k_mhe_solve.json)
        f = os.path.join(ROOT, "profiles", "traffic_k_mhe_solve.json")
    try:
        d = json.load(open(f))
    except Exception:
        return None, f"profiles/traffic_{kernel}.json absent"
    if d.get("kernel") != kernel or int(d.get("batch", -1)) != int(batch) or d.get("csrc_sha1") != bench.csrc_sha1():
        return None, f"profiles/traffic_{kernel}.json is for {d.get('kernel')} at batch {d.get('batch')}, csrc {d.get('csrc_sha1')}"
    return d.get("hbm_bytes_per_launch"), f"profiles/traffic_{kernel}.json"


def run(name, maker, B, steps, pipelined_leg=True, streams_kw=None, **kw):
    p = maker()
    p.ekf_rate = p.rate
    for k, v in kw.items():
        setattr(p, k, v)
    W = max(p.N + 10, 64)   # (past the transient of the first vision intervals: ticks 40-56 on the N = 20 shapes, see bench.py)
    HIST = 16 if streams_kw else 0   # ticks behind the timed region whose iteration counts are read (a read synchronises: not in the timed region)
    s = make_streams(p, B, W + steps + HIST, **(streams_kw or {}))
    sd = streams_to_device(s)
    est = BatchedEstimator(p, B)
    # as bench.py: the small kernels are timed over the last warm-up steps, the timed region brackets only the solve launches
    for k in range(W - 8):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    est.timing_enable(1)
    for k in range(W - 8, W):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    tim_warm = est.timing_read()
    est.timing_enable(2 if int(p.est_type) == 0 else 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + steps):
        est.push_stream_step(sd, k)
        est.step(k)
    est.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tim = est.timing_read()
    est.timing_enable(False)
    for cls in ("ekf", "assemble"):
        if tim[cls][1] == 0:
            tim[cls] = tim_warm[cls]
    info = est.solver_info()
    o = est.get()
    kernel = est.solve_kernel_name(True)
    li = est.launch_info()
    hist = None
    if HIST:  # ADMM iterations per instance and launch: how far from lock-step the fleet is
        import numpy as np
        its = []
        for k in range(W + steps, W + steps + HIST):
            est.push_stream_step(sd, k)
            est.step(k)
            its.append(est.solver_info()["iters"].copy())
        its = np.array(its)
        vals, cnt = np.unique(its, return_counts=True)
        hist = {"iterations": {int(v): round(float(c) / its.size, 4) for v, c in zip(vals, cnt)}, "mean": float(its.mean()),
                "max_over_mean_per_launch": float((its.max(axis=1) / its.mean(axis=1)).mean()), "ticks": HIST}
    est.close()
    piped = None
    if int(p.est_type) == 0 and pipelined_leg:  # the same steps with consecutive steps overlapped (bit-identical results; reported beside the in-order figure)
        q = p.copy()
        q.solve_pipeline = 1
        est = BatchedEstimator(q, B)
        for k in range(W):
            est.push_stream_step(sd, k)
            est.step(k)
        est.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(W, W + steps):
            est.push_stream_step(sd, k)
            est.step(k)
        est.sync()
        torch.cuda.synchronize()
        piped = B * steps / (time.perf_counter() - t1)
        est.close()
    roof = None
    if kernel and tim["solve"][1] > 0 and int(p.leg_odom_type) == 0:
        avg_s = tim["solve"][0] / tim["solve"][1] * 1e-3
        b_alg = alg_bytes_per_step(p)
        traffic, src = measured_traffic(kernel, B)
        roof = {"bound": "hbm", "achieved": b_alg * B / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": b_alg * B / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "kernel": kernel,
                "avg_launch_ms": avg_s * 1e3, "launches": tim["solve"][1], "alg_bytes_per_step": b_alg, "units_per_launch": B,
                "solve_workgroups": li["solve_workgroups"],
                # what the kernel really moves beyond L2 (TCC FETCH / WRITE: Infinity-Cache hits included), as a rate: for the
                # shapes whose factor streams from the slab every iteration THIS is the resource they run against
                "traffic_rate_gbs": (traffic / avg_s / 1e9) if traffic else None,
                "traffic_over_algorithmic": (traffic / (b_alg * B)) if traffic else None}
    elif kernel and tim["solve"][1] > 0:
        # SURVEY 8(d) prices leg_odom_type 0 only; the foot-state variant carries 3 L more states per window step: no contract figure
        avg_s = tim["solve"][0] / tim["solve"][1] * 1e-3
        traffic, src = measured_traffic(kernel, B)
        roof = {"kernel": kernel, "avg_launch_ms": avg_s * 1e3, "launches": tim["solve"][1],
                "alg_bytes_per_step": None, "note": "SURVEY.md 8(d) has no byte figure for leg_odom_type 1",
                "traffic": traffic, "traffic_source": src, "traffic_rate_gbs": (traffic / avg_s / 1e9) if traffic else None}
    print(json.dumps({"shape": name, "roofline": roof, "kernel_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in tim.items()}, "legs": p.num_legs, "N": p.N, "batch": B, "steps": steps,
                      "estimator_steps_per_s": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "with_step_pipelining_steps_per_s": piped,
                      "iteration_histogram": hist, "solve_kernel": kernel, "lib": os.environ.get("DEKF_LIB", "product"),
                      "mean_iters": float(info["iters"].mean()), "solved_frac": float((o["status"] == 1).mean()),
                      "polish_accepted_frac": float((info["polish_status"] == 1).mean())}), flush=True)


SHAPES = {  # the single-shape form (profilers put `python3 tools/bench_shapes.py <shape>` behind `--`)
    "go1": ("go1 N=20 (bench line shape)", go1_params, 4096, 30, {}),
    "go1_8192": ("go1 N=20, per-rank batch of the 8-GPU config", go1_params, 8192, 20, {}),
    "go1_65536": ("go1 N=20, the WHOLE batch of the 8-GPU config (65 536) on one GPU", go1_params, 65536, 10, {}),
    "cassie": ("cassie N=20", cassie_params, 4096, 30, {}),
    "pogox": ("pogox N=100", pogox_params, 1024, 12, {}),
    "go1foot": ("go1 with foot-position states (leg_odom_type 1, 21-dim blocks)", go1_params, 4096, 8, dict(leg_odom_type=1)),
    # a fleet that is not in lock-step (round 6): every robot's camera on its own clock, latency U[10, 60] ms, gait 1-3 Hz (streams.py: desync)
    "go1_desync": ("go1 N=20, de-synchronised fleet (own camera clock, latency 10-60 ms, gait 1-3 Hz per robot)", go1_params, 4096, 100,
                   dict(streams_kw=dict(desync=True))),
    "go1_mixed": ("go1 N=20, MIXED fleet (cameras at 5-50 Hz on their own clocks, every tenth robot blind, gait 1-3 Hz)", go1_params, 4096, 100,
                  dict(streams_kw=dict(desync=2))),
    "go1_desync_8192": ("go1 N=20, de-synchronised fleet, per-rank batch of the 8-GPU config", go1_params, 8192, 60, dict(streams_kw=dict(desync=True))),
}

if __name__ == "__main__":
    if len(sys.argv) > 1:
        nm, maker, B, steps, kw = SHAPES[sys.argv[1]]
        # (profilers wrap this form: the in-order launches only, so that 'the last launches' of a trace are what the line is about)
        run(nm, maker, B, int(sys.argv[2]) if len(sys.argv) > 2 else steps, pipelined_leg=("desync" in sys.argv[1] or "mixed" in sys.argv[1]), **kw)
        sys.exit(0)
    run("go1 N=20 (bench line shape)", go1_params, 4096, 100)
    run(*SHAPES["go1_desync"][:4], **SHAPES["go1_desync"][4])
    run("go1 N=20, per-rank batch of the 8-GPU config", go1_params, 8192, 60)
    run("go1 N=20, the WHOLE batch of the 8-GPU config (65 536) on one GPU", go1_params, 65536, 20)
    run("cassie N=20", cassie_params, 4096, 100)
    run("pogox N=100", pogox_params, 1024, 60)
    run("go1 N=20 with osqp.polish (the node's declared default; k_mhe_solve_r3_4_n20_pol)", go1_params, 4096, 100, polish=1)
    run("pogox N=100 with osqp.polish", pogox_params, 1024, 40, polish=1)
    run("go1 KF mode (est_type 1: recursion instead of the QP)", go1_params, 4096, 200, est_type=1)
    run("go1 KF mode, batch 65536", go1_params, 65536, 100, est_type=1)
    run("go1 with foot-position states (leg_odom_type 1, 21-dim blocks)", go1_params, 4096, 40, leg_odom_type=1)
    run("go1 foot-position states, KF mode", go1_params, 4096, 100, leg_odom_type=1, est_type=1)
