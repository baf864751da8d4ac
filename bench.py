#!/usr/bin/env python3
"""bench.py — estimator-steps/s of the EKF + 20-step-MHE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one estimator-step for every robot instance of the batch: one EKF timer tick
(predict + accel-correct, VO rewind when a pose is due) + one full MHE update(T) (term
construction, VO bounds, marginalisation, Ruiz scaling, factorisation, OSQP-style ADMM to
eps 1e-6, extraction) — SURVEY.md §8(d).  Workload at N=1: BASELINE.json configs[1]
"Go1, batch=4096 synthetic IMU+encoder+vision streams, 20-step MHE, 1xMI355X".  Instances are
independent, so N GPUs run N shards of 4096 (weak scaling) and exchange only the fused
base-velocity estimates (one RCCL all-gather per step).  Sensor logs are synthetic
(decentralized_ekf_mhe_amd/streams.py) and resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# SURVEY.md §8(d) contract figure: algorithmic bytes per estimator-step, Go1, N = 20
B_ALG_GO1 = 5736
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/traffic_k_mhe_solve.json, produced by tools/collect_traffic.sh); None if absent"""
    path = os.path.join(ROOT, "profiles", "traffic_k_mhe_solve.json")
    try:
        with open(path) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except Exception:
        return None


def usable_cores():
    """cores this process may really use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(params, steps, seed_first):
    """the oracle (CPU restatement of the reference algorithm) timed on this host's cores: one thread
    (the reference's deployment: one pinned core per robot) and the best of a few thread counts"""
    import oracle_lib
    from decentralized_ekf_mhe_amd.streams import make_streams
    cores = usable_cores()
    one = make_streams(params, 2, steps, first_instance=seed_first)
    _, _, _, secs1 = oracle_lib.run_streams(params, one, nthreads=1)
    best = (2 * steps / secs1, 1, 2)
    for nt in sorted({min(cores, 8), min(cores, 32), cores}):
        if nt <= 1:
            continue
        inst = 2 * nt
        s = make_streams(params, inst, steps, first_instance=seed_first)
        _, _, _, secs = oracle_lib.run_streams(params, s, nthreads=nt)
        if inst * steps / secs > best[0]:
            best = (inst * steps / secs, nt, inst)
    return {"value": best[0], "unit": "estimator-steps/s", "cores": best[1], "kind": "port",
            "single_thread_value": 2 * steps / secs1, "usable_cores": cores,
            "sample": f"{best[2]} Go1 instances x {steps} steps (T=0..{steps - 1}, window fill included) of the same "
                      f"synthetic logs through oracle/liboracle.so (fp64 restatement of the Eigen+OSQP path) on "
                      f"{best[1]} threads (best of 8/32/all usable cores); single_thread_value = 2 instances on 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from decentralized_ekf_mhe_amd import go1_params
    from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, new_unique_id, streams_to_device
    from decentralized_ekf_mhe_amd.streams import make_streams

    p = go1_params()
    p.ekf_rate = p.rate  # one EKF tick per estimator-step, so its dt is the step
    B, W, K = args.batch, args.warmup, args.steps
    # The metric is quoted at steady state: full window AND visual-odometry intervals active (from about tick 40 on the
    # solves need 75 ADMM iterations instead of 50).  The default warm-up of 50 steps covers that; if the caller asks for
    # fewer, the missing ticks run as untimed SETUP in front of the W warm-up steps, so that the timed region is always
    # exactly K steps of the same steady state.
    STEADY_FROM = 50
    assert STEADY_FROM >= p.N + 1
    fill = max(0, STEADY_FROM - W)

    t_gen = time.time()
    s = make_streams(p, B, fill + W + K, first_instance=rank * B)
    sd = streams_to_device(s, device=f"cuda:{local_rank}")
    t_gen = time.time() - t_gen

    est = BatchedEstimator(p, B, device=local_rank)
    vb_all, gather_path = None, "none"
    if world > 1 and not args.no_allgather:
        # the estimator's own RCCL communicator (dekf_comm_init); should it fail to come up on some rank, every
        # rank falls back to torch.distributed's communicator for the same exchange, and the line says which ran
        uid, ok = None, 0
        if rank == 0:
            try:
                uid = new_unique_id()
            except Exception as e:  # noqa: BLE001
                print(f"[rank 0] dekf_comm_unique_id failed: {e}", file=sys.stderr)
        ids = [uid]
        dist.broadcast_object_list(ids, src=0)  # every rank reaches this, whatever happened on rank 0
        if ids[0] is not None:
            try:
                est.comm_init(world, rank, ids[0])
                ok = 1
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] dekf_comm_init failed: {e}", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_path = "dekf_allgather_vb (RCCL, second stream)" if int(flag.item()) == 1 else "torch.distributed.all_gather_into_tensor"
        vb_all = torch.empty((world, B, 3), dtype=torch.float64, device=f"cuda:{local_rank}")
    own_comm = gather_path.startswith("dekf")
    vb_mine = torch.empty((B, 3), dtype=torch.float64, device=f"cuda:{local_rank}") if (vb_all is not None and not own_comm) else None

    def run(k0, k1):
        for k in range(k0, k1):
            est.push_stream_step(sd, k)
            est.step(k)
            if vb_all is None:
                continue
            if own_comm:
                est.allgather_vb(vb_all)
            else:
                est.get_into(v_b=vb_mine)
                est.sync()
                dist.all_gather_into_tensor(vb_all.view(world * B, 3), vb_mine)

    run(0, fill)      # window fill that the requested warm-up does not cover (0 with the defaults)
    run(fill, fill + W)
    est.sync()
    est.timing_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(fill + W, fill + W + K)
    est.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tim = est.timing_read()
    est.timing_enable(False)

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = est.get()
    info = est.solver_info()
    solved = float((out["status"] == 1).mean())
    v_err = float(np.abs(out["x"][:, 3:6] - s["gt_v_s"][fill + W + K - 1]).max())

    if rank == 0:
        value = world * B * K / elapsed
        solve_ms, solve_n = tim["solve"]
        avg_solve_s = solve_ms / max(solve_n, 1) * 1e-3
        achieved = B_ALG_GO1 * B / avg_solve_s / 1e9
        line = {
            "metric": "estimator-steps/sec (EKF+MHE, 20-step window)",
            "value": value, "unit": "estimator-steps/s", "n_gpus": world, "steps": K, "warmup": W, "window_fill_steps_before_warmup": fill,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Go1, batch=4096 per GPU synthetic IMU+encoder+vision streams, 20-step MHE "
                                   "(BASELINE.json configs[1])",
                       "robot": "go1", "legs": 4, "N": int(p.N), "batch_per_gpu": B, "global_batch": world * B,
                       "eps": 1e-6, "parallelism": f"instances sharded x{world}, RCCL all-gather of v_b per step", "allgather": gather_path},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic() if (B == 4096 and world == 1) else None,
                         "kernel": "k_mhe_solve", "avg_launch_ms": avg_solve_s * 1e3, "launches": solve_n,
                         "alg_bytes_per_step": B_ALG_GO1, "units_per_launch": B},
            "kernel_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in tim.items()},
            "solver": {"mean_iters": float(info["iters"].mean()), "max_iters": int(info["iters"].max()),
                       "solved_frac": solved, "max_abs_v_err_vs_truth": v_err},
            "stream_gen_s": t_gen,
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N = 1 only (rank 0's host cores, same run)
            line["cpu_baseline"] = cpu_baseline(p, min(fill + W + K, 150), seed_first=0)
        print(json.dumps(line))
    est.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
