#!/usr/bin/env python3
"""bench.py — estimator-steps/s of the EKF + 20-step-MHE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: called plainly (no WORLD_SIZE in the environment), `main()` starts the N ranks itself as a
child `python -m torch.distributed.run` — before this process has imported torch or made any HIP call — and exits with
the child's code; rank 0 of the child prints the line.

One "step" = one estimator-step for every robot instance of the batch: one EKF timer tick
(predict + accel-correct, VO rewind when a pose is due) + one full MHE update(T) (term
construction, VO bounds, marginalisation, Ruiz scaling, factorisation, OSQP-style ADMM to
eps 1e-6, extraction) — SURVEY.md §8(d).  Workload at N=1: BASELINE.json configs[1]
"Go1, batch=4096 synthetic IMU+encoder+vision streams, 20-step MHE, 1xMI355X".  Instances are
independent (the reference runs one process per robot, go1_launch.py:18-25), so N GPUs run N
shards and exchange only the fused base-velocity estimates (one RCCL all-gather per step).  The
shard is 4096 instances at N = 1 (configs[1]) and 8192 at N > 1, so that N = 8 is configs[3]
"Go1, batch=65 536 sharded across 8xMI355X"; --batch overrides both.  Sensor logs are synthetic
(decentralized_ekf_mhe_amd/streams.py) and resident in HBM before the timed region.

The rank choreography (`run_bench`) takes everything that touches a device through a `BenchEnv`, so that
tests/test_bench_orchestration.py can run the very same code on two gloo ranks with a stand-in estimator; `main()`
always builds the real environment (HIP estimator through the C ABI, RCCL) and refuses to start without a GPU.
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
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_TFLOPS = 78.6  # MI355X_MICROARCH.md: fp64 vector peak (the path computes in fp64)
# dependent-issue floor of one 9x9 chain step: nine v_fmac_f64_dpp, each occupying its SIMD for 16 cycles
# (quarter-rate DPP f64), back to back on one wavefront (tools/probes/dpp_chain_probe.hip)
CHAIN_STEP_FLOOR_CYCLES = 9 * 16
TRAFFIC_FILE = os.path.join("profiles", "traffic_k_mhe_solve.json")  # refreshed by tools/final_profiles.sh for the kernel of this round


def measured_traffic(kernel, batch):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  The PMC passes cannot run inside
    this process (rocprofv3 wraps the command), so the figure is the committed result of tools/collect_traffic.sh
    (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, corrected as MI355X_MICROARCH.md prescribes).  It is only quoted
    when that file was collected for the very kernel this run launched (`kernel`: dekf_solve_kernel_name) with the same
    code (the file records a hash of csrc/) at the same batch; otherwise (None, why)."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as fh:
            d = json.load(fh)
    except Exception:
        return None, f"{TRAFFIC_FILE} absent"
    if d.get("kernel") != kernel or int(d.get("batch", -1)) != int(batch):
        return None, f"{TRAFFIC_FILE} is for {d.get('kernel')} at batch {d.get('batch')}, this run launched {kernel} at batch {batch}"
    if d.get("csrc_sha1") != csrc_sha1():
        return None, f"{TRAFFIC_FILE} was collected for another revision of csrc/ ({d.get('csrc_sha1')})"
    return d.get("hbm_bytes_per_launch"), f"{TRAFFIC_FILE} ({d.get('collected', 'rocprofv3 --pmc passes')})"


def csrc_sha1():
    """hash over the kernel sources: ties a committed PMC figure to the code it was measured on"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "decentralized_ekf_mhe_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_flops(L, N, iters, factorizations, checks, scaling_passes=10):
    """fp64 operations (FMA = 2) one estimator-step needs in the STRUCTURED algorithm as it is implemented (slack
    blocks eliminated, block-tridiagonal 9x9 system in the window states) — what roofline.flop_frac
    prices against the fp64 vector peak.  Counted per phase:
      solve    two-sided block substitution: (K-1) forward + K (g = S^-1 f) + (K-1) outward 9x9 mat-vecs
      rows     per 3-row block: A_x x (3..30), two slack-block applies (18 / 36 each), 17 per row of updates, w (9)
      x cols   9 K entries, ~8 each
      factor   slack-block inverses, T_kk / C_k assembly (9 K columns x ~400), per block: Schur update 2.9 k,
               Gauss-Jordan 1.5 k, W = C S^-1 1.5 k
      Ruiz     per pass ~ (rows + columns) x 40;   residual check ~ 1.3 x one row phase
      assemble + marginalise (21/24-dim Gauss-Jordan ~ 2 dim^3) + EKF tick"""
    K = N
    mv9 = 2 * 81
    solve = (3 * K - 2) * mv9
    rows = K * L * 102 + 2 * (K - 1) * 177 + 2 * (K - 1) * 105
    xcols = 9 * K * 8
    per_iter = solve + rows + xcols
    factor = (K * L * 60 + (K - 1) * 420) + 9 * K * 400 + K * (2900 + 1500 + 1500)
    ruiz = scaling_passes * (K * (3 * L + 12) + K * (21 + 3 * L)) * 40 // 3
    check = int(1.3 * rows) + 9 * K * 30
    dim = 12 + 3 * L
    assemble = 2 * dim ** 3 + 2 * 9 ** 3 + 4000 + 1200
    return {"per_iteration": per_iter, "total": iters * per_iter + factorizations * factor + ruiz + checks * check + assemble}


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


def cpu_baseline(params, total_steps, seed_first, gpu_x_final=None):
    """The checker leg.  The oracle (CPU restatement of the reference algorithm, oracle/) is timed on this host's
    cores — one thread (the reference's deployment: one pinned core per robot) and all usable cores — and, because
    its multi-thread sample is the first instances of the very logs the GPU just ran, it also yields the error of the
    timed batch's final states against the oracle (`max_rel_err_vs_oracle`)."""
    import oracle_lib
    from decentralized_ekf_mhe_amd.streams import make_streams
    cores = usable_cores()
    short = min(total_steps, 150)
    one = make_streams(params, 2, short, first_instance=seed_first)
    _, _, _, secs1 = oracle_lib.run_streams(params, one, nthreads=1)
    single = 2 * short / secs1
    nt = max(1, min(cores, 32))
    inst = max(16, nt)
    s = make_streams(params, inst, total_steps, first_instance=seed_first)
    x_ref, _, _, secs = oracle_lib.run_streams(params, s, nthreads=nt)
    multi = inst * total_steps / secs
    best = (multi, nt, inst, total_steps) if multi >= single else (single, 1, 2, short)
    out = {"value": best[0], "unit": "estimator-steps/s", "cores": best[1], "kind": "port",
           "single_thread_value": single, "usable_cores": cores,
           "sample": f"{best[2]} Go1 instances x {best[3]} steps (T = 0.., window fill included) of the same synthetic "
                     f"logs through oracle/liboracle.so on {best[1]} thread(s); single_thread_value = 2 instances x "
                     f"{short} steps on 1 thread.  The port is an fp64 restatement of the Eigen+OSQP path that keeps "
                     f"H and A as DENSE matrices, re-slices them at every marginalisation and factors a generic sparse "
                     f"KKT matrix per step: slower than Eigen-sparse + QDLDL would be, stated as a baseline only"}
    # A CPU build that EXPLOITS THE STRUCTURE, beside the port: the device cores themselves (same block-tridiagonal algorithm, same
    # code: decentralized_ekf_mhe_amd/csrc/*_core.h) compiled lane-sequentially for the host with g++ -O3 -march=native
    # (tests/hostsim: test infrastructure, never a fallback of the product) on the same logs — what a CPU implementation that never
    # materialises the QP costs, as opposed to the Eigen + OSQP-shaped port above.
    try:
        import hostsim_lib
        t_b = time.time()
        hostsim_lib.build(fast=True)
        t_b = time.time() - t_b
        _, s1 = hostsim_lib.run_streams_timed(params, one, nthreads=1)
        inst_s = 4 * nt
        ss = make_streams(params, inst_s, total_steps, first_instance=seed_first)
        xs, sm = hostsim_lib.run_streams_timed(params, ss, nthreads=nt)
        n_cmp = min(inst, inst_s)
        out["structured"] = {"value": inst_s * total_steps / sm, "unit": "estimator-steps/s", "cores": nt, "single_thread_value": 2 * short / s1,
                             "kind": "the product's own cores (block-tridiagonal ADMM, QP never materialised) built lane-sequentially for the host: "
                                     "g++ -O3 -march=native, tests/hostsim",
                             "sample": f"{inst_s} Go1 instances x {total_steps} steps on {nt} thread(s); single_thread_value = 2 instances x {short} steps",
                             "max_abs_diff_vs_port": float(np.abs(xs[:n_cmp] - x_ref[total_steps - 1, :n_cmp]).max()), "build_s": round(t_b, 1)}
    except Exception as e:  # noqa: BLE001  (an extra beside the contract's figure: it must not cost the line)
        out["structured"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
    if gpu_x_final is not None:
        n = min(16, inst, len(gpu_x_final))
        ref, got = x_ref[total_steps - 1, :n], gpu_x_final[:n]
        rel, norm = 0.0, 0.0
        for blk in (slice(0, 3), slice(3, 6), slice(6, 9)):
            num = np.abs(got[:, blk] - ref[:, blk]).max(axis=-1)
            den = np.abs(ref[:, blk]).max(axis=-1)
            rel = max(rel, float((num / np.maximum(den, 1e-12))[den > 1e-3].max(initial=0.0)))
            norm = max(norm, float((num / (1e-4 * den + 1e-6)).max()))
        out["error_vs_oracle"] = {"instances": n, "at_step": total_steps - 1, "max_rel_err": rel,
                                  "max_err_over_tolerance": norm,
                                  "tolerance": "|x - oracle|_inf <= 1e-4 |oracle|_inf + 1e-6 per p / v / bias block"}
    return out


def pipelined_leg(env, p, B, sd, total, K):
    """The same K steps with dekf_params.solve_pipeline = 1 (the solve of step T on a second stream out
    of its own set of buffers, so that the pushes, the EKF tick, the term construction and the solve of step T + 1 start under its
    last round; results bit-identical, tests/test_gpu_configs.py).  Reported NEXT TO `value`, never as it: consecutive solve
    launches overlap in this mode, so a launch duration — what `roofline` is priced on — is no longer the cost of a launch, and
    a caller that needs step T's estimate before it can produce step T + 1's samples (closed loop) cannot use it."""
    p2 = p.copy()
    p2.solve_pipeline = 1
    est = env.make_estimator(p2, B)
    try:
        for k in range(total - K):
            est.push_stream_step(sd, k)
            est.step(k)
        est.sync()
        env.device_sync()
        t0 = time.perf_counter()
        for k in range(total - K, total):
            est.push_stream_step(sd, k)
            est.step(k)
        est.sync()
        env.device_sync()
        dt = time.perf_counter() - t0
        solved = float((est.get()["status"] == 1).mean())
    finally:
        est.close()
    return {"value": B * K / dt, "unit": "estimator-steps/s", "ms_per_step": dt / K * 1e3, "steps": K, "solved_frac": solved,
            "what": "dekf_params.solve_pipeline = 1: consecutive steps overlap on two streams (same results bit for bit); "
                    "open-loop replay only, so it is reported beside `value`, not as it"}


def runtime_versions():
    out = {}
    try:
        import torch
        out["torch"] = torch.__version__
        out["hip"] = getattr(torch.version, "hip", None)
    except Exception:  # noqa: BLE001
        pass
    try:
        out["rocm"] = open("/opt/rocm/.info/version").read().strip()
    except Exception:  # noqa: BLE001
        out["rocm"] = None
    try:
        from decentralized_ekf_mhe_amd import capi
        out["hip_runtime_version"] = int(capi.load().dekf_hip_runtime_version())
    except Exception:  # noqa: BLE001
        pass
    return out


def pipelined_leg_multi(env, dist, p, B, sd, total, K, world, rank):
    """pipelined_leg on every rank of a multi-GPU run, with the per-step all-gather of the pipelined handle; MAX over ranks"""
    import torch
    p2 = p.copy()
    p2.solve_pipeline = 1
    out = {"value": None, "unit": "estimator-steps/s", "steps": K,
           "what": "dekf_params.solve_pipeline = 1 on every rank, with the per-step RCCL all-gather of the pipelined handle"}
    est = None
    ok = 1
    try:
        est = env.make_estimator(p2, B)
        uid = [env.new_unique_id() if rank == 0 else None]
    except Exception as e:  # noqa: BLE001
        ok, uid = 0, [None]
        out["error"] = f"{type(e).__name__}: {e}"
    dist.broadcast_object_list(uid, src=0)
    flag = torch.tensor([ok if uid[0] is not None else 0], dtype=torch.int32, device=env.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # every rank reaches the collective below, or none does
    if int(flag.item()) != 1:
        if est is not None:
            est.close()
        out.setdefault("error", "a rank could not create its pipelined handle")
        return out
    try:
        est.comm_init(world, rank, uid[0])
        vb_all = torch.empty((2, world, B, 3), dtype=torch.float64, device=env.device)   # two buffers: the exchange of step T is still in flight at T + 1
        for k in range(total - K):
            est.push_stream_step(sd, k); est.step(k); est.allgather_vb(vb_all[k & 1])
        est.sync(); env.device_sync(); dist.barrier()
        t0 = time.perf_counter()
        for k in range(total - K, total):
            est.push_stream_step(sd, k); est.step(k); est.allgather_vb(vb_all[k & 1])
        est.sync(); env.device_sync(); dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=env.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = est.get()["v_b"]
        last = (total - 1) & 1
        good = torch.tensor([1 if np.array_equal(vb_all[last, rank].cpu().numpy(), mine) else 0], dtype=torch.int32, device=env.device)
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        out.update(value=world * B * K / dt, ms_per_step=dt / K * 1e3, own_shard_round_trips_on_every_rank=bool(int(good.item())))
    finally:
        est.close()
    return out


def default_batch(gpus):
    """instances per GPU when --batch is not given: configs[1] at one GPU; at N > 1 the per-rank share of configs[3]
    (65 536 over 8 GPUs), so that the driver's plain `bench.py --gpus 8` IS that configuration"""
    return 4096 if gpus == 1 else 8192


def workload_name(B, world, N):
    """config.workload from what actually runs; names the BASELINE.json configuration when it is exactly one"""
    base = f"Go1, batch={world * B} synthetic IMU+encoder+vision streams, {N}-step MHE"
    if world == 1:
        tag = " (BASELINE.json configs[1])" if (B == 4096 and N == 20) else ""
        return f"{base}, 1xMI355X{tag}"
    tag = ""
    if world * B == 65536 and world == 8 and N == 20:
        tag = " (BASELINE.json configs[3])"
    elif B == 8192 and N == 20:
        tag = " (the per-rank share of BASELINE.json configs[3] on every rank)"
    return f"{base}, {B} per GPU sharded across {world}xMI355X with RCCL all-gather over xGMI{tag}"


class BenchEnv:
    """what run_bench needs from the outside: the device, the collective backend and the estimator factory"""

    def __init__(self, device, backend, make_estimator, to_device, new_unique_id, preflight, device_sync, real=True):
        self.device, self.backend = device, backend
        self.make_estimator, self.to_device, self.new_unique_id = make_estimator, to_device, new_unique_id
        self.preflight, self.device_sync, self.real = preflight, device_sync, real


def agree_on_gather(env, dist, est, world, rank):
    """Which all-gather runs, decided TOGETHER.  ncclCommInitRank is a blocking collective: a rank that skipped it
    (library missing, no unique id) would leave the others hanging inside it.  So the ranks first agree (MIN-reduce
    of a pre-flight flag: RCCL loadable here, unique id received) whether to try the estimator's own communicator at
    all; only then does every rank call dekf_comm_init, and a failure THERE is fatal and loud on the failing rank.
    If the pre-flight fails anywhere, all ranks use torch.distributed's communicator for the same exchange."""
    import torch
    uid = None
    if rank == 0:
        try:
            uid = env.new_unique_id()
        except Exception as e:  # noqa: BLE001
            print(f"[rank 0] dekf_comm_unique_id failed: {e}", file=sys.stderr)
    ids = [uid]
    dist.broadcast_object_list(ids, src=0)  # every rank reaches this, whatever happened on rank 0
    ok = 1 if (ids[0] is not None and env.preflight()) else 0
    flag = torch.tensor([ok], dtype=torch.int32, device=env.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        return "torch.distributed.all_gather_into_tensor"
    est.comm_init(world, rank, ids[0])  # collective; raises on failure (no silent second path from here on)
    return "dekf_allgather_vb (RCCL, second stream)"


def run_bench(args, env, rank, world):
    """everything between process-group creation and the JSON line; returns the line (rank 0) or None"""
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
    from decentralized_ekf_mhe_amd import go1_params
    from decentralized_ekf_mhe_amd.streams import make_streams

    p = go1_params()
    p.ekf_rate = p.rate  # one EKF tick per estimator-step, so its dt is the step
    if getattr(args, "pipeline", False):
        p.solve_pipeline = 1
    B, W, K = args.batch, args.warmup, args.steps
    # The metric is quoted at steady state: full window AND visual-odometry intervals active.  The first vision intervals enter the
    # window around tick 40 and the solves settle at 75 ADMM iterations only by tick 57 (ticks 50-56 still average 83 … 75 iterations
    # and 2.3 … 1.9 ms per launch against 1.85 ms afterwards: tools/probes/per_tick_solve_time.py, profiles/r05_per_tick_solve_time.txt
    # — a 20-step timed region that started at tick 50 measured that transient, 3.6 % low).  The ticks the requested warm-up does not
    # cover run as untimed SETUP in front of the W warm-up steps, so that the timed region is always exactly K steps of the same
    # steady state whatever W and K are.
    STEADY_FROM = 64
    assert STEADY_FROM >= p.N + 1
    fill = max(0, STEADY_FROM - W)
    total = fill + W + K

    t_gen = time.time()
    s = make_streams(p, B, total, first_instance=rank * B)
    sd = env.to_device(s)
    t_gen = time.time() - t_gen

    est = env.make_estimator(p, B)
    vb_all, gather_path = None, "none"
    if world > 1 and not args.no_allgather:
        gather_path = agree_on_gather(env, dist, est, world, rank)
        vb_all = torch.empty((world, B, 3), dtype=torch.float64, device=env.device)
    own_comm = gather_path.startswith("dekf")
    vb_mine = torch.empty((B, 3), dtype=torch.float64, device=env.device) if (vb_all is not None and not own_comm) else None
    # What the COMMUNICATOR says (not what this script passed in): ncclCommCount / ncclCommUserRank, and one exchange of the rank
    # numbers through the handle's own communicator and stream, checked slot by slot.  A line with comm_world == n_gpus and
    # comm_ranks_seen == n_gpus proves that RCCL connected that many ranks.
    comm_world = comm_rank = ranks_seen = None
    if own_comm:
        # (never executed on more than one GPU before the driver's first multi-GPU run: whatever goes wrong HERE must not cost the line —
        # the fields then say what happened instead of a number.  A mismatch against the launcher's world / rank is reported, not asserted.)
        try:
            comm_world, comm_rank = est.comm_info()
            if (comm_world, comm_rank) != (world, rank):
                comm_world = f"MISMATCH: communicator says world {comm_world} rank {comm_rank}, launcher says {world} / {rank}"
            ranks_seen = est.comm_ranks_seen()   # collective
        except Exception as e:  # noqa: BLE001
            ranks_seen = f"{type(e).__name__}: {e}"

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
    # Kernel times by HIP events on the handle's stream.  An event pair costs the stream about 7 us (measured: 1.955 against 1.934 ms
    # per step with six pairs on / off), so the TIMED region brackets only the dominant kernel — the MHE solve, what roofline.achieved
    # is priced on — and the small kernels (EKF tick, term construction) are timed over the last warm-up steps instead.
    w_all = min(W, 20)
    run(fill, fill + W - w_all)
    est.sync()
    est.timing_enable(1)
    run(fill + W - w_all, fill + W)
    est.sync()
    tim_warm = est.timing_read()
    est.timing_enable(2)
    if world > 1:
        dist.barrier()
    env.device_sync()
    t0 = time.perf_counter()
    run(fill + W, total)
    est.sync()
    env.device_sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tim = est.timing_read()
    est.timing_enable(False)
    for cls in ("ekf", "assemble"):   # (not bracketed in the timed region: the warm-up's figures)
        if tim[cls][1] == 0:
            tim[cls] = tim_warm[cls]

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=env.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = est.get()
    info = est.solver_info()
    solved = float((out["status"] == 1).mean())
    v_err = float(np.abs(out["x"][:, 3:6] - s["gt_v_s"][total - 1]).max())

    line = None
    if rank == 0:
        value = world * B * K / elapsed
        solve_ms, solve_n = tim["solve"]
        avg_solve_s = solve_ms / max(solve_n, 1) * 1e-3
        achieved = B_ALG_GO1 * B / avg_solve_s / 1e9
        mean_iters = float(info["iters"].mean())
        mean_refactor = float(info["rho_updates"].mean())
        ct = max(int(p.check_termination), 1)
        fl = algorithmic_flops(int(p.num_legs), int(p.N), mean_iters, 1.0 + mean_refactor, mean_iters / ct, int(p.scaling_iters))
        per_gpu_rate = B * K / elapsed
        # cycles one workgroup spends per solve: a launch runs B / grid solves back to back on each persistent workgroup
        li = est.launch_info()
        cycles_per_solve = avg_solve_s * li["clock_hz"] / max(1.0, float(np.ceil(B / max(li["solve_workgroups"], 1))))
        # dependent mat-vec steps of one block-tridiagonal solve: (N - 2) / 2 forward + 1 joint middle + (N - 2) / 2 outward
        # on each side of the two-sided solve, the two sides in lock step
        chain_floor = mean_iters * (int(p.N) - 1) * CHAIN_STEP_FLOOR_CYCLES
        kernel = est.solve_kernel_name(True)
        traffic, traffic_src = measured_traffic(kernel, B) if (world == 1 and env.real) else (None, None)
        line = {
            "metric": "estimator-steps/sec (EKF+MHE, 20-step window)",
            "value": value, "unit": "estimator-steps/s", "n_gpus": world, "steps": K, "warmup": W, "window_fill_steps_before_warmup": fill,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_name(B, world, int(p.N)),
                       "robot": "go1", "legs": 4, "N": int(p.N), "batch_per_gpu": B, "global_batch": world * B,
                       "eps": 1e-6, "parallelism": f"instances sharded x{world}, RCCL all-gather of v_b per step", "allgather": gather_path,
                       "comm_world": comm_world, "comm_ranks_seen": ranks_seen},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_rate_gbs": (traffic / avg_solve_s / 1e9) if traffic else None,
                         "traffic_over_algorithmic": (traffic / (B_ALG_GO1 * B)) if traffic else None,
                         "kernel": kernel, "avg_launch_ms": avg_solve_s * 1e3, "launches": solve_n,
                         "alg_bytes_per_step": B_ALG_GO1, "units_per_launch": B,
                         # the contract's roofline is the HBM one; what actually limits this kernel is the chain of
                         # dependent mat-vec steps inside every ADMM iteration, so both honest fractions ride along:
                         "limiter": "the CU's throughput on a dependent chain (ADMM iterations x dependent 9x9 mat-vec steps; three resident solves saturate a CU: "
                                    "0.278 / 0.352 / 0.439 ms per solve with 0 / 2 / 3 neighbours, DESIGN.md section 4.6), not HBM",
                         # a rocprofv3 --stats table of this command also lists k_mhe_marginalize_early with a long wall time: it is
                         # background work on a second stream at the least priority (the next step's arrival cost, ~0.05 ms of machine
                         # time) whose workgroups wait for the slots this kernel frees in its last round
                         "concurrent_background_kernel": "k_mhe_marginalize_early (second stream, least priority; wall time = waiting for slots)",
                         "flops_per_step": fl["total"], "flops_per_admm_iteration": fl["per_iteration"],
                         "flop_peak_tflops": FP64_VECTOR_TFLOPS,
                         "flop_frac": fl["total"] * per_gpu_rate / (FP64_VECTOR_TFLOPS * 1e12),
                         "solve_workgroups": li["solve_workgroups"], "workgroups_per_cu": li["solve_workgroups"] / max(li["compute_units"], 1),
                         "chain_floor_cycles_per_solve": chain_floor, "measured_cycles_per_solve": cycles_per_solve,
                         "chain_floor_frac": chain_floor / cycles_per_solve if cycles_per_solve > 0 else None},
            "kernel_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in tim.items() if k != "allgather"},
            # the exchange itself, event-timed on the handle's communication stream over the timed steps (it overlaps the next step)
            "allgather_ms_per_step": (tim["allgather"][0] / tim["allgather"][1]) if tim.get("allgather", (0, 0))[1] else None,
            "solver": {"mean_iters": mean_iters, "max_iters": int(info["iters"].max()),
                       "mean_refactorisations": mean_refactor,
                       "solved_frac": solved, "max_abs_v_err_vs_truth": v_err},
            "stream_gen_s": t_gen,
            # the overlap of the look-ahead arrival cost (and of solve_pipeline = 1) rests on HIP runtime behaviour (dekf_capi.hip): which
            # runtime produced this line (guarded by tests/test_gpu_configs.py::test_the_overlap_the_throughput_rests_on_is_still_there)
            "runtime": runtime_versions() if env.real else None,
        }
        if not env.real:
            line["stand_in"] = True  # tests only: no estimator ran, nothing in this line is a measurement
        if not args.no_cpu_baseline and world == 1 and env.real:  # the CPU leg runs at N = 1 only (rank 0's host cores, same run)
            cb = cpu_baseline(p, total, seed_first=0, gpu_x_final=out["x"])
            err = cb.pop("error_vs_oracle", None)
            line["cpu_baseline"] = cb
            if err:
                line["solver"]["max_rel_err_vs_oracle"] = err["max_rel_err"]
                line["solver"]["error_vs_oracle"] = err
    est.close()
    if line is not None and world == 1 and env.real and not getattr(args, "pipeline", False) and getattr(args, "pipelined_leg", False):
        try:   # an extra beside the contract's line: whatever goes wrong here must not cost the line itself
            line["with_step_pipelining"] = pipelined_leg(env, p, B, sd, total, K)
        except Exception as e:  # noqa: BLE001
            line["with_step_pipelining"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
    if world > 1 and own_comm and not getattr(args, "pipeline", False) and getattr(args, "pipelined_leg", False):
        # N > 1 with --pipelined-leg: the same K steps once more with solve_pipeline = 1 AND its all-gather (one event per output set: the
        # exchange of step T waits for the solve that produced v_b on the communication stream only) — every rank takes part, rank 0
        # reports it beside `value`.  This is the only place the pipelined all-gather meets more than one rank.  Opt-in like the N = 1
        # leg: a second communicator and a second pass have never run on more than one GPU, and the plain command the driver times must
        # not be able to hang in an extra.
        res = pipelined_leg_multi(env, dist, p, B, sd, total, K, world, rank)
        if line is not None:
            line["with_step_pipelining"] = res
    return line


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(args, argv):
    """`bench.py --gpus N` called plainly for N > 1: one process per GPU is started HERE, as a child
    `python -m torch.distributed.run`, and this process only waits for it.  Nothing in this process has imported torch
    or touched HIP at this point (a process that has initialised the GPU must never exec or fork workers), and the
    children are fresh interpreters.  Rank 0 of the child prints the JSON line on the inherited stdout."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default 4096 at --gpus 1, else 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--pipelined-leg", action="store_true", help="after the line's own K steps, run them once more with solve_pipeline = 1 "
                    "and report that as with_step_pipelining beside value (at N > 1: on every rank, with the pipelined handle's all-gather).  Off by default: the extra launches of the same "
                    "kernel overlap each other, and a rocprofv3 summary of the default command is to hold the in-order launches only")
    ap.add_argument("--no-pipelined-leg", action="store_true", help=argparse.SUPPRESS)  # (accepted, the default)
    ap.add_argument("--pipeline", action="store_true", help="dekf_params.solve_pipeline = 1: consecutive steps overlap (A/B; the launch "
                    "durations the roofline is priced on then overlap too, so the default keeps the steps in order)")
    # test hook (tests/test_bench_orchestration.py): "module:function" returning a BenchEnv for (rank, local_rank, world);
    # such a line is marked "stand_in": true and is never a measurement
    ap.add_argument("--bench-env", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.batch is None:
        args.batch = default_batch(args.gpus)
    return args


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")

    if args.bench_env:
        import importlib
        mod, fn = args.bench_env.split(":")
        env = getattr(importlib.import_module(mod), fn)(rank, local_rank, world)
        assert not env.real
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(env.backend)
        line = run_bench(args, env, rank, world)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    assert local_rank < torch.cuda.device_count(), f"rank {rank}: no GPU {local_rank} on this node"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from decentralized_ekf_mhe_amd import capi
    from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, new_unique_id, streams_to_device

    def preflight():
        try:
            capi.load()
            new_unique_id()  # RCCL reachable through dlopen in this process
            return True
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] RCCL pre-flight failed: {e}", file=sys.stderr)
            return False

    dev = torch.device("cuda", local_rank)
    env = BenchEnv(device=dev, backend="nccl",
                   make_estimator=lambda p, B: BatchedEstimator(p, B, device=local_rank),
                   to_device=lambda s: streams_to_device(s, device=f"cuda:{local_rank}"),
                   new_unique_id=new_unique_id, preflight=preflight, device_sync=torch.cuda.synchronize)
    line = run_bench(args, env, rank, world)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
