"""bench.py's rank choreography (`run_bench`, `agree_on_gather`) on two, four and eight gloo ranks, on CPU.

The 8-GPU runs are the driver's to launch, so the code path `bench.py --gpus N` takes for N > 1 — unique-id
broadcast, the MIN-reduced pre-flight that decides which all-gather runs, the per-step exchange, the barrier-bracketed
timed region, the MAX-reduce of the elapsed time and the rank-0 JSON line — is executed here first, with a stand-in
estimator that lives in tests/bench_stand_in.py (the product has no such thing: bench.py's main() always builds the HIP estimator
and refuses to start without a GPU)."""
import argparse
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
from bench_stand_in import make_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, preflight_fails_on, slow_s=0.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    made = []
    env = make_env(rank, rank, world, preflight_fails_on=preflight_fails_on, made=made)
    args = argparse.Namespace(gpus=world, steps=7, warmup=5, batch=6, no_cpu_baseline=True, no_allgather=False, pipelined_leg=True)
    if slow_s:
        make0 = env.make_estimator

        def make_slow(p, B):
            e = make0(p, B)
            e.slow_rank_delay_s = slow_s
            return e
        env.make_estimator = make_slow
    line = bench.run_bench(args, env, rank, world)
    est = made[0]
    assert est.k == 64 + 7 - 1 and est.timed_steps == 7          # 59 fill + 5 warm-up + exactly 7 timed steps
    if preflight_fails_on is None:
        assert est.comm == "up" and est.gathers == 71 and est.layout_checked == 71
    else:
        assert est.comm is None and est.gathers == 0                 # nobody entered the collective init
    assert (line is None) == (rank != 0)
    if rank == 0:
        with open(os.path.join(out_dir, "line.json"), "w") as fh:
            json.dump(line, fh)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, preflight_fails_on, world=2, slow_s=0.0):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), preflight_fails_on, slow_s), nprocs=world, join=True)
    with open(tmp_path / "line.json") as fh:
        return json.load(fh)


def test_two_ranks_own_communicator(tmp_path):
    d = _run(tmp_path, None)
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 5 and d["window_fill_steps_before_warmup"] == 59
    assert d["config"]["allgather"].startswith("dekf_allgather_vb") and d["config"]["global_batch"] == 12
    assert abs(d["value"] - 2 * 6 * 7 / (d["ms_per_step"] * 7e-3)) / d["value"] < 1e-9
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-15
    assert rf["traffic"] is None and 0 < rf["flop_frac"] < 1 and rf["chain_floor_frac"] > 0
    assert "cpu_baseline" not in d and d["scaling"] == "weak" and d["vs_baseline"] is None
    # the line proves how many ranks the communicator connected, times the exchange, and carries the pipelined pass with ITS all-gather
    assert d["config"]["comm_world"] == 2 and d["config"]["comm_ranks_seen"] == 2
    assert d["allgather_ms_per_step"] == pytest.approx(0.02) and "allgather" not in d["kernel_ms_per_step"]
    wp = d["with_step_pipelining"]
    assert wp["value"] > 0 and wp["own_shard_round_trips_on_every_rank"] is True and wp["steps"] == 7


def test_preflight_failure_on_one_rank_moves_every_rank_to_the_fallback(tmp_path):
    d = _run(tmp_path, 1)
    assert d["config"]["allgather"] == "torch.distributed.all_gather_into_tensor"
    assert d["config"]["comm_world"] is None and d["config"]["comm_ranks_seen"] is None and "with_step_pipelining" not in d


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_ranks_layout_timing_and_preflight(tmp_path, world):
    """the 8-GPU configuration (BASELINE.json configs[3]) is the driver's to launch; its choreography runs here on gloo ranks:
    every rank ends up with the fleet's v_b as [world][B][3] (checked inside the stand-in at every step, on every rank), the line
    reports the MAX over ranks of the timed region (the last rank is made slow), value is the whole job's"""
    slow = 0.02
    d = _run(tmp_path, None, world=world, slow_s=slow)
    assert d["n_gpus"] == world and d["config"]["global_batch"] == 6 * world and d["config"]["allgather"].startswith("dekf_allgather_vb")
    assert d["ms_per_step"] >= 1e3 * slow                       # the slow rank's time, not rank 0's
    assert abs(d["value"] - world * 6 * 7 / (d["ms_per_step"] * 7e-3)) / d["value"] < 1e-9
    assert d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["config"]["comm_world"] == world and d["config"]["comm_ranks_seen"] == world
    assert d["with_step_pipelining"]["value"] > 0 and d["with_step_pipelining"]["own_shard_round_trips_on_every_rank"] is True


def test_preflight_failure_on_the_last_of_eight_ranks_moves_all_to_the_fallback(tmp_path):
    d = _run(tmp_path, 7, world=8)
    assert d["n_gpus"] == 8 and d["config"]["allgather"] == "torch.distributed.all_gather_into_tensor"


def test_flop_model_is_consistent():
    f = bench.algorithmic_flops(4, 20, 75, 2.0, 3.0)
    assert f["per_iteration"] == (3 * 20 - 2) * 162 + 20 * 4 * 102 + 38 * 177 + 38 * 105 + 9 * 20 * 8
    assert 2.0e6 < f["total"] < 4.0e6
    assert bench.algorithmic_flops(4, 20, 150, 2.0, 6.0)["total"] > 1.9 * f["total"] - 1.0e6


def _bench_cli(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    return r


def test_plain_gpus_2_self_launches_its_ranks_before_touching_a_device():
    """the driver calls `python bench.py --gpus N` plainly: main() must start N ranks itself (child torch.distributed.run)
    and rank 0 of the child must print the one JSON line — here with the gloo stand-in, end to end through main()"""
    r = _bench_cli(["--gpus", "2", "--steps", "4", "--warmup", "3", "--bench-env", "bench_stand_in:make_env"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 3 and d["stand_in"] is True
    # default shard at N > 1 is the per-rank share of configs[3]
    assert d["config"]["batch_per_gpu"] == 8192 and d["config"]["global_batch"] == 16384
    assert "8192 per GPU sharded across 2xMI355X" in d["config"]["workload"] and "per-rank share" in d["config"]["workload"]
    assert d["config"]["allgather"].startswith("dekf_allgather_vb")


def test_a_failing_rank_fails_the_self_launched_run():
    r = _bench_cli(["--gpus", "2", "--steps", "2", "--warmup", "1", "--bench-env", "bench_stand_in:no_such_factory"])
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_plain_run_without_a_gpu_refuses_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this box has a GPU")
    r = _bench_cli(["--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_workload_names_and_default_shards():
    assert bench.default_batch(1) == 4096 and bench.default_batch(2) == 8192 and bench.default_batch(8) == 8192
    assert bench.parse_args([]).batch == 4096 and bench.parse_args(["--gpus", "8"]).batch == 8192
    assert bench.parse_args(["--gpus", "8", "--batch", "512"]).batch == 512
    w1 = bench.workload_name(4096, 1, 20)
    assert "batch=4096" in w1 and "configs[1]" in w1 and "1xMI355X" in w1
    assert "configs" not in bench.workload_name(8192, 1, 20) and "batch=8192" in bench.workload_name(8192, 1, 20)
    w8 = bench.workload_name(8192, 8, 20)
    assert "batch=65536" in w8 and "configs[3]" in w8 and "8xMI355X" in w8
    assert "configs[3]" not in bench.workload_name(4096, 8, 20).replace("per-rank share of BASELINE.json configs[3]", "")
