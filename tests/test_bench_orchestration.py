"""bench.py's rank choreography (`run_bench`, `agree_on_gather`) on two gloo ranks, on CPU.

The 8-GPU runs are the driver's to launch, so the code path `bench.py --gpus N` takes for N > 1 — unique-id
broadcast, the MIN-reduced pre-flight that decides which all-gather runs, the per-step exchange, the barrier-bracketed
timed region, the MAX-reduce of the elapsed time and the rank-0 JSON line — is executed here first, with a stand-in
estimator that lives in THIS file (the product has no such thing: bench.py's main() always builds the HIP estimator
and refuses to start without a GPU)."""
import argparse
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench


class StandInEstimator:
    """records what run_bench asks of an estimator; v_b of step k is a known function of (rank, instance, k)"""

    def __init__(self, p, B, rank, world, comm_fails=False):
        self.B, self.rank, self.world, self.k = B, rank, world, -1
        self.comm_fails, self.comm, self.timing = comm_fails, None, False
        self.timed_steps, self.gathers = 0, 0

    def vb(self):
        inst = np.arange(self.B)[:, None] + self.rank * self.B
        return inst * 1000.0 + self.k + np.arange(3)[None, :] * 0.25

    def push_stream_step(self, sd, k):
        assert sd["tag"] == "device-streams"

    def step(self, k):
        assert k == self.k + 1
        self.k = k
        self.timed_steps += self.timing

    def sync(self):
        pass

    def comm_init(self, world, rank, uid):
        assert (world, rank) == (self.world, self.rank) and uid == b"stand-in-id"
        if self.comm_fails:
            raise RuntimeError("stand-in communicator refused")
        self.comm = "up"

    def allgather_vb(self, out):
        assert self.comm == "up"
        dist.all_gather_into_tensor(out.view(self.world * self.B, 3), torch.from_numpy(self.vb()))
        self.gathers += 1

    def get_into(self, v_b=None):
        v_b.copy_(torch.from_numpy(self.vb()))

    def timing_enable(self, on):
        self.timing = bool(on)

    def timing_read(self):
        return {"ekf": (0.01 * self.timed_steps, self.timed_steps), "assemble": (0.07 * self.timed_steps, self.timed_steps),
                "solve": (3.5 * self.timed_steps, self.timed_steps)}

    def launch_info(self):
        return dict(solve_workgroups=512, compute_units=256, clock_hz=2.4e9)

    def get(self):
        x = np.zeros((self.B, 9))
        x[:, 3] = 0.5
        return dict(x=x, v_b=self.vb(), status=np.ones(self.B, np.int32))

    def solver_info(self):
        return dict(iters=np.full(self.B, 75, np.int32), rho_updates=np.ones(self.B, np.int32))

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, preflight_fails_on):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    made = []

    def make(p, B):
        made.append(StandInEstimator(p, B, rank, world))
        return made[-1]

    env = bench.BenchEnv(device=torch.device("cpu"), backend="gloo", make_estimator=make,
                         to_device=lambda s: {"tag": "device-streams"}, new_unique_id=lambda: b"stand-in-id",
                         preflight=lambda: rank != preflight_fails_on, device_sync=lambda: None, real=False)
    args = argparse.Namespace(gpus=world, steps=7, warmup=5, batch=6, no_cpu_baseline=True, no_allgather=False)
    line = bench.run_bench(args, env, rank, world)
    est = made[0]
    assert est.k == 50 + 7 - 1 and est.timed_steps == 7          # 45 fill + 5 warm-up + exactly 7 timed steps
    if preflight_fails_on is None:
        assert est.comm == "up" and est.gathers == 57
    else:
        assert est.comm is None and est.gathers == 0                 # nobody entered the collective init
    assert (line is None) == (rank != 0)
    if rank == 0:
        with open(os.path.join(out_dir, "line.json"), "w") as fh:
            json.dump(line, fh)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, preflight_fails_on):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), preflight_fails_on), nprocs=world, join=True)
    with open(tmp_path / "line.json") as fh:
        return json.load(fh)


def test_two_ranks_own_communicator(tmp_path):
    d = _run(tmp_path, None)
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 5 and d["window_fill_steps_before_warmup"] == 45
    assert d["config"]["allgather"].startswith("dekf_allgather_vb") and d["config"]["global_batch"] == 12
    assert abs(d["value"] - 2 * 6 * 7 / (d["ms_per_step"] * 7e-3)) / d["value"] < 1e-9
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-15
    assert rf["traffic"] is None and 0 < rf["flop_frac"] < 1 and rf["chain_floor_frac"] > 0
    assert "cpu_baseline" not in d and d["scaling"] == "weak" and d["vs_baseline"] is None


def test_preflight_failure_on_one_rank_moves_every_rank_to_the_fallback(tmp_path):
    d = _run(tmp_path, 1)
    assert d["config"]["allgather"] == "torch.distributed.all_gather_into_tensor"


def test_flop_model_is_consistent():
    f = bench.algorithmic_flops(4, 20, 75, 2.0, 3.0)
    assert f["per_iteration"] == (3 * 20 - 2) * 162 + 20 * 4 * 102 + 38 * 177 + 38 * 105 + 9 * 20 * 8
    assert 2.0e6 < f["total"] < 4.0e6
    assert bench.algorithmic_flops(4, 20, 150, 2.0, 6.0)["total"] > 1.9 * f["total"] - 1.0e6
