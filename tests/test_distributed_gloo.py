"""N > 1 path on CPU: two gloo ranks each own a shard of the fleet (instances are independent, so
the only exchange is the all-gather of the fused base velocities).  What is checked here is what
bench.py and dekf_allgather_vb rely on: per-rank log generation (first_instance offsets) equals the
corresponding slice of a single global generation, and the gathered [world][B][3] layout equals the
single-process result.  The arithmetic on each rank is the hostsim build of the device cores (no GPU
in this container); the RCCL call itself is exercised by the driver's multi-GPU bench."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hostsim_lib as HS
from decentralized_ekf_mhe_amd import go1_params
from decentralized_ekf_mhe_amd.streams import make_streams

B_PER_RANK, K = 2, 26


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = go1_params()
    p.ekf_rate = p.rate
    s = make_streams(p, B_PER_RANK, K, first_instance=rank * B_PER_RANK)
    hs = HS.HostSim(p, B_PER_RANK)
    gathered = []
    for k in range(K):
        hs.feed(s, k)
        hs.step(k)
        vb = torch.from_numpy(hs.get()["v_b"].copy())
        allv = torch.empty((world * B_PER_RANK, 3), dtype=torch.float64)  # rank-major, like ncclAllGather
        dist.all_gather_into_tensor(allv, vb)
        gathered.append(allv.numpy().reshape(world, B_PER_RANK, 3).copy())
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction of bench.py
    assert t.item() == world
    if rank == 0:
        np.save(out_path, np.array(gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    world = 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    gathered = np.load(out)  # [K][world][B][3]
    p = go1_params()
    p.ekf_rate = p.rate
    s = make_streams(p, world * B_PER_RANK, K)
    hs = HS.HostSim(p, world * B_PER_RANK)
    for k in range(K):
        hs.feed(s, k)
        hs.step(k)
        ref = hs.get()["v_b"].reshape(world, B_PER_RANK, 3)
        assert np.array_equal(gathered[k], ref), k


def test_shard_generation_is_a_slice_of_the_global_one():
    p = go1_params()
    g = make_streams(p, 6, 12)
    for r in range(3):
        sh = make_streams(p, 2, 12, first_instance=2 * r)
        for key in ("accel", "gyro", "J", "qdot", "vo_dp", "vo_q", "imu_t", "contact"):
            assert np.array_equal(sh[key], g[key][:, 2 * r:2 * r + 2]), (r, key)
