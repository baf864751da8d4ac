"""Stand-in estimator for bench.py's rank choreography on CPU ranks (gloo).  TEST infrastructure: it records what
`bench.run_bench` asks of an estimator and returns known values; the product has no such thing (bench.py's real
environment builds the HIP estimator and refuses to start without a GPU).  `make_env` is what
`bench.py --bench-env bench_stand_in:make_env` loads in tests/test_bench_orchestration.py."""
import numpy as np
import torch
import torch.distributed as dist


class StandInEstimator:
    """records what run_bench asks of an estimator; v_b of step k is a known function of (rank, instance, k)"""

    def __init__(self, p, B, rank, world, comm_fails=False):
        self.B, self.rank, self.world, self.k = B, rank, world, -1
        self.comm_fails, self.comm, self.timing = comm_fails, None, False
        self.timed_steps, self.gathers, self.layout_checked = 0, 0, 0
        self.timing_mode_seen = None
        self.slow_rank_delay_s = 0.0  # tests: the LAST rank sleeps this long per timed step, so that the MAX over ranks is what the line reports

    def vb(self, rank=None):
        inst = np.arange(self.B)[:, None] + (self.rank if rank is None else rank) * self.B
        return inst * 1000.0 + self.k + np.arange(3)[None, :] * 0.25

    def push_stream_step(self, sd, k):
        assert sd["tag"] == "device-streams"

    def step(self, k):
        assert k == self.k + 1
        self.k = k
        self.timed_steps += self.timing
        if self.timing and self.slow_rank_delay_s and self.rank == self.world - 1:
            import time
            time.sleep(self.slow_rank_delay_s)

    def sync(self):
        pass

    def comm_init(self, world, rank, uid):
        assert (world, rank) == (self.world, self.rank) and uid == b"stand-in-id"
        if self.comm_fails:
            raise RuntimeError("stand-in communicator refused")
        self.comm = "up"

    def comm_info(self):
        assert self.comm == "up"
        return self.world, self.rank

    def comm_ranks_seen(self):
        got = [None] * self.world
        dist.all_gather_object(got, self.rank)     # (collective, like the real one)
        return len({r for i, r in enumerate(got) if r == i})

    def allgather_vb(self, out):
        assert self.comm == "up"
        dist.all_gather_into_tensor(out.view(self.world * self.B, 3), torch.from_numpy(self.vb()))
        self.gathers += 1
        # the fleet's estimates as every rank must hold them: [world][B][3], rank r's shard at out[r]
        assert tuple(out.shape) == (self.world, self.B, 3)
        for r in range(self.world):
            assert np.array_equal(out[r].numpy(), self.vb(r)), (self.rank, r, self.k)
        self.layout_checked += 1

    def get_into(self, v_b=None):
        v_b.copy_(torch.from_numpy(self.vb()))

    def timing_enable(self, on):
        self.timing = int(on) == 2   # bench.py brackets the TIMED region with mode 2 (solve launches only); mode 1 is the warm-up's
        self.timing_mode_seen = int(on)

    def timing_read(self):
        return {"ekf": (0.01 * self.timed_steps, self.timed_steps), "assemble": (0.07 * self.timed_steps, self.timed_steps),
                "solve": (3.5 * self.timed_steps, self.timed_steps), "allgather": (0.02 * self.gathers, self.gathers)}

    def solve_kernel_name(self, full_window=True):
        return "stand-in"

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


def make_env(rank, local_rank, world, preflight_fails_on=None, made=None):
    import bench

    def make(p, B):
        e = StandInEstimator(p, B, rank, world)
        if made is not None:
            made.append(e)
        return e

    return bench.BenchEnv(device=torch.device("cpu"), backend="gloo", make_estimator=make,
                          to_device=lambda s: {"tag": "device-streams"}, new_unique_id=lambda: b"stand-in-id",
                          preflight=lambda: rank != preflight_fails_on, device_sync=lambda: None, real=False)
