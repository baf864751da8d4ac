"""The four-per-CU solve kernels of round 6 (`k_mhe_solve_r4_{4,2}_n20[_pol]`: workgroups of THREE wavefronts — the solve wavefront and
two workers — selected by `solve_workgroups_per_cu = 4`; mhe_admm_core.h: admm_chunk_r4).

Same operations on the same operands as the three-workgroup kernels, so the contract is BIT identity with them on every output
(states, v_b, quaternion, residuals, iteration, rho-update and polish counts) at every tick — plus one direct every-tick oracle
parity run, so that the family does not hang on another family's test alone.  The reference's solve: MheSrb.cpp:340-349 (OSQP), set
up per tick by MheSrb.cpp:272-338.  The instance queue the solve kernels take their work from (kernels.hip: DEKF_QUEUE_LOOP) is
exercised with batches that are not multiples of the persistent grid, across a reset, and back to back with another handle."""
import numpy as np
import pytest

import oracle_lib as O
from decentralized_ekf_mhe_amd import cassie_params, go1_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
from test_gpu_parity import _params
from test_gpu_r3_parity import check_every_tick, run_tiled

pytestmark = pytest.mark.gpu
OUT_KEYS = ("x", "v_b", "quat", "status")
INFO_KEYS = ("iters", "rho_updates", "pri_res", "dua_res", "polish_status")


def _pair(p, B):
    ests = []
    for cap in (0, 4):
        q = p.copy()
        q.solve_workgroups_per_cu = cap
        ests.append(BatchedEstimator(q, B))
    return ests


def _identical_every_tick(p, s, B, K, names):
    sd = streams_to_device(s)
    e3, e4 = _pair(p, B)
    assert e3.solve_kernel_name(True) == names[0] and e4.solve_kernel_name(True) == names[1], (e3.solve_kernel_name(True), e4.solve_kernel_name(True))
    assert e4.launch_info()["solve_workgroups"] == min(B, 4 * e4.launch_info()["compute_units"])
    iters = []
    for k in range(K):
        outs = []
        for e in (e3, e4):
            e.push_stream_step(sd, k)
            e.step(k)
            outs.append((e.get(), e.solver_info()))
        (o3, i3), (o4, i4) = outs
        for key in OUT_KEYS:
            assert np.array_equal(o3[key], o4[key]), (k, key)
        for key in INFO_KEYS:
            assert np.array_equal(i3[key], i4[key]), (k, key)
        iters.append(i4["iters"].copy())
    e3.close(); e4.close()
    return np.array(iters)


@pytest.mark.parametrize("maker,names", [(go1_params, ("k_mhe_solve_r3_4_n20", "k_mhe_solve_r4_4_n20")),
                                         (cassie_params, ("k_mhe_solve_r3_2_n20", "k_mhe_solve_r4_2_n20"))], ids=["go1", "cassie"])
def test_r4_gives_the_bits_of_r3(maker, names):
    """1100 distinct logs (more than the 1024 slots of the four-per-CU grid: some workgroups take a second instance from the queue),
    window fill + 50 full windows: VO rows active, solves with one and two rho updates"""
    p = _params(maker)
    B, K = 1100, p.N + 50
    it = _identical_every_tick(p, make_streams(p, B, K), B, K, names)
    assert it[p.N:].min() >= 25 and it[p.N:].max() >= 100


def test_r4_with_polishing_gives_the_bits_of_r3():
    p = _params(go1_params, polish=1)
    B, K = 1040, p.N + 25
    _identical_every_tick(p, make_streams(p, B, K), B, K, ("k_mhe_solve_r3_4_n20_pol", "k_mhe_solve_r4_4_n20_pol"))


@pytest.mark.parametrize("cap,adapt", [(40, 1), (60, 0)])
def test_r4_iteration_cap_and_fixed_rho_give_the_bits_of_r3(cap, adapt):
    """chunks that end AT the cap (not at a multiple of the termination check), with and without a refactorisation in between"""
    p = _params(go1_params, max_qp_iter=cap, adapt_rho=adapt)
    B, K = 1030, p.N + 12
    it = _identical_every_tick(p, make_streams(p, B, K), B, K, ("k_mhe_solve_r3_4_n20", "k_mhe_solve_r4_4_n20"))
    assert it[p.N:].max() == cap


def test_r4_every_tick_matches_oracle_on_flight_phases_and_vo_dropouts():
    """the inputs of test_gpu_r3_parity.py::test_r3_flight_phases_vo_dropouts_late_and_long_vo_intervals straight against the oracle"""
    p = _params(go1_params)
    p.solve_workgroups_per_cu = 4
    D, K = 48, 130
    s = make_streams(p, D, K, vo_rate=30.0)
    slow = make_streams(p, D, K, vo_rate=3.75, vo_latency=0.06)
    for key in ("vo_mask", "vo_t_pre", "vo_t_now", "vo_dp", "vo_t_pose", "vo_q"):
        s[key][:, 24:] = slow[key][:, 24:]
    s["contact"][25:55, 0:12] = 0.0
    s["contact"][25:55, 24:36] = 0.0
    s["contact"][60:85, 12:24] = 1.0
    s["contact"][90:125, 36:48] = 1.0
    s["vo_mask"][40:80, 1::2] = 0
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K, reps=22, family="_r4_")          # 1056 instances on 1024 workgroups
    assert g["kernel"] == "k_mhe_solve_r4_4_n20"
    check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, p.N, iters_equal=0.97)


def test_instance_queue_serves_every_instance_once_whatever_the_batch():
    """Batches that leave the persistent grid a ragged last round, a reset in between and a second handle launching in between: every
    launch must find the queue at zero (the last workgroup of the launch before reset it) and serve each instance exactly once —
    an instance served twice or not at all shows up as a difference to the same logs run in a batch that fits the grid."""
    p = _params(go1_params)
    K = p.N + 6
    s = make_streams(p, 96, K)

    def tile(reps):
        return {k: (np.ascontiguousarray(np.tile(v, (1, reps) + (1,) * (v.ndim - 2))) if isinstance(v, np.ndarray) else v) for k, v in s.items()}

    ref = BatchedEstimator(p, 96)
    sd_ref = streams_to_device(s)
    xr = []
    for k in range(K):
        ref.push_stream_step(sd_ref, k); ref.step(k)
        xr.append(ref.get()["x"].copy())
    ref.close()
    for reps, cap in ((9, 0), (11, 4), (17, 0)):            # 864 on 768 slots, 1056 on 1024, 1632 on 768
        q = p.copy(); q.solve_workgroups_per_cu = cap
        est, other = BatchedEstimator(q, 96 * reps), BatchedEstimator(q, 96 * reps)
        sd = streams_to_device(tile(reps))
        for rnd in range(2):
            for k in range(K):
                est.push_stream_step(sd, k); est.step(k)
                other.push_stream_step(sd, k); other.step(k)
                x = est.get()["x"].reshape(reps, 96, -1)
                assert np.array_equal(x, np.broadcast_to(np.asarray(xr[k]), x.shape)), (reps, cap, rnd, k)
            est.reset(); other.reset()
        est.close(); other.close()
