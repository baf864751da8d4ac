"""The BENCHMARK kernels under every-tick oracle parity.

`dekf_create` launches the three-workgroup kernels `k_mhe_solve_r3_{4,2}_n20[_pol]` for every full window of the N = 20 shapes (since
round 6 at every batch: they are 21 % faster than the two-workgroup kernels even alone on a CU; until then only above 512 instances,
so that the oracle tests of tests/test_gpu_parity.py, B <= 37, ran the `_ll_` / `_lg_` kernels).  Here the batch is 832 (= 13 x 64 distinct logs: more than 768 slots, so some workgroups take a second instance), the oracle
runs the 64 (or fewer) distinct logs, and EVERY tick is compared: state blocks, v_b, quaternion, iteration counts — plus bit
identity of the tiles (an instance's result must not depend on the slot it ran in).  The solve the reference runs every tick:
MheSrb.cpp:340-349 (OSQP), set up per tick by MheSrb.cpp:272-338.

Tolerances as everywhere (tests/test_gpu_parity.py): per 3-block 1e-4 |oracle|_inf + 1e-6, quaternion 1e-9 absolute."""
import numpy as np
import pytest

import oracle_lib as O
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params
from decentralized_ekf_mhe_amd.estimator import BatchedEstimator, streams_to_device
from decentralized_ekf_mhe_amd.streams import make_streams
from test_gpu_parity import ATOL, RTOL, _params, block_err

pytestmark = pytest.mark.gpu
REPS = 13


def _tile(s, reps):
    return {k: (np.ascontiguousarray(np.tile(v, (1, reps) + (1,) * (v.ndim - 2))) if isinstance(v, np.ndarray) else v)
            for k, v in s.items()}


def run_tiled(p, s, K, reps=REPS, want_polish=False, family="_r3_", min_batch=512):
    """the distinct logs of `s` tiled `reps` times over one batch; every tick read back.  Returns per-tick arrays of the FIRST
    tile after checking that every other tile carries the same bits."""
    D = s["imu_t"].shape[1]
    B = D * reps
    assert B > min_batch
    est = BatchedEstimator(p, B)
    name_full, name_fill = est.solve_kernel_name(True), est.solve_kernel_name(False)
    assert family in name_full, name_full
    assert family not in name_fill, name_fill
    sd = streams_to_device(_tile(s, reps))
    xs, qs, vbs, its, sts, pols, pris = [], [], [], [], [], [], []
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
        o, info = est.get(), est.solver_info()
        for key in ("x", "v_b", "quat", "status"):
            t = o[key].reshape((reps, D) + o[key].shape[1:])
            assert np.array_equal(t, np.broadcast_to(t[0], t.shape)), (k, key)
        for key in ("iters", "rho_updates", "polish_status"):
            t = info[key].reshape(reps, D)
            assert np.array_equal(t, np.broadcast_to(t[0], t.shape)), (k, key)
        xs.append(o["x"][:D]); qs.append(o["quat"][:D]); vbs.append(o["v_b"][:D]); sts.append(o["status"][:D])
        its.append(info["iters"][:D]); pols.append(info["polish_status"][:D]); pris.append(info["pri_res"][:D])
    est.close()
    out = dict(x=np.array(xs), quat=np.array(qs), v_b=np.array(vbs), iters=np.array(its), status=np.array(sts),
               polish_status=np.array(pols), pri_res=np.array(pris), kernel=name_full)
    return out


def check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, N, iters_equal=0.98):
    assert np.abs(g["quat"] - q_ref).max() < 1e-9
    assert (g["status"][1:] == 1).all()
    # window-fill ticks (two-workgroup kernel) and full windows (three-workgroup kernel) separately, so that a failure names its kernel
    for name, sl in (("fill", slice(1, N)), ("full", slice(N, None))):
        assert block_err(g["x"][sl], x_ref[sl]) <= 1.0, name
        assert np.abs(g["v_b"][sl] - vb_ref[sl]).max() <= RTOL * np.abs(vb_ref).max() + ATOL, name
        assert (g["iters"][sl] == it_ref[sl]).mean() >= iters_equal, (name, (g["iters"][sl] == it_ref[sl]).mean())


@pytest.mark.parametrize("maker,kernel", [(go1_params, "k_mhe_solve_r3_4_n20"), (cassie_params, "k_mhe_solve_r3_2_n20")], ids=["go1", "cassie"])
def test_r3_every_tick_matches_oracle(maker, kernel):
    """64 distinct trot logs, 150 ticks (130 of them full windows: VO rows active, one and two rho updates per solve)"""
    p = _params(maker)
    D, K = 64, 150
    s = make_streams(p, D, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K)
    assert g["kernel"] == kernel
    check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, p.N)
    assert g["iters"][p.N:].max() >= 100      # solves with two refactorisations are among them


def test_r3_flight_phases_vo_dropouts_late_and_long_vo_intervals():
    """the inputs of test_gpu_parity.py::test_flight_phases_vo_dropouts_and_long_vo_intervals (flight longer than the window: every
    Meas weight at the swing value 1e-14; all feet down; a camera that drops out for 40 ticks; VO frame pairs 8 x longer than usual,
    delivered 12 ticks late) through the three-workgroup kernel's row loops, every tick"""
    p = _params(go1_params)
    D, K = 48, 130
    s = make_streams(p, D, K, vo_rate=30.0)
    slow = make_streams(p, D, K, vo_rate=3.75, vo_latency=0.06)
    for key in ("vo_mask", "vo_t_pre", "vo_t_now", "vo_dp", "vo_t_pose", "vo_q"):
        s[key][:, 24:] = slow[key][:, 24:]         # the second half of the instances gets the slow, late camera
    s["contact"][25:55, 0:12] = 0.0                # flight: 30 ticks > N
    s["contact"][25:55, 24:36] = 0.0
    s["contact"][60:85, 12:24] = 1.0               # all four feet down
    s["contact"][90:125, 36:48] = 1.0
    s["vo_mask"][40:80, 1::2] = 0                  # dropout on every other robot
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K, reps=18)                # 864 instances
    check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, p.N, iters_equal=0.97)


@pytest.mark.parametrize("cap,adapt", [(40, 1), (60, 0), (40, 0), (60, 1)])
def test_r3_iteration_cap_and_fixed_rho(cap, adapt):
    """osqp.maxQPIter below what convergence needs, with and without adaptive rho: the chunked row loops of the three-workgroup kernel
    must stop AT the cap (not at a multiple of the 25-iteration check) and hand back the oracle's unconverged iterate"""
    p = _params(go1_params, max_qp_iter=cap, adapt_rho=adapt)
    D, K = 16, 44
    s = make_streams(p, D, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K, reps=40)                # 640 instances
    it, st = g["iters"][1:], g["status"][1:]
    assert np.array_equal(it, it_ref[1:]) and it.max() == cap
    capped = it == cap
    assert (st[~capped] == 1).all() and np.isin(st[capped], (1, 2)).all() and (st[capped] == 2).any()
    assert block_err(g["x"][1:], x_ref[1:]) <= 1.0
    assert np.abs(g["quat"] - q_ref).max() < 1e-9


@pytest.mark.parametrize("maker,kernel", [(go1_params, "k_mhe_solve_r3_4_n20_pol"), (cassie_params, "k_mhe_solve_r3_2_n20_pol")], ids=["go1", "cassie"])
def test_r3_polish_matches_oracle(maker, kernel):
    """osqp.polish = 1 (the node's declared default, EstSub.cpp:188) on the three-workgroup kernels: polishing runs through
    admm_chunk_r3 with sigma = delta, rho = 1 / delta, alpha = 1 from a cold start whose z sits on the bounds.  Status-agreement rule
    and thresholds of tests/test_polish.py."""
    from test_polish import _oracle_run, _status_agrees, _tol_units
    p = _params(maker, polish=1)
    D, K = 16, 60
    s = make_streams(p, D, K)
    x_ref, st_ref, res_ref = _oracle_run(p, s, K)
    g = run_tiled(p, s, K, reps=40)
    assert g["kernel"] == kernel
    assert (g["status"][1:] == 1).all()
    full = slice(p.N, None)
    pol, ref = g["polish_status"][full], st_ref[full]
    assert _status_agrees(pol, ref), np.argwhere(~((pol == ref) | ((pol == -1) & (ref == 1))))[:5]
    assert (pol == ref).mean() >= 0.99, (pol == ref).mean()
    assert (ref == 1).any() and (pol == 1).any()
    assert _tol_units(g["x"][1:], x_ref[1:]) <= 1.0
    both = (pol == 1) & (ref == 1)
    scale = max(1.0, np.abs(x_ref).max())
    assert np.abs(g["x"][full][both] - x_ref[full][both]).max() <= 1e-8 * scale
    assert np.all(g["pri_res"][full][both] <= 1e-9)


def test_rr_pogox_every_tick_matches_oracle():
    """PogoX (1 leg, N = 100): full windows run k_mhe_solve_rr_1 (row state in registers at a run-time horizon, two workgroups per
    CU) once the batch exceeds the 256 slots of the generic kernel, which still runs the 99 window-fill ticks.  8 distinct logs
    tiled to 320 instances, 135 ticks, every tick against the oracle."""
    p = _params(pogox_params)
    D, K = 8, 135
    s = make_streams(p, D, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K, reps=40, family="_rr_", min_batch=256)
    assert g["kernel"] == "k_mhe_solve_rr_1"
    check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, p.N)


def test_rr_pogox_polish_matches_oracle():
    """osqp.polish on k_mhe_solve_rr_1_pol: the polishing iterations run through admm_chunk_rr (cold start with z on the bounds,
    sigma = delta, rho = 1 / delta, alpha = 1).  Rules of tests/test_polish.py."""
    from test_polish import _oracle_run, _status_agrees, _tol_units
    p = _params(pogox_params, polish=1)
    D, K = 4, 118
    s = make_streams(p, D, K)
    x_ref, st_ref, res_ref = _oracle_run(p, s, K)
    g = run_tiled(p, s, K, reps=72, family="_rr_", min_batch=256)
    assert g["kernel"] == "k_mhe_solve_rr_1_pol"
    assert (g["status"][1:] == 1).all()
    full = slice(p.N, None)
    pol, ref = g["polish_status"][full], st_ref[full]
    assert _status_agrees(pol, ref)
    assert (pol == ref).mean() >= 0.99, (pol == ref).mean()
    assert _tol_units(g["x"][1:], x_ref[1:]) <= 1.0


def test_rr_other_window_length_and_caps():
    """the rows-in-registers kernel at another run-time horizon (N = 90: three tiles of Dyn lane pairs instead of four, other fill
    of the last tiles; shorter windows fit the generic placement twice per CU and keep it) and with an iteration cap below
    convergence; a cap on the resident workgroups (solve_workgroups_per_cu = 1) must fall back to the generic kernel and give the
    same estimates to rounding"""
    p = _params(pogox_params, N=90, max_qp_iter=60)
    D, K = 8, 116
    s = make_streams(p, D, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K, reps=36, family="_rr_", min_batch=256)
    assert np.abs(g["quat"] - q_ref).max() < 1e-9
    assert np.array_equal(g["iters"][1:], it_ref[1:]) and g["iters"][p.N:].max() == 60
    assert block_err(g["x"][1:], x_ref[1:]) <= 1.0
    q1 = p.copy()
    q1.solve_workgroups_per_cu = 1
    est = BatchedEstimator(q1, D * 36)
    assert "_rr_" not in est.solve_kernel_name(True)
    sd = streams_to_device(_tile(s, 36))
    for k in range(K):
        est.push_stream_step(sd, k)
        est.step(k)
    o = est.get()
    est.close()
    assert block_err(o["x"][:D], g["x"][-1]) <= 0.01


def test_rr_pogox_flight_vo_dropout_and_late_vo():
    """inputs the hopping logs do not produce, through the rows-in-registers kernel at every tick: the leg in the air for 120 ticks
    (longer than the window: every Meas weight at the swing value), on the ground for 60, a camera that drops out for 50 ticks, and a
    slow, late camera (3.75 Hz, 60 ms) on half of the robots"""
    p = _params(pogox_params)
    D, K = 8, 150
    s = make_streams(p, D, K, vo_rate=30.0)
    slow = make_streams(p, D, K, vo_rate=3.75, vo_latency=0.06)
    for key in ("vo_mask", "vo_t_pre", "vo_t_now", "vo_dp", "vo_t_pose", "vo_q"):
        s[key][:, 4:] = slow[key][:, 4:]
    s["contact"][10:130, 0:2] = 0.0
    s["contact"][40:100, 2:4] = 1.0
    s["vo_mask"][60:110, 1::2] = 0
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K, reps=36, family="_rr_", min_batch=256)
    check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, p.N, iters_equal=0.97)


@pytest.mark.parametrize("desync", [1, 2], ids=["own_camera_clocks", "mixed_rates_and_blind_robots"])
def test_r3_fleet_that_is_not_in_lock_step_matches_oracle_every_tick(desync):
    """The reference deploys one process per robot: every camera delivers on its own clock (EstSub.cpp:45-56) and OSQP stops when THAT
    robot has converged (MheSrb.cpp:340-349).  64 distinct logs with per-robot VO phase, latency U[10, 60] ms and gait 1-3 Hz
    (streams.py: desync) — and, second case, per-robot frame rates of 5-50 Hz with every tenth robot blind, so that one launch mixes
    solves of 50, 75 and 100 iterations — tiled to 832 instances on the three-workgroup kernel; the instance queue hands the
    workgroups their instances in whatever order they finish.  Every tick against the oracle, tiles bit-identical."""
    p = _params(go1_params)
    D, K = 64, 140
    s = make_streams(p, D, K, desync=desync)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=16, want_iters=True)
    g = run_tiled(p, s, K)
    assert g["kernel"] == "k_mhe_solve_r3_4_n20"
    check_every_tick(g, x_ref, vb_ref, q_ref, it_ref, p.N, iters_equal=0.97)
    full = g["iters"][p.N + 30:]
    if desync == 2:
        assert len(np.unique(full)) >= 2 and (full == 50).mean() > 0.05, np.unique(full, return_counts=True)   # a launch really mixes iteration counts
