"""The device cores (decentralized_ekf_mhe_amd/csrc/*_core.h), compiled lane-sequentially for
the host (tests/hostsim, -DDEKF_HOSTSIM), against the oracle.  This is what can be checked
without a GPU: the structured arithmetic and every index of the kernels.  The same checks run
on the real wavefronts in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import hostsim_lib as HS
import oracle_lib as O
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params
from decentralized_ekf_mhe_amd.streams import make_streams

RTOL, ATOL = 1e-4, 1e-6


def _params(maker, **kw):
    p = maker()
    p.ekf_rate = p.rate
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _run(p, s, B, K):
    hs = HS.HostSim(p, B)
    outs = []
    for k in range(K):
        hs.feed(s, k)
        hs.step(k)
        outs.append(hs.get())
    return hs, outs


def _check(outs, x_ref, q_ref, it_ref=None, x_tol=1e-6):
    for k, o in enumerate(outs):
        assert np.abs(o["quat"] - q_ref[k]).max() < 1e-12, k
        if k == 0:
            continue
        assert (o["status"] == 1).all(), (k, o["status"])
        for blk in (slice(0, 3), slice(3, 6), slice(6, 9)):
            num = np.abs(o["x"][:, blk] - x_ref[k][:, blk]).max()
            assert num <= x_tol * (RTOL * np.abs(x_ref[k][:, blk]).max() + ATOL) / ATOL * 1.0 or num <= RTOL * np.abs(x_ref[k][:, blk]).max() + ATOL, (k, blk, num)
        if it_ref is not None:
            assert np.array_equal(o["iters"], it_ref[k]), (k, o["iters"], it_ref[k])


def test_go1_with_vo_through_marginalisation():
    p = _params(go1_params)
    B, K = 3, 70
    s = make_streams(p, B, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=3, want_iters=True)
    hs, outs = _run(p, s, B, K)
    _check(outs, x_ref, q_ref, it_ref)
    # far tighter than the repo tolerance in practice
    worst = max(np.abs(o["x"] - x_ref[k]).max() for k, o in enumerate(outs))
    assert worst < 1e-5
    assert max(np.abs(o["v_b"] - vb_ref[k]).max() for k, o in enumerate(outs)) < 1e-5
    assert np.abs(outs[-1]["p_vo"]).max() > 0  # VO branch ran


def test_arrival_cost_and_scaling_match_oracle():
    """M_p, n_p after marginalisation and the Ruiz vectors D, E against the oracle's generic ones"""
    p = _params(go1_params, N=6)
    K = 16
    s = make_streams(p, 1, K)
    pipe = O.Pipe(p)
    hs = HS.HostSim(p, 1)
    for k in range(K):
        pipe.feed(s, k, 0); pipe.step(k)
        hs.feed(s, k); hs.step(k)
        if k >= p.N:
            M, n = pipe.est.arrival()
            M2, n2 = hs.arrival()
            assert np.abs(M2[0] - M).max() <= 1e-9 * np.abs(M).max(), k
            assert np.abs(n2[0] - n).max() <= 1e-9 * max(np.abs(n).max(), 1e-30), k
        if k >= 1:
            D, E, c = pipe.est.scaling()
            D2, E2 = hs.scaling(len(D), len(E))
            assert np.abs(D2 / D - 1).max() < 1e-9 and np.abs(E2 / E - 1).max() < 1e-9, k


@pytest.mark.parametrize("maker,N,K", [(cassie_params, 8, 30), (pogox_params, 30, 50)])
def test_other_shapes(maker, N, K):
    p = _params(maker, N=N)
    s = make_streams(p, 2, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=2, want_iters=True)
    hs, outs = _run(p, s, 2, K)
    _check(outs, x_ref, q_ref, it_ref)


def test_kf_mode():
    p = _params(go1_params, est_type=1)
    s = make_streams(p, 2, 40)
    x_ref, vb_ref, q_ref, _ = O.run_streams(p, s)
    hs, outs = _run(p, s, 2, 40)
    for k, o in enumerate(outs):
        assert np.abs(o["x"] - x_ref[k]).max() <= 1e-10 * max(1.0, np.abs(x_ref[k]).max()), k
        assert np.abs(o["v_b"] - vb_ref[k]).max() <= 1e-10, k


def test_ekf_rewind_matches_oracle():
    """EKF core alone at 500 Hz with VO poses of assorted latencies (rel = 0, 1, many)"""
    p = _params(go1_params)
    p.ekf_rate = 500
    rng = np.random.default_rng(3)
    B, K = 5, 300
    e = [O.Ekf(p) for _ in range(B)]
    hs = HS.HostSim(p, B)
    t = np.arange(K)[:, None] * 0.002 + rng.uniform(0, 1e-4, (K, B))
    gyr = rng.normal(0, 0.3, (K, B, 3))
    acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.5, (K, B, 3))
    L = HS.lib()
    dummy = dict(p_foot=np.zeros((B, 4, 3)), J=np.zeros((B, 4, 3, 3)), qd=np.zeros((B, 4, 3)), c=np.zeros((B, 4)))
    for k in range(K):
        L.hs_push_imu(hs.h, HS._p(np.ascontiguousarray(t[k])), HS._p(np.ascontiguousarray(acc[k])), HS._p(np.ascontiguousarray(gyr[k])))
        mask = np.zeros(B, np.int32)
        tv, qv = np.zeros(B), np.zeros((B, 4))
        for b in range(B):
            e[b].set_imu(t[k, b], acc[k, b], gyr[k, b])
            if k > 30 and rng.uniform() < 0.15:
                lat = int(rng.integers(0, 25))
                q = rng.normal(size=4) * 0.02 + [1, 0, 0, 0]
                q /= np.linalg.norm(q)
                mask[b], tv[b], qv[b] = 1, t[k - lat, b] + 1e-6, q
                e[b].set_vo(tv[b], q)
        if mask.any():
            L.hs_push_vo(hs.h, HS._p(mask), HS._p(tv), HS._p(tv), HS._p(np.zeros((B, 3))), HS._p(tv), HS._p(qv))
        L.hs_ekf_step(hs.h)
        for b in range(B):
            e[b].step()
        q_hs = hs.get()["quat"]
        P_hs = hs.ekf_cov()
        for b in range(B):
            q, P = e[b].get()
            assert np.abs(q_hs[b] - q).max() < 1e-11, (k, b)
            assert np.abs(P_hs[b] - P).max() < 1e-11 * max(1e-6, np.abs(P).max()) + 1e-18, (k, b)
