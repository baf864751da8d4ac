"""Pins for the quaternion-EKF oracle (KA4 of SURVEY.md §8c): closed-form numpy
cross-implementation, zero-innovation fixed point, the reproduced quat_2_W defect, and
the VO rewind/replay bookkeeping including its off-by-one."""
import numpy as np

import oracle_lib as O
import ref_numpy as RN
from decentralized_ekf_mhe_amd import go1_params


def _consts(p):
    dt = 1.0 / p.ekf_rate
    Cg = np.diag(np.array(p.ekf_process_std[:3]) ** 2)
    Ca = np.diag(np.array(p.ekf_gravity_meas_std[:3]) ** 2)
    Cv = np.diag(np.array(p.ekf_vo_meas_std[:4]) ** 2)
    return dt, Cg, Ca, Cv


def test_single_calls_match_numpy():
    p = go1_params()
    dt, Cg, Ca, Cv = _consts(p)
    e = O.Ekf(p)
    rng = np.random.default_rng(1)
    for _ in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        A = rng.normal(size=(4, 4)) * 1e-3
        P = A @ A.T + 1e-6 * np.eye(4)
        w, a, qv = rng.normal(size=3), rng.normal(size=3) * 3 + [0, 0, 9.8], rng.normal(size=4)
        for (got, want) in ((e.predict(q, w, P), RN.ekf_predict(q, P, w, dt, Cg)),
                            (e.correct(q, a, P), RN.ekf_correct(q, P, a, Ca)),
                            (e.vo_correct(q, qv, P), RN.ekf_vo_correct(q, P, qv, Cv))):
            assert np.max(np.abs(got[0] - want[0])) < 1e-12
            assert np.max(np.abs(got[1] - want[1])) < 1e-12 * max(1.0, np.max(np.abs(want[1])))


def test_w_matrix_defect_is_reproduced():
    """with the textbook W the predicted covariance differs; the oracle must follow the
    reference's W (rows 2 and 3 as orien_ekf.cpp:289-291 leaves them)"""
    p = go1_params()
    dt, Cg, _, _ = _consts(p)
    e = O.Ekf(p)
    q = np.array([0.9, 0.1, -0.3, 0.2])
    q /= np.linalg.norm(q)
    P = np.diag([1e-6] * 4)
    _, Pp = e.predict(q, np.zeros(3), P)
    w, x, y, z = q
    W_ref = 0.5 * dt * np.array([[-x, -y, -z], [w, -z, y], [z, x, w], [-y, 0, 0]])
    W_txt = 0.5 * dt * np.array([[-x, -y, -z], [w, -z, y], [z, w, -x], [-y, x, w]])
    assert np.allclose(Pp, P + W_ref @ Cg @ W_ref.T, rtol=0, atol=1e-18)
    assert not np.allclose(Pp, P + W_txt @ Cg @ W_txt.T, rtol=0, atol=1e-12)


def test_zero_innovation_fixed_point():
    """omega = 0 and a = R(q)' g  =>  the quaternion does not move"""
    p = go1_params()
    e = O.Ekf(p)
    q = np.array([0.95, 0.05, -0.2, 0.1])
    q /= np.linalg.norm(q)
    a = RN.quat_to_rot(q).T @ np.array([0, 0, 9.81])
    P = np.eye(4) * 1e-6
    for _ in range(20):
        qp, Pp = e.predict(q, np.zeros(3), P)
        q2, P = e.correct(qp, a, Pp)
        assert np.max(np.abs(q2 - q)) < 1e-14
        assert np.max(np.abs(P - P.T)) < 1e-12 * np.max(np.abs(P))


def _numpy_ekf_run(p, t, acc, gyr, vo):
    """independent timer-step loop with the reference's rewind semantics"""
    dt, Cg, Ca, Cv = _consts(p)
    q = np.array(p.ekf_quaternion_init[:4])
    P = np.diag(np.array(p.ekf_init_std[:4]) ** 2)
    hist = []
    out = []
    for k in range(len(t)):
        hist.append((t[k], acc[k], gyr[k], q.copy(), P.copy()))
        if k in vo:
            tv, qv = vo[k]
            times = [h[0] for h in hist]
            idx = int(np.searchsorted(times, tv, side="right")) - 1
            if idx >= 0:
                rel = (len(hist) - 1) - idx
                q, P = hist[idx][3].copy(), hist[idx][4].copy()
                for i in range(rel - 1):
                    q, P = RN.ekf_predict(q, P, hist[idx + i][2], dt, Cg)
                    q, P = RN.ekf_correct(q, P, hist[idx + i][1], Ca)
                    if i == 0:
                        q, P = RN.ekf_vo_correct(q, P, qv, Cv)
        q, P = RN.ekf_predict(q, P, gyr[k], dt, Cg)
        q, P = RN.ekf_correct(q, P, acc[k], Ca)
        out.append((q.copy(), P.copy()))
    return out


def test_trace_with_vo_rewind():
    p = go1_params()
    rng = np.random.default_rng(7)
    K = 400
    dt = 1.0 / p.ekf_rate
    t = np.arange(K) * dt + rng.uniform(0, 1e-4, K)
    gyr = 0.3 * np.sin(np.arange(K)[:, None] * 0.05 + np.arange(3)) + rng.normal(0, 0.03, (K, 3))
    acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.3, (K, 3))
    vo = {}
    for k in range(40, K, 17):
        lat = rng.integers(0, 20)            # rel = 0, 1 (no-ops) up to 19 replayed steps
        qv = rng.normal(size=4) * 0.01 + [1, 0, 0, 0]
        vo[k] = (t[k - lat] + 1e-6, qv / np.linalg.norm(qv))
    vo[5] = (-1.0, np.array([1.0, 0, 0, 0]))  # older than every IMU sample: dropped
    want = _numpy_ekf_run(p, t, acc, gyr, vo)
    e = O.Ekf(p)
    replays = []
    for k in range(K):
        e.set_imu(t[k], acc[k], gyr[k])
        if k in vo:
            e.set_vo(*vo[k])
        e.step()
        replays.append(e.last_replay())
        q, P = e.get()
        assert np.max(np.abs(q - want[k][0])) < 1e-11, k
        assert np.max(np.abs(P - want[k][1])) < 1e-11 * max(1e-6, np.max(np.abs(P))) + 1e-18, k
    assert max(replays) >= 15 and replays[5] == 0
    # a VO sample that lands on the newest or second-newest IMU sample changes nothing (rel < 2)
    zero_lat = [k for k in vo if k >= 40 and replays[k] == 0]
    assert len(zero_lat) >= 1
