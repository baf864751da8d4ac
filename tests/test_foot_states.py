"""leg_odom_type = 1: the foot positions are states (dim_state = 9 + 3 legs, 21 on Go1; DecentralEst.cpp:20, 101-113,
310-325, 432-451, 550-563).  CPU side: the oracle's own pins for that branch (KA1: MHE optimum == KF on the same log,
against an independent numpy model) and the lane-sequential build of the device cores against the oracle — plus one
finding about the reference's arithmetic: its covariance-form marginalisation (MheSrb.cpp:527-651) loses the
measurement information of a foot that has been through a swing phase (process covariance dt^2 * 1e14 next to a
measurement covariance of 1e-4), so the oracle, which follows those formulas, drifts about 1e-5 away from the
never-marginalised problem; the device cores fold the same step in information form and stay within 1e-9 of it."""
import numpy as np
import pytest

import hostsim_lib as HS
import oracle_lib as O
import ref_numpy as RN
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params
from decentralized_ekf_mhe_amd.streams import make_streams

RTOL, ATOL = 1e-4, 1e-6


def _params(maker=go1_params, **kw):
    p = maker()
    p.ekf_rate = p.rate
    p.leg_odom_type = 1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def block_err(x, ref):
    """worst |x - ref| / (RTOL |ref| + ATOL) over the 3-vector blocks of the state (p, v, bias, one per foot)"""
    nb = x.shape[-1] // 3
    d = np.abs(x - ref).reshape(x.shape[:-1] + (nb, 3)).max(axis=-1)
    r = np.abs(ref).reshape(ref.shape[:-1] + (nb, 3)).max(axis=-1)
    return float((d / (RTOL * r + ATOL)).max())


def _hostsim(p, s, B, K):
    hs = HS.HostSim(p, B)
    xs, its, sts, vbs = [], [], [], []
    for k in range(K):
        hs.feed(s, k)
        hs.step(k)
        o = hs.get()
        xs.append(o["x"]); its.append(o["iters"]); sts.append(o["status"]); vbs.append(o["v_b"])
    return np.array(xs), np.array(its), np.array(sts), np.array(vbs)


def test_dimensions_and_first_window():
    p = _params()
    assert p.dim_state == 21
    s = make_streams(p, 1, 4, vo=False)
    pipe = O.Pipe(p)
    dims = []
    for k in range(4):
        pipe.feed(s, k, 0)
        pipe.step(k)
        H, g, A, l, u = pipe.est.qp()
        dims.append((H.shape[0], A.shape[0]))
    assert dims == [(33, 12), (33 + 57, 12 + 36), (33 + 2 * 57, 12 + 72), (33 + 3 * 57, 12 + 108)]
    x = pipe.est.get()[0]
    # a foot's state is its position in the world frame: base position + R * (IMU-to-foot vector)
    R = pipe.est.rotation()
    for leg in range(4):
        assert np.abs(x[9 + 3 * leg:12 + 3 * leg] - (x[0:3] + R @ s["p_foot"][3, 0, leg])).max() < 5e-2


def test_oracle_qp_matches_block_form_with_foot_states():
    p = _params()
    s = make_streams(p, 1, 7, vo=False)
    qps = {}
    pipe = O.Pipe(p)
    quats = []
    for k in range(7):
        pipe.feed(s, k, 0)
        pipe.step(k)
        quats.append(pipe.quat())
        qps[k] = pipe.est.qp()
    for T in (1, 3, 6):
        H, g, A, l, u = qps[T]
        H2, g2, A2, l2, u2 = RN.window_qp(p, s, 0, np.array(quats), T)
        assert H.shape == H2.shape and A.shape == A2.shape
        nb = H.shape[0] // 3
        dH = np.abs(H - H2).reshape(nb, 3, nb, 3).max(axis=(1, 3))
        sH = np.abs(H2).reshape(nb, 3, nb, 3).max(axis=(1, 3))
        assert np.all(dH <= 1e-9 * sH)
        assert np.max(np.abs(A - A2)) < 1e-13
        assert np.max(np.abs(g - g2)) <= 1e-9 * np.abs(g2).max()      # the prior pulls every foot to its first measurement
        fin = np.abs(l2) < 1e20
        assert np.array_equal(fin, np.abs(l) < 1e20) and np.max(np.abs(l[fin] - l2[fin])) < 1e-12


def test_mhe_equals_kf_with_foot_states_ka1():
    """KA1 for leg_odom_type 1: exact optimum of the oracle's window == the independent numpy KF, while nothing has been
    marginalised (T < N; afterwards the reference's marginalisation formula limits the agreement, see below)"""
    p = _params(N=30)
    nsteps = 30
    s = make_streams(p, 1, nsteps, vo=False)
    pipe = O.Pipe(p)
    quats, exact, admm = [], {}, {}
    for k in range(nsteps):
        pipe.feed(s, k, 0)
        pipe.step(k)
        quats.append(pipe.quat())
        if k >= 1:
            H, g, A, l, u = pipe.est.qp()
            exact[k] = RN.kkt_exact(H, g, A, l, u)[0][-33:-12]
            admm[k] = pipe.est.get()[0].copy()
    xs_np, _ = RN.kalman_filter(p, s, 0, np.array(quats))
    for k in exact:
        # the covariance-form KF is the ill-conditioned side here (swing feet: 1e10 amplification of rounding,
        # test_kf_with_foot_states_and_its_conditioning), so the identity is checked to the repo tolerance on the
        # foot blocks and 100 x tighter on position / velocity / bias
        assert block_err(exact[k][:9], xs_np[k][:9]) <= 1e-2, k
        assert block_err(exact[k], xs_np[k]) <= 1.0, k
        assert block_err(admm[k], exact[k]) <= 1.0, k              # KA2: the eps = 1e-6 iterate
    pk = _params(N=30, est_type=1)
    xk, _, _, _ = O.run_streams(pk, s, nthreads=1)
    xs_ref, _ = RN.kalman_filter(pk, s, 0, np.array(quats), ref_double_init=True)
    assert block_err(xk[:12, 0], xs_ref[:12]) <= 1.0               # oracle KF (est_type 1) == numpy KF, while well conditioned


@pytest.mark.parametrize("maker,N,B,K", [(go1_params, 20, 3, 55), (cassie_params, 8, 2, 30), (pogox_params, 11, 2, 35)])
def test_device_cores_match_oracle_with_foot_states(maker, N, B, K):
    p = _params(maker, N=N)
    s = make_streams(p, B, K)
    x_ref, vb_ref, q_ref, _, it_ref = O.run_streams(p, s, nthreads=4, want_iters=True)
    x, it, st, vb = _hostsim(p, s, B, K)
    assert (st[1:] == 1).all()
    assert block_err(x[1:], x_ref[1:]) <= 1.0
    assert (it[1:] == it_ref[1:]).mean() > 0.98
    assert np.abs(vb[1:] - vb_ref[1:]).max() <= RTOL * np.abs(vb_ref).max() + ATOL


def test_kf_with_foot_states_and_its_conditioning():
    """est_type 1 with foot-position states.  A swinging foot gets dt^2 * 1e14 of process covariance per step, so the
    innovation covariance mixes 1e9 with 1e-4 and the covariance-form filter amplifies rounding by ~1e10: two textbook
    float64 implementations of the very same recursion (the oracle's C++ and ref_numpy's) already disagree at the
    1e-6 .. 1e-5 relative level on the same log.  The device cores are therefore held to 10 x the repo tolerance
    against the oracle in this mode (type 0, where nothing of the kind happens, is held to 1e-9)."""
    p = _params(est_type=1)
    B, K = 3, 40
    s = make_streams(p, B, K)
    x_ref, vb_ref, q_ref, _ = O.run_streams(p, s, nthreads=3)
    spread = 0.0
    for b in range(B):
        xs_np, _ = RN.kalman_filter(p, s, b, q_ref[:, b], ref_double_init=True)
        spread = max(spread, block_err(xs_np[1:], x_ref[1:, b]))
    assert 1e-3 < spread < 10.0, spread                     # conditioning of the model, not of an implementation
    x, it, st, vb = _hostsim(p, s, B, K)
    assert block_err(x[1:], x_ref[1:]) <= 10.0
    assert block_err(x[1:6], x_ref[1:6]) <= 1.0


def test_the_two_arrival_cost_forms_against_the_oracle_and_the_never_marginalised_problem():
    """instance 5 of the synthetic fleet sends a foot through a swing phase around tick 41; once that step leaves the
    window (T = 62) the oracle's states leave the exact optimum of the never-marginalised problem by ~1e-5: the reference's
    covariance-form Schur complement (MheSrb.cpp:527-651) evaluates the information a foot regains at touch-down through
    a 1e20 - 1e20 = 1e6 cancellation.  The device cores with the DEFAULT form (dekf_params.arrival_cost_form = 0: the same
    saddle inverse, M through its lower triangle, row-pivoted inverse) reproduce the oracle to 1e-9 — drift included; the
    information form (arrival_cost_form = 1) stays at 1e-9 of the exact optimum instead, inside the tolerance of both."""
    K = 68
    res = {}
    for form in (0, 1):
        p = _params(arrival_cost_form=form)
        s = make_streams(p, 1, K, first_instance=5, vo=True)
        pipe, hs = O.Pipe(p), HS.HostSim(p, 1)
        quats = []
        for k in range(K):
            pipe.feed(s, k, 0)
            pipe.step(k)
            hs.feed(s, k)
            hs.step(k)
            quats.append(pipe.quat())
        H, g, A, l, u = RN.window_qp(p, s, 0, np.array(quats), K - 1, vo=True)
        truth = RN.kkt_exact(H, g, A, l, u)[0][-33:-12]
        res[form] = (hs.get()["x"][0], pipe.est.get()[0], truth)
    x_ref_form, x_orc, truth = res[0]
    x_info_form = res[1][0]
    err_orc = np.abs(x_orc - truth).max()
    assert 1e-6 < err_orc < 1e-4, err_orc                                   # the reference formula's own drift
    assert np.abs(x_ref_form - x_orc).max() < 2e-9                          # reference form: the oracle, drift and all
    assert block_err(x_ref_form[None], x_orc[None]) <= 0.01
    err_info = np.abs(x_info_form - truth).max()
    assert err_info < 2e-8 and err_orc > 20 * err_info, (err_info, err_orc)  # information form: the exact optimum
    assert block_err(x_info_form[None], x_orc[None]) <= 1.0                 # and still inside the tolerance of the oracle


def foot_state_spread(p, s, x_ref, nthreads=8):
    """How far apart do two equally legitimate fp64 evaluations of the REFERENCE's type-1 arrival cost (MheSrb.cpp:527-651) land on
    this log?  The oracle once more with every entry of the saddle matrix S moved by one unit in the last place before Eigen's
    pivoted inverse (oracle/densemat.hpp, variant 5: what another summation order or one fused multiply-add while BUILDING S does),
    against the oracle as it is (x_ref).  Returns (base, foot): worst error over tolerance of the base blocks and of the
    foot-position blocks.  This is the yardstick the device's allowance on the foot blocks is tied to (tests/test_gpu_foot_states.py,
    tools/stress_parity.py)."""
    with O.marg_inverse_variant(5):
        x_ulp, _, _, _ = O.run_streams(p, s, nthreads=nthreads)
    base = block_err(x_ulp[1:, :, :9], x_ref[1:, :, :9])
    foot = block_err(x_ulp[1:, :, 9:], x_ref[1:, :, 9:])
    return base, foot


FOOT_ALLOWANCE = {0: 3.0, 1: 5.0}


def foot_allowance(form):
    """Allowance, in units of the stated tolerance, for the foot-position blocks of leg_odom_type 1 against the oracle, by form of the
    arrival cost (dekf_params.arrival_cost_form).  Derived from what a CONSISTENT re-evaluation of the reference formula moves
    (test_reference_formula_spread_on_foot_states): the oracle's own saddle inverse in another pivot order, or in long double, lands
    1.5 x the tolerance away from its natural-order result on these blocks.  Form 0 (the reference's formula, another elimination on
    the device): twice that, 3 x — measured 0.9 x (lane-sequential build, 4 x 240 ticks), 1.83 x (GPU, 32 x 2000 ticks of a 5 Hz
    gait, profiles/r05_type1_long_parity.jsonl).  Form 1 (information form: the same cost by another route, no 1e20 - 1e20
    cancellation, so it does not reproduce the reference's cancellation noise): 5 x — measured 2.6 x / 3.24 x.
    The one-ulp yardstick of round 4 (foot_state_spread: 13 x on the foot blocks) is reported, no longer used as a limit: moving
    EVERY entry of S by one ulp at EVERY tick is a random walk no implementation performs — on the long run it puts 36 x on the BASE
    blocks, where the device is at 0.16 x — so it bounds the formula's sensitivity from above and calibrates nothing."""
    return FOOT_ALLOWANCE[int(form)]


def test_reference_formula_spread_on_foot_states():
    """The evidence behind the allowance on the foot-position blocks.  32 swing phases of a 5 Hz gait,
    8 robots x 400 ticks, the ORACLE against itself:
      * pivot order: Eigen's natural-order pivoted inverse of S against the same elimination on the reversed matrix, and against
        long double: the two alternatives agree with each other to 0.05 of the tolerance and differ from the natural order by
        ~1.5 x the tolerance on the foot blocks — the reference's own evaluation is the outlier of the three;
      * one unit in the last place on the entries of S: an order of magnitude more, and even the base states move by about the
        tolerance.
    So "within 1e-4 of the reference" is not defined for the foot blocks of this variant to better than ~1.5 x: the reference formula
    evaluates the information a foot regains at touch-down through a 1e20 - 1e20 = 1e6 cancellation.  The allowance on these
    blocks (foot_allowance: 3 x for the reference form, 5 x for the information form) is tied to the pivot-order figure, not to the
    one-ulp figure."""
    p = _params()
    B, K = 8, 400
    s = make_streams(p, B, K, gait_hz=5.0)
    x0, _, _, _ = O.run_streams(p, s, nthreads=8)
    with O.marg_inverse_variant(2):
        x2, _, _, _ = O.run_streams(p, s, nthreads=8)
    with O.marg_inverse_variant(1):
        x1, _, _, _ = O.run_streams(p, s, nthreads=8)
    foot = lambda a, b: block_err(a[1:, :, 9:], b[1:, :, 9:])
    base = lambda a, b: block_err(a[1:, :, :9], b[1:, :, :9])
    assert foot(x1, x2) <= 0.1 and base(x1, x2) <= 0.05                    # long double == reversed order
    piv = foot(x0, x2)
    assert 1.0 <= piv <= 3.0, piv                                          # the reference's pivot order against them: 1.54 measured
    assert base(x0, x2) <= 0.25
    ulp_base, ulp_foot = foot_state_spread(p, s, x0)
    assert ulp_foot >= 5.0, ulp_foot                                       # 13 measured: one ulp in S
    assert 0.5 <= ulp_base <= 5.0, ulp_base                                # 1.2 measured: even the base states feel it
    # the device cores (lane-sequential build), default form, against the oracle on the first four robots: inside that spread
    Bh, Kh = 4, 240
    sh = {k: (np.ascontiguousarray(v[:Kh, :Bh]) if isinstance(v, np.ndarray) and v.shape[:2] == (K, B) else v) for k, v in s.items()}
    x, it, st, vb = _hostsim(p, sh, Bh, Kh)
    assert (st[1:] == 1).all()
    dev = block_err(x[1:, :, 9:], x0[1:Kh, :Bh, 9:])
    assert dev <= foot_allowance(0) and dev <= 2.0 * piv, (dev, piv, ulp_foot)   # 0.92 measured
    assert block_err(x[1:, :, :9], x0[1:Kh, :Bh, :9]) <= 1.0
    # the information form on the same log: 2.6 measured on the foot blocks, base blocks 0.09
    pi = _params()
    pi.arrival_cost_form = 1
    xi, _, sti, _ = _hostsim(pi, sh, Bh, Kh)
    assert (sti[1:] == 1).all()
    assert block_err(xi[1:, :, 9:], x0[1:Kh, :Bh, 9:]) <= foot_allowance(1)
    assert block_err(xi[1:, :, :9], x0[1:Kh, :Bh, :9]) <= 1.0


def test_spd_inverses_in_another_elimination_order_change_nothing():
    """Test knob variants 3 and 4 (oracle/est_oracle.hpp: the SPD inverses M^-1, Q^-1, R^-1 that feed the saddle matrix run their
    Cholesky solve in the reversed elimination order, read from the LOWER triangle of M as the reference's SimplicialLLT does;
    4: the pivoted inverse of S reversed as well).  S then differs in its last bits only: variant 3 stays with the reference's
    evaluation, variant 4 with the reversed-pivot one (variant 2) — and nothing diverges, which it does within three swing phases
    when both triangles of M feed the inverse."""
    p = _params()
    B, K = 4, 300
    s = make_streams(p, B, K, gait_hz=5.0)
    runs = {}
    for v in (0, 2, 3, 4):
        with O.marg_inverse_variant(v):
            runs[v], _, _, _ = O.run_streams(p, s, nthreads=4)
        assert np.isfinite(runs[v]).all()
    foot = lambda a, b: block_err(a[1:, :, 9:], b[1:, :, 9:])
    base = lambda a, b: block_err(a[1:, :, :9], b[1:, :, :9])
    assert foot(runs[3], runs[0]) <= 1.0 and base(runs[3], runs[0]) <= 0.1, (foot(runs[3], runs[0]), base(runs[3], runs[0]))
    assert foot(runs[4], runs[2]) <= 1.0 and base(runs[4], runs[2]) <= 0.1, (foot(runs[4], runs[2]), base(runs[4], runs[2]))
    assert foot(runs[4], runs[0]) <= 3.0       # the pivot-order figure (1.5 x), whatever feeds S
