"""Independent numpy cross-implementations used to pin the C++ oracle (tests only).

None of this shares code with oracle/*.hpp: the QP here is written down directly in its
block form, the KF is the textbook recursion, the EKF uses closed
forms.  Agreement between the two independent restatements is what stands in for the
reference tests that do not exist.
"""
import numpy as np

INF = 1e30


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def quat_to_rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def kkt_exact(H, g, A, l, u):
    """Exact optimum of min 1/2 x'Hx + g'x  s.t. rows with l == u hold with equality and
    rows with infinite bounds are free (the only two row kinds this QP ever has)."""
    eq = (u - l) < 1e-9
    free = (l < -1e20) & (u > 1e20)
    assert np.all(eq | free), "QP has a finite inequality row"
    Ae, be = A[eq], l[eq]
    n, me = H.shape[0], Ae.shape[0]
    K = np.zeros((n + me, n + me))
    K[:n, :n] = H
    K[:n, n:] = Ae.T
    K[n:, :n] = Ae
    rhs = np.concatenate([-g, be])
    # equilibrate before the solve: weights span 1e-14 .. 4e10
    d = 1.0 / np.sqrt(np.maximum(np.abs(K).max(axis=1), 1e-300))
    Ks = K * d[:, None] * d[None, :]
    sol = np.linalg.solve(Ks, rhs * d) * d
    # one step of refinement in the original scaling
    r = rhs - K @ sol
    sol = sol + np.linalg.solve(Ks, r * d) * d
    return sol[:n], sol[n:]


class Model:
    """Constants of DecentralizedEstimation::initialize; both leg-odometry types: 0 foot-velocity pseudo-measurements
    (9 states), 1 foot positions as extra states (9 + 3 L; DecentralEst.cpp:20, 101-113, 310-325, 432-451, 550-563)."""

    def __init__(self, p):
        self.p = p
        self.L, self.nj, self.N = p.num_legs, p.joints_per_leg, p.N
        self.ft = p.leg_odom_type
        self.ns = 9 + 3 * self.L * self.ft
        self.dt = 1.0 / p.rate
        sq = lambda a, n=3: np.diag(np.array(a[:n]) ** 2)
        self.C_p, self.C_accel, self.C_bias = sq(p.p_process_std), sq(p.accel_input_std), sq(p.accel_bias_std)
        self.C_gyro = sq(p.gyro_input_std)
        self.C_enc_pos, self.C_enc_vel = sq(p.joint_position_std, self.nj), sq(p.joint_velocity_std, self.nj)
        self.C_swing, self.C_slide = sq(p.foot_swing_std), sq(p.foot_slide_std)
        self.Q_vo = np.diag(1.0 / np.array(p.vo_p_std[:3]) ** 2)
        prior = [np.array(p.p_init_std[:3]), np.array(p.v_init_std[:3]), np.array(p.accel_bias_init_std[:3])]
        prior += [np.array(p.foot_init_std[:3])] * (self.L * self.ft)
        self.C_prior = np.diag(np.concatenate(prior) ** 2)
        self.A_meas = np.zeros((3 * self.L, self.ns))
        for i in range(self.L):
            if self.ft:
                self.A_meas[3 * i:3 * i + 3, 0:3] = -np.eye(3)
                self.A_meas[3 * i:3 * i + 3, 9 + 3 * i:12 + 3 * i] = np.eye(3)
            else:
                self.A_meas[3 * i:3 * i + 3, 3:6] = np.eye(3)

    def x_prior(self, b_meas0):
        """prior mean: zero, except that every foot starts at its first measurement R p_foot (:310-325)"""
        x = np.zeros(self.ns)
        if self.ft:
            x[9:] = b_meas0
        return x

    def sample(self, s, k, b, quat):
        """measurement-side quantities of step k: R, a_s, omega, b_meas, C_meas (+ the contact flags, which decide
        the process noise of the foot states of the NEXT dynamics step)"""
        R = quat_to_rot(np.asarray(quat, float))
        a_s = R @ s["accel"][k, b] + np.array([0, 0, -9.81])
        om = s["gyro"][k, b]
        b_meas = np.zeros(3 * self.L)
        C_meas = np.zeros((3 * self.L, 3 * self.L))
        for i in range(self.L):
            J, pf, qd = s["J"][k, b, i], s["p_foot"][k, b, i], s["qdot"][k, b, i]
            blk = slice(3 * i, 3 * i + 3)
            if self.ft:
                b_meas[blk] = R @ pf
                C_meas[blk, blk] = R @ (J @ self.C_enc_pos @ J.T) @ R.T
                continue
            b_meas[blk] = -R @ J @ qd - R @ np.cross(om, pf)
            if s["contact"][k, b, i] == 0.0:
                C_meas[blk, blk] = self.C_swing
            else:
                G = np.hstack([-J, -skew(om) @ J, skew(pf)])
                Cm = np.zeros((2 * self.nj + 3,) * 2)
                Cm[:self.nj, :self.nj] = self.C_enc_vel
                Cm[self.nj:2 * self.nj, self.nj:2 * self.nj] = self.C_enc_pos
                Cm[2 * self.nj:, 2 * self.nj:] = self.C_gyro
                C_meas[blk, blk] = R @ G @ Cm @ G.T @ R.T
        self.last_contact = np.array(s["contact"][k, b], float)
        return R, a_s, om, b_meas, C_meas

    def dynamics(self, R, a_s, contact=None):
        """A, b and the process covariance G C_in G' of one step (foot states: identity dynamics, dt^2 R C R' with
        C = foot_slide in contact, foot_swing otherwise)"""
        dt, ns = self.dt, self.ns
        A = np.eye(ns)
        A[0:3, 3:6] = dt * np.eye(3)
        A[0:3, 6:9] = -dt * dt / 2 * R
        A[3:6, 6:9] = -dt * R
        b = np.concatenate([-dt * dt / 2 * a_s, -dt * a_s, np.zeros(ns - 6)])
        G = np.zeros((ns, ns))
        G[0:3, 0:3] = dt * R
        G[0:3, 3:6] = 0.5 * dt * dt * R
        G[3:6, 3:6] = dt * R
        G[6:9, 6:9] = dt * np.eye(3)
        Cin = np.zeros((ns, ns))
        Cin[0:3, 0:3], Cin[3:6, 3:6], Cin[6:9, 6:9] = self.C_p, self.C_accel, self.C_bias
        for i in range(self.L * self.ft):
            blk = slice(9 + 3 * i, 12 + 3 * i)
            G[blk, blk] = dt * R
            Cin[blk, blk] = self.C_slide if contact[i] != 0.0 else self.C_swing
        return A, b, G @ Cin @ G.T


def kalman_filter(p, s, b, quats, ref_double_init=False):
    """Textbook KF on the same model; returns x[K,ns], C[K,ns,ns].
    ref_double_init=True reproduces est_type 1 of the reference exactly: initialize() runs
    InitializeKF() AND UpdateKF() (DecentralEst.cpp:140-141), i.e. the first sample is
    predicted-through and corrected a second time before update(1)."""
    m = Model(p)
    ns = m.ns
    K = s["imu_t"].shape[0]
    xs, Cs = np.zeros((K, ns)), np.zeros((K, ns, ns))
    x, Cc = None, m.C_prior.copy()
    prev = None
    for k in range(K):
        R, a_s, om, b_meas, C_meas = m.sample(s, k, b, quats[k])
        contact = m.last_contact
        if k == 0:
            x = m.x_prior(b_meas)
        if k > 0:
            A, bd, Cd = m.dynamics(*prev)
            x = A @ x - bd
            Cc = A @ Cc @ A.T + Cd
        S = m.A_meas @ Cc @ m.A_meas.T + C_meas
        Kg = Cc @ m.A_meas.T @ np.linalg.inv(S)
        x = x + Kg @ (b_meas - m.A_meas @ x)
        Cc = (np.eye(ns) - Kg @ m.A_meas) @ Cc
        if k == 0 and ref_double_init:
            A, bd, Cd = m.dynamics(R, a_s, contact)
            x = A @ x - bd
            Cc = A @ Cc @ A.T + Cd
            S = m.A_meas @ Cc @ m.A_meas.T + C_meas
            Kg = Cc @ m.A_meas.T @ np.linalg.inv(S)
            x = x + Kg @ (b_meas - m.A_meas @ x)
            Cc = (np.eye(ns) - Kg @ m.A_meas) @ Cc
        xs[k], Cs[k] = x, Cc
        prev = (R, a_s, contact)
    return xs, Cs


class VoTrack:
    """Independent numpy statement of the visual-odometry branch of GetMeasurement + UpdateVOConstraints
    (DecentralEst.cpp:883-945, 987-1009) and of the cubic Bezier over the last four accumulated VO positions
    (Bezier_simple.cpp:12-82).  Deliberately written differently from oracle/est_oracle.hpp: absolute sample
    indices (sample k of the log = discrete time k) with a sliding lower limit instead of trimmed stacks,
    np.searchsorted for upper_bound, the Bernstein form of the cubic, and the bounds of ALL steps kept in one dict
    keyed by discrete time ({k: -(node_{i+1} - node_i)}, the value UpdateVOConstraints writes into l and u)."""

    def __init__(self, N, dt):
        self.N, self.dt = N, dt
        self.times, self.R = [], []
        self.p_acc = np.zeros(3)
        self.way, self.way_t = [], []
        self.bounds = {}

    @staticmethod
    def bernstein(u, P):
        v = 1.0 - u
        return v ** 3 * P[0] + 3 * v * v * u * P[1] + 3 * v * u * u * P[2] + u ** 3 * P[3]

    def vo_event(self, T, t_pre, t_now, dp_body):
        """a frame pair latched before the sample of update(T) is taken (stack = samples 0..T-1)"""
        n = len(self.times)
        if n == 0:
            return False                                  # stays pending in the reference; the logs never do this
        lo = max(0, n - (4 * self.N + 1))                   # the stacks are trimmed to 4N+1 entries (:963)
        tt = np.array(self.times[lo:n])
        j = int(np.searchsorted(tt, t_pre, side="right"))
        if j == 0:
            return True                                   # too early: discarded (:898-904)
        i_pre = lo + j - 1
        i_now = lo + int(np.searchsorted(tt, t_now, side="right")) - 1
        self.p_acc = self.p_acc + self.R[i_pre] @ np.asarray(dp_body)
        self.way.append(self.p_acc.copy())
        self.way_t.append(t_now)
        self.way, self.way_t = self.way[-4:], self.way_t[-4:]
        w0 = n - min(self.N, T)                           # first sample of the window
        i0 = max(w0, i_pre)
        if i_now > w0 and len(self.way) == 4:
            span = self.way_t[-1] - self.way_t[0]
            u0 = (self.times[i0] - self.way_t[0]) / span
            nodes = [self.bernstein(u0 + (self.dt / span) * i, self.way) for i in range(i_now - i0 + 1)]
            for i in range(len(nodes) - 1):
                self.bounds[i0 + i] = -(nodes[i + 1] - nodes[i])
        return True

    def push(self, t, R):
        self.times.append(float(t))
        self.R.append(np.array(R))


def window_qp(p, s, b, quats, T, vo=False, return_track=False):
    """The UN-MARGINALISED QP after update(T) over ALL steps 0..T (for T < N this is the window itself), in the
    layout of SURVEY.md Appendix A: variables [x0 v0 | w0 c0 x1 v1 | ...], rows [M0 | D0 V0 M1 | ...].
    vo=True: VO rows carry the equality bounds VoTrack writes, the others stay +-1e30.  Both leg-odometry types."""
    m = Model(p)
    L = m.L
    nm, ns, nc = 3 * L, m.ns, 3
    sv, sc = ns + nm + ns + nc, nm + ns + nc
    n = (ns + nm) + T * sv
    mm = nm + T * sc
    H, g = np.zeros((n, n)), np.zeros(n)
    A, l, u = np.zeros((mm, n)), np.zeros(mm), np.zeros(mm)
    samples, contacts = [], []
    for k in range(T + 1):
        samples.append(m.sample(s, k, b, quats[k]))
        contacts.append(m.last_contact)
    track = VoTrack(m.N, m.dt)
    for k in range(T + 1):
        if vo and k >= 1 and s["vo_mask"][k, b]:
            track.vo_event(k, s["vo_t_pre"][k, b], s["vo_t_now"][k, b], s["vo_dp"][k, b])
        track.push(s["imu_t"][k, b], samples[k][0])

    def xo(k):
        return 0 if k == 0 else (ns + nm) + (k - 1) * sv + ns + nc

    def vo_(k):
        return xo(k) + ns

    def wo(k):  # w_k, c_k live in the block created at update(k+1)
        return (ns + nm) + k * sv

    Qp = np.linalg.inv(m.C_prior)
    H[0:ns, 0:ns] = Qp
    g[0:ns] = -Qp @ m.x_prior(samples[0][3])
    for k in range(T + 1):
        R, a_s, om, b_meas, C_meas = samples[k]
        r0 = 0 if k == 0 else nm + (k - 1) * sc + ns + nc
        A[r0:r0 + nm, xo(k):xo(k) + ns] = m.A_meas
        A[r0:r0 + nm, vo_(k):vo_(k) + nm] = -np.eye(nm)
        l[r0:r0 + nm] = u[r0:r0 + nm] = b_meas
        H[vo_(k):vo_(k) + nm, vo_(k):vo_(k) + nm] = np.linalg.inv(C_meas)
        if k < T:
            Ad, bd, Cd = m.dynamics(R, a_s, contacts[k])
            rd = nm + k * sc
            A[rd:rd + ns, xo(k):xo(k) + ns] = Ad
            A[rd:rd + ns, wo(k):wo(k) + ns] = -np.eye(ns)
            A[rd:rd + ns, xo(k + 1):xo(k + 1) + ns] = -np.eye(ns)
            l[rd:rd + ns] = u[rd:rd + ns] = bd
            Qd = np.zeros((ns, ns))
            Qd[0:6, 0:6] = np.linalg.inv(Cd[0:6, 0:6])
            Qd[6:9, 6:9] = np.diag(1.0 / np.array(p.accel_bias_std[:3]) ** 2) / m.dt ** 2
            for i in range(L * m.ft):
                blk = slice(9 + 3 * i, 12 + 3 * i)
                Qd[blk, blk] = np.linalg.inv(Cd[blk, blk])
            H[wo(k):wo(k) + ns, wo(k):wo(k) + ns] = Qd
            rc = rd + ns
            co = wo(k) + ns
            A[rc:rc + 3, xo(k):xo(k) + 3] = np.eye(3)
            A[rc:rc + 3, xo(k + 1):xo(k + 1) + 3] = -np.eye(3)
            A[rc:rc + 3, co:co + 3] = -np.eye(3)
            if k in track.bounds:
                l[rc:rc + 3] = u[rc:rc + 3] = track.bounds[k]
            else:
                l[rc:rc + 3], u[rc:rc + 3] = -INF, INF
            H[co:co + 3, co:co + 3] = R @ m.Q_vo @ R.T
    if return_track:
        return (H, g, A, l, u), track
    return H, g, A, l, u


# ---------------------------------------------------------------- EKF, closed forms
def ekf_predict(q, P, gyro, dt, C_gyro):
    wx, wy, wz = gyro
    Om = np.array([[0, -wx, -wy, -wz], [wx, 0, wz, -wy], [wy, -wz, 0, wx], [wz, wy, -wx, 0]])
    F = np.eye(4) + dt / 2 * Om
    w, x, y, z = q
    # the reference's W, including its mis-assigned last row (orien_ekf.cpp:270-294)
    W = 0.5 * dt * np.array([[-x, -y, -z], [w, -z, y], [z, x, w], [-y, 0, 0]])
    qp = F @ q
    Pp = F @ P @ F.T + W @ C_gyro @ W.T
    return qp / np.linalg.norm(qp), Pp


def ekf_correct(q, P, accel, C_accel, g=9.81):
    R = quat_to_rot(q)
    a_hat = R.T @ np.array([0, 0, g])
    w, x, y, z = q
    H = 2 * g * np.array([[-y, z, -w, x], [x, w, z, y], [w, -x, -y, z]])
    rel = np.linalg.norm(accel) / g
    S = H @ P @ H.T + rel ** 2 * C_accel
    K = P @ H.T @ np.linalg.inv(S)
    qc = q + K @ (accel - a_hat)
    Pc = (np.eye(4) - K @ H) @ P
    return qc / np.linalg.norm(qc), Pc


def ekf_vo_correct(q, P, q_vo, C_vo):
    K = P @ np.linalg.inv(P + C_vo)
    qc = q + K @ (q_vo - q)
    Pc = (np.eye(4) - K) @ P
    return qc / np.linalg.norm(qc), Pc
