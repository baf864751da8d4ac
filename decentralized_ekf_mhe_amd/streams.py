"""Synthetic sensor logs for batches of legged robots (SURVEY.md §8(d) recipe).

Every instance ``i`` draws from ``numpy.random.default_rng(0x5EED0000 + i)``, so a shard
``[first, first + batch)`` of a larger fleet can be generated independently on each rank
and the oracle and the HIP path can be fed identical logs.

Arrays are ``[nsteps][batch][...]`` float64, i.e. one contiguous slab per estimator step,
which is what ``dekf_push_*`` consumes.  Step ``k`` carries what the reference's ROS callbacks
would have latched into ``robot_store`` by the k-th 5 ms timer tick
(go1Sub.cpp:30-126, EstSub.cpp:45-56, orien_ekf.cpp:48-75).
"""
import numpy as np

SEED0 = 0x5EED0000
G_S = np.array([0.0, 0.0, -9.81])


def _rot_zyx(yaw, pitch, roll):
    """R_sb = Rz(yaw) Ry(pitch) Rx(roll); inputs [...], output [..., 3, 3]."""
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    R = np.empty(yaw.shape + (3, 3))
    R[..., 0, 0] = cy * cp
    R[..., 0, 1] = cy * sp * sr - sy * cr
    R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp
    R[..., 1, 1] = sy * sp * sr + cy * cr
    R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp
    R[..., 2, 1] = cp * sr
    R[..., 2, 2] = cp * cr
    return R


def _quat_from_rot(R):
    """[..., 3, 3] -> [..., 4] (w x y z), w >= 0 branch-free enough for small tilts + yaw."""
    t = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    w = 0.5 * np.sqrt(np.maximum(1.0 + t, 1e-12))
    x = (R[..., 2, 1] - R[..., 1, 2]) / (4 * w)
    y = (R[..., 0, 2] - R[..., 2, 0]) / (4 * w)
    z = (R[..., 1, 0] - R[..., 0, 1]) / (4 * w)
    return np.stack([w, x, y, z], axis=-1)


def leg_kinematics(q, leg, nj):
    """Serial-chain leg used only to synthesise (p_imu_2_foot, J): joint 0 rotates about
    body x at the hip, joints 1.. about the (rotated) y axis, equal link lengths.
    q [..., nj] -> p [..., 3], J [..., 3, nj].  Not the Go1 FROST model (that one is the
    reference's own code and only serves as golden vectors, tests/golden/go1_kin.npz)."""
    sx = (1.0, 1.0, -1.0, -1.0)[leg % 4]
    sy = (-1.0, 1.0, -1.0, 1.0)[leg % 4]
    hip = np.array([sx * 0.1881, sy * 0.04675, 0.0])
    l1 = sy * 0.08
    ln = 0.426 / max(nj - 1, 1)
    shp = q.shape[:-1]
    c0, s0 = np.cos(q[..., 0]), np.sin(q[..., 0])
    Rx = np.zeros(shp + (3, 3))
    Rx[..., 0, 0] = 1
    Rx[..., 1, 1] = c0
    Rx[..., 1, 2] = -s0
    Rx[..., 2, 1] = s0
    Rx[..., 2, 2] = c0
    ay = Rx[..., :, 1]  # rotated y axis
    origins = [np.broadcast_to(hip, shp + (3,))]
    axes = [np.broadcast_to(np.array([1.0, 0, 0]), shp + (3,))]
    o = hip + Rx @ np.array([0.0, l1, 0.0])
    ang = np.zeros(shp)
    for j in range(1, nj):
        origins.append(o)
        axes.append(ay)
        ang = ang + q[..., j]
        # link along -z of the frame rotated by Rx * Ry(ang)
        d = np.stack([-np.sin(ang), np.zeros(shp), -np.cos(ang)], axis=-1) * ln
        o = o + np.einsum("...ij,...j->...i", Rx, d)
    p = o
    J = np.stack([np.cross(a, p - og) for a, og in zip(axes, origins)], axis=-1)
    return p, J


def make_streams(params, batch, nsteps, first_instance=0, vo=True, vo_rate=30.0, vo_latency=0.03,
                 seed0=SEED0, gait_hz=2.0, desync=False):
    """Return a dict of [nsteps][batch][...] arrays (+ ground truth under 'gt_*').  gait_hz: contact cycles per second
    (2 Hz trot, 60 % duty); the long-run parity cases raise it to pack more swing phases into a log.
    desync: a fleet that is NOT in lock-step, as the reference deploys it — every robot's camera runs on its own clock
    (EstSub.cpp:45-56: vo_callback latches whenever ITS front-end delivers), so per instance the VO frame phase is U[0, 1 / vo_rate),
    the delivery latency U[10, 60] ms and the gait 1-3 Hz (drawn from a second generator per instance: the default streams keep
    their bits); desync=2 additionally draws every camera's frame rate from U[5, 50] Hz and leaves every tenth robot without vision.
    With the default the whole batch receives its vision intervals at the same ticks."""
    L, nj = params.num_legs, params.joints_per_leg
    dt = 1.0 / params.rate
    B, K = batch, nsteps
    rngs = [np.random.default_rng(seed0 + first_instance + i) for i in range(B)]
    ph = np.stack([r.uniform(0, 2 * np.pi, 8) for r in rngs])  # [B, 8]
    bias = np.stack([r.normal(0, 0.05, 3) for r in rngs])  # [B, 3]

    if desync:
        rng2 = [np.random.default_rng(seed0 + 0x10000000 + first_instance + i) for i in range(B)]
        vo_phase = np.array([r.uniform(0.0, 1.0 / vo_rate) for r in rng2])
        vo_lat = np.array([r.uniform(0.010, 0.060) for r in rng2])
        gait_hz = np.array([r.uniform(1.0, 3.0) for r in rng2])  # [B]
        vo_rate_b = np.full(B, float(vo_rate))
        if desync == 2:  # ... and a MIXED fleet: every camera at its own frame rate, U[5, 50] Hz, every tenth robot blind
            vo_rate_b = np.array([r.uniform(5.0, 50.0) for r in rng2])
            vo_phase = vo_phase * vo_rate / vo_rate_b
            blind = np.array([r.uniform() < 0.1 for r in rng2])
            vo_phase = np.where(blind, 1e9, vo_phase)
    else:
        vo_phase = np.zeros(B)
        vo_lat = np.full(B, float(vo_latency))
        vo_rate_b = np.full(B, float(vo_rate))
    gait_b = np.broadcast_to(np.asarray(gait_hz, dtype=np.float64), (B,))
    k = np.arange(K)[:, None]  # [K,1]
    jitter = np.stack([r.uniform(0, 0.2e-3, K) for r in rngs], axis=1)  # [K,B]
    t = k * dt + 0 * ph[None, :, 0]  # [K,B] nominal time
    imu_t = t + jitter

    def vel(tt):
        return np.stack([0.5 + 0.2 * np.sin(np.pi * tt + ph[:, 0]),
                         0.1 * np.sin(1.4 * np.pi * tt + ph[:, 1]),
                         0.05 * np.sin(4 * np.pi * tt + ph[:, 2])], axis=-1)

    def acc(tt):
        return np.stack([0.2 * np.pi * np.cos(np.pi * tt + ph[:, 0]),
                         0.14 * np.pi * np.cos(1.4 * np.pi * tt + ph[:, 1]),
                         0.2 * np.pi * np.cos(4 * np.pi * tt + ph[:, 2])], axis=-1)

    def pos(tt):
        return np.stack([0.5 * tt - 0.2 / np.pi * (np.cos(np.pi * tt + ph[:, 0]) - np.cos(ph[:, 0])),
                         -0.1 / (1.4 * np.pi) * (np.cos(1.4 * np.pi * tt + ph[:, 1]) - np.cos(ph[:, 1])),
                         -0.05 / (4 * np.pi) * (np.cos(4 * np.pi * tt + ph[:, 2]) - np.cos(ph[:, 2]))], axis=-1)

    def euler(tt):
        roll = 0.1 * np.sin(2 * np.pi * tt + ph[:, 3])
        pitch = 0.1 * np.sin(2 * np.pi * tt + ph[:, 4])
        yaw = 0.2 * tt
        droll = 0.2 * np.pi * np.cos(2 * np.pi * tt + ph[:, 3])
        dpitch = 0.2 * np.pi * np.cos(2 * np.pi * tt + ph[:, 4])
        dyaw = 0.2 + 0 * tt
        return roll, pitch, yaw, droll, dpitch, dyaw

    roll, pitch, yaw, droll, dpitch, dyaw = euler(t)
    R = _rot_zyx(yaw, pitch, roll)  # [K,B,3,3]
    omega = np.stack([droll - dyaw * np.sin(pitch),
                      dpitch * np.cos(roll) + dyaw * np.sin(roll) * np.cos(pitch),
                      -dpitch * np.sin(roll) + dyaw * np.cos(roll) * np.cos(pitch)], axis=-1)
    v_s = vel(t)
    a_s = acc(t)
    n_gyro = np.stack([r.normal(0, 0.03, (K, 3)) for r in rngs], axis=1)
    n_acc = np.stack([r.normal(0, 0.025, (K, 3)) for r in rngs], axis=1)
    gyro = omega + n_gyro
    accel = np.einsum("kbji,kbj->kbi", R, a_s - G_S) + bias[None] + n_acc

    # legs
    gait_phase = {4: (0.0, 0.5, 0.5, 0.0), 2: (0.0, 0.5), 1: (0.0,)}.get(L, tuple(i / L for i in range(L)))
    contact = np.empty((K, B, L))
    p_foot = np.empty((K, B, L, 3))
    J = np.empty((K, B, L, 3, nj))
    qd = np.empty((K, B, L, nj))
    q_joint = np.empty((K, B, L, nj))
    nominal = np.array([0.0] + [0.8, -1.6] * nj)[:nj]
    n_qd = np.stack([r.normal(0, 1.0, (K, L, nj)) for r in rngs], axis=1)
    for leg in range(L):
        cyc = (gait_b * t + gait_phase[leg] + ph[:, 5] / (2 * np.pi)) % 1.0
        contact[:, :, leg] = (cyc < 0.6).astype(np.float64)
        q = nominal + 0.3 * np.sin(2 * np.pi * gait_b[:, None] * t[..., None] + ph[:, 6, None] + leg + np.arange(nj))
        q_joint[:, :, leg] = q
        p, Jl = leg_kinematics(q, leg, nj)
        p_foot[:, :, leg] = p
        J[:, :, leg] = Jl
        rhs = -np.einsum("kbji,kbj->kbi", R, v_s) - np.cross(omega, p)
        qd_st = np.einsum("kbij,kbj->kbi", np.linalg.pinv(Jl), rhs)
        st = contact[:, :, leg, None]
        qd[:, :, leg] = st * (qd_st + 0.22 * n_qd[:, :, leg]) + (1 - st) * n_qd[:, :, leg]
    foot_force = 20.0 + 180.0 * contact

    out = dict(imu_t=imu_t, accel=accel, gyro=gyro, p_foot=p_foot, J=J, qdot=qd, contact=contact,
               q_joint=q_joint, foot_force=foot_force,
               gt_v_s=v_s, gt_R=R, gt_quat=_quat_from_rot(R), gt_p=pos(t), gt_bias=bias)

    # visual odometry: frame pairs at vo_rate, delivered vo_latency after the newer frame
    vo_mask = np.zeros((K, B), dtype=np.int32)
    vo_t_pre = np.zeros((K, B))
    vo_t_now = np.zeros((K, B))
    vo_dp = np.zeros((K, B, 3))
    vo_t_pose = np.zeros((K, B))
    vo_q = np.zeros((K, B, 4))
    vo_q[..., 0] = 1.0
    if vo:
        n_dp = [r.normal(0, 1.5e-5, (K, 3)) for r in rngs]
        n_q = [r.normal(0, 1e-4, (K, 4)) for r in rngs]
        nframes = int(np.floor((K - 1) * dt * float(np.max(vo_rate_b)))) + 1
        ib = np.arange(B)
        for f in range(1, nframes):
            tp = vo_phase + (f - 1) / vo_rate_b                                 # [B]: every instance's own camera clock
            tn = vo_phase + f / vo_rate_b
            kk = np.ceil((tn + vo_lat) / dt - 1e-9).astype(np.int64)            # the tick at which the pair is delivered
            ok = kk < K
            if not ok.any():
                break
            r_, p_, y_, _, _, _ = euler(tp)
            R_pre = _rot_zyx(y_, p_, r_)
            r2, p2, y2, _, _, _ = euler(tn)
            q_now = _quat_from_rot(_rot_zyx(y2, p2, r2))
            dp = np.einsum("bji,bj->bi", R_pre, pos(tn) - pos(tp)) + np.stack([n[f] for n in n_dp])
            qn = q_now + np.stack([n[f] for n in n_q])
            qn = qn / np.linalg.norm(qn, axis=-1, keepdims=True)
            kv, iv = kk[ok], ib[ok]
            vo_mask[kv, iv] = 1
            vo_t_pre[kv, iv] = tp[ok]
            vo_t_now[kv, iv] = tn[ok]
            vo_t_pose[kv, iv] = tn[ok]
            vo_dp[kv, iv] = dp[ok]
            vo_q[kv, iv] = qn[ok]
    out.update(vo_mask=vo_mask, vo_t_pre=vo_t_pre, vo_t_now=vo_t_now, vo_dp=vo_dp,
               vo_t_pose=vo_t_pose, vo_q=vo_q)
    for key, val in out.items():
        if isinstance(val, np.ndarray) and val.dtype != np.int32:
            out[key] = np.ascontiguousarray(val, dtype=np.float64)
    return out
