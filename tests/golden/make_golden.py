#!/usr/bin/env python3
"""Regenerates the committed fixtures in tests/golden/ (run in the build container, where
/root/reference exists; nothing here runs on the GPU box).

go1_kin.npz   TRUE reference vectors: inputs/outputs of the reference's own FROST-generated Go1
              kinematics (src/go1_example/src/Expressions/*.cc compiled as oracle/_ref/libgo1kin.so),
              driven exactly as go1Sub::lo_callback drives them (go1Sub.cpp:64-125).
ekf_trace.npz / mhe_trace.npz
              regression pins produced by the CPU oracle (the reference ships no vectors for these
              paths and cannot be built here — SURVEY.md §8c); inputs + expected outputs.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def go1_kin():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libgo1kin.so"))
    dp = C.POINTER(C.c_double)

    def call(sym, var, n):
        f = getattr(lib, sym)
        f.argtypes, f.restype = [dp, dp], None
        out = np.zeros(n)
        f(out.ctypes.data_as(dp), var.ctypes.data_as(dp))
        return out

    legs = ["FR", "FL", "RR", "RL"]
    rng = np.random.default_rng(20240510)
    n = 96
    q = rng.uniform(-1.0, 1.0, (n, 12)) * np.tile([0.8, 1.2, 1.0], 4) + np.tile([0.0, 0.8, -1.6], 4)
    q[0] = 0.0
    p_ib = np.array([0.01592, 0.06659, 0.00617])
    p = np.zeros((n, 4, 3))
    J = np.zeros((n, 4, 3, 3))
    for i in range(n):
        var = np.zeros(22)
        for l in range(4):
            var[6 + 4 * l:6 + 4 * l + 3] = q[i, 3 * l:3 * l + 3]
        for l, name in enumerate(legs):
            p[i, l] = call(f"_ZN11SymFunction{len(name) + 9}{name}_foot_rawEPdPKd", var, 3) + p_ib
            Jfull = call(f"_ZN11SymFunction{len(name) + 6}J_{name}_rawEPdPKd", var, 66).reshape(22, 3).T  # 3x22 col-major
            J[i, l] = Jfull[:, 6 + 4 * l:6 + 4 * l + 3]
    np.savez_compressed(os.path.join(HERE, "go1_kin.npz"), joint_position=q, p_ib=p_ib, p_imu_2_foot=p, J_imu_2_foot=J)
    print("go1_kin.npz", p.shape, J.shape)


def ekf_trace():
    import oracle_lib as O
    from decentralized_ekf_mhe_amd import go1_params
    p = go1_params()  # 500 Hz
    rng = np.random.default_rng(77)
    K = 1000
    t = np.arange(K) / p.ekf_rate + rng.uniform(0, 1e-4, K)
    gyro = 0.4 * np.sin(np.arange(K)[:, None] * 0.02 + np.arange(3)) + rng.normal(0, 0.03, (K, 3))
    accel = np.array([0, 0, 9.81]) + rng.normal(0, 0.3, (K, 3))
    vo_step = np.zeros(K, np.int32)
    vo_t = np.zeros(K)
    vo_q = np.zeros((K, 4))
    for k in range(50, K, 17):       # ~30 Hz, latency 0..40 ms -> rel 0, 1, ..., 20
        lat = int(rng.integers(0, 21))
        q = rng.normal(size=4) * 0.01 + [1, 0, 0, 0]
        vo_step[k], vo_t[k], vo_q[k] = 1, t[k - lat] + 1e-6, q / np.linalg.norm(q)
    e = O.Ekf(p)
    qs, Ps, rp = np.zeros((K, 4)), np.zeros((K, 4, 4)), np.zeros(K, np.int32)
    for k in range(K):
        e.set_imu(t[k], accel[k], gyro[k])
        if vo_step[k]:
            e.set_vo(vo_t[k], vo_q[k])
        e.step()
        qs[k], Ps[k] = e.get()
        rp[k] = e.last_replay()
    np.savez_compressed(os.path.join(HERE, "ekf_trace.npz"), t=t, gyro=gyro, accel=accel, vo_step=vo_step, vo_t=vo_t,
                        vo_q=vo_q, quat=qs, cov=Ps, replay=rp)
    print("ekf_trace.npz", K, "max replay", rp.max())


def mhe_trace():
    import oracle_lib as O
    from decentralized_ekf_mhe_amd import go1_params
    from decentralized_ekf_mhe_amd.streams import make_streams
    p = go1_params()
    p.ekf_rate = p.rate
    B, K = 3, 64
    s = make_streams(p, B, K)
    x, vb, q, _, it = O.run_streams(p, s, want_iters=True)
    keep = ("imu_t", "accel", "gyro", "p_foot", "J", "qdot", "contact", "vo_mask", "vo_t_pre", "vo_t_now", "vo_dp",
            "vo_t_pose", "vo_q")
    np.savez_compressed(os.path.join(HERE, "mhe_trace.npz"), x=x, v_b=vb, quat=q, iters=it, **{k: s[k] for k in keep})
    print("mhe_trace.npz", x.shape)


if __name__ == "__main__":
    go1_kin()
    ekf_trace()
    mhe_trace()
