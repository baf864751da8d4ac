// go1_kin.h — Go1 leg odometry front-end on the device: what go1Sub::lo_callback computes
// on the host (src/go1_example/src/go1Sub.cpp:53-126) from raw joint states — foot position
// and 3x3 Jacobian per leg in the IMU frame, contact from the foot-force threshold.
// The reference gets FK/J from ~5 kLoC of FROST/Mathematica-generated expressions
// (src/go1_example/src/Expressions/*.cc); here the same serial chain is written out by hand
// (hip abduction about x, thigh and calf about y; link constants identified from, and
// checked against, golden vectors of the reference's own compiled code:
// tests/golden/go1_kin.npz, tests/test_golden.py::test_go1_kinematics_against_reference_vectors).
// Leg order FR, FL, RR, RL; joint order hip, thigh, calf (go1Sub.cpp:84-85).
#pragma once
#include "cfg.h"

namespace dekf {

constexpr double GO1_HIP_X = 0.1881, GO1_HIP_Y = 0.04675, GO1_ABD = 0.08, GO1_THIGH = 0.213, GO1_CALF = 0.213;

// q[3] -> p[3] (without p_ib), J[3][3] row-major
DEKF_FN void go1_leg_fk(int leg, const double* q, double* p, double* J) {
    const double sx = (leg < 2) ? 1.0 : -1.0;
    const double sy = (leg & 1) ? 1.0 : -1.0;
    const double hip[3] = {sx * GO1_HIP_X, sy * GO1_HIP_Y, 0.0};
    const double c0 = cos(q[0]), s0 = sin(q[0]);
    const double a1 = q[1], a2 = q[1] + q[2];
    // link vectors in the hip-roll frame, then rotated by Rx(q0): (x, y, z) -> (x, c0 y - s0 z, s0 y + c0 z)
    auto rx = [&](double x, double y, double z, double* o) { o[0] = x; o[1] = c0 * y - s0 * z; o[2] = s0 * y + c0 * z; };
    double d0[3], d1[3], d2[3];
    rx(0.0, sy * GO1_ABD, 0.0, d0);
    rx(-GO1_THIGH * sin(a1), 0.0, -GO1_THIGH * cos(a1), d1);
    rx(-GO1_CALF * sin(a2), 0.0, -GO1_CALF * cos(a2), d2);
    double o1[3], o2[3];
    for (int i = 0; i < 3; ++i) { o1[i] = hip[i] + d0[i]; o2[i] = o1[i] + d1[i]; p[i] = o2[i] + d2[i]; }
    const double ax[3] = {1.0, 0.0, 0.0}, ay[3] = {0.0, c0, s0};
    double r0[3] = {p[0] - hip[0], p[1] - hip[1], p[2] - hip[2]};
    double r1[3] = {p[0] - o1[0], p[1] - o1[1], p[2] - o1[2]};
    double r2[3] = {p[0] - o2[0], p[1] - o2[1], p[2] - o2[2]};
    double c[3];
    cross3(ax, r0, c); J[0] = c[0]; J[3] = c[1]; J[6] = c[2];
    cross3(ay, r1, c); J[1] = c[0]; J[4] = c[1]; J[7] = c[2];
    cross3(ay, r2, c); J[2] = c[0]; J[5] = c[1]; J[8] = c[2];
}

// raw joint_position[12], joint_velocity[12], foot_force[4] of instance b -> sensor latch
DEKF_FN void go1_leg_odometry(const DevState& s, int b, const double* jp, const double* jv, const double* force,
                              double threshold, const double* p_ib) {
    for (int leg = 0; leg < 4; ++leg) {
        double p[3], J[9];
        go1_leg_fk(leg, jp + 12 * (size_t)b + 3 * leg, p, J);
        for (int i = 0; i < 3; ++i) s.p_foot[(size_t)b * 12 + 3 * leg + i] = p[i] + p_ib[i];
        for (int i = 0; i < 9; ++i) s.J[(size_t)b * 36 + 9 * leg + i] = J[i];
        for (int i = 0; i < 3; ++i) s.qdot[(size_t)b * 12 + 3 * leg + i] = jv[12 * (size_t)b + 3 * leg + i];
        s.contact[(size_t)b * 4 + leg] = force[4 * (size_t)b + leg] >= threshold ? 1.0 : 0.0;
    }
}

}  // namespace dekf
