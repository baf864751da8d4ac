// ekf_core.h — quaternion EKF, one lane (thread) per robot instance.
//
// Replaces orien_ekf::timerCallback and what it calls
//   (src/orien_est/src/orien_ekf.cpp:77-106; predict :108-123, accel correct :125-142,
//    VO correct :144-154, history + rewind :156-212, helpers :214-357).
// Reproduced on purpose: the W matrix as quat_2_W really builds it (rows 2/3, :289-291), the
// rel-1 replay loop with the VO correction inside it at i == 0 (:191-205).
// Bounded where the reference is not: the history is a ring of cfg.ekf_hist samples instead
// of ever-growing vectors; a VO pose older than the ring is dropped like one older than the
// first sample is (:176-183).
// State is field-major ([field][B]) so the 64 lanes of a wave read consecutive doubles.
#pragma once
#include "cfg.h"
#include "smallmat.h"

namespace dekf {

constexpr int EKF_HIST_REC = 27;  // gyro3 accel3 t q4 P16

struct Ekf4 {
    double q[4];
    double P[16];
};

DEKF_FN void ekf_normalize(double* q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}

DEKF_FN void ekf_predict(const DevCfg& c, Ekf4& e, const double* w) {
    const double h = 0.5 * c.ekf_dt;
    const double* q = e.q;
    // F = I + dt/2 Omega(w)
    double F[16] = {1, -h * w[0], -h * w[1], -h * w[2],
                    h * w[0], 1, h * w[2], -h * w[1],
                    h * w[1], -h * w[2], 1, h * w[0],
                    h * w[2], h * w[1], -h * w[0], 1};
    // W as the reference assembles it: rows [-x -y -z; w -z y; z x w; -y 0 0] * dt/2
    double W[12] = {-q[1], -q[2], -q[3], q[0], -q[3], q[2], q[3], q[1], q[0], -q[2], 0.0, 0.0};
    for (int i = 0; i < 12; ++i) W[i] *= h;
    double qn[4];
    for (int i = 0; i < 4; ++i) qn[i] = F[4 * i] * q[0] + F[4 * i + 1] * q[1] + F[4 * i + 2] * q[2] + F[4 * i + 3] * q[3];
    double FP[16], Pn[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += F[4 * i + t] * e.P[4 * t + j];
            FP[4 * i + j] = s;
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += FP[4 * i + t] * F[4 * j + t];
            for (int t = 0; t < 3; ++t) s += W[3 * i + t] * c.ekf_Cgyro[t] * W[3 * j + t];
            Pn[4 * i + j] = s;
        }
    for (int i = 0; i < 4; ++i) e.q[i] = qn[i];
    for (int i = 0; i < 16; ++i) e.P[i] = Pn[i];
    ekf_normalize(e.q);
}

DEKF_FN void ekf_correct(const DevCfg& c, Ekf4& e, const double* a) {
    const double g = 9.81;
    double R[9];
    quat_to_rot(e.q, R);
    double ah[3] = {R[6] * g, R[7] * g, R[8] * g};  // R' * (0,0,g)
    double w = e.q[0], x = e.q[1], y = e.q[2], z = e.q[3];
    double H[12] = {-g * y, g * z, -g * w, g * x,
                    g * x, g * w, g * z, g * y,
                    g * w, -g * x, -g * y, g * z};
    for (int i = 0; i < 12; ++i) H[i] *= 2.0;
    double rel2 = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) / (g * g);
    double PHt[12];  // 4x3
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += e.P[4 * i + t] * H[4 * j + t];
            PHt[3 * i + j] = s;
        }
    double S[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += H[4 * i + t] * PHt[3 * t + j];
            S[3 * i + j] = s + (i == j ? rel2 * c.ekf_Caccel[i] : 0.0);
        }
    inv_small<3>(S, 3);
    double K[12];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) K[3 * i + j] = PHt[3 * i] * S[j] + PHt[3 * i + 1] * S[3 + j] + PHt[3 * i + 2] * S[6 + j];
    double inn[3] = {a[0] - ah[0], a[1] - ah[1], a[2] - ah[2]};
    for (int i = 0; i < 4; ++i) e.q[i] += K[3 * i] * inn[0] + K[3 * i + 1] * inn[1] + K[3 * i + 2] * inn[2];
    double IKH[16], Pn[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int t = 0; t < 3; ++t) s -= K[3 * i + t] * H[4 * t + j];
            IKH[4 * i + j] = s;
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += IKH[4 * i + t] * e.P[4 * t + j];
            Pn[4 * i + j] = s;
        }
    for (int i = 0; i < 16; ++i) e.P[i] = Pn[i];
    ekf_normalize(e.q);
}

DEKF_FN void ekf_vo_correct(const DevCfg& c, Ekf4& e, const double* qv) {
    double S[16];
    for (int i = 0; i < 16; ++i) S[i] = e.P[i];
    for (int i = 0; i < 4; ++i) S[5 * i] += c.ekf_Cvo[i];
    inv_small<4>(S, 4);
    double K[16], Pn[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += e.P[4 * i + t] * S[4 * t + j];
            K[4 * i + j] = s;
        }
    double d[4] = {qv[0] - e.q[0], qv[1] - e.q[1], qv[2] - e.q[2], qv[3] - e.q[3]};
    for (int i = 0; i < 4; ++i) e.q[i] += K[4 * i] * d[0] + K[4 * i + 1] * d[1] + K[4 * i + 2] * d[2] + K[4 * i + 3] * d[3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int t = 0; t < 4; ++t) s += ((i == t ? 1.0 : 0.0) - K[4 * i + t]) * e.P[4 * t + j];
            Pn[4 * i + j] = s;
        }
    for (int i = 0; i < 16; ++i) e.P[i] = Pn[i];
    ekf_normalize(e.q);
}

// one timer tick of instance b; `count` = ticks done so far (uniform over the batch)
DEKF_FN void ekf_tick(const DevCfg& c, const DevState& s, int b, int count) {
    const size_t B = (size_t)c.B;
    const int H = c.ekf_hist;
    Ekf4 e;
    for (int i = 0; i < 4; ++i) e.q[i] = s.ekf_q[i * B + b];
    for (int i = 0; i < 16; ++i) e.P[i] = s.ekf_P[i * B + b];
    double w[3], a[3];
    for (int i = 0; i < 3; ++i) { w[i] = s.gyro[3 * (size_t)b + i]; a[i] = s.accel[3 * (size_t)b + i]; }
    double t = s.imu_t[b];
    // get_measurement: push (gyro, accel, t, q, P)
    double* slot = s.ekf_hist + (size_t)(count % H) * EKF_HIST_REC * B + b;
    for (int i = 0; i < 3; ++i) { slot[i * B] = w[i]; slot[(3 + i) * B] = a[i]; }
    slot[6 * B] = t;
    for (int i = 0; i < 4; ++i) slot[(7 + i) * B] = e.q[i];
    for (int i = 0; i < 16; ++i) slot[(11 + i) * B] = e.P[i];
    if (s.ekf_vo_flag[b]) {
        s.ekf_vo_flag[b] = 0;
        double tv = s.ekf_vo_t[b];
        double qv[4];
        for (int i = 0; i < 4; ++i) qv[i] = s.ekf_vo_q[4 * (size_t)b + i];
        int oldest = count + 1 > H ? count + 1 - H : 0;
        // newest stored sample with time <= tv  (upper_bound - 1 on a sorted stack)
        int idx = -1;
        for (int i = count; i >= oldest; --i) {
            double ti = i == count ? t : s.ekf_hist[((size_t)(i % H) * EKF_HIST_REC + 6) * B + b];
            if (ti <= tv) { idx = i; break; }
        }
        if (idx >= 0) {
            int rel = count - idx;
            const double* hs = s.ekf_hist + (size_t)(idx % H) * EKF_HIST_REC * B + b;
            for (int i = 0; i < 4; ++i) e.q[i] = hs[(7 + i) * B];
            for (int i = 0; i < 16; ++i) e.P[i] = hs[(11 + i) * B];
            for (int i = 0; i < rel - 1; ++i) {
                const double* hi = s.ekf_hist + (size_t)((idx + i) % H) * EKF_HIST_REC * B + b;
                double wi[3] = {hi[0], hi[B], hi[2 * B]};
                double ai[3] = {hi[3 * B], hi[4 * B], hi[5 * B]};
                ekf_predict(c, e, wi);
                ekf_correct(c, e, ai);
                if (i == 0) ekf_vo_correct(c, e, qv);
            }
        }
    }
    ekf_predict(c, e, w);
    ekf_correct(c, e, a);
    for (int i = 0; i < 4; ++i) { s.ekf_q[i * B + b] = e.q[i]; s.quat[4 * (size_t)b + i] = e.q[i]; }
    for (int i = 0; i < 16; ++i) s.ekf_P[i * B + b] = e.P[i];
}

}  // namespace dekf
