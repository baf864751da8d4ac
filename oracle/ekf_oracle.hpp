// TEST INFRASTRUCTURE ONLY — part of the CPU oracle (see oracle/README.md).
//
// fp64 restatement of the orien_est quaternion EKF, single instance:
//   timer step            src/orien_est/src/orien_ekf.cpp:77-106
//   predict               :108-123   (gyro_2_Ohm :214-228, quat_2_W :270-294)
//   accel correct         :125-142   (quat_2_Rot :296-305, quat_2_H :307-329)
//   VO correct            :144-154
//   history + VO rewind   :156-212
// Reproduced on purpose:
//   * quat_2_W assigns W(2,1), W(2,2) twice and never W(3,1), W(3,2)  (:289-291)
//   * the replay loop runs rel-1 iterations and applies the VO correction inside it at
//     i == 0, so nothing happens when rel < 2                          (:191-205)
//   * the history vectors are never trimmed.
// parity: the reference ships no test for this path; pins are tests/test_oracle_ekf.py
// (zero-innovation fixed point, numpy cross-implementation, golden trace).
#pragma once
#include <algorithm>
#include <vector>

#include "densemat.hpp"

namespace orc {

struct EkfParams {
    double init_std[4], process_std[3], gravity_meas_std[3], vo_meas_std[4], quaternion_init[4];
    int rate;
};

class EkfOracle {
  public:
    double dt;
    Vec gravity{0, 0, 9.81};
    Mat Cov_q, C_gyro, C_accel, C_vo;
    Vec quat;  // w x y z
    // latched sensor sample
    Vec accel_b{0, 0, 0}, gyro_b{0, 0, 0};
    double imu_time = 0;
    bool init_imu = false;
    // latched VO pose
    bool vo_new = false;
    double vo_time = 0;
    Vec vo_quat{1, 0, 0, 0};
    int discrete_time = 0;
    int last_replay = 0;  // diagnostics: replay iterations of the last step
    // history
    std::vector<Vec> gyro_hist, accel_hist, quat_hist;
    std::vector<Mat> cov_hist;
    std::vector<double> time_hist;
    std::vector<int> dtime_hist;

    explicit EkfOracle(const EkfParams& p) {
        dt = 1.0 / (double)p.rate;
        Cov_q = Mat(4, 4); C_gyro = Mat(3, 3); C_accel = Mat(3, 3); C_vo = Mat(4, 4);
        for (int i = 0; i < 4; ++i) { Cov_q(i, i) = p.init_std[i] * p.init_std[i]; C_vo(i, i) = p.vo_meas_std[i] * p.vo_meas_std[i]; }
        for (int i = 0; i < 3; ++i) { C_gyro(i, i) = p.process_std[i] * p.process_std[i]; C_accel(i, i) = p.gravity_meas_std[i] * p.gravity_meas_std[i]; }
        quat = Vec(p.quaternion_init, p.quaternion_init + 4);
    }

    void set_imu(double t, const double* a, const double* w) {
        imu_time = t;
        accel_b = Vec(a, a + 3);
        gyro_b = Vec(w, w + 3);
        init_imu = true;
    }
    void set_vo(double t, const double* q_wxyz) {
        vo_time = t;
        vo_quat = Vec(q_wxyz, q_wxyz + 4);
        vo_new = true;
    }

    static void normalize(Vec& q) {
        double n = norm2(q);
        for (auto& v : q) v /= n;
    }
    static Mat omega_matrix(const Vec& w) {
        Mat O(4, 4);
        O(0, 1) = -w[0]; O(0, 2) = -w[1]; O(0, 3) = -w[2];
        O(1, 0) = w[0];  O(2, 0) = w[1];  O(3, 0) = w[2];
        O(1, 2) = w[2];  O(1, 3) = -w[1]; O(2, 3) = w[0];
        O(2, 1) = -w[2]; O(3, 1) = w[1];  O(3, 2) = -w[0];
        return O;
    }
    Mat w_matrix(const Vec& q) const {
        Mat W(4, 3);
        W(0, 0) = -q[1]; W(0, 1) = -q[2]; W(0, 2) = -q[3];
        W(1, 0) = q[0];  W(1, 1) = -q[3]; W(1, 2) = q[2];
        W(2, 0) = q[3];  W(2, 1) = q[0];  W(2, 2) = -q[1];
        W(3, 0) = -q[2];
        // the reference's last two assignments land on row 2, not row 3
        W(2, 1) = q[1];
        W(2, 2) = q[0];
        return (0.5 * dt) * W;
    }
    Mat h_matrix(const Vec& q) const {
        double w = q[0], x = q[1], y = q[2], z = q[3];
        double g0 = gravity[0], g1 = gravity[1], g2 = gravity[2];
        Mat H(3, 4);
        H(0, 0) = g0 * w + g1 * z - g2 * y;  H(0, 1) = g0 * x + g1 * y + g2 * z;
        H(0, 2) = -g0 * y + g1 * x - g2 * w; H(0, 3) = -g0 * z + g1 * w + g2 * x;
        H(1, 0) = -g0 * z + g1 * w + g2 * x; H(1, 1) = g0 * y - g1 * x + g2 * w;
        H(1, 2) = g0 * x + g1 * y + g2 * z;  H(1, 3) = -g0 * w - g1 * z + g2 * y;
        H(2, 0) = g0 * y - g1 * x + g2 * w;  H(2, 1) = g0 * z - g1 * w - g2 * x;
        H(2, 2) = g0 * w + g1 * z - g2 * y;  H(2, 3) = g0 * x + g1 * y + g2 * z;
        return 2.0 * H;
    }

    void predict(Vec& q_pred, const Vec& q, const Vec& gyro, const Mat& Cov, Mat& Cov_pred) const {
        Mat W = w_matrix(q);
        Mat F = Mat::identity(4) + (dt / 2) * omega_matrix(gyro);
        q_pred = F * q;
        Cov_pred = F * Cov * F.T() + W * C_gyro * W.T();
        normalize(q_pred);
    }
    void correct(Vec& q_corr, const Vec& q_pred, const Vec& accel, const Mat& Cov_pred, Mat& Cov_corr) const {
        Mat R = quat_to_rot_normalized(q_pred[0], q_pred[1], q_pred[2], q_pred[3]);
        Vec a_hat = R.T() * gravity;
        Mat H = h_matrix(q_pred);
        double rel = norm2(accel) / norm2(gravity);
        Mat S = H * Cov_pred * H.T() + (rel * rel) * C_accel;
        Mat K = Cov_pred * H.T() * inverse(S);
        q_corr = q_pred + K * (accel - a_hat);
        Cov_corr = (Mat::identity(4) - K * H) * Cov_pred;
        normalize(q_corr);
    }
    void vo_correct(Vec& q_corr, const Vec& q_pred, const Vec& q_vo, const Mat& Cov_pred, Mat& Cov_corr) const {
        Mat H = Mat::identity(4);
        Mat K = Cov_pred * H.T() * inverse(H * Cov_pred * H.T() + C_vo);
        q_corr = q_pred + K * (q_vo - q_pred);
        Cov_corr = (Mat::identity(4) - K * H) * Cov_pred;
        normalize(q_corr);
    }

    void get_measurement() {
        gyro_hist.push_back(gyro_b);
        accel_hist.push_back(accel_b);
        time_hist.push_back(imu_time);
        dtime_hist.push_back(discrete_time);
        quat_hist.push_back(quat);
        cov_hist.push_back(Cov_q);
        last_replay = 0;
        if (vo_new && !time_hist.empty()) {
            vo_new = false;
            auto it = std::upper_bound(time_hist.begin(), time_hist.end(), vo_time);
            if (it == time_hist.begin()) return;  // VO older than every stored IMU sample: dropped
            int idx = (int)(it - time_hist.begin()) - 1;
            int rel = dtime_hist.back() - dtime_hist[idx];
            quat = quat_hist[idx];
            Cov_q = cov_hist[idx];
            Vec qp, qc;
            Mat Cp, Cc;
            for (int i = 0; i < rel - 1; ++i) {
                predict(qp, quat, gyro_hist[idx + i], Cov_q, Cp);
                correct(qc, qp, accel_hist[idx + i], Cp, Cc);
                if (i == 0) {
                    qp = qc; Cp = Cc;
                    vo_correct(qc, qp, vo_quat, Cp, Cc);
                }
                quat = qc;
                Cov_q = Cc;
                last_replay++;
            }
        }
    }

    void step() {
        if (!init_imu) return;
        get_measurement();
        Vec qp, qc;
        Mat Cp, Cc;
        predict(qp, quat, gyro_b, Cov_q, Cp);
        correct(qc, qp, accel_b, Cp, Cc);
        quat = qc;
        Cov_q = Cc;
        discrete_time++;
    }
};

}  // namespace orc
