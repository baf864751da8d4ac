// go1_shim_demo.cpp — the reference's call pattern (EstSub.cpp:58-91: fill robot_store, then
// mhe.initialize() at T == 0 and mhe.update(T) afterwards) against the MI355X library through the
// source-compatible shim.  Reads a binary sensor log (81 doubles per step, see tests/test_cpp_shim.py)
// and prints x_MHE_ and v_MHE_b_ per step.
//
//   g++ -std=c++17 -O2 examples/go1_shim_demo.cpp -o go1_shim_demo \
//       -Ldecentralized_ekf_mhe_amd/csrc -ldekf -Wl,-rpath,$PWD/decentralized_ekf_mhe_amd/csrc
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../decentralized_ekf_mhe_amd/cpp/DecentralEst.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s log.bin steps [est_type]\n", argv[0]); return 2; }
    const int K = std::atoi(argv[2]);
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("log"); return 2; }
    std::vector<double> log((size_t)K * 81);
    if (std::fread(log.data(), sizeof(double), log.size(), f) != log.size()) { std::fprintf(stderr, "short log\n"); return 2; }
    std::fclose(f);

    auto store = std::make_shared<robot_store>();
    auto params = std::make_shared<robot_params>();
    // parameters_go1.yaml
    params->p_init_std_ = {0.001, 0.001, 0.001};
    params->v_init_std_ = {0.001, 0.001, 0.001};
    params->foot_init_std_ = {0.001, 0.001, 0.001};
    params->accel_bias_init_std_ = {0.0001, 0.0001, 0.0001};
    params->p_process_std_ = {0.001, 0.001, 0.001};
    params->accel_input_std_ = {0.025, 0.025, 0.02};
    params->gyro_input_std_ = {0.03, 0.03, 0.03};
    params->accel_bias_std_ = {0.07, 0.02, 0.03};
    params->quaternion_ib_ = {1.0, 0.0, 0.0, 0.0};
    params->p_ib_ = {0.01592, 0.06659, 0.00617};
    params->joint_position_std_ = {0.04, 0.04, 0.04};
    params->joint_velocity_std_ = {0.22, 0.22, 0.22};
    params->foot_slide_std_ = {0.003, 0.003, 0.003};
    params->foot_swing_std_ = {1.0e7, 1.0e7, 1.0e7};
    params->vo_p_std_ = {0.000015, 0.000015, 0.000015};
    if (argc > 3) params->est_type_ = std::atoi(argv[3]);

    store->joint_states_velocity_ = VectorXd(12);
    store->contact_ = VectorXd(4);
    store->p_imu_2_foot_ = MatrixXd(12, 1);
    store->J_imu_2_foot_ = MatrixXd(12, 3);

    DecentralizedEstimation mhe;
    for (int T = 0; T < K; ++T) {
        const double* r = log.data() + (size_t)T * 81;
        store->imu_time_ = r[0];
        for (int i = 0; i < 3; ++i) { store->accel_b_(i) = r[1 + i]; store->angular_b_(i) = r[4 + i]; }
        store->quaternion_.w() = r[7]; store->quaternion_.x() = r[8]; store->quaternion_.y() = r[9]; store->quaternion_.z() = r[10];
        for (int i = 0; i < 12; ++i) {
            store->p_imu_2_foot_(i, 0) = r[11 + i];
            for (int j = 0; j < 3; ++j) store->J_imu_2_foot_(i, j) = r[23 + 3 * i + j];
            store->joint_states_velocity_(i) = r[59 + i];
        }
        for (int i = 0; i < 4; ++i) store->contact_(i) = r[71 + i];
        if (r[75] != 0.0) {
            store->vo_new_ = true;
            store->vo_time_pre_ = r[76];
            store->vo_time_now_ = r[77];
            for (int i = 0; i < 3; ++i) store->vo_p_body_pre_2_body_(i) = r[78 + i];
        }
        if (T == 0) mhe.initialize(store, params);
        else mhe.update(T);
        const VectorXd& x = params->est_type_ == 0 ? mhe.x_MHE_ : mhe.x_KF_;
        const Vector3d& v = params->est_type_ == 0 ? mhe.v_MHE_b_ : mhe.v_KF_b_;
        std::printf("%d", T);
        for (int i = 0; i < 9; ++i) std::printf(" %.17g", x(i));
        for (int i = 0; i < 3; ++i) std::printf(" %.17g", v(i));
        std::printf(" %d\n", mhe.solver_iters_);
    }
    return 0;
}
