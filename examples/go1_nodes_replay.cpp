// go1_nodes_replay.cpp — both nodes of the reference's Go1 launch file (orien_sub and est_sub,
// src/go1_example/launch/go1_launch.py) wired together in one process without ROS2: a recorded sequence
// of "messages" and timer ticks is replayed through the node cores, the imu/filter message orien_sub
// publishes is handed to est_sub's subscription, and est_sub writes the reference's log files.
//
//   go1_nodes_replay params.yaml events.bin time_init log_dir
//
// events.bin: records of 32 doubles, [kind, clock, payload...]
//   kind 1  unitree/imu            linear_acceleration[3] angular_velocity[3]      (to both nodes)
//   kind 2  /unitree/joint_state   position[16] velocity[12]
//   kind 4  orb/vo                 stamp_pre stamp_now x y z
//   kind 5  /mocap/RigidBody       position[3] velocity[3] quaternion wxyz[4]
//   kind 6  orb/pos                stamp x y z w
//   kind 7  orien_sub timer tick   (publishes imu/filter -> est_sub)
//   kind 0  est_sub timer tick
// Prints one line per est_sub tick that ran: T, x (9), v_body (3), solver status, and one line
// "q k w x y z" per orien_sub tick.
//
//   g++ -std=c++17 -O2 examples/go1_nodes_replay.cpp -o go1_nodes_replay
//       -Ldecentralized_ekf_mhe_amd/csrc -ldekf -Wl,-rpath,$PWD/decentralized_ekf_mhe_amd/csrc
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../decentralized_ekf_mhe_amd/cpp/est_node_core.hpp"
#include "../decentralized_ekf_mhe_amd/cpp/orien_node_core.hpp"
#include "../decentralized_ekf_mhe_amd/cpp/ros_params.hpp"

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s params.yaml events.bin time_init log_dir\n", argv[0]); return 2; }
    const double time_init = std::atof(argv[3]);
    std::FILE* f = std::fopen(argv[2], "rb");
    if (!f) { std::perror("events"); return 2; }
    std::vector<double> ev;
    double rec[32];
    while (std::fread(rec, sizeof(double), 32, f) == 32) ev.insert(ev.end(), rec, rec + 32);
    std::fclose(f);

    try {
        dekf_ros::ParamNode est_params = dekf_ros::ParamNode::from_file(argv[1], "est_sub");
        dekf_ros::ParamNode orien_params = dekf_ros::ParamNode::from_file(argv[1], "orien_sub");
        robotSub::Go1NodeCore est(est_params, time_init);
        orien_ekf::OrienNodeCore orien(orien_params, time_init);
        est.log_dir_ = argv[4];
        for (const std::string& name : est_params.undeclared_overrides()) std::fprintf(stderr, "est_sub: unknown parameter %s\n", name.c_str());

        int orien_ticks = 0;
        for (size_t e = 0; e < ev.size() / 32; ++e) {
            const double* r = ev.data() + 32 * e;
            const int kind = (int)r[0];
            const double clock = r[1];
            const double* p = r + 2;
            if (kind == 1) {
                orien.imu_callback(clock, p, p + 3);
                est.imu_callback(clock, p, p + 3);
            } else if (kind == 2) {
                est.lo_callback(std::vector<double>(p, p + 16), std::vector<double>(p + 16, p + 28));
            } else if (kind == 4) {
                est.vo_callback(p[0], p[1], p[2], p[3], p[4]);
            } else if (kind == 5) {
                est.mocap_callback(p, p + 3, p + 6);
            } else if (kind == 6) {
                orien.vo_pose_callback(p[0], p[1], p[2], p[3], p[4]);
            } else if (kind == 7) {
                const orien_ekf::FilterMsg& m = orien.timerCallback();
                std::printf("q %d %.17g %.17g %.17g %.17g\n", orien_ticks++, m.orientation_wxyz[0], m.orientation_wxyz[1],
                            m.orientation_wxyz[2], m.orientation_wxyz[3]);
                est.orien_filter_callback(m.orientation_wxyz[1], m.orientation_wxyz[2], m.orientation_wxyz[3], m.orientation_wxyz[0]);
            } else if (kind == 0) {
                if (!est.timerCallback()) continue;
                const VectorXd& x = est.robot_params_->est_type_ == 0 ? est.mhe.x_MHE_ : est.mhe.x_KF_;
                const Vector3d& v = est.robot_params_->est_type_ == 0 ? est.mhe.v_MHE_b_ : est.mhe.v_KF_b_;
                std::printf("x %d", est.discrete_time_ - 1);
                for (int i = 0; i < 9; ++i) std::printf(" %.17g", x(i));
                for (int i = 0; i < 3; ++i) std::printf(" %.17g", v(i));
                std::printf(" %d\n", est.mhe.solver_status_);
            } else {
                std::fprintf(stderr, "unknown record kind %d\n", kind);
                return 2;
            }
        }
        est.logger.done_logging();
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "error: %s\n", ex.what());
        return 1;
    }
    return 0;
}
