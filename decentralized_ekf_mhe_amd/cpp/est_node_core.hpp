// est_node_core.hpp — everything the reference's `est_sub` node does between its ROS2 callbacks and the
// estimator, with no ROS2 type in it.
//
//   robotSub::paramsWrapper          parameter names and defaults of EstSub.cpp:123-208
//   robotSub::EstNodeCore            robotSub (EstSub.cpp:8-121): orientation and VO latches, the timer
//                                    tick (gate on 10 IMU messages, initialize at T = 0, update(T) after,
//                                    logging from tick N + 1 on), the 27-double log row
//   robotSub::Go1NodeCore            go1Sub (go1Sub.cpp:8-153): IMU, joint-state and mocap latches
//
// The ROS2 shells in ros2/ only convert messages to the plain arguments used here, so the logic that
// decides WHAT is estimated and WHEN is the part that is compiled and tested without ROS2
// (tests/test_node_cores.py).  Estimation itself runs in libdekf.so on the GPU through DecentralEst.hpp.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "DecentralEst.hpp"
#include "data_logger.hpp"

namespace EigenUtils {
// EigenUtils.hpp:99-123: roll, pitch, yaw of a (w, x, y, z) quaternion
inline void QuaternionToEuler(const Quaterniond& q, Vector3d& euler) {
    const double qw = q.w(), qx = q.x(), qy = q.y(), qz = q.z();
    euler(0) = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
    const double sinp = 2 * (qw * qy - qz * qx);
    euler(1) = std::fabs(sinp) >= 1 ? std::copysign(M_PI / 2, sinp) : std::asin(sinp);
    euler(2) = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
}
}  // namespace EigenUtils

namespace robotSub {

// The est_sub parameter set (names and declared defaults: EstSub.cpp:123-208) as tables of
// (parameter name, default, robot_params member); `Node` is anything with rclcpp::Node's
// declare_parameter / get_parameter: dekf_ros::ParamNode (ros_params.hpp) or the rclcpp::Node itself.
namespace param_tables {
using V = std::vector<double>;
struct VecEntry { const char* name; V def; V robot_params::*member; };
struct NumEntry { const char* name; double def; double robot_params::*member; };
struct IntEntry { const char* name; int def; int robot_params::*member; };
struct FlagEntry { const char* name; bool def; bool robot_params::*member; };
inline const std::vector<VecEntry>& vectors() {
    static const std::vector<VecEntry> t = {
        {"prior.p_init_std", V(3, 0.001), &robot_params::p_init_std_},
        {"prior.v_init_std", V(3, 0.001), &robot_params::v_init_std_},
        {"prior.foot_init_std", V(3, 0.001), &robot_params::foot_init_std_},
        {"prior.accel_bias_init_std", V(3, 0.001), &robot_params::accel_bias_init_std_},
        {"process.p_process_std", V(3, 0.01), &robot_params::p_process_std_},
        {"process.accel_input_std", V{0.01, 0.04, 0.001}, &robot_params::accel_input_std_},
        {"process.gyro_input_std", V(3, 0.01), &robot_params::gyro_input_std_},
        {"process.accel_bias_process_std", V{1.0, 1.0, 0.1}, &robot_params::accel_bias_std_},
        {"leg_odom.quaternion_ib", V{1.0, 0.0, 0.0, 0.0}, &robot_params::quaternion_ib_},
        {"leg_odom.p_ib", V(3, 0.0), &robot_params::p_ib_},
        {"leg_odom.joint_position_std", V(3, 0.01), &robot_params::joint_position_std_},
        {"leg_odom.joint_velocity_std", V(3, 0.01), &robot_params::joint_velocity_std_},
        {"leg_odom.foot_slide_std", V(3, 0.001), &robot_params::foot_slide_std_},
        {"leg_odom.foot_swing_std", V(3, 10000.0), &robot_params::foot_swing_std_},
        {"visual_odom.vo_p_std", V(3, 0.001), &robot_params::vo_p_std_},
    };
    return t;
}
inline const std::vector<NumEntry>& numbers() {
    static const std::vector<NumEntry> t = {
        {"leg_odom.contact_effort_theshold", 150.0, &robot_params::contact_effort_theshold_},
        {"osqp.rho", 0.1, &robot_params::rho_},
        {"osqp.alpha", 1.6, &robot_params::alpha_},
        {"osqp.delta", 0.00001, &robot_params::delta_},
        {"osqp.sigma", 0.00001, &robot_params::sigma_},
        {"osqp.primTol", 0.000001, &robot_params::primTol_},
        {"osqp.dualTol", 0.000001, &robot_params::dualTol_},
        {"osqp.realtiveTol", 1e-3, &robot_params::realtiveTol_},
        {"osqp.absTol", 1e-3, &robot_params::absTol_},
        {"osqp.timeLimit", 0.005, &robot_params::timeLimit_},
    };
    return t;
}
inline const std::vector<IntEntry>& integers() {
    static const std::vector<IntEntry> t = {
        {"leg_odom.num_leg", 4, &robot_params::num_legs_},
        {"leg_odom.leg_odom_type", 0, &robot_params::leg_odom_type_},
        {"estimation.rate", 50, &robot_params::rate_},
        {"estimation.N", 50, &robot_params::N_},
        {"estimation.est_type", 0, &robot_params::est_type_},
        {"osqp.maxQPIter", 1000, &robot_params::maxQPIter_},
    };
    return t;
}
inline const std::vector<FlagEntry>& flags() {
    static const std::vector<FlagEntry> t = {
        {"osqp.verbose", true, &robot_params::verbose_},
        {"osqp.adaptRho", true, &robot_params::adaptRho_},
        {"osqp.polish", true, &robot_params::polish_},
    };
    return t;
}
}  // namespace param_tables

template <class Node>
void paramsWrapper(Node& node, robot_params& rp, std::string& log_name, int& timer_interval_ms) {
    node.declare_parameter("log_name", std::string("exp"));
    log_name = node.get_parameter("log_name").as_string();
    node.declare_parameter("estimation.interval", 20);  // period of the node's wall timer, ms
    timer_interval_ms = (int)node.get_parameter("estimation.interval").as_int();
    for (const auto& e : param_tables::vectors()) {
        node.declare_parameter(e.name, e.def);
        rp.*(e.member) = node.get_parameter(e.name).as_double_array();
    }
    for (const auto& e : param_tables::numbers()) {
        node.declare_parameter(e.name, e.def);
        rp.*(e.member) = node.get_parameter(e.name).as_double();
    }
    for (const auto& e : param_tables::integers()) {
        node.declare_parameter(e.name, e.def);
        rp.*(e.member) = (int)node.get_parameter(e.name).as_int();
    }
    for (const auto& e : param_tables::flags()) {
        node.declare_parameter(e.name, e.def);
        rp.*(e.member) = node.get_parameter(e.name).as_bool();
    }
}

class EstNodeCore {
  public:
    // `time_init`: the node's clock at construction in seconds (EstSub.cpp:27); every stamp handed to
    // the callbacks below is on the same clock.
    template <class Node>
    EstNodeCore(Node& node, double time_init, int device = 0) : time_init_(time_init), device_(device) {
        robot_store_ = std::make_shared<robot_store>();
        robot_params_ = std::make_shared<robot_params>();
        paramsWrapper(node, *robot_params_, log_name_, timer_interval_ms_);
        // (The DECLARED default of osqp.polish is true, EstSub.cpp:188, while the Go1 parameter file sets it false,
        // parameters_go1.yaml:44: both run — the solve kernels polish when asked to, mhe_solve_core.h.)
        const char* home = std::getenv("HOME");
        log_dir_ = std::string(home ? home : ".") + "/log_exp/";  // data_logger.hpp:53-57 of the reference
        x_logged_ = &mhe.x_MHE_;
    }
    EstNodeCore(const EstNodeCore&) = delete;
    EstNodeCore& operator=(const EstNodeCore&) = delete;

    // EstSub.cpp:34-43, topic imu/filter (sensor_msgs/Imu orientation)
    void orien_filter_callback(double x, double y, double z, double w) {
        robot_store_->quaternion_.x() = x;
        robot_store_->quaternion_.y() = y;
        robot_store_->quaternion_.z() = z;
        robot_store_->quaternion_.w() = w;
        EigenUtils::QuaternionToEuler(robot_store_->quaternion_, filter_euler_);
    }
    // EstSub.cpp:45-56, topic orb/vo (custom_msgs/VoRealtiveTransform): header_pre.stamp, header.stamp
    // as seconds (sec + nanosec / 1e9), relative translation body_pre -> body
    void vo_callback(double stamp_pre, double stamp_now, double x_relative, double y_relative, double z_relative) {
        robot_store_->vo_new_ = true;
        robot_store_->vo_time_pre_ = stamp_pre - time_init_;
        robot_store_->vo_time_now_ = stamp_now - time_init_;
        robot_store_->vo_p_body_pre_2_body_(0) = x_relative;
        robot_store_->vo_p_body_pre_2_body_(1) = y_relative;
        robot_store_->vo_p_body_pre_2_body_(2) = z_relative;
    }
    // EstSub.cpp:58-91.  Returns true when the estimator ran on this tick.
    bool timerCallback() {
        const auto start = std::chrono::steady_clock::now();
        bool ran = false;
        if (imu_msg_num_ >= 10) {
            if (discrete_time_ == 0) {
                mhe.initialize(robot_store_, robot_params_, device_);
                for (int i = 0; i < 3; ++i) gt_p_offset_(i) = robot_store_->gt_p_(i);
            } else {
                mhe.update(discrete_time_);
            }
            discrete_time_++;
            ran = true;
            if (discrete_time_ == robot_params_->N_ + 1) init_logging();
            if (discrete_time_ > robot_params_->N_ + 1) logger.spin_logging();
        }
        last_callback_seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        return ran;
    }
    // EstSub.cpp:93-121: the seven log variables, 27 doubles per row
    void init_logging() {
        const bool kf = robot_params_->est_type_ != 0;
        logger.init(log_name_, log_dir_);
        logger.add_data_vectorXd(gt_p_.data(), 3, "pose");
        logger.add_data_vectorXd(gt_v_b_.data(), 3, "GT_v");
        logger.add_data_vectorXd((kf ? mhe.v_KF_b_ : mhe.v_MHE_b_).data(), 3, "v_body");
        x_logged_ = kf ? &mhe.x_KF_ : &mhe.x_MHE_;
        logger.add_data_vectorXd(x_logged_->data(), (unsigned)x_logged_->size(), "x_MHE");
        logger.add_data_vectorXd(mhe.p_vo_accmulate_.data(), 3, "p_vo_accmulate_");
        logger.add_data_vectorXd(filter_euler_.data(), 3, "filter_euler_");
        logger.add_data_vectorXd(gt_euler_.data(), 3, "gt_euler_");
    }

    DecentralizedEstimation mhe;
    Data_Logger logger;
    std::shared_ptr<robot_store> robot_store_;
    std::shared_ptr<robot_params> robot_params_;

    double time_init_ = 0;
    int imu_msg_num_ = 0;
    int discrete_time_ = 0;
    int timer_interval_ms_ = 20;  // estimation.interval: period of the wall timer
    std::string log_name_ = "exp";
    std::string log_dir_;         // new: the reference hard-wires $HOME/log_exp/
    double last_callback_seconds_ = 0.0;  // the reference prints 1 / this every tick

    Vector3d gt_p_offset_, gt_p_, gt_v_b_;
    Quaterniond gt_quaternion_;
    Vector3d gt_euler_, filter_euler_;

  protected:
    int device_ = 0;
    VectorXd* x_logged_ = nullptr;
};

class Go1NodeCore : public EstNodeCore {
  public:
    template <class Node>
    Go1NodeCore(Node& node, double time_init, int device = 0) : EstNodeCore(node, time_init, device) {
        mhe.go1_raw_joints_ = true;  // forward kinematics, Jacobians and contact on the device (f2)
        robot_store_->contact_ = VectorXd::Zero(robot_params_->num_legs_);
    }
    // go1Sub.cpp:30-51, topic /unitree/imu.  `now` is the node clock at arrival (the reference stamps the
    // sample with its own clock, not with the message header).
    void imu_callback(double now, const double linear_acceleration[3], const double angular_velocity[3]) {
        robot_store_->imu_time_ = now - time_init_;
        for (int i = 0; i < 3; ++i) {
            robot_store_->accel_b_(i) = linear_acceleration[i];
            robot_store_->angular_b_(i) = angular_velocity[i];
        }
        imu_msg_num_++;
    }
    // go1Sub.cpp:53-126, topic /unitree/joint_state: position = 12 joint angles + 4 foot forces
    // (the contact test reads position[12 + leg], :76), velocity = 12 joint rates.
    void lo_callback(const std::vector<double>& position, const std::vector<double>& velocity) {
        if (position.size() < 16 || velocity.size() < 12) throw std::invalid_argument("joint_state: need 16 positions and 12 velocities");
        robot_store_->joint_states_position_.resize((int)position.size());
        robot_store_->joint_states_velocity_.resize((int)velocity.size());
        for (size_t i = 0; i < position.size(); ++i) robot_store_->joint_states_position_((int)i) = position[i];
        for (size_t i = 0; i < velocity.size(); ++i) robot_store_->joint_states_velocity_((int)i) = velocity[i];
        for (int i = 0; i < robot_params_->num_legs_; ++i)
            robot_store_->contact_(i) = position[(size_t)(12 + i)] >= robot_params_->contact_effort_theshold_ ? 1.0 : 0.0;
    }
    // go1Sub.cpp:128-153, topic /mocap/RigidBody: position[3], velocity[3], quaternion (w, x, y, z)
    void mocap_callback(const double position[3], const double velocity[3], const double quaternion_wxyz[4]) {
        for (int i = 0; i < 3; ++i) { robot_store_->gt_p_(i) = position[i]; robot_store_->gt_v_s_(i) = velocity[i]; }
        gt_quaternion_.w() = quaternion_wxyz[0];
        gt_quaternion_.x() = quaternion_wxyz[1];
        gt_quaternion_.y() = quaternion_wxyz[2];
        gt_quaternion_.z() = quaternion_wxyz[3];
        EigenUtils::QuaternionToEuler(gt_quaternion_, gt_euler_);
        for (int i = 0; i < 3; ++i) gt_p_(i) = robot_store_->gt_p_(i) - gt_p_offset_(i);
        // gt_v_b_ = R(gt_quaternion_.normalized()) * gt_v_s_   (:152, R itself as in the reference)
        double w = quaternion_wxyz[0], x = quaternion_wxyz[1], y = quaternion_wxyz[2], z = quaternion_wxyz[3];
        const double n = std::sqrt(w * w + x * x + y * y + z * z);
        w /= n; x /= n; y /= n; z /= n;
        const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
        for (int i = 0; i < 3; ++i) gt_v_b_(i) = R[i][0] * velocity[0] + R[i][1] * velocity[1] + R[i][2] * velocity[2];
    }
};

}  // namespace robotSub
