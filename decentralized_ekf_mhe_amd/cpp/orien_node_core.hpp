// orien_node_core.hpp — the reference's `orien_sub` node (src/orien_est/src/orien_ekf.cpp:8-106) without
// ROS2 types: parameter names and defaults (:13-25), the VO-pose and IMU latches (:48-75) and the timer
// tick that runs the quaternion EKF and fills the imu/filter message (:77-106).  The EKF arithmetic
// (history, VO rewind and replay, predict, accel-correct) is dekf_ekf_step in libdekf.so.
#pragma once
#include <string>
#include <vector>

#include "DecentralEst.hpp"

namespace orien_ekf {

// what timerCallback publishes on imu/filter (sensor_msgs/Imu): orientation, and the latched IMU sample
struct FilterMsg {
    double orientation_wxyz[4];
    double linear_acceleration[3];
    double angular_velocity[3];
};

// orien_ekf.cpp:13-25.  `Node`: dekf_ros::ParamNode or rclcpp::Node.
template <class Node>
void paramsWrapper(Node& node, dekf_params& d) {
    using V = std::vector<double>;
    node.declare_parameter("init_std", V{0.001, 0.001, 0.001, 0.001});
    node.declare_parameter("process_std", V{0.1, 0.1, 0.1});
    node.declare_parameter("gravity_meas_std", V{4.0, 4.0, 4.0});
    node.declare_parameter("vo_meas_std", V{0.0001, 0.0001, 0.0001, 0.0001});
    node.declare_parameter("quaternion_init", V{1.0, 0.0, 0.0, 0.0});
    node.declare_parameter("rate", 500);
    const V q_init_std = node.get_parameter("init_std").as_double_array();
    const V gyro_std = node.get_parameter("process_std").as_double_array();
    const V accel_std = node.get_parameter("gravity_meas_std").as_double_array();
    const V vo_std = node.get_parameter("vo_meas_std").as_double_array();
    const V q_init = node.get_parameter("quaternion_init").as_double_array();
    if (q_init_std.size() < 4 || gyro_std.size() < 3 || accel_std.size() < 3 || vo_std.size() < 4 || q_init.size() < 4)
        throw std::invalid_argument("orien_sub: init_std/vo_meas_std/quaternion_init need 4 entries, process_std/gravity_meas_std 3");
    for (int i = 0; i < 4; ++i) {
        d.ekf_init_std[i] = q_init_std[(size_t)i];
        d.ekf_vo_meas_std[i] = vo_std[(size_t)i];
        d.ekf_quaternion_init[i] = q_init[(size_t)i];
    }
    for (int i = 0; i < 3; ++i) {
        d.ekf_process_std[i] = gyro_std[(size_t)i];
        d.ekf_gravity_meas_std[i] = accel_std[(size_t)i];
    }
    d.ekf_rate = (int)node.get_parameter("rate").as_int();
}

class OrienNodeCore {
  public:
    template <class Node>
    OrienNodeCore(Node& node, double time_init, int device = 0) : time_init_(time_init) {
        dekf_default_params(&prm_);
        paramsWrapper(node, prm_);
        dt_ = 1.0 / static_cast<double>(prm_.ekf_rate);
        for (int i = 0; i < 4; ++i) msg_.orientation_wxyz[i] = prm_.ekf_quaternion_init[i];
        for (int i = 0; i < 3; ++i) msg_.linear_acceleration[i] = msg_.angular_velocity[i] = 0.0;
        ekf_.reset(new OrientationEkf(prm_, device));
    }
    // period of the wall timer in microseconds (orien_ekf.cpp:44)
    int timer_period_us() const { return int(dt_ * 1e6); }

    // orien_ekf.cpp:48-60, topic orb/pos (geometry_msgs/PoseStamped): header stamp in seconds, orientation
    void vo_pose_callback(double stamp, double x, double y, double z, double w) {
        const double q[4] = {w, x, y, z};
        ekf_->vo_pose_callback(stamp - time_init_, q);
        init_vo = 1;
    }
    // orien_ekf.cpp:62-75, topic unitree/imu; `now` = node clock at arrival
    void imu_callback(double now, const double linear_acceleration[3], const double angular_velocity[3]) {
        ekf_->imu_callback(now - time_init_, linear_acceleration, angular_velocity);
        for (int i = 0; i < 3; ++i) {
            msg_.linear_acceleration[i] = linear_acceleration[i];
            msg_.angular_velocity[i] = angular_velocity[i];
        }
        init_imu = 1;
    }
    // orien_ekf.cpp:77-106: no filter step before the first IMU sample, but a message on every tick
    const FilterMsg& timerCallback() {
        if (init_imu) {
            ekf_->timerCallback();
            ekf_->quaternion(msg_.orientation_wxyz);
            discrete_time_++;
        }
        return msg_;
    }

    int init_imu = 0, init_vo = 0;
    int discrete_time_ = 0;
    double dt_ = 0.002;
    double time_init_ = 0;
    dekf_params prm_;

  private:
    std::unique_ptr<OrientationEkf> ekf_;
    FilterMsg msg_;
};

}  // namespace orien_ekf
