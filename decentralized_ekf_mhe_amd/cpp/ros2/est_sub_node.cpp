// est_sub_node.cpp — ROS2 shell of the estimator node `est_sub` for the Go1.
// NOT built in this repository's image (no ROS2 there): everything with behaviour lives in
// ../est_node_core.hpp, which is compiled and tested without ROS2; this file only turns messages into
// the plain arguments of robotSub::Go1NodeCore.
//
// Same node name, topics, message types and parameter names as the reference
// (src/decentral_legged_est/src/EstSub.cpp:8-28, src/go1_example/src/go1Sub.cpp:8-24,155-166):
//   sub  imu/filter            sensor_msgs/Imu                      orientation from orien_sub
//   sub  orb/vo                custom_msgs/VoRealtiveTransform      sparsely integrated VO
//   sub  /unitree/imu          sensor_msgs/Imu
//   sub  /unitree/joint_state  sensor_msgs/JointState               12 angles + 4 foot forces in `position`
//   sub  /mocap/RigidBody      optitrack_broadcast/Mocap            ground truth, logged only
//   wall timer of `estimation.interval` ms; publishes nothing (output = log files under ~/log_exp/)
#include <chrono>
#include <functional>
#include <memory>
#include <vector>

#include <rclcpp/rclcpp.hpp>
#include "custom_msgs/msg/vo_realtive_transform.hpp"
#include "optitrack_broadcast/msg/mocap.hpp"
#include "sensor_msgs/msg/imu.hpp"
#include "sensor_msgs/msg/joint_state.hpp"

#include "../est_node_core.hpp"

namespace robotSub {

static double clock_seconds() { return static_cast<double>(rclcpp::Clock().now().nanoseconds()) / 1e9; }
template <class Stamp> static double stamp_seconds(const Stamp& s) { return static_cast<double>(s.sec) + static_cast<double>(s.nanosec) / 1e9; }

class go1Sub : public rclcpp::Node {
  public:
    explicit go1Sub(const std::string& name) : Node(name), core_(*this, clock_seconds()) {
        orien_filter_sub_ = create_subscription<sensor_msgs::msg::Imu>(
            "imu/filter", 10, [this](const sensor_msgs::msg::Imu::SharedPtr msg) {
                core_.orien_filter_callback(msg->orientation.x, msg->orientation.y, msg->orientation.z, msg->orientation.w);
            });
        vo_sub_ = create_subscription<custom_msgs::msg::VoRealtiveTransform>(
            "orb/vo", 10, [this](const custom_msgs::msg::VoRealtiveTransform::SharedPtr msg) {
                core_.vo_callback(stamp_seconds(msg->header_pre.stamp), stamp_seconds(msg->header.stamp), msg->x_relative,
                                  msg->y_relative, msg->z_relative);
            });
        imu_sub_ = create_subscription<sensor_msgs::msg::Imu>(
            "/unitree/imu", 10, [this](const sensor_msgs::msg::Imu::SharedPtr msg) {
                const double a[3] = {msg->linear_acceleration.x, msg->linear_acceleration.y, msg->linear_acceleration.z};
                const double w[3] = {msg->angular_velocity.x, msg->angular_velocity.y, msg->angular_velocity.z};
                core_.imu_callback(clock_seconds(), a, w);
            });
        lo_sub_ = create_subscription<sensor_msgs::msg::JointState>(
            "/unitree/joint_state", 10,
            [this](const sensor_msgs::msg::JointState::SharedPtr msg) { core_.lo_callback(msg->position, msg->velocity); });
        mocap_sub_ = create_subscription<optitrack_broadcast::msg::Mocap>(
            "/mocap/RigidBody", 10, [this](const optitrack_broadcast::msg::Mocap::SharedPtr msg) {
                const double p[3] = {msg->position[0], msg->position[1], msg->position[2]};
                const double v[3] = {msg->velocity[0], msg->velocity[1], msg->velocity[2]};
                const double q[4] = {msg->quaternion[0], msg->quaternion[1], msg->quaternion[2], msg->quaternion[3]};
                core_.mocap_callback(p, v, q);
            });
        timer_ = create_wall_timer(std::chrono::milliseconds(core_.timer_interval_ms_), [this]() {
            core_.timerCallback();
            RCLCPP_DEBUG(get_logger(), "%.1f Hz", 1.0 / core_.last_callback_seconds_);
        });
    }

  private:
    Go1NodeCore core_;
    rclcpp::Subscription<sensor_msgs::msg::Imu>::SharedPtr orien_filter_sub_, imu_sub_;
    rclcpp::Subscription<custom_msgs::msg::VoRealtiveTransform>::SharedPtr vo_sub_;
    rclcpp::Subscription<sensor_msgs::msg::JointState>::SharedPtr lo_sub_;
    rclcpp::Subscription<optitrack_broadcast::msg::Mocap>::SharedPtr mocap_sub_;
    rclcpp::TimerBase::SharedPtr timer_;
};

}  // namespace robotSub

int main(int argc, char** argv) {
    rclcpp::init(argc, argv);
    rclcpp::spin(std::make_shared<robotSub::go1Sub>("est_sub"));
    rclcpp::shutdown();
    return 0;
}
