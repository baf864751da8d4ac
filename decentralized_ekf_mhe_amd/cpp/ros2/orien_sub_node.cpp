// orien_sub_node.cpp — ROS2 shell of the orientation node `orien_sub`.
// NOT built in this repository's image (no ROS2 there); the behaviour is in ../orien_node_core.hpp.
//
// Same node name, topics and parameters as the reference (src/orien_est/src/orien_ekf.cpp:8-46,359-366):
//   sub  orb/pos      geometry_msgs/PoseStamped   VO orientation, applied by rewind and replay
//   sub  unitree/imu  sensor_msgs/Imu
//   pub  imu/filter   sensor_msgs/Imu             filtered orientation + the latched IMU sample
//   wall timer of 1 / rate seconds
#include <chrono>
#include <memory>

#include <rclcpp/rclcpp.hpp>
#include "geometry_msgs/msg/pose_stamped.hpp"
#include "sensor_msgs/msg/imu.hpp"

#include "../orien_node_core.hpp"

namespace orien_ekf {

static double clock_seconds() { return static_cast<double>(rclcpp::Clock().now().nanoseconds()) / 1e9; }

class orien_ekf_node : public rclcpp::Node {
  public:
    explicit orien_ekf_node(const std::string& name) : Node(name), core_(*this, clock_seconds()) {
        vo_pose_sub_ = create_subscription<geometry_msgs::msg::PoseStamped>(
            "orb/pos", 10, [this](const geometry_msgs::msg::PoseStamped::SharedPtr msg) {
                const double stamp = static_cast<double>(msg->header.stamp.sec) + static_cast<double>(msg->header.stamp.nanosec) / 1e9;
                core_.vo_pose_callback(stamp, msg->pose.orientation.x, msg->pose.orientation.y, msg->pose.orientation.z,
                                       msg->pose.orientation.w);
            });
        imu_sub_ = create_subscription<sensor_msgs::msg::Imu>(
            "unitree/imu", 10, [this](const sensor_msgs::msg::Imu::SharedPtr msg) {
                const double a[3] = {msg->linear_acceleration.x, msg->linear_acceleration.y, msg->linear_acceleration.z};
                const double w[3] = {msg->angular_velocity.x, msg->angular_velocity.y, msg->angular_velocity.z};
                core_.imu_callback(clock_seconds(), a, w);
            });
        publisher_filter_ = create_publisher<sensor_msgs::msg::Imu>("imu/filter", 10);
        timer_ = create_wall_timer(std::chrono::microseconds(core_.timer_period_us()), [this]() {
            const FilterMsg& f = core_.timerCallback();
            sensor_msgs::msg::Imu out;  // header stamp left at zero, as in the reference (:90)
            out.orientation.w = f.orientation_wxyz[0];
            out.orientation.x = f.orientation_wxyz[1];
            out.orientation.y = f.orientation_wxyz[2];
            out.orientation.z = f.orientation_wxyz[3];
            out.linear_acceleration.x = f.linear_acceleration[0];
            out.linear_acceleration.y = f.linear_acceleration[1];
            out.linear_acceleration.z = f.linear_acceleration[2];
            out.angular_velocity.x = f.angular_velocity[0];
            out.angular_velocity.y = f.angular_velocity[1];
            out.angular_velocity.z = f.angular_velocity[2];
            publisher_filter_->publish(out);
        });
    }

  private:
    OrienNodeCore core_;
    rclcpp::Subscription<geometry_msgs::msg::PoseStamped>::SharedPtr vo_pose_sub_;
    rclcpp::Subscription<sensor_msgs::msg::Imu>::SharedPtr imu_sub_;
    rclcpp::Publisher<sensor_msgs::msg::Imu>::SharedPtr publisher_filter_;
    rclcpp::TimerBase::SharedPtr timer_;
};

}  // namespace orien_ekf

int main(int argc, char* argv[]) {
    rclcpp::init(argc, argv);
    rclcpp::spin(std::make_shared<orien_ekf::orien_ekf_node>("orien_sub"));
    rclcpp::shutdown();
    return 0;
}
