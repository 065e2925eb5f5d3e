// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <ros/ros.h>
#include <vector>
namespace std_msgs {
template <class A> struct Header_ { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
typedef Header_<std::allocator<void>> Header;
}  // namespace std_msgs
namespace sensor_msgs {
template <class A> struct Image_ {
  std_msgs::Header_<A> header;
  uint32_t height = 0, width = 0;
  std::string encoding;
  uint8_t is_bigendian = 0;
  uint32_t step = 0;
  std::vector<uint8_t, typename A::template rebind<uint8_t>::other> data;
  typedef std::shared_ptr<const Image_<A>> ConstPtr;
};
typedef Image_<std::allocator<void>> Image;
typedef std::shared_ptr<const Image> ImageConstPtr;
}  // namespace sensor_msgs
