// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <sensor_msgs/Image.h>
namespace sensor_msgs {
struct RegionOfInterest { uint32_t x_offset = 0, y_offset = 0, height = 0, width = 0; bool do_rectify = false; };
}
namespace stereo_msgs {
template <class A> struct DisparityImage_ {
  std_msgs::Header_<A> header;
  sensor_msgs::Image_<A> image;
  float f = 0, T = 0;
  sensor_msgs::RegionOfInterest valid_window;
  float min_disparity = 0, max_disparity = 0, delta_d = 0;
  typedef std::shared_ptr<const DisparityImage_<A>> ConstPtr;
};
typedef DisparityImage_<std::allocator<void>> DisparityImage;
typedef std::shared_ptr<const DisparityImage> DisparityImageConstPtr;
}  // namespace stereo_msgs
