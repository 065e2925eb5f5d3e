// ros_shim.hpp -- ROS-free stand-ins for the few message types the node
// touches, field-for-field the same as sensor_msgs/Image, sensor_msgs/
// PointField, sensor_msgs/PointCloud2 and std_msgs/Header, so that
// Disparity2PCloud (disparity_to_point_cloud_amd.hpp) can be built, run and
// tested in an image that has no ROS.  With ROS present the node is
// instantiated with the real message types instead (ros/ directory).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "image_prep.hpp"
#include "pinned_allocator.hpp"

namespace d2pc_shim {

struct Time {
  uint32_t sec = 0, nsec = 0;
  bool operator==(const Time &o) const { return sec == o.sec && nsec == o.nsec; }
};

struct Header {
  uint32_t seq = 0;
  Time stamp;
  std::string frame_id;
};

struct Image {  // sensor_msgs/Image
  Header header;
  uint32_t height = 0, width = 0;
  std::string encoding;
  uint8_t is_bigendian = 0;
  uint32_t step = 0;
  std::vector<uint8_t> data;
  typedef std::shared_ptr<const Image> ConstPtr;
};

struct PointField {  // sensor_msgs/PointField
  enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};

template <class ByteAlloc>
struct PointCloud2_ {  // sensor_msgs/PointCloud2_<ContainerAllocator>
  Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t, ByteAlloc> data;
  bool is_dense = false;
};
typedef PointCloud2_<std::allocator<uint8_t>> PointCloud2;
// the payload in page-locked memory the kernels store into directly (pinned_allocator.hpp)
typedef PointCloud2_<d2pc::PinnedAllocator<uint8_t>> PinnedPointCloud2;

struct RegionOfInterest {  // sensor_msgs/RegionOfInterest
  uint32_t x_offset = 0, y_offset = 0, height = 0, width = 0;
  bool do_rectify = false;
};

struct DisparityImage {  // stereo_msgs/DisparityImage
  Header header;
  Image image;             // 32FC1 disparities
  float f = 0.f;           // focal length, pixels
  float T = 0.f;           // baseline, world units
  RegionOfInterest valid_window;
  float min_disparity = 0.f, max_disparity = 0.f;
  float delta_d = 0.f;
  typedef std::shared_ptr<const DisparityImage> ConstPtr;
};

template <class Cloud>
struct MsgsT {
  typedef d2pc_shim::Image Image;
  typedef d2pc_shim::DisparityImage DisparityImage;
  typedef d2pc_shim::PointField PointField;
  typedef Cloud PointCloud2;
  // cpp:50 + cpp:55-57 without cv_bridge / OpenCV
  static d2pc::Mono8 prepare(const Image &msg, int median_ksize) {
    return d2pc::median_blur(d2pc::to_mono8(msg), median_ksize);
  }
};
typedef MsgsT<PointCloud2> Msgs;              // pageable payload, as the reference's message
typedef MsgsT<PinnedPointCloud2> PinnedMsgs;  // zero-copy payload

}  // namespace d2pc_shim
