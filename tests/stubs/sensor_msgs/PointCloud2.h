// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <sensor_msgs/Image.h>
namespace sensor_msgs {
template <class A> struct PointField_ {
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};
typedef PointField_<std::allocator<void>> PointField;
template <class A> struct PointCloud2_ {
  std_msgs::Header_<A> header;
  uint32_t height = 0, width = 0;
  std::vector<PointField_<A>, typename A::template rebind<PointField_<A>>::other> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t, typename A::template rebind<uint8_t>::other> data;
  bool is_dense = false;
};
typedef PointCloud2_<std::allocator<void>> PointCloud2;
}  // namespace sensor_msgs
