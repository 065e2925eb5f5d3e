// replay_main.cpp -- ROS-free harness around Disparity2PCloud::DisparityCb.
//   d2pc_replay prep  <in.raw> <w> <h> <mono8|mono16> <out.raw>
//        host plumbing only (toCvCopy + medianBlur 11): no GPU needed
//   d2pc_replay cloud <in.raw> <w> <h> <mono8|mono16> <out.bin> [compact] [hostmedian] [step=N] [bigendian] [name=value ...]
//        name=value sets a private parameter (~fx_ ~fy_ ~cx_ ~cy_ ~base_line_), as a launch file would
//        full callback; writes PointCloud2 metadata (text) then the payload
// <in.raw> holds the sensor_msgs/Image data bytes (row-major, step = w*bpp).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "disparity_to_point_cloud_amd.hpp"
#include "ros_shim.hpp"

static std::vector<uint8_t> slurp(const char *path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static bool has_flag(int argc, char **argv, const char *flag) {
  for (int i = 7; i < argc; ++i)
    if (!strcmp(argv[i], flag)) return true;
  return false;
}

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: see replay_main.cpp\n"); return 2; }
  const std::string cmd = argv[1], enc = argv[5];
  auto img = std::make_shared<d2pc_shim::Image>();
  img->width = uint32_t(atoi(argv[3]));
  img->height = uint32_t(atoi(argv[4]));
  img->encoding = enc;
  img->step = img->width * (enc == "mono16" ? 2u : 1u);
  for (int i = 7; i < argc; ++i)  // step=<bytes>: padded rows; bigendian: byte-swapped 16-bit samples
    if (!strncmp(argv[i], "step=", 5)) img->step = uint32_t(atoi(argv[i] + 5));
  img->is_bigendian = has_flag(argc, argv, "bigendian") ? 1 : 0;
  img->data = slurp(argv[2]);
  img->header.stamp.sec = 1234;
  img->header.stamp.nsec = 5678;
  img->header.frame_id = "left_cam";
  try {
    if (cmd == "prep") {
      d2pc::Mono8 m = d2pc::median_blur(d2pc::to_mono8(*img), 11);
      std::ofstream(argv[6], std::ios::binary).write(reinterpret_cast<const char *>(m.pix.data()), std::streamsize(m.pix.size()));
      return 0;
    }
    if (cmd == "cloud") {
      d2pc_shim::PointCloud2 got;
      int published = 0;
      d2pc::ParamSource nh;  // d2pcloud.launch sets no params: defaults apply
      for (int i = 7; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (eq && strncmp(argv[i], "step=", 5)) nh.values[std::string(argv[i], size_t(eq - argv[i]))] = atof(eq + 1);
      }
      d2pc::Disparity2PCloudT<d2pc_shim::Msgs> node(
          nh, [&](const d2pc_shim::PointCloud2 &pc) { got = pc; ++published; }, 0, nullptr,
          has_flag(argc, argv, "compact") ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY, false,
          !has_flag(argc, argv, "hostmedian"));
      node.DisparityCb(img);
      if (published != 1) { fprintf(stderr, "nothing published\n"); return 3; }
      std::ofstream o(argv[6], std::ios::binary);
      o << "height " << got.height << " width " << got.width << " point_step " << got.point_step << " row_step "
        << got.row_step << " is_bigendian " << got.is_bigendian << " is_dense " << got.is_dense << " frame_id "
        << got.header.frame_id << " stamp " << got.header.stamp.sec << "." << got.header.stamp.nsec << " fields";
      for (auto &f : got.fields) o << " " << f.name << ":" << f.offset << ":" << int(f.datatype) << ":" << f.count;
      o << "\n";
      o.write(reinterpret_cast<const char *>(got.data.data()), std::streamsize(got.data.size()));
      return 0;
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 4;
  }
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 2;
}
