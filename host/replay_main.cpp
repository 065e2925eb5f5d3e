// replay_main.cpp -- ROS-free harness around Disparity2PCloud::DisparityCb.
//   d2pc_replay prep  <in.raw> <w> <h> <mono8|mono16> <out.raw>
//        host plumbing only (toCvCopy + medianBlur 11): no GPU needed
//   d2pc_replay cloud <in.raw> <w> <h> <mono8|mono16> <out.bin> [compact] [hostmedian] [step=N] [bigendian] [name=value ...]
//        name=value sets a private parameter (~fx_ ~fy_ ~cx_ ~cy_ ~base_line_), as a launch file would
//        full callback; writes PointCloud2 metadata (text) then the payload
//        extra flags: pinned  = publish sensor_msgs::PointCloud2_<PinnedAllocator> (kernels store into output.data)
//   d2pc_replay dispimage <in.f32> <w> <h> 32FC1 <out.bin> f=<px> T=<m> min_disparity=<d> [compact] [pinned] [step=N]
//        the stereo_msgs/DisparityImage callback (hpp:65 TODO): calibration from the message
//   d2pc_replay both <in.raw> <w> <h> <mono8|mono16> <out.bin> di=<in.f32> diw=<w> dih=<h> f=<px> T=<m> min_disparity=<d> [compact] [pinned]
//        ONE node with both topics live: DisparityImageCb, then DisparityCb, then DisparityImageCb again.  Writes the
//        DisparityCb cloud to <out.bin> and the second DisparityImage cloud to <out.bin>.di -- the two calibrations
//        (stereoRectify's Q_ and the message's f, T, min_disparity) must not leak into each other
//        dimode=compact|parity puts the DisparityImage context into that output mode after the first DisparityImage
//   d2pc_replay drop <in.raw> <w> <h> <mono8|mono16> <out.bin> [compact] [pinned]
//        a device-side failure in mid-stream: frame 1 is published; before frame 2 the node's context is given border 0,
//        so that the ROI outgrows the cloud DisparityCb sized for border 40 (cpp:70,72) and the ABI answers
//        D2PC_ERR_CAPACITY -- logged, frame dropped; border 40 again, frame 3 is published and written to <out.bin>
//   d2pc_replay latency <in.raw> <w> <h> <mono8|mono16> <frames> [compact] [pinned]
//        per-frame wall time of DisparityCb over <frames> calls (median, p10, p90 in microseconds)
//   d2pc_replay --gpus N | --device D [...]
//        the multi-GPU deployment in one process: RCCL broadcast of the calibration, one thread + context + frame
//        queue per GPU, counters all-reduced (host/multi_gpu.hpp)
// Every single-frame command takes device=<HIP ordinal> (default 0).
// <in.raw> holds the sensor_msgs/Image data bytes (row-major, step = w*bpp).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "disparity_to_point_cloud_amd.hpp"
#include "multi_gpu.hpp"
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

template <class Cloud>
static void write_cloud(const char *path, const Cloud &got) {
  std::ofstream o(path, std::ios::binary);
  o << "height " << got.height << " width " << got.width << " point_step " << got.point_step << " row_step "
    << got.row_step << " is_bigendian " << got.is_bigendian << " is_dense " << got.is_dense << " frame_id "
    << got.header.frame_id << " stamp " << got.header.stamp.sec << "." << got.header.stamp.nsec << " fields";
  for (auto &f : got.fields) o << " " << f.name << ":" << f.offset << ":" << int(f.datatype) << ":" << f.count;
  o << "\n";
  o.write(reinterpret_cast<const char *>(got.data.data()), std::streamsize(got.data.size()));
}

static int device_from(int argc, char **argv) {
  for (int i = 7; i < argc; ++i)
    if (!strncmp(argv[i], "device=", 7)) return atoi(argv[i] + 7);
  return 0;
}

static d2pc::ParamSource params_from(int argc, char **argv) {
  d2pc::ParamSource nh;  // d2pcloud.launch sets no params: defaults apply
  for (int i = 7; i < argc; ++i) {
    const char *eq = strchr(argv[i], '=');
    if (eq && strncmp(argv[i], "step=", 5)) nh.values[std::string(argv[i], size_t(eq - argv[i]))] = atof(eq + 1);
  }
  return nh;
}

// One callback through a node instantiated with the message policy M; returns the published cloud.
template <class M, class Fn>
static int run_node(int argc, char **argv, const char *out_path, Fn call) {
  typename M::PointCloud2 got;
  int published = 0;
  d2pc::Disparity2PCloudT<M> node(
      params_from(argc, argv), [&](const typename M::PointCloud2 &pc) { got = pc; ++published; }, device_from(argc, argv), nullptr,
      has_flag(argc, argv, "compact") ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY, false, !has_flag(argc, argv, "hostmedian"));
  call(node);
  if (published != 1) { fprintf(stderr, "nothing published\n"); return 3; }
  write_cloud(out_path, got);
  return 0;
}

template <class M>
static int run_latency(int argc, char **argv, const std::shared_ptr<d2pc_shim::Image> &img, int frames) {
  size_t last = 0;
  d2pc::Disparity2PCloudT<M> node(
      params_from(argc, argv), [&](const typename M::PointCloud2 &pc) { last = pc.data.size(); }, device_from(argc, argv), nullptr,
      has_flag(argc, argv, "compact") ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY, false, true);
  for (int i = 0; i < 20; ++i) node.DisparityCb(img);
  std::vector<double> us;
  for (int i = 0; i < frames; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    node.DisparityCb(img);
    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  std::sort(us.begin(), us.end());
  printf("DisparityCb %ux%u %s %s payload: median %.1f us  p10 %.1f  p90 %.1f  (%zu bytes published)\n", img->width,
         img->height, img->encoding.c_str(), has_flag(argc, argv, "pinned") ? "pinned  " : "pageable", us[us.size() / 2],
         us[us.size() / 10], us[us.size() * 9 / 10], last);
  return 0;
}

int main(int argc, char **argv) {
  {
    const d2pc_multi::Options mo = d2pc_multi::parse(argc, argv);
    if (mo.gpus > 0 || !mo.error.empty()) return d2pc_multi::run(mo);
  }
  if (argc < 7) { fprintf(stderr, "usage: see replay_main.cpp\n"); return 2; }
  const std::string cmd = argv[1], enc = argv[5];
  auto img = std::make_shared<d2pc_shim::Image>();
  img->width = uint32_t(atoi(argv[3]));
  img->height = uint32_t(atoi(argv[4]));
  img->encoding = enc;
  img->step = img->width * (enc == "mono16" ? 2u : enc == "32FC1" ? 4u : 1u);
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
      auto call = [&](auto &node) { node.DisparityCb(img); };
      return has_flag(argc, argv, "pinned") ? run_node<d2pc_shim::PinnedMsgs>(argc, argv, argv[6], call)
                                            : run_node<d2pc_shim::Msgs>(argc, argv, argv[6], call);
    }
    if (cmd == "dispimage") {
      auto dm = std::make_shared<d2pc_shim::DisparityImage>();
      dm->header = img->header;
      dm->image = *img;
      dm->image.step = img->width * 4u;
      for (int i = 7; i < argc; ++i) {
        if (!strncmp(argv[i], "step=", 5)) dm->image.step = uint32_t(atoi(argv[i] + 5));
        if (!strncmp(argv[i], "f=", 2)) dm->f = float(atof(argv[i] + 2));
        if (!strncmp(argv[i], "T=", 2)) dm->T = float(atof(argv[i] + 2));
        if (!strncmp(argv[i], "min_disparity=", 14)) dm->min_disparity = float(atof(argv[i] + 14));
      }
      auto call = [&](auto &node) { node.DisparityImageCb(dm); };
      return has_flag(argc, argv, "pinned") ? run_node<d2pc_shim::PinnedMsgs>(argc, argv, argv[6], call)
                                            : run_node<d2pc_shim::Msgs>(argc, argv, argv[6], call);
    }
    if (cmd == "both") {
      auto dm = std::make_shared<d2pc_shim::DisparityImage>();
      dm->header = img->header;
      dm->image.encoding = "32FC1";
      const char *di_path = nullptr;
      for (int i = 7; i < argc; ++i) {
        if (!strncmp(argv[i], "di=", 3)) di_path = argv[i] + 3;
        if (!strncmp(argv[i], "diw=", 4)) dm->image.width = uint32_t(atoi(argv[i] + 4));
        if (!strncmp(argv[i], "dih=", 4)) dm->image.height = uint32_t(atoi(argv[i] + 4));
        if (!strncmp(argv[i], "f=", 2)) dm->f = float(atof(argv[i] + 2));
        if (!strncmp(argv[i], "T=", 2)) dm->T = float(atof(argv[i] + 2));
        if (!strncmp(argv[i], "min_disparity=", 14)) dm->min_disparity = float(atof(argv[i] + 14));
      }
      if (!di_path) { fprintf(stderr, "both: di=<file> missing\n"); return 2; }
      dm->image.step = dm->image.width * 4u;
      dm->image.data = slurp(di_path);
      const std::string di_out = std::string(argv[6]) + ".di";
      auto run = [&](auto tag) {
        typedef decltype(tag) M;
        std::vector<typename M::PointCloud2> got;
        d2pc::Disparity2PCloudT<M> node(
            params_from(argc, argv), [&](const typename M::PointCloud2 &pc) { got.push_back(pc); }, device_from(argc, argv), nullptr,
            has_flag(argc, argv, "compact") ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY, false, true);
        node.DisparityImageCb(dm);
        // dimode=compact|parity: the DisparityImage context in ANOTHER mode than the node's own from here on -- each cloud's
        // is_dense and size must then be its producer's (finish_and_publish takes the producing context)
        for (int i = 7; i < argc; ++i)
          if (!strncmp(argv[i], "dimode=", 7) &&
              d2pc_set_mode(node.disparity_image_context(), !strcmp(argv[i] + 7, "compact") ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY) != D2PC_OK)
            return 3;
        node.DisparityCb(img);
        node.DisparityImageCb(dm);
        if (got.size() != 3) { fprintf(stderr, "expected three clouds, got %zu\n", got.size()); return 3; }
        write_cloud(argv[6], got[1]);
        write_cloud(di_out.c_str(), got[2]);
        return 0;
      };
      return has_flag(argc, argv, "pinned") ? run(d2pc_shim::PinnedMsgs()) : run(d2pc_shim::Msgs());
    }
    if (cmd == "drop") {
      auto run = [&](auto tag) {
        typedef decltype(tag) M;
        std::vector<typename M::PointCloud2> got;
        d2pc::Disparity2PCloudT<M> node(
            params_from(argc, argv), [&](const typename M::PointCloud2 &pc) { got.push_back(pc); }, device_from(argc, argv), nullptr,
            has_flag(argc, argv, "compact") ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY, false, true);
        node.DisparityCb(img);
        if (d2pc_set_border(node.context(), 0) != D2PC_OK) return 3;
        node.DisparityCb(img);  // D2PC_ERR_CAPACITY (PARITY) -- must neither throw nor publish
        const size_t after_bad = got.size(), dropped = node.frames_dropped();
        if (d2pc_set_border(node.context(), 40) != D2PC_OK) return 3;
        node.DisparityCb(img);
        printf("published %zu, after the failing frame %zu, dropped %zu\n", got.size(), after_bad, dropped);
        if (got.size() != after_bad + 1 || got.empty()) { fprintf(stderr, "the frame after the failure was not published\n"); return 3; }
        write_cloud(argv[6], got.back());
        return 0;
      };
      return has_flag(argc, argv, "pinned") ? run(d2pc_shim::PinnedMsgs()) : run(d2pc_shim::Msgs());
    }
    if (cmd == "latency") {
      const int frames = atoi(argv[6]);
      return has_flag(argc, argv, "pinned") ? run_latency<d2pc_shim::PinnedMsgs>(argc, argv, img, frames)
                                            : run_latency<d2pc_shim::Msgs>(argc, argv, img, frames);
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 4;
  }
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 2;
}
