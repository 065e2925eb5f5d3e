// image_prep.hpp -- the host-side plumbing DisparityCb keeps on the CPU:
//   cpp:50     cv_bridge::toCvCopy(*msg, "mono8")
//   cpp:55-57  cv::medianBlur(image, filtered, 11)
// In a ROS build these two calls stay cv_bridge / OpenCV calls (see
// ros/disparity_to_point_cloud_node.cpp); this header provides equivalents for
// the ROS-free harness so the whole callback can be exercised end to end.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace d2pc {

struct Mono8 {
  int width = 0, height = 0;
  std::vector<uint8_t> pix;  // contiguous rows
};

// cv_bridge semantics for a "mono8" request [upstream cv_bridge.cpp]:
//   mono8 / 8UC1   -> copy
//   mono16 / 16UC1 published as mono16 -> convertTo(CV_8U, 255./65535.)
//   anything else  -> cv_bridge::Exception (here: std::runtime_error)
template <class Image>
Mono8 to_mono8(const Image &msg) {
  Mono8 out;
  out.width = int(msg.width);
  out.height = int(msg.height);
  out.pix.resize(size_t(out.width) * out.height);
  if (msg.encoding == "mono8") {
    if (msg.step < msg.width || msg.data.size() < size_t(msg.step) * msg.height)
      throw std::runtime_error("image data smaller than step*height");
    for (int y = 0; y < out.height; ++y)
      memcpy(&out.pix[size_t(y) * out.width], &msg.data[size_t(y) * msg.step], size_t(out.width));
  } else if (msg.encoding == "mono16") {
    if (msg.step < 2 * msg.width || msg.data.size() < size_t(msg.step) * msg.height)
      throw std::runtime_error("image data smaller than step*height");
    const float a = float(255. / 65535.);
    for (int y = 0; y < out.height; ++y) {
      const uint8_t *row = &msg.data[size_t(y) * msg.step];
      for (int x = 0; x < out.width; ++x) {
        uint16_t v;
        memcpy(&v, row + 2 * x, 2);
        if (msg.is_bigendian) v = uint16_t((v >> 8) | (v << 8));
        const long r = lrintf(float(v) * a);  // cvRound: nearest-even
        out.pix[size_t(y) * out.width + x] = uint8_t(r < 0 ? 0 : r > 255 ? 255 : r);
      }
    }
  } else {
    throw std::runtime_error("[" + msg.encoding + "] is not a color format. but [mono8] is. "
                             "Conversion between depth and color not supported.");
  }
  return out;
}

// k x k median on 8-bit, BORDER_REPLICATE (what cv::medianBlur computes):
// per row a sliding 256-bin histogram, k column taps leave and k enter per step.
inline Mono8 median_blur(const Mono8 &src, int ksize) {
  if (ksize <= 1) return src;
  Mono8 dst;
  dst.width = src.width;
  dst.height = src.height;
  dst.pix.resize(src.pix.size());
  const int r = ksize / 2, half = ksize * ksize / 2, W = src.width, H = src.height;
  if (W == 0 || H == 0) return dst;
  auto at = [&](int y, int x) {
    y = y < 0 ? 0 : y >= H ? H - 1 : y;
    x = x < 0 ? 0 : x >= W ? W - 1 : x;
    return src.pix[size_t(y) * W + x];
  };
  for (int y = 0; y < H; ++y) {
    int hist[256] = {0};
    for (int dy = -r; dy <= r; ++dy)
      for (int dx = -r; dx <= r; ++dx) ++hist[at(y + dy, dx)];
    for (int x = 0; x < W; ++x) {
      if (x > 0)
        for (int dy = -r; dy <= r; ++dy) {
          --hist[at(y + dy, x - 1 - r)];
          ++hist[at(y + dy, x + r)];
        }
      int acc = 0, m = 0;
      for (; m < 256; ++m) {
        acc += hist[m];
        if (acc > half) break;
      }
      dst.pix[size_t(y) * W + x] = uint8_t(m);
    }
  }
  return dst;
}

}  // namespace d2pc
