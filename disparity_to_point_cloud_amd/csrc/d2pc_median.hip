// d2pc_median.hip -- k x k median of an 8-bit image on gfx950 (k odd, <= 11),
// BORDER_REPLICATE: the device form of cv::medianBlur(img, out, 11) at
// reference src/disparity_to_point_cloud.cpp:55-57.
//
// Bit-plane radix select (no sorting, no histograms), two window rows per
// register:
//  1.  A block owns TW x 64 output pixels.  Every input row of the tile (+halo)
//      is turned into eight 32-bit PLANE WORDS, bit j of plane b = bit b of the
//      pixel in column c0-r+j, by one lane per row with in-register bit
//      transposes (four 8x8 bit transposes and two 4x4 byte transposes: ~110
//      integer ops for all eight words of a row).  The words are cut into
//      overlapping 16-bit SEGMENTS (one every 17-k columns, so that every pixel
//      finds its whole k-bit window inside one segment) and the segments of
//      rows i and i+1 (the neighbouring lane's: one DPP move) are packed into
//      one dword in LDS: pair[b][segment][i] = seg(i) | seg(i+1) << 16.
//  2.  The median of the k*k window of pixel (y,x) is found MSB-first (small
//      windows: for two horizontally adjacent pixels per thread, which share
//      every pair word).  The candidate set of a pixel is
//      ceil(k/2) registers, each holding the k-bit masks of TWO window rows; per
//      plane and register: ones = cand & pair word, n1 +=
//      popcount, and after the rank test cand &= pair ^ flip -- three integer
//      instructions (v_and, v_bcnt accumulate, v_bitop3) for two rows.
//      k = 11: 8 planes x 6 registers x 3 = 144 ops per pixel (one row per
//      register: 264; compare-and-count: 2 x 968).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d2pc_launch.hpp"
#include "d2pc_median_tile.hpp"

namespace d2pc {

template <int KS>
__global__ __launch_bounds__(MedianShape<KS>::THREADS) void k_median_u8(const uint8_t *__restrict__ src,
                                                                        uint8_t *__restrict__ dst,
                                                                        const MedianArgs a) {
  using S = MedianShape<KS>;
  __shared__ __attribute__((aligned(16))) uint32_t s_pair[S::LDS_WORDS];
  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int c0 = int(a.out_x0) + int(tx) * S::TW, y0 = int(a.out_y0) + int(ty) * S::TH;
  median_tile<KS>(src + uint64_t(f) * a.src_frame_stride, dst + uint64_t(f) * a.dst_frame_stride, a, c0, y0, s_pair,
                         threadIdx.x);
}

// ---------------------------------------------------------------------------
// cv_bridge::toCvCopy(msg, "mono8") on a mono16 image (reference
// src/disparity_to_point_cloud.cpp:50): image.convertTo(CV_8U, 255./65535.) =
// saturate_cast<uchar>(cvRound(v * (float)(255./65535.))), product in float,
// round-half-to-even.  Eight pixels per thread (one 16-byte load, one 8-byte store when aligned).
// ---------------------------------------------------------------------------
// Two index forms (tools/mono16_bench.py; the first form of this kernel recovered (x, y, frame) from one flat 64-bit
// index -- three 64-bit divisions per thread, more instructions than the eight conversions):
//   ROWS   grid = (groups of 8 pixels / block, rows, frames): no division; the last wave of every row is part empty,
//          so it serves rows that fill their waves to >= 90 % (4K: 480 groups, 1080p: 240)
//   flat   grid = (groups x rows / 256, frames): one 32-bit division per thread; every wave full (752 pixels = 94 groups)
template <bool ROWS>
__global__ __launch_bounds__(kBlock) void k_mono16_to_mono8(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                            const MedianArgs a) {
  const uint32_t groups = (a.width + 7u) / 8u;
  uint32_t y, gx;
  if constexpr (ROWS) {
    gx = blockIdx.x * blockDim.x + threadIdx.x;
    y = blockIdx.y;
    if (gx >= groups) return;
  } else {
    const uint32_t i = blockIdx.x * uint32_t(kBlock) + threadIdx.x;
    if (i >= groups * a.height) return;
    y = i / groups;
    gx = i - y * groups;
  }
  const float scale = float(255. / 65535.);
  auto cvt = [&](uint32_t v) {
    const float r = __builtin_rintf(float(v) * scale);  // default rounding mode: nearest-even
    return uint32_t(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
  };
  for (uint32_t f = ROWS ? blockIdx.z : blockIdx.y; f < a.n_frames; f += ROWS ? gridDim.z : gridDim.y) {
    const uint8_t *srow = src + uint64_t(f) * a.src_frame_stride + uint64_t(y) * a.src_row_stride;
    uint8_t *drow = dst + uint64_t(f) * a.dst_frame_stride + uint64_t(y) * a.dst_row_stride;
    const uint32_t x = gx * 8u;
    const uint8_t *sp = srow + 2u * x;
    uint8_t *dp = drow + x;
    if (x + 8u <= a.width && (reinterpret_cast<uintptr_t>(sp) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dp) & 7u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4 *>(sp);  // 8 x uint16
      uint2 o;
      o.x = cvt(v.x & 0xffffu) | (cvt(v.x >> 16) << 8) | (cvt(v.y & 0xffffu) << 16) | (cvt(v.y >> 16) << 24);
      o.y = cvt(v.z & 0xffffu) | (cvt(v.z >> 16) << 8) | (cvt(v.w & 0xffffu) << 16) | (cvt(v.w >> 16) << 24);
      *reinterpret_cast<uint2 *>(dp) = o;
    } else {
      for (uint32_t k = 0; k < 8u && x + k < a.width; ++k)
        dp[k] = uint8_t(cvt(reinterpret_cast<const uint16_t *>(sp)[k]));
    }
  }
}

hipError_t launch_mono16_to_mono8(const void *src, void *dst, const MedianArgs &a, hipStream_t stream) {
  const uint32_t groups = (a.width + 7u) / 8u;
  if (groups == 0 || a.height == 0 || a.n_frames == 0) return hipErrorInvalidValue;
  const uint32_t frames = a.n_frames < 65535u ? a.n_frames : 65535u;
  const uint8_t *s8 = static_cast<const uint8_t *>(src);
  uint8_t *d8 = static_cast<uint8_t *>(dst);
  const uint32_t waves = (groups + 63u) / 64u;
  if (groups * 10u >= waves * 64u * 9u && a.height <= 65535u) {  // rows fill their waves: the division-free grid
    const uint32_t block = waves * 64u < uint32_t(kBlock) ? waves * 64u : uint32_t(kBlock);
    hipLaunchKernelGGL(k_mono16_to_mono8<true>, dim3((groups + block - 1) / block, a.height, frames), dim3(block), 0, stream, s8, d8, a);
  } else {
    if (uint64_t(groups) * a.height > 0xffffffffull) return hipErrorInvalidValue;
    const uint32_t per_frame = groups * a.height;
    hipLaunchKernelGGL(k_mono16_to_mono8<false>, dim3((per_frame + kBlock - 1) / kBlock, frames), dim3(kBlock), 0, stream, s8, d8, a);
  }
  return hipGetLastError();
}

bool median_ksize_supported(int k) { return k == 3 || k == 5 || k == 7 || k == 9 || k == 11; }

namespace {
template <int KS>
hipError_t launch_k(const uint8_t *s, uint8_t *d, MedianArgs a, hipStream_t stream) {
  using S = MedianShape<KS>;
  a.tiles_x = (a.out_w + S::TW - 1) / S::TW;
  a.tiles_y = (a.out_h + S::TH - 1) / S::TH;
  const uint64_t blocks = uint64_t(a.tiles_x) * a.tiles_y * a.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_median_u8<KS>, dim3(uint32_t(blocks)), dim3(S::THREADS), 0, stream, s, d, a);
  return hipGetLastError();
}
}  // namespace

// The bit-sliced kernel works on 256 x 32 tiles, three or four 4-wave blocks per CU, ~8 (3 x 3) to ~20 us (11 x 11)
// per block: it pays once the launch has a good fraction of a chipful of tiles (profiles/r02_median_bitsliced.txt,
// 11 x 11: 312 tiles 0.86x, 512 tiles 1.11x, 15,600 tiles 1.46x the per-pixel kernel; 3 x 3: 156 tiles 1.01x, 15,600
// tiles 2.84x); below that (one native 752x480 frame is 39 tiles, 4-7 us) the per-pixel kernel's small tiles spread better.
constexpr uint64_t bs_min_tiles(int ksize) { return ksize <= 3 ? 192 : ksize <= 5 ? 320 : 448; }

bool median_uses_bs(const MedianArgs &a, int ksize) {
  return median_ksize_supported(ksize) && a.out_w != 0 && a.out_h != 0 &&
         (a.algo == 2 || a.algo == 3 || (a.algo == 0 && median_bs_tiles(a) >= bs_min_tiles(ksize)));  // (3: experiment build, the lane-pair select)
}

hipError_t launch_median(const void *src, void *dst, const MedianArgs &args, int ksize, hipStream_t stream) {
  const uint8_t *s = static_cast<const uint8_t *>(src);
  uint8_t *d = static_cast<uint8_t *>(dst);
  MedianArgs a = args;
  if (a.out_w == 0 || a.out_h == 0) {  // whole image
    a.out_x0 = a.out_y0 = 0;
    a.out_w = a.width;
    a.out_h = a.height;
  }
  if (a.out_x0 + a.out_w > a.width || a.out_y0 + a.out_h > a.height) return hipErrorInvalidValue;
  if (median_uses_bs(a, ksize)) return launch_median_bs(src, dst, a, ksize, stream);
  switch (ksize) {
    case 3: return launch_k<3>(s, d, a, stream);
    case 5: return launch_k<5>(s, d, a, stream);
    case 7: return launch_k<7>(s, d, a, stream);
    case 9: return launch_k<9>(s, d, a, stream);
    case 11: return launch_k<11>(s, d, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace d2pc
