// d2pc_median_bs.hip -- the bit-sliced k x k median as a kernel of its own (tile body and design notes:
// d2pc_median_bs_tile.hpp); second device form of cv::medianBlur(img, out, 11) at reference
// src/disparity_to_point_cloud.cpp:55-57.
#include "d2pc_median_bs_tile.hpp"

namespace d2pc {

template <int KS>
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(D2PC_BS_WAVES))) void k_median_bs_u8(
    const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const MedianArgs a) {
  using S = MedianBsShape<KS>;
  __shared__ __attribute__((aligned(16))) uint32_t s_w[S::W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_raw[S::RAW_WORDS];
  const uint32_t tid = threadIdx.x;
  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int x0 = int(a.out_x0) + int(tx) * S::TW, y0 = int(a.out_y0) + int(ty) * S::TH;  // first output pixel
  median_bs_tile<KS>(src + uint64_t(f) * a.src_frame_stride, a, x0, y0, s_w, s_raw, tid);
  uint8_t *fdst = dst + uint64_t(f) * a.dst_frame_stride;
  {
    const uint8_t *ob = reinterpret_cast<const uint8_t *>(s_w);
    const uint32_t x_end = a.out_x0 + a.out_w, y_end = a.out_y0 + a.out_h;
    for (uint32_t c = tid; c < uint32_t(S::TW * S::TH / 16); c += uint32_t(S::THREADS)) {
      const uint32_t r = c / uint32_t(S::TW / 16), xo = 16u * (c - r * uint32_t(S::TW / 16));
      const uint32_t oy = uint32_t(y0) + r, ox = uint32_t(x0) + xo;
      if (oy >= y_end || ox >= x_end) continue;
      uint8_t *o = fdst + uint64_t(oy) * a.dst_row_stride + ox;
      const uint8_t *i = ob + r * uint32_t(S::OUT_STRIDE) + xo;
      if (ox + 16u <= x_end) {
        __builtin_memcpy(o, i, 16);
      } else {
        for (uint32_t k = 0; ox + k < x_end; ++k) o[k] = i[k];
      }
    }
  }
}

#if D2PC_EXPERIMENTS
// the lane-pair form of the select (d2pc_median_bs_tile.hpp, select2): 512 threads per tile, four waves per SIMD
template <int KS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_median_bs2_u8(
    const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const MedianArgs a) {
  using S = MedianBsShape<KS>;
  __shared__ __attribute__((aligned(16))) uint32_t s_w[S::W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_raw[S::RAW_WORDS];
  const uint32_t tid = threadIdx.x;
  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int x0 = int(a.out_x0) + int(tx) * S::TW, y0 = int(a.out_y0) + int(ty) * S::TH;
  median_bs2_tile<KS>(src + uint64_t(f) * a.src_frame_stride, a, x0, y0, s_w, s_raw, tid);
  uint8_t *fdst = dst + uint64_t(f) * a.dst_frame_stride;
  const uint8_t *ob = reinterpret_cast<const uint8_t *>(s_w);
  const uint32_t x_end = a.out_x0 + a.out_w, y_end = a.out_y0 + a.out_h;
  for (uint32_t c = tid; c < uint32_t(S::TW * S::TH / 16); c += 512u) {
    const uint32_t r = c / uint32_t(S::TW / 16), xo = 16u * (c - r * uint32_t(S::TW / 16));
    const uint32_t oy = uint32_t(y0) + r, ox = uint32_t(x0) + xo;
    if (oy >= y_end || ox >= x_end) continue;
    uint8_t *o = fdst + uint64_t(oy) * a.dst_row_stride + ox;
    const uint8_t *i = ob + r * uint32_t(S::OUT_STRIDE) + xo;
    if (ox + 16u <= x_end) {
      __builtin_memcpy(o, i, 16);
    } else {
      for (uint32_t k = 0; ox + k < x_end; ++k) o[k] = i[k];
    }
  }
}
#endif

namespace {
template <int KS>
hipError_t launch_bs(const uint8_t *s, uint8_t *d, MedianArgs a, hipStream_t stream) {
  using S = MedianBsShape<KS>;
  a.tiles_x = (a.out_w + S::TW - 1) / S::TW;
  a.tiles_y = (a.out_h + S::TH - 1) / S::TH;
  const uint64_t blocks = uint64_t(a.tiles_x) * a.tiles_y * a.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
#if D2PC_EXPERIMENTS
  if (a.algo == 3) {  // the lane-pair select (experiment)
    hipLaunchKernelGGL(k_median_bs2_u8<KS>, dim3(uint32_t(blocks)), dim3(512), 0, stream, s, d, a);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(k_median_bs_u8<KS>, dim3(uint32_t(blocks)), dim3(S::THREADS), 0, stream, s, d, a);
  return hipGetLastError();
}
}  // namespace

// `a` arrives with the output rectangle resolved (launch_median does that)
uint64_t median_bs_tiles(const MedianArgs &a) {
  using S = MedianBsShape<11>;
  return uint64_t((a.out_w + S::TW - 1) / S::TW) * ((a.out_h + S::TH - 1) / S::TH) * a.n_frames;
}
hipError_t launch_median_bs(const void *src, void *dst, const MedianArgs &a, int ksize, hipStream_t stream) {
  const uint8_t *s = static_cast<const uint8_t *>(src);
  uint8_t *d = static_cast<uint8_t *>(dst);
  switch (ksize) {
    case 3: return launch_bs<3>(s, d, a, stream);
    case 5: return launch_bs<5>(s, d, a, stream);
    case 7: return launch_bs<7>(s, d, a, stream);
    case 9: return launch_bs<9>(s, d, a, stream);
    case 11: return launch_bs<11>(s, d, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace d2pc
