// d2pc_median.hip -- k x k median of an 8-bit image on gfx950 (k odd, <= 11),
// BORDER_REPLICATE: the device form of cv::medianBlur(img, out, 11) at
// reference src/disparity_to_point_cloud.cpp:55-57.
//
// Wave-ballot bit-plane radix select (no sorting, no histograms):
//  1. A block owns 16 x 64 output pixels.  For every input row of the tile
//     (+halo) a half-wave holds 32 consecutive pixels, one per lane, and ONE
//     __ballot per bit plane turns the row into a 32-bit word whose bit j is
//     that plane's bit of column c0-r+j: 8 ballots give all planes of two
//     rows.  The words go to LDS: plane[b][row].
//  2. The median of the k*k window of pixel (y,x) is found MSB-first.  The
//     candidate set is k row masks (k consecutive bits starting at bit x);
//     per plane: ones = cand & plane word, n1 = popcount (v_bcnt accumulates),
//     the rank decides whether the median's bit is 0 or 1 and the candidates
//     shrink to the matching half.  8 planes x k rows x ~4 integer ops:
//     ~370 ops per pixel for k = 11, against 2 x 968 for compare-and-count.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d2pc_launch.hpp"

namespace d2pc {

constexpr int kMedTileW = 16, kMedTileH = 64;

template <int KS>
__global__ __launch_bounds__(kBlock) void k_median_u8(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                      const MedianArgs a) {
  constexpr int R = KS / 2;
  constexpr int IN_ROWS = kMedTileH + 2 * R;
  constexpr int PAIRS = (IN_ROWS + 1) / 2;
  __shared__ uint32_t s_plane[8][2 * PAIRS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int c0 = int(tx) * kMedTileW, y0 = int(ty) * kMedTileH;
  const uint8_t *fsrc = src + uint64_t(f) * a.src_frame_stride;
  uint8_t *fdst = dst + uint64_t(f) * a.dst_frame_stride;

  // ---- 1. bit planes of the tile's input rows, two rows per ballot ----------
  // All of a wave's row loads are issued before the first is used: the loop
  // is latency-bound otherwise (one ~2 us round trip per row pair, ten pairs).
  const int j = int(lane & 31u);
  int ix = c0 - R + j;
  ix = ix < 0 ? 0 : ix >= int(a.width) ? int(a.width) - 1 : ix;  // replicate
  constexpr int PER_WAVE = (PAIRS + kBlock / 64 - 1) / (kBlock / 64);
  uint32_t v[PER_WAVE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int rp = int(wave) + i * (kBlock / 64);
    int iy = y0 - R + 2 * rp + int(lane >> 5);
    iy = iy < 0 ? 0 : iy >= int(a.height) ? int(a.height) - 1 : iy;  // also keeps rp >= PAIRS in bounds
    v[i] = fsrc[uint64_t(iy) * a.src_row_stride + uint32_t(ix)];
  }
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int rp = int(wave) + i * (kBlock / 64);
    if (rp < PAIRS) {  // wave-uniform
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint64_t m = __ballot((v[i] >> p) & 1u);
        if (lane == 0) {
          s_plane[p][2 * rp] = uint32_t(m);
          s_plane[p][2 * rp + 1] = uint32_t(m >> 32);
        }
      }
    }
  }
  __syncthreads();

  // ---- 2. radix select per output pixel ---------------------------------------
  constexpr uint32_t kField = (1u << KS) - 1u;
#pragma unroll 1
  for (int i = 0; i < kMedTileW * kMedTileH / kBlock; ++i) {
    const uint32_t p = tid + uint32_t(i) * kBlock;
    const uint32_t y = p >> 4, x = p & 15u;
    uint32_t cand[KS];
#pragma unroll
    for (int r = 0; r < KS; ++r) cand[r] = kField << x;
    uint32_t rank = uint32_t(KS * KS / 2) + 1u;  // 1-based rank of the median
    uint32_t ncand = uint32_t(KS * KS);
    uint32_t med = 0;
#pragma unroll
    for (int pl = 7; pl >= 0; --pl) {
      uint32_t ones[KS];
      uint32_t n1 = 0;
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        ones[r] = cand[r] & s_plane[pl][y + uint32_t(r)];
        n1 += uint32_t(__popc(ones[r]));
      }
      const uint32_t n0 = ncand - n1;
      const bool bit1 = rank > n0;  // the median is among the elements whose bit is 1
#pragma unroll
      for (int r = 0; r < KS; ++r) cand[r] = bit1 ? ones[r] : (cand[r] ^ ones[r]);
      rank = bit1 ? rank - n0 : rank;
      ncand = bit1 ? n1 : n0;
      med |= bit1 ? (1u << pl) : 0u;
    }
    const uint32_t oy = uint32_t(y0) + y, ox = uint32_t(c0) + x;
    if (oy < a.height && ox < a.width) fdst[uint64_t(oy) * a.dst_row_stride + ox] = uint8_t(med);
  }
}

bool median_ksize_supported(int k) { return k == 3 || k == 5 || k == 7 || k == 9 || k == 11; }

hipError_t launch_median(const void *src, void *dst, const MedianArgs &a0, int ksize, hipStream_t stream) {
  MedianArgs a = a0;
  a.tiles_x = (a.width + kMedTileW - 1) / kMedTileW;
  a.tiles_y = (a.height + kMedTileH - 1) / kMedTileH;
  const uint64_t blocks = uint64_t(a.tiles_x) * a.tiles_y * a.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid{uint32_t(blocks)}, block{kBlock};
  const uint8_t *s = static_cast<const uint8_t *>(src);
  uint8_t *d = static_cast<uint8_t *>(dst);
  switch (ksize) {
    case 3: hipLaunchKernelGGL(k_median_u8<3>, grid, block, 0, stream, s, d, a); break;
    case 5: hipLaunchKernelGGL(k_median_u8<5>, grid, block, 0, stream, s, d, a); break;
    case 7: hipLaunchKernelGGL(k_median_u8<7>, grid, block, 0, stream, s, d, a); break;
    case 9: hipLaunchKernelGGL(k_median_u8<9>, grid, block, 0, stream, s, d, a); break;
    case 11: hipLaunchKernelGGL(k_median_u8<11>, grid, block, 0, stream, s, d, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace d2pc
