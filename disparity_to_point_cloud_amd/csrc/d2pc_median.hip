// d2pc_median.hip -- k x k median of an 8-bit image on gfx950 (k odd, <= 11),
// BORDER_REPLICATE: the device form of cv::medianBlur(img, out, 11) at
// reference src/disparity_to_point_cloud.cpp:55-57.
//
// Bit-plane radix select (no sorting, no histograms):
//  1. A block owns 16 x 64 output pixels.  Every input row of the tile (+halo)
//     is turned into eight 32-bit PLANE WORDS, bit j of plane b = bit b of the
//     pixel in column c0-r+j, by one lane per row with in-register bit
//     transposes.  The words go to LDS: plane[b][row].
//  2. The median of the k*k window of pixel (y,x) is found MSB-first.  The
//     candidate set is k row masks (k consecutive bits starting at bit x);
//     per plane: ones = cand & plane word, n1 = popcount (v_bcnt accumulates),
//     the rank decides whether the median's bit is 0 or 1 and the candidates
//     shrink to the matching half.  8 planes x k rows x 3 integer ops (and,
//     bcnt-accumulate, bitop3): ~280 ops per pixel for k = 11, against 2 x 968
//     for compare-and-count.
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
  const uint32_t tid = threadIdx.x;

  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int c0 = int(tx) * kMedTileW, y0 = int(ty) * kMedTileH;
  const uint8_t *fsrc = src + uint64_t(f) * a.src_frame_stride;
  uint8_t *fdst = dst + uint64_t(f) * a.dst_frame_stride;

  // ---- 1. bit planes of the tile's input rows: one LANE per row -----------------
  // Thread t < IN_ROWS fetches the 32 pixels of input row t and bit-transposes them in registers
  // (four 8x8 bit transposes + a 4x4 byte transpose per half: ~110 integer ops for all eight plane
  // words of the row).  One ballot per plane and row pair did the same with ~20 instructions per
  // row pair in EVERY wave: 1600 wave instructions per block against ~300 now.
  if (tid < uint32_t(IN_ROWS)) {
    int iy = y0 - R + int(tid);
    iy = iy < 0 ? 0 : iy >= int(a.height) ? int(a.height) - 1 : iy;  // replicate
    const uint8_t *row = fsrc + uint64_t(iy) * a.src_row_stride;
    const int cl = c0 - R;  // column of bit 0
    uint32_t px[8];         // pixels cl .. cl+31, four per dword
    if (cl >= 0 && cl + 31 < int(a.width)) {  // block-uniform: interior tile
      __builtin_memcpy(px, row + cl, 32);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        px[j] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int ix = cl + 4 * j + k;
          ix = ix < 0 ? 0 : ix >= int(a.width) ? int(a.width) - 1 : ix;  // replicate
          px[j] |= uint32_t(row[ix]) << (8 * k);
        }
      }
    }
    // 8 pixels (lo = pixels 0..3, hi = 4..7) -> byte p of (lo, hi) = bit p of the 8 pixels
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      uint32_t lo = px[j], hi = px[j + 1], t;
      t = (lo ^ (lo >> 7)) & 0x00aa00aau, lo ^= t ^ (t << 7);
      t = (hi ^ (hi >> 7)) & 0x00aa00aau, hi ^= t ^ (t << 7);
      t = (lo ^ (lo >> 14)) & 0x0000ccccu, lo ^= t ^ (t << 14);
      t = (hi ^ (hi >> 14)) & 0x0000ccccu, hi ^= t ^ (t << 14);
      t = (lo ^ ((lo >> 28) | (hi << 4))) & 0xf0f0f0f0u;
      lo ^= t ^ (t << 28);
      hi ^= t >> 4;
      px[j] = lo, px[j + 1] = hi;
    }
    // plane p = byte p of the four blocks: a 4x4 byte transpose of the lows (planes 0..3) and of the highs
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t b0 = px[h], b1 = px[2 + h], b2 = px[4 + h], b3 = px[6 + h];
      const uint32_t a0 = __builtin_amdgcn_perm(b1, b0, 0x05010400u), a1 = __builtin_amdgcn_perm(b1, b0, 0x07030602u);
      const uint32_t a2 = __builtin_amdgcn_perm(b3, b2, 0x05010400u), a3 = __builtin_amdgcn_perm(b3, b2, 0x07030602u);
      s_plane[4 * h + 0][tid] = __builtin_amdgcn_perm(a2, a0, 0x05040100u);
      s_plane[4 * h + 1][tid] = __builtin_amdgcn_perm(a2, a0, 0x07060302u);
      s_plane[4 * h + 2][tid] = __builtin_amdgcn_perm(a3, a1, 0x05040100u);
      s_plane[4 * h + 3][tid] = __builtin_amdgcn_perm(a3, a1, 0x07060302u);
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
      uint32_t word[KS];
      uint32_t n1 = 0;
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        word[r] = s_plane[pl][y + uint32_t(r)];
        n1 += uint32_t(__popc(cand[r] & word[r]));
      }
      const uint32_t n0 = ncand - n1;
      const bool bit1 = rank > n0;  // the median is among the elements whose bit is 1
      // keep the candidates whose bit equals the median's: cand & (word ^ flip) is ONE v_bitop3 per row
      // (as an intrinsic: written with operators, LLVM folds it into the next plane's AND and spends
      // a fourth instruction per row on the shared word ^ flip)
      const uint32_t flip = bit1 ? 0u : ~0u;
      if (pl > 0) {
#pragma unroll
        for (int r = 0; r < KS; ++r) cand[r] = __builtin_amdgcn_bitop3_b32(word[r], cand[r], flip, 0x48);
      }
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
