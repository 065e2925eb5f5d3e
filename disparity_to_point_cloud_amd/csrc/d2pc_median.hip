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

namespace d2pc {

namespace {

template <int KS>
struct MedianShape {
  static constexpr int R = KS / 2;
  static constexpr int SEG_PX = 17 - KS;           // output pixels served by one 16-bit segment
  static constexpr int NSEG = 16 / SEG_PX + 1;     // segments cut from a 32-bit plane word
  static constexpr int TW = NSEG * SEG_PX;         // tile width  (k=11: 18, 9: 24, 7: 20, 5: 24, 3: 28)
  static constexpr int TH = 64;                    // tile height
  static constexpr int IN_ROWS = TH + 2 * R;
  static constexpr int NREG = (KS + 1) / 2;        // row pairs per window
  // Small windows: a thread selects for TWO horizontally adjacent pixels (they share every pair word:
  // 3x3 +6 %, same device); at 9x9 and 11x11 the second pixel's registers cost a wave of occupancy (-3 %).
  static constexpr int NPX = KS <= 5 ? 2 : 1;
  static constexpr int ITEMS = TW / NPX * TH;
  static constexpr int THREADS = ITEMS % 256 == 0 ? 256 : ITEMS % 192 == 0 ? 192 : ITEMS % 320 == 0 ? 320 : 128;
  static_assert(SEG_PX % 2 == 0 && TW % 2 == 0, "a pixel pair never straddles two segments");
  static_assert((NSEG - 1) * SEG_PX + 16 <= 32, "segments must lie inside the plane word");
  static_assert(TW + KS - 1 <= 32, "the tile's windows must lie inside the plane word");
  static_assert(ITEMS % THREADS == 0, "whole passes over the tile");
  static_assert(IN_ROWS <= 2 * 63 && THREADS >= 128, "two waves of 63 row pairs cover the input rows");
};

}  // namespace

template <int KS>
__global__ __launch_bounds__(MedianShape<KS>::THREADS) void k_median_u8(const uint8_t *__restrict__ src,
                                                                        uint8_t *__restrict__ dst,
                                                                        const MedianArgs a) {
  using S = MedianShape<KS>;
  constexpr int R = S::R, IN_ROWS = S::IN_ROWS;
  __shared__ uint32_t s_pair[8][S::NSEG][IN_ROWS];
  const uint32_t tid = threadIdx.x;

  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int c0 = int(a.out_x0) + int(tx) * S::TW, y0 = int(a.out_y0) + int(ty) * S::TH;
  const uint8_t *fsrc = src + uint64_t(f) * a.src_frame_stride;
  uint8_t *fdst = dst + uint64_t(f) * a.dst_frame_stride;

  // ---- 1. packed plane segments of the tile's input rows: one LANE per row ---------
  // Wave w takes input rows 63w .. 63w+63: lane l needs the plane words of the row below it, which
  // lane l+1 of the SAME wave holds (one DPP move per plane); lane 63 only serves as that partner,
  // its own row is lane 0 of the next wave.  Two waves cover the <= 74 rows.
  const uint32_t lane = tid & 63u;
  const uint32_t in_row = (tid >> 6) * 63u + lane;
  if ((tid >> 6) * 63u < uint32_t(IN_ROWS)) {  // wave-uniform
    int iy = y0 - R + int(in_row);
    iy = iy < 0 ? 0 : iy >= int(a.height) ? int(a.height) - 1 : iy;  // replicate (also keeps rows past IN_ROWS in bounds)
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
    uint32_t plane[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t b0 = px[h], b1 = px[2 + h], b2 = px[4 + h], b3 = px[6 + h];
      const uint32_t a0 = __builtin_amdgcn_perm(b1, b0, 0x05010400u), a1 = __builtin_amdgcn_perm(b1, b0, 0x07030602u);
      const uint32_t a2 = __builtin_amdgcn_perm(b3, b2, 0x05010400u), a3 = __builtin_amdgcn_perm(b3, b2, 0x07030602u);
      plane[4 * h + 0] = __builtin_amdgcn_perm(a2, a0, 0x05040100u);
      plane[4 * h + 1] = __builtin_amdgcn_perm(a2, a0, 0x07060302u);
      plane[4 * h + 2] = __builtin_amdgcn_perm(a3, a1, 0x05040100u);
      plane[4 * h + 3] = __builtin_amdgcn_perm(a3, a1, 0x07060302u);
    }
    // 16-bit segments of this row and the next, packed
    const bool writer = lane < 63u && in_row < uint32_t(IN_ROWS);
#pragma unroll
    for (int pl = 0; pl < 8; ++pl) {
      const uint32_t below = uint32_t(__builtin_amdgcn_update_dpp(0, int(plane[pl]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
#pragma unroll
      for (int sg = 0; sg < S::NSEG; ++sg) {
        const uint32_t v = ((plane[pl] >> (sg * S::SEG_PX)) & 0xffffu) | ((below >> (sg * S::SEG_PX)) << 16);
        if (writer) s_pair[pl][sg][in_row] = v;
      }
    }
  }
  __syncthreads();

  // ---- 2. radix select, NPX horizontally adjacent pixels per thread ------------------------------
  constexpr uint32_t kField = (1u << KS) - 1u;
  constexpr int NPX = S::NPX;
#pragma unroll 1
  for (int it = 0; it < S::ITEMS / S::THREADS; ++it) {
    const uint32_t p = tid + uint32_t(it) * uint32_t(S::THREADS);
    const uint32_t y = p / uint32_t(S::TW / NPX), x = uint32_t(NPX) * (p - y * uint32_t(S::TW / NPX));
    const uint32_t sg = x / uint32_t(S::SEG_PX), xs = x - sg * uint32_t(S::SEG_PX);
    uint32_t cand[NPX][S::NREG];
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
      const uint32_t one_row = kField << (xs + uint32_t(q));  // <= 16 bits by construction
#pragma unroll
      for (int j = 0; j < S::NREG; ++j) cand[q][j] = one_row | (one_row << 16);
      if (KS & 1) cand[q][S::NREG - 1] = one_row;  // the last register holds one window row only
    }
    // With c candidates left and the median the (a+1)-th smallest of them, only d = c - a - 1 has to
    // be carried: the median's bit is 1  <=>  zeros <= a  <=>  z = d - ones < 0; then d stays (c and a
    // shrink by the same number of zeros), otherwise d = z.  Five integer ops per plane.
    int32_t d[NPX], acc[NPX];  // acc: minus the median, built MSB first
#pragma unroll
    for (int q = 0; q < NPX; ++q) d[q] = KS * KS - (KS * KS / 2 + 1), acc[q] = 0;
#pragma unroll
    for (int pl = 7; pl >= 0; --pl) {
      uint32_t word[S::NREG];
#pragma unroll
      for (int j = 0; j < S::NREG; ++j) word[j] = s_pair[pl][sg][y + 2u * uint32_t(j)];
#pragma unroll
      for (int q = 0; q < NPX; ++q) {
        uint32_t n1 = 0;
#pragma unroll
        for (int j = 0; j < S::NREG; ++j) n1 += uint32_t(__popc(cand[q][j] & word[j]));
        const int32_t z = d[q] - int32_t(n1);
        const int32_t is1 = z >> 31;  // all ones when the median's bit is 1
        // keep the candidates whose bit equals the median's: cand & ~(word ^ is1) is ONE v_bitop3 per
        // register (as an intrinsic: written with operators, LLVM folds it into the next plane's AND
        // and spends a fourth instruction on the shared term)
        if (pl > 0) {
#pragma unroll
          for (int j = 0; j < S::NREG; ++j)
            cand[q][j] = __builtin_amdgcn_bitop3_b32(word[j], cand[q][j], uint32_t(is1), 0x84);
        }
        d[q] = z + (int32_t(n1) & is1);
        acc[q] = (acc[q] << 1) + is1;
      }
    }
    const uint32_t oy = uint32_t(y0) + y, ox = uint32_t(c0) + x;
    if (oy < a.out_y0 + a.out_h) {
      uint8_t *o = fdst + uint64_t(oy) * a.dst_row_stride + ox;
#pragma unroll
      for (int q = 0; q < NPX; ++q)
        if (ox + uint32_t(q) < a.out_x0 + a.out_w) o[q] = uint8_t(-acc[q]);
    }
  }
}

// ---------------------------------------------------------------------------
// cv_bridge::toCvCopy(msg, "mono8") on a mono16 image (reference
// src/disparity_to_point_cloud.cpp:50): image.convertTo(CV_8U, 255./65535.) =
// saturate_cast<uchar>(cvRound(v * (float)(255./65535.))), product in float,
// round-half-to-even.  Eight pixels per thread (one 16-byte load, one 8-byte store when aligned).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_mono16_to_mono8(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                            const MedianArgs a) {
  const uint32_t groups = (a.width + 7u) / 8u;  // eight pixels per thread and row
  const uint64_t total = uint64_t(groups) * a.height * a.n_frames;
  const float scale = float(255. / 65535.);
  auto cvt = [&](uint32_t v) {
    const float r = __builtin_rintf(float(v) * scale);  // default rounding mode: nearest-even
    return uint32_t(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
  };
  for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < total; i += uint64_t(gridDim.x) * kBlock) {
    const uint32_t gx = uint32_t(i % groups);
    const uint64_t r = i / groups;
    const uint32_t y = uint32_t(r % a.height), f = uint32_t(r / a.height);
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
  const uint64_t total = uint64_t((a.width + 7u) / 8u) * a.height * a.n_frames;
  if (total == 0) return hipErrorInvalidValue;
  const uint64_t want = (total + kBlock - 1) / kBlock;
  const uint32_t grid = uint32_t(want < 65536u ? want : 65536u);
  hipLaunchKernelGGL(k_mono16_to_mono8, dim3(grid), dim3(kBlock), 0, stream, static_cast<const uint8_t *>(src),
                     static_cast<uint8_t *>(dst), a);
  return hipGetLastError();
}

bool median_ksize_supported(int k) { return k == 3 || k == 5 || k == 7 || k == 9 || k == 11; }

namespace {
template <int KS>
hipError_t launch_k(const uint8_t *s, uint8_t *d, MedianArgs a, hipStream_t stream) {
  using S = MedianShape<KS>;
  if (a.out_w == 0 || a.out_h == 0) {  // whole image
    a.out_x0 = a.out_y0 = 0;
    a.out_w = a.width;
    a.out_h = a.height;
  }
  if (a.out_x0 + a.out_w > a.width || a.out_y0 + a.out_h > a.height) return hipErrorInvalidValue;
  a.tiles_x = (a.out_w + S::TW - 1) / S::TW;
  a.tiles_y = (a.out_h + S::TH - 1) / S::TH;
  const uint64_t blocks = uint64_t(a.tiles_x) * a.tiles_y * a.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_median_u8<KS>, dim3(uint32_t(blocks)), dim3(S::THREADS), 0, stream, s, d, a);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_median(const void *src, void *dst, const MedianArgs &a, int ksize, hipStream_t stream) {
  const uint8_t *s = static_cast<const uint8_t *>(src);
  uint8_t *d = static_cast<uint8_t *>(dst);
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
