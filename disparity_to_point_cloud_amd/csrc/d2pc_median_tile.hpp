// d2pc_median_tile.hpp -- the tile body of the per-pixel median kernel (d2pc_median.hip).
//
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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d2pc_launch.hpp"

namespace d2pc {

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
  // LDS: pair[segment][input row][plane] -- the eight planes of a row pair side by side, so that the select
  // loop reads four planes with one ds_read_b128 (256 B/clk/CU; ds_read2_b32/_b64 get half that) at an
  // immediate offset from ONE base address per pixel; the segments are four dwords (banks) apart so that
  // the three segments a 16-lane group touches do not collide
  static constexpr int SG_STRIDE = IN_ROWS * 8 + 4;
  static constexpr int LDS_WORDS = NSEG * SG_STRIDE;
  static_assert(SEG_PX % 2 == 0 && TW % 2 == 0, "a pixel pair never straddles two segments");
  static_assert((NSEG - 1) * SEG_PX + 16 <= 32, "segments must lie inside the plane word");
  static_assert(TW + KS - 1 <= 32, "the tile's windows must lie inside the plane word");
  static_assert(ITEMS % THREADS == 0, "whole passes over the tile");
  static_assert(IN_ROWS <= 2 * 63 && THREADS >= 128, "two waves of 63 row pairs cover the input rows");
};

// One TW x 64 output tile at (c0, y0) of one frame.  Called by EVERY thread of the block (it contains the
// block barrier); threads tid >= MedianShape<KS>::THREADS only take part in the barriers.
template <int KS>
__device__ __forceinline__ void median_tile(const uint8_t *__restrict__ fsrc, uint8_t *__restrict__ fdst,
                                            const MedianArgs &a, const int c0, const int y0,
                                            uint32_t (&s_pair)[MedianShape<KS>::LDS_WORDS], const uint32_t tid) {
  using S = MedianShape<KS>;
  constexpr int R = S::R, IN_ROWS = S::IN_ROWS;
  const bool active = tid < uint32_t(S::THREADS);  // wave-uniform (THREADS is a multiple of 64)
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
    uint32_t below[8];
#pragma unroll
    for (int pl = 0; pl < 8; ++pl)
      below[pl] = uint32_t(__builtin_amdgcn_update_dpp(0, int(plane[pl]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
#pragma unroll
    for (int sg = 0; sg < S::NSEG; ++sg) {
#pragma unroll
      for (int pl = 0; pl < 8; pl += 4) {
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = ((plane[pl + i] >> (sg * S::SEG_PX)) & 0xffffu) | ((below[pl + i] >> (sg * S::SEG_PX)) << 16);
        if (writer) *reinterpret_cast<uint4 *>(&s_pair[sg * S::SG_STRIDE + int(in_row) * 8 + pl]) = make_uint4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  __syncthreads();

  // ---- 2. radix select, NPX horizontally adjacent pixels per thread ------------------------------
  constexpr uint32_t kField = (1u << KS) - 1u;
  constexpr int NPX = S::NPX;
  constexpr uint32_t TWP = S::TW / NPX;  // items per tile row; item p = tid + it * THREADS sits at (p / TWP, NPX * (p % TWP))
  uint32_t y = tid / TWP, x = uint32_t(NPX) * (tid - y * TWP);
#pragma unroll 1
  for (int it = 0; active && it < S::ITEMS / S::THREADS; ++it) {
    const uint32_t sg = x / uint32_t(S::SEG_PX), xs = x - sg * uint32_t(S::SEG_PX);
    const uint32_t *base = &s_pair[sg * uint32_t(S::SG_STRIDE) + y * 8u];
    uint32_t cand[NPX][S::NREG];
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
      const uint32_t one_row = kField << (xs + uint32_t(q));  // <= 16 bits by construction
#pragma unroll
      for (int j = 0; j < S::NREG; ++j) cand[q][j] = one_row | (one_row << 16);
      if (KS & 1) cand[q][S::NREG - 1] = one_row;  // the last register holds one window row only
    }
    // With c candidates left and the median the (a+1)-th smallest of them, only d = c - a - 1 has to be
    // carried, as m = -d - 1 < 0: the popcounts of a plane accumulate ON m (v_bcnt's addend), e = m + ones,
    // and the median's bit is 0  <=>  ones <= d  <=>  e < 0.  Then d becomes d - ones (m = e), otherwise it
    // stays: m = max_u32(e, m) covers both (e >= m as signed numbers; a non-negative e is the smaller
    // unsigned).  Three integer ops per plane next to the three per register.
    int32_t m[NPX], acc[NPX];  // acc: minus the number of zero bits' weights = median - 255, built MSB first
#pragma unroll
    for (int q = 0; q < NPX; ++q) m[q] = -(KS * KS - (KS * KS / 2 + 1)) - 1, acc[q] = 0;
#pragma unroll
    for (int ph = 1; ph >= 0; --ph) {  // planes 7..4, then 3..0: four planes of a row pair are ONE ds_read_b128
      uint4 w4[S::NREG];
#pragma unroll
      for (int j = 0; j < S::NREG; ++j) w4[j] = *reinterpret_cast<const uint4 *>(base + 16 * j + 4 * ph);
#pragma unroll
      for (int h = 3; h >= 0; --h) {
        uint32_t word[S::NREG];
#pragma unroll
        for (int j = 0; j < S::NREG; ++j) word[j] = h == 3 ? w4[j].w : h == 2 ? w4[j].z : h == 1 ? w4[j].y : w4[j].x;
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
          uint32_t e = uint32_t(m[q]);
#pragma unroll
          for (int j = 0; j < S::NREG; ++j) {
            // a chain of accumulating popcounts (written with operators, LLVM re-associates it into a tree
            // with extra additions)
            const uint32_t ones = cand[q][j] & word[j];
            asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(e) : "v"(ones), "v"(e));
          }
          const int32_t is0 = int32_t(e) >> 31;  // all ones when the median's bit is 0
          // keep the candidates whose bit equals the median's: cand & (word ^ is0) is ONE v_bitop3 per
          // register (as an intrinsic: written with operators, LLVM folds it into the next plane's AND
          // and spends a fourth instruction on the shared term)
          if (ph > 0 || h > 0) {
#pragma unroll
            for (int j = 0; j < S::NREG; ++j)
              cand[q][j] = __builtin_amdgcn_bitop3_b32(word[j], cand[q][j], uint32_t(is0), 0x48);
          }
          m[q] = int32_t(e > uint32_t(m[q]) ? e : uint32_t(m[q]));
          acc[q] = (acc[q] << 1) + is0;
        }
      }
    }
    const uint32_t oy = uint32_t(y0) + y, ox = uint32_t(c0) + x;
    if (oy < a.out_y0 + a.out_h) {
      uint8_t *o = fdst + uint64_t(oy) * a.dst_row_stride + ox;
#pragma unroll
      for (int q = 0; q < NPX; ++q)
        if (ox + uint32_t(q) < a.out_x0 + a.out_w) {
          o[q] = uint8_t(255 + acc[q]);
        }
    }
    // the next item, THREADS further on
    y += uint32_t(S::THREADS) / TWP;
    x += uint32_t(NPX) * (uint32_t(S::THREADS) % TWP);
    if (x >= uint32_t(S::TW)) x -= uint32_t(S::TW), ++y;
  }
}

}  // namespace d2pc
