// d2pc_median_bs_tile.hpp -- the tile body of d2pc_median_bs.hip, shared with the tile-fused callback kernel
// (k_callback_bs in d2pc_callback.hip).
//
// d2pc_median_bs.hip -- k x k median of an 8-bit image on gfx950, BIT-SLICED ACROSS PIXELS: the second
// device form of cv::medianBlur(img, out, 11) at reference src/disparity_to_point_cloud.cpp:55-57
// (SURVEY.md section 8(f) #1), used for large launches; d2pc_median.hip (one pixel per thread) serves the rest.
//
// Why a second kernel.  tools/valu_rate.hip: on gfx950 the bitwise VALU instructions (v_and, v_xor,
// v_bitop3) issue every 2 cycles per SIMD, v_bcnt_u32_b32 and most other integer instructions every 4.  The
// per-pixel radix select spends a third of its instructions and half of its issue cycles on popcounts.
// Here a thread owns 32 output pixels of one row, ONE BIT PER PIXEL in every register, and the whole select
// is v_and / v_bitop3:
//   * cand[dy][dx] (k*k words): bit j set <=> window position (dy, dx) of pixel j is still a candidate
//   * per bit plane, MSB first: M = cand & W(dy, dx) marks the candidates whose bit is 1; the k*k words M are
//     summed PER BIT POSITION by a carry-save adder tree (a full adder = 2 v_bitop3 on 3 words) into an 8-bit
//     number held as 8 words; the rank test, the rank update and the candidate update cand &= W ^ is0 are
//     bitwise as well.  Per pixel and plane: k*k * (1 + ~2 + 1) / 32 = 15 two-cycle instructions for 11 x 11,
//     against 6 * 3 + 5 = 23 instructions (6 of them four-cycle) per pixel.
//   * W(dy, dx) is the bit plane of the pixels (x_j + dx, y + dy).  A thread's pixels are S = 8 COLUMNS APART,
//     x_j = X0 + 8 j + t (t = 0..7), so that shifting the window by dx moves to ANOTHER WORD instead of shifting
//     bits: with W_u[bit j] = pixel X0 + 8 j + u, thread t reads the words u = t .. t + k - 1 of a row, and
//     W_{u+8} = W_u >> 1 (plus one pixel of the next tile).  18 words per row and plane serve 256 pixels.
//   * LDS: W[plane][row][24 dwords] (18 used; the stride makes the ds_read_b64 of a 32-lane group -- 4 threads
//     x 8 rows -- hit 64 different banks).  Even and odd t run in different waves: a thread reads the six
//     aligned pairs from word (t & ~1) on and uses words [par .. par + 11) of them, so `par` must be uniform.
//   * both passes over the window (count, then candidate update) read W from LDS: 2 * 66 ds_read_b64 per
//     plane and thread; the k*k candidate words stay in registers (121 + ~40: two or three waves per SIMD).
//
// The result is bit-identical to d2pc_median.hip and to the oracle (tests/test_median_gpu.py runs both).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d2pc_launch.hpp"

// A/B switches (make variant NAME=x DEFS=-D...; tools/ab_callback.py): waves per SIMD the bit-sliced kernels are compiled for,
// and how many of the window's rows keep their plane words in registers from the count pass to the candidate update (each
// such row saves one LDS read of 12 words per plane and costs 12 registers: 3 waves per SIMD leave room for none)
#ifndef D2PC_BS_WAVES
#define D2PC_BS_WAVES 3
#endif
#ifndef D2PC_BS_KEEP_ROWS
#define D2PC_BS_KEEP_ROWS 0
#endif
#ifndef D2PC_BS_NO_LDS
#define D2PC_BS_NO_LDS 0
#endif
#ifndef D2PC_BS_PEEL_MSB
#define D2PC_BS_PEEL_MSB 1
#endif
#ifndef D2PC_BS_TRIM_READS
#define D2PC_BS_TRIM_READS 1
#endif
// D2PC_BS_GROUP_ROWS = GR: GR adjacent lanes of the select take GR vertically adjacent output rows of the same columns and visit
// their window rows by ABSOLUTE row (see select): most of a group's LDS reads are then the same words, served by one bank access.
// Interleaved on one device (profiles/r06_ab_median_group_rows.txt): PARITY body 578.3 -> 569.8 / 567.0 / 566.2 us for GR = 2 / 4 / 8
// (shader clock 1.89 -> 1.92 GHz at the same 1,395 W), the bare median kernel 434.5 -> 426.8 / 424.2 / 420.9; the persistent COMPACT body
// is best at 4 (8 costs it five spilled registers).  1 = rounds 2-5's mapping (lane = (t >> 1) + 4 x row).
#ifndef D2PC_BS_GROUP_ROWS
#define D2PC_BS_GROUP_ROWS 4
#endif
// D2PC_BS_PRIO = 1 (the product since round 6): the stages AROUND the select (staging, plane words, bytes back, the callers' count /
// scatter / epilogue stages) run at a raised wave priority (s_setprio 3), the select at the default.  A SIMD arbitrates vector issue
// between its waves by priority, then age: the few, latency-bound instructions of those stages used to queue behind the co-resident
// blocks' select streams (the plane-word stage took 8,200 cycles for ~260 instructions per thread); ahead of them the stages end
// sooner, more of a block's life is select, and the chip -- at its socket power cap under this kernel -- runs a little wider and
// slower: PARITY body 586.7 -> 573.0 us (2.06 -> 2.00 GHz), the bare median kernel 448 -> 433 us, the COMPACT bodies 678 -> 642 /
// 715 -> 689 us, interleaved on one device (profiles/r06_ab_callback_prio.txt).  0 = everything at the default priority.
#ifndef D2PC_BS_PRIO
#define D2PC_BS_PRIO 1
#endif
#if D2PC_BS_PRIO
#define D2PC_BS_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define D2PC_BS_SETPRIO(p)
#endif

namespace d2pc {

#ifdef D2PC_DIAG
// diagnostic build: shader-clock sums per stage of the tile body, added by wave 0's lane 0 of every block
// (tools/diag_callback.py): [0] tiles, [1] rows requested -> staged in LDS, [2] plane words, [3] select, [4] bytes back,
// [5] the caller's epilogue (COMPACT body: table + barrier), [6] block lifetime, [7] the same in 100 MHz ticks, COMPACT body:
// [8] count + barrier, [9] publish + place (wave 0; the last wave waits here), [10] scatter, [11] tile kept + barriers;
// [16..31] the same stamps for the block's LAST wave
// (256 slots of 32 words, two 128-byte lines each, picked by the block index: thousands of blocks adding to ONE word serialise
// at the memory side and tripled the kernel's time in the first form of this diagnostic)
inline __device__ unsigned long long g_bs_diag[256][32];
#define D2PC_BS_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define D2PC_BS_ADD(i, v) \
  do { if ((tid & 63u) == 0u && ((tid >> 6) == 0u || (tid >> 6) == 3u)) atomicAdd(&g_bs_diag[blockIdx.x & 255u][(i) + ((tid >> 6) == 3u ? 16 : 0)], (unsigned long long)(v)); } while (0)
#else
#define D2PC_BS_STAMP(x)
#define D2PC_BS_ADD(i, v)
#endif

template <int KS>
struct MedianBsShape {
  static constexpr int R = KS / 2;
  static constexpr int S = 8;                     // column distance of a thread's pixels
  static constexpr int TW = 32 * S;               // tile width (output pixels)
  static constexpr int TH = 32;                   // tile height
  static constexpr int THREADS = S * TH;          // one thread per (t, row)
  static constexpr int IN_ROWS = TH + KS - 1;
  static constexpr int NW = S + KS - 1;           // words W_0 .. W_{NW-1} per plane and row
  static constexpr int ROW_STRIDE = 24;           // dwords; 24 r mod 64 takes 8 different multiples of 8 for 8 rows
  static constexpr int PLANE_STRIDE = IN_ROWS * ROW_STRIDE;
  static constexpr int W_WORDS = 8 * PLANE_STRIDE;
  static constexpr int NREAD = (KS + 2) / 2;      // aligned word pairs that cover words [par, par + KS)
  static constexpr int RAW_STRIDE = 272;          // bytes per staged input row (>= TW + 2 * 8 = every byte the gather touches)
  static constexpr int RAW_WORDS = IN_ROWS * RAW_STRIDE / 4;
  static constexpr int D0 = KS * KS - (KS * KS / 2 + 1);  // candidates above the median at the start
  static_assert(KS % 2 == 1 && KS >= 3 && KS <= 11, "k*k < 128: the rank arithmetic is 7 + 1 bits");
  static_assert(2 * NREAD + 6 <= ROW_STRIDE && NW <= ROW_STRIDE, "reads stay inside the row's slot");
  static_assert(RAW_WORDS * 4 >= 8 * THREADS * 4, "the staged rows' space later holds 8 result words per thread");
  static constexpr int OUT_STRIDE = TW + 16;      // bytes per staged output row: rows 4 banks apart (byte stores of 8 rows x 2 dwords)
  static_assert(W_WORDS * 4 >= OUT_STRIDE * TH, "W's space later holds the tile's output bytes");
};

namespace bs {

template <uint32_t TABLE>  // bit (4a + 2b + c) of TABLE is the result for the input bits (a, b, c)
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, TABLE);
}

// 32 pixels (px[k] = pixels 4k .. 4k+3, one per byte) <-> eight plane words (bit j of plane[b] = bit b of pixel j).
// Both steps are transposes (8 x 8 bits inside a register pair, 4 x 4 bytes across four registers), hence
// their own inverses: to_planes runs them one way, to_pixels the other.
__device__ __forceinline__ void bit_transpose_8x8(uint32_t &lo, uint32_t &hi) {
  uint32_t t;
  t = (lo ^ (lo >> 7)) & 0x00aa00aau, lo ^= t ^ (t << 7);
  t = (hi ^ (hi >> 7)) & 0x00aa00aau, hi ^= t ^ (t << 7);
  t = (lo ^ (lo >> 14)) & 0x0000ccccu, lo ^= t ^ (t << 14);
  t = (hi ^ (hi >> 14)) & 0x0000ccccu, hi ^= t ^ (t << 14);
  t = (lo ^ ((lo >> 28) | (hi << 4))) & 0xf0f0f0f0u;
  lo ^= t ^ (t << 28);
  hi ^= t >> 4;
}
__device__ __forceinline__ void byte_transpose_4x4(const uint32_t b0, const uint32_t b1, const uint32_t b2, const uint32_t b3,
                                                   uint32_t (&o)[4]) {
  const uint32_t a0 = __builtin_amdgcn_perm(b1, b0, 0x05010400u), a1 = __builtin_amdgcn_perm(b1, b0, 0x07030602u);
  const uint32_t a2 = __builtin_amdgcn_perm(b3, b2, 0x05010400u), a3 = __builtin_amdgcn_perm(b3, b2, 0x07030602u);
  o[0] = __builtin_amdgcn_perm(a2, a0, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(a2, a0, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(a3, a1, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(a3, a1, 0x07060302u);
}
__device__ __forceinline__ void to_planes(uint32_t (&px)[8], uint32_t (&plane)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) bit_transpose_8x8(px[j], px[j + 1]);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t o[4];
    byte_transpose_4x4(px[h], px[2 + h], px[4 + h], px[6 + h], o);
#pragma unroll
    for (int i = 0; i < 4; ++i) plane[4 * h + i] = o[i];
  }
}
__device__ __forceinline__ void to_pixels(const uint32_t (&plane)[8], uint32_t (&px)[8]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t o[4];
    byte_transpose_4x4(plane[4 * h], plane[4 * h + 1], plane[4 * h + 2], plane[4 * h + 3], o);
#pragma unroll
    for (int i = 0; i < 4; ++i) px[2 * i + h] = o[i];
  }
#pragma unroll
  for (int j = 0; j < 8; j += 2) bit_transpose_8x8(px[j], px[j + 1]);
}

// Carry-save counter: per bit position, how many of the words added so far had that bit set.  Level L holds
// up to two pending words of weight 2^L; a third makes a full adder whose carry moves one level up.  Every
// n[] is a compile-time constant once the loops around add() are unrolled, so this is straight-line code.
struct Csa {
  uint32_t a[8], b[8];
  int n[8];
};
template <int L>
__device__ __forceinline__ void csa_add(Csa &c, const uint32_t x) {
  if constexpr (L < 8) {
    if (c.n[L] == 0) {
      c.a[L] = x, c.n[L] = 1;
    } else if (c.n[L] == 1) {
      c.b[L] = x, c.n[L] = 2;
    } else {
      const uint32_t s = bitop3<0x96>(c.a[L], c.b[L], x);   // a ^ b ^ x
      const uint32_t cy = bitop3<0xe8>(c.a[L], c.b[L], x);  // majority
      c.a[L] = s, c.n[L] = 1;
      csa_add<L + 1>(c, cy);
    }
  }
}
// the pending words of all levels -> the binary digits s[0..7] of the count (count < 256)
__device__ __forceinline__ void csa_finish(Csa &c, uint32_t (&s)[8]) {
  bool have_carry = false;
  uint32_t carry = 0;
#pragma unroll
  for (int L = 0; L < 8; ++L) {
    const int inputs = c.n[L] + (have_carry ? 1 : 0);
    if (inputs == 0) {
      s[L] = 0, have_carry = false;
    } else if (inputs == 1) {
      s[L] = have_carry ? carry : c.a[L], have_carry = false;
    } else if (inputs == 2) {
      const uint32_t x = c.a[L], y = have_carry ? carry : c.b[L];
      s[L] = x ^ y, carry = x & y, have_carry = true;
    } else {
      s[L] = bitop3<0x96>(c.a[L], c.b[L], carry);
      carry = bitop3<0xe8>(c.a[L], c.b[L], carry), have_carry = true;
    }
  }
}

// One ds_read_b64 (256 B/clk/CU).  Volatile: left to itself the compiler drops the half of the first or last
// pair that the parity does not use and re-pairs the rest as ds_read2_b32, which runs at half that rate.
__device__ __forceinline__ uint2 ld_pair(const uint32_t *p) {
  typedef const volatile __attribute__((address_space(3))) uint64_t *lds_u64;  // (volatile hides the address space)
  const uint64_t q = *(lds_u64)(p);
  return make_uint2(uint32_t(q), uint32_t(q >> 32));
}

// The select of one thread: 32 pixels of output row `row`, columns 8 j + t.  `w_row` points at word (t & ~1)
// of plane 0, input row `row` (the window's first row).  Returns nothing: the median's bit planes go to
// bits_out[plane * THREADS].
// PAR >= 0 (the select: which of the NREAD aligned pairs' 2 NREAD words it uses is known: [PAR, PAR + KS)): the pair whose other half
// is never used is read as ONE word -- 11 dwords per row instead of 12 at 11 x 11.  The select's LDS reads are 28 % of the callback
// body's power (profiles/r06_energy_probe.txt).  PAR < 0: all pairs whole.
template <int KS, int PAR = -1>
__device__ __forceinline__ void ld_row(const uint32_t *p, uint32_t (&w)[2 * MedianBsShape<KS>::NREAD]) {
#if D2PC_BS_NO_LDS  // ENERGY PROBE ONLY (wrong results): the select's instructions without its LDS reads -- whatever the registers hold
#pragma unroll
  for (int i = 0; i < 2 * MedianBsShape<KS>::NREAD; ++i) asm volatile("; no LDS read" : "=v"(w[i]));
  (void)p;
#else
  constexpr int NR = MedianBsShape<KS>::NREAD;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool lo_used = PAR < 0 || (2 * i >= PAR && 2 * i < PAR + KS), hi_used = PAR < 0 || (2 * i + 1 >= PAR && 2 * i + 1 < PAR + KS);
    if (D2PC_BS_TRIM_READS && lo_used != hi_used) {
      typedef const volatile __attribute__((address_space(3))) uint32_t *lds_u32;
      w[2 * i + (hi_used ? 1 : 0)] = *(lds_u32)(p + 2 * i + (hi_used ? 1 : 0));
    } else if (lo_used || hi_used) {
      const uint2 v = ld_pair(p + 2 * i);
      w[2 * i] = v.x, w[2 * i + 1] = v.y;
    }
  }
#endif
}

template <int KS, int PAR, int GR = 1>
__device__ __forceinline__ void select(const uint32_t *__restrict__ w_row, uint32_t *__restrict__ bits_out, const uint32_t g = 0u) {
  // GR > 1 (D2PC_BS_GROUP_ROWS): GR adjacent LANES take GR vertically adjacent output rows of the same columns; `w_row` is the row of the
  // group's FIRST lane and `g` this lane's place in the group.  The window rows are visited by ABSOLUTE row: slot k holds the row R = k
  // (mod KS) of the lane's window [g, g + KS) -- row k for the lanes with g <= k, a later one for the others -- so that in most slots all the
  // lanes of a group ask LDS for the SAME words and one bank access serves them.  The order in which a window's rows are counted does not
  // matter.  Slots k >= GR - 1 are the same row for every lane; the others' offsets are per-lane registers (GR - 1 of them).
  using S = MedianBsShape<KS>;
  constexpr int NWORD = 2 * S::NREAD;
  constexpr int NOFF = (GR - 1 < KS ? GR - 1 : KS);   // slots whose row depends on the lane
  uint32_t off[NOFF > 0 ? NOFF : 1];
#pragma unroll
  for (int k = 0; k < NOFF; ++k) {
    const uint32_t wraps = g > uint32_t(k) ? (g - uint32_t(k) + uint32_t(KS - 1)) / uint32_t(KS) : 0u;   // R = k + KS * ceil((g - k) / KS)
    off[k] = (uint32_t(k) + uint32_t(KS) * wraps) * uint32_t(S::ROW_STRIDE);
  }
  auto row_off = [&](int dy) -> uint32_t { return dy < NOFF ? off[dy] : uint32_t(dy * S::ROW_STRIDE); };
  uint32_t cand[KS][KS];
  // mm = 127 - d as seven words (d = candidates above the median at the start): count + mm >= 128  <=>  count > d  <=>
  // the median's bit is 1; otherwise d becomes d - count, i.e. mm becomes the sum's low seven digits.
  uint32_t mm[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) mm[k] = ((127 - S::D0) >> k) & 1 ? 0xffffffffu : 0u;
  constexpr int KEEP = D2PC_BS_KEEP_ROWS < KS ? D2PC_BS_KEEP_ROWS : KS;
  uint32_t kept[KEEP > 0 ? KEEP : 1][NWORD];
#if D2PC_BS_PEEL_MSB
  // The MOST SIGNIFICANT plane on its own (round 6): every tap is still a candidate, so its count needs no AND with the candidate
  // words, its update is one XOR that CREATES them (cand = w ^ is0), and the k*k moves that used to set them to all-ones are gone:
  // 2 k*k instructions fewer per thread and tile (242 of ~4,800 at 11 x 11; under the power cap operations pay linearly).
  {
    const uint32_t *wp = w_row + 7 * S::PLANE_STRIDE;
    Csa c;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.a[k] = k < 7 ? mm[k] : 0u, c.b[k] = 0u, c.n[k] = k < 7 ? 1 : 0;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
      uint32_t w[NWORD];
      ld_row<KS, PAR>(wp + row_off(dy), w);
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) csa_add<0>(c, w[dx + PAR]);
    }
    uint32_t s[8];
    csa_finish(c, s);
    bits_out[7 * S::THREADS] = s[7];
    const uint32_t is0 = ~s[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) mm[k] = bitop3<0xca>(is0, s[k], mm[k]);  // is0 ? s : mm
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
      uint32_t w[NWORD];
      ld_row<KS, PAR>(wp + row_off(dy), w);
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) cand[dy][dx] = w[dx + PAR] ^ is0;
    }
  }
  constexpr int kFirstPlane = 6;
#else
#pragma unroll
  for (int dy = 0; dy < KS; ++dy)
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) cand[dy][dx] = 0xffffffffu;
  constexpr int kFirstPlane = 7;
#endif
#pragma unroll 1
  for (int pl = kFirstPlane; pl >= 0; --pl) {
    const uint32_t *wp = w_row + pl * S::PLANE_STRIDE;
    Csa c;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.a[k] = k < 7 ? mm[k] : 0u, c.b[k] = 0u, c.n[k] = k < 7 ? 1 : 0;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
      if (dy < KEEP) {
        ld_row<KS, PAR>(wp + row_off(dy), kept[dy]);
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) csa_add<0>(c, cand[dy][dx] & kept[dy][dx + PAR]);
      } else {
        uint32_t w[NWORD];
        ld_row<KS, PAR>(wp + row_off(dy), w);
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) csa_add<0>(c, cand[dy][dx] & w[dx + PAR]);
      }
    }
    uint32_t s[8];
    csa_finish(c, s);
    bits_out[pl * S::THREADS] = s[7];
    if (pl == 0) break;  // the candidates are not needed any more
    const uint32_t is0 = ~s[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) mm[k] = bitop3<0xca>(is0, s[k], mm[k]);  // is0 ? s : mm
    // second pass: the candidate update.  (Loading each row's words one visit ahead, which costs the last
    // spare registers, changed nothing: with three waves per SIMD the loop is bound by instruction issue,
    // not by the LDS round trip -- a build without any LDS read in this loop takes the same time.)
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
      if (dy < KEEP) {
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) cand[dy][dx] = bitop3<0x48>(kept[dy][dx + PAR], cand[dy][dx], is0);
      } else {
        uint32_t w[NWORD];
        ld_row<KS, PAR>(wp + row_off(dy), w);
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) cand[dy][dx] = bitop3<0x48>(w[dx + PAR], cand[dy][dx], is0);  // cand & (w ^ is0)
      }
    }
  }
}


#if D2PC_EXPERIMENTS
// ---- EXPERIMENT (round 4's verdict, item 3c; experiment build only): the candidate window split over a LANE PAIR -------
// The select above keeps k*k candidate words per thread (121 at 11 x 11): 159 VGPRs, three waves per SIMD -- the worst
// occupancy for two-cycle instructions (profiles/r02_valu_rates.txt: 3.03 cycles per instruction at 3 waves, 2.75 at 2, 2.68
// at 4) and too few waves to cover the waits (4.0 SIMD-cycles per instruction measured, profiles/r05_callback_counters.json).
// Here two adjacent lanes (h = lane & 1) serve ONE (t, row): lane h owns the window rows dy = r + (KS / 2) h, r = 0 .. KS / 2
// -- the middle row is shared, by columns: h = 0 owns its columns 0 .. KS / 2, h = 1 the rest; the words a lane does not own
// start as "no candidate" and never count.  Each lane sums its own candidates with the same carry-save tree, the two 8-digit
// sums are exchanged (8 DPP quad_perm moves) and added bit-sliced (15 v_bitop3), so both lanes know the median's bit and the
// new rank and update their own half.  (KS / 2 + 1) * KS candidate words per lane (66 at 11 x 11) -> four waves per SIMD;
// ~+20 % instructions per pixel (the shared row's unowned words, the exchange, the rank seed's mask).
template <int KS, int PAR>
__device__ __forceinline__ void select2(const uint32_t *__restrict__ w_row, uint32_t *__restrict__ bits_out, const uint32_t h) {
  using S = MedianBsShape<KS>;
  constexpr int NWORD = 2 * S::NREAD, HR = KS / 2 + 1;  // local rows per lane
  const uint32_t own0 = h ? 0u : 0xffffffffu, own1 = ~own0;  // all ones in the lanes of half 0 / half 1
  uint32_t cand[HR][KS];
#pragma unroll
  for (int r = 0; r < HR; ++r)
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      // half 0: local row HR-1 is the shared row dy = KS/2, it owns columns 0 .. KS/2; half 1: local row 0 is the shared row, columns KS/2+1 ..
      uint32_t v = 0xffffffffu;
      if (r == HR - 1 && dx > KS / 2) v = own1;       // (half 0 does not own these; for half 1 this is its last row dy = KS-1: owned)
      if (r == 0 && dx <= KS / 2) v = own0;           // (half 1 does not own these; for half 0 this is dy = 0: owned)
      cand[r][dx] = v;
    }
  uint32_t mm[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) mm[k] = ((127 - S::D0) >> k) & 1 ? 0xffffffffu : 0u;
#pragma unroll 1
  for (int pl = 7; pl >= 0; --pl) {
    const uint32_t *wp = w_row + pl * S::PLANE_STRIDE;
    Csa c;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.a[k] = k < 7 ? (mm[k] & own0) : 0u, c.b[k] = 0u, c.n[k] = k < 7 ? 1 : 0;  // the rank seed counts once: in half 0
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      uint32_t w[NWORD];
      ld_row<KS>(wp + r * S::ROW_STRIDE, w);
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) csa_add<0>(c, cand[r][dx] & w[dx + PAR]);
    }
    uint32_t mine[8], s[8];
    csa_finish(c, mine);
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // own + partner's sum, digit by digit (both lanes of the pair form the same total)
      const uint32_t other = uint32_t(__builtin_amdgcn_update_dpp(0, int(mine[k]), 0xB1, 0xf, 0xf, false));  // quad_perm:[1,0,3,2]
      if (k == 0) {
        s[0] = mine[0] ^ other, carry = mine[0] & other;
      } else {
        s[k] = bitop3<0x96>(mine[k], other, carry);
        if (k < 7) carry = bitop3<0xe8>(mine[k], other, carry);
      }
    }
    if (h == 0) bits_out[pl * 256] = s[7];
    if (pl == 0) break;
    const uint32_t is0 = ~s[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) mm[k] = bitop3<0xca>(is0, s[k], mm[k]);
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      uint32_t w[NWORD];
      ld_row<KS>(wp + r * S::ROW_STRIDE, w);
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) cand[r][dx] = bitop3<0x48>(w[dx + PAR], cand[r][dx], is0);
    }
  }
}
#endif  // D2PC_EXPERIMENTS

}  // namespace bs

// Stage 1a of one 256 x 32 tile whose first output pixel is (x0, y0) of frame `fsrc`: the tile's input rows (+ halo),
// replicated at the image edges, requested as 16-byte runs into registers -- all of a thread's loads issued together
// (a loop with the edge test inside serialises a dozen global round trips per block: a third of the kernel's time).
// Separate from the rest so that a software-pipelined caller can have the NEXT tile's rows in flight while it
// finishes the current one (k_callback_bs_compact_pipe).
template <int KS>
struct MedianBsRows {
  static constexpr int RUNS = MedianBsShape<KS>::RAW_STRIDE / 16;
  static constexpr int TOTAL = MedianBsShape<KS>::IN_ROWS * RUNS;
  static constexpr int PER_THREAD = (TOTAL + MedianBsShape<KS>::THREADS - 1) / MedianBsShape<KS>::THREADS;
  uint4 v[PER_THREAD];
};

template <int KS>
__device__ __forceinline__ void median_bs_load(MedianBsRows<KS> &rows, const uint8_t *__restrict__ fsrc, const MedianArgs &a,
                                               const int x0, const int y0, const uint32_t tid) {
  using S = MedianBsShape<KS>;
  using RW = MedianBsRows<KS>;
  constexpr int R = S::R;
  const bool interior = x0 - R >= 0 && x0 - R + S::RAW_STRIDE <= int(a.width);  // block-uniform
#pragma unroll
  for (int k = 0; k < RW::PER_THREAD; ++k) {
    const uint32_t c = tid + uint32_t(k * S::THREADS);
    const uint32_t r = c / uint32_t(RW::RUNS), i16 = 16u * (c - r * uint32_t(RW::RUNS));
    int iy = y0 - R + int(r);
    iy = iy < 0 ? 0 : iy >= int(a.height) ? int(a.height) - 1 : iy;  // (also keeps the rows of c >= TOTAL in bounds)
    const uint8_t *row = fsrc + uint64_t(iy) * a.src_row_stride;
    const int x = x0 - R + int(i16);
    if (interior) {
      __builtin_memcpy(&rows.v[k], row + x, 16);
    } else {
      uint32_t d[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d[q] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ix = x + 4 * q + e;
          ix = ix < 0 ? 0 : ix >= int(a.width) ? int(a.width) - 1 : ix;
          d[q] |= uint32_t(row[ix]) << (8 * e);
        }
      }
      rows.v[k] = make_uint4(d[0], d[1], d[2], d[3]);
    }
  }
}

template <int KS>
__device__ __forceinline__ void median_bs_tile_from(const MedianBsRows<KS> &rows, uint32_t (&s_w)[MedianBsShape<KS>::W_WORDS],
                                                    uint32_t (&s_raw)[MedianBsShape<KS>::RAW_WORDS], const uint32_t tid);

// Stages 1-4 of one tile: afterwards (the function ends with the block barrier) the tile's filtered bytes lie in s_w,
// read as bytes, OUT_STRIDE per row, and s_raw is free.  Called by all THREADS threads of the block.
template <int KS>
__device__ __forceinline__ void median_bs_tile(const uint8_t *__restrict__ fsrc, const MedianArgs &a, const int x0, const int y0,
                                               uint32_t (&s_w)[MedianBsShape<KS>::W_WORDS],
                                               uint32_t (&s_raw)[MedianBsShape<KS>::RAW_WORDS], const uint32_t tid) {
  MedianBsRows<KS> rows;
  median_bs_load<KS>(rows, fsrc, a, x0, y0, tid);
  median_bs_tile_from<KS>(rows, s_w, s_raw, tid);
}

// Stages 1b-4: the requested rows as bytes into LDS, then plane words, the select, and the bytes back.
template <int KS>
__device__ __forceinline__ void median_bs_tile_from(const MedianBsRows<KS> &rows, uint32_t (&s_w)[MedianBsShape<KS>::W_WORDS],
                                                    uint32_t (&s_raw)[MedianBsShape<KS>::RAW_WORDS], const uint32_t tid) {
  using S = MedianBsShape<KS>;
  using RW = MedianBsRows<KS>;
  D2PC_BS_STAMP(d0);
  D2PC_BS_SETPRIO(3);
#pragma unroll
  for (int k = 0; k < RW::PER_THREAD; ++k) {
    const uint32_t c = tid + uint32_t(k * S::THREADS);
    if (c < uint32_t(RW::TOTAL)) reinterpret_cast<uint4 *>(s_raw)[c] = rows.v[k];
  }
  __syncthreads();
  D2PC_BS_STAMP(d1);

  // ---- 2. plane words: item (row r, u) gathers the pixels 8 j + u of the row --------------------------
  {
    const uint8_t *raw = reinterpret_cast<const uint8_t *>(s_raw);
    for (uint32_t item = tid; item < uint32_t(S::IN_ROWS * S::S); item += uint32_t(S::THREADS)) {
      const uint32_t r = item >> 3, u = item & 7u;
      const uint8_t *rp = raw + r * uint32_t(S::RAW_STRIDE) + u;
      uint32_t px[8], plane[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        px[k] = uint32_t(rp[32 * k]) | (uint32_t(rp[32 * k + 8]) << 8) | (uint32_t(rp[32 * k + 16]) << 16) |
                (uint32_t(rp[32 * k + 24]) << 24);
      const uint32_t e1 = rp[256], e2 = rp[264];  // the next tile's first pixels at this u: bit 32 and 33 of the row
      bs::to_planes(px, plane);
      uint32_t *wr = s_w + r * uint32_t(S::ROW_STRIDE) + u;
#pragma unroll
      for (int pl = 0; pl < 8; ++pl) {
        const uint32_t hi = ((e1 >> pl) & 1u) | (((e2 >> pl) & 1u) << 1);
        wr[pl * S::PLANE_STRIDE] = plane[pl];
        if (u + 8u < uint32_t(S::NW)) wr[pl * S::PLANE_STRIDE + 8] = __builtin_amdgcn_alignbit(hi, plane[pl], 1);
        if (S::NW > 16 && u + 16u < uint32_t(S::NW)) wr[pl * S::PLANE_STRIDE + 16] = __builtin_amdgcn_alignbit(hi, plane[pl], 2);
      }
    }
  }
  __syncthreads();
  D2PC_BS_STAMP(d2);

  // ---- 3. the select: wave = one parity of t, 16 rows; lane = (t >> 1) + 4 * row -----------------------
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  // lane = place in the row group + GR x (t >> 1) + 4 GR x group: GR = 1 is rounds 2-5's mapping (lane = (t >> 1) + 4 x row)
  constexpr uint32_t GR = uint32_t(D2PC_BS_GROUP_ROWS);
  static_assert(GR == 1 || GR == 2 || GR == 4 || GR == 8 || GR == 16, "a wave holds 16 rows of one column parity");
  const uint32_t par = wave & 1u, g = lane % GR, t = 2u * ((lane / GR) & 3u) + par, row0 = 16u * (wave >> 1) + GR * (lane / (4u * GR));
  const uint32_t row = row0 + g;
  {
    D2PC_BS_SETPRIO(0);
    const uint32_t *w_row = s_w + row0 * uint32_t(S::ROW_STRIDE) + (t - par);
    uint32_t *bits_out = s_raw + tid;  // the staged bytes are no longer needed
    if (par) bs::select<KS, 1, int(GR)>(w_row, bits_out, g);
    else bs::select<KS, 0, int(GR)>(w_row, bits_out, g);
    D2PC_BS_SETPRIO(3);
  }
  __syncthreads();  // every wave has finished reading W
  D2PC_BS_STAMP(d3);

  // ---- 4. bit planes -> bytes, staged in W's space, stored in 16-byte runs ---------------------------
  {
    uint32_t plane[8], px[8];
#pragma unroll
    for (int pl = 0; pl < 8; ++pl) plane[pl] = s_raw[pl * S::THREADS + tid];
    bs::to_pixels(plane, px);
    uint8_t *ob = reinterpret_cast<uint8_t *>(s_w) + row * uint32_t(S::OUT_STRIDE) + t;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) ob[8 * (4 * k + q)] = uint8_t(px[k] >> (8 * q));
  }
  __syncthreads();
  D2PC_BS_STAMP(d4);
  D2PC_BS_ADD(1, d1 - d0);
  D2PC_BS_ADD(2, d2 - d1);
  D2PC_BS_ADD(3, d3 - d2);
  D2PC_BS_ADD(4, d4 - d3);
}

#if D2PC_EXPERIMENTS
// The tile body around select2: the same four stages with 512 threads per 256 x 32 tile (two lanes per (t, row)).
// Experiment build only (k_median_bs2_u8, median_algo 3).
template <int KS>
__device__ __forceinline__ void median_bs2_tile(const uint8_t *__restrict__ fsrc, const MedianArgs &a, const int x0, const int y0,
                                                uint32_t (&s_w)[MedianBsShape<KS>::W_WORDS],
                                                uint32_t (&s_raw)[MedianBsShape<KS>::RAW_WORDS], const uint32_t tid) {
  using S = MedianBsShape<KS>;
  constexpr int NT = 512, R = S::R, RUNS = S::RAW_STRIDE / 16, TOTAL = S::IN_ROWS * RUNS, PER = (TOTAL + NT - 1) / NT;
  // ---- 1. input rows (+ halo), replicated at the image edges, as 16-byte runs: all loads first
  uint4 v[PER];
  const bool interior = x0 - R >= 0 && x0 - R + S::RAW_STRIDE <= int(a.width);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const uint32_t c = tid + uint32_t(k * NT);
    const uint32_t r = c / uint32_t(RUNS), i16 = 16u * (c - r * uint32_t(RUNS));
    int iy = y0 - R + int(r);
    iy = iy < 0 ? 0 : iy >= int(a.height) ? int(a.height) - 1 : iy;
    const uint8_t *row = fsrc + uint64_t(iy) * a.src_row_stride;
    const int x = x0 - R + int(i16);
    if (interior) {
      __builtin_memcpy(&v[k], row + x, 16);
    } else {
      uint32_t d[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d[q] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ix = x + 4 * q + e;
          ix = ix < 0 ? 0 : ix >= int(a.width) ? int(a.width) - 1 : ix;
          d[q] |= uint32_t(row[ix]) << (8 * e);
        }
      }
      v[k] = make_uint4(d[0], d[1], d[2], d[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const uint32_t c = tid + uint32_t(k * NT);
    if (c < uint32_t(TOTAL)) reinterpret_cast<uint4 *>(s_raw)[c] = v[k];
  }
  __syncthreads();
  // ---- 2. plane words (as in median_bs_tile_from)
  {
    const uint8_t *raw = reinterpret_cast<const uint8_t *>(s_raw);
    for (uint32_t item = tid; item < uint32_t(S::IN_ROWS * S::S); item += uint32_t(NT)) {
      const uint32_t r = item >> 3, u = item & 7u;
      const uint8_t *rp = raw + r * uint32_t(S::RAW_STRIDE) + u;
      uint32_t px[8], plane[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        px[k] = uint32_t(rp[32 * k]) | (uint32_t(rp[32 * k + 8]) << 8) | (uint32_t(rp[32 * k + 16]) << 16) |
                (uint32_t(rp[32 * k + 24]) << 24);
      const uint32_t e1 = rp[256], e2 = rp[264];
      bs::to_planes(px, plane);
      uint32_t *wr = s_w + r * uint32_t(S::ROW_STRIDE) + u;
#pragma unroll
      for (int pl = 0; pl < 8; ++pl) {
        const uint32_t hi = ((e1 >> pl) & 1u) | (((e2 >> pl) & 1u) << 1);
        wr[pl * S::PLANE_STRIDE] = plane[pl];
        if (u + 8u < uint32_t(S::NW)) wr[pl * S::PLANE_STRIDE + 8] = __builtin_amdgcn_alignbit(hi, plane[pl], 1);
        if (S::NW > 16 && u + 16u < uint32_t(S::NW)) wr[pl * S::PLANE_STRIDE + 16] = __builtin_amdgcn_alignbit(hi, plane[pl], 2);
      }
    }
  }
  __syncthreads();
  // ---- 3. the select: wave = one parity of t, 8 rows; lane = h + 2 * ((t >> 1) + 4 * row)
  {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t par = wave & 1u, h = lane & 1u, t = 2u * ((lane >> 1) & 3u) + par, row = 8u * (wave >> 1) + (lane >> 3);
    const uint32_t *w_row = s_w + (row + uint32_t(KS / 2) * h) * uint32_t(S::ROW_STRIDE) + (t - par);
    uint32_t *bits_out = s_raw + (row * 8u + t);  // logical thread (t, row); 8 plane words 256 apart
    if (par) bs::select2<KS, 1>(w_row, bits_out, h);
    else bs::select2<KS, 0>(w_row, bits_out, h);
  }
  __syncthreads();
  // ---- 4. bit planes -> bytes, staged in W's space (the first 256 threads: one per (t, row))
  if (tid < 256u) {
    const uint32_t row = tid >> 3, t = tid & 7u;
    uint32_t plane[8], px[8];
#pragma unroll
    for (int pl = 0; pl < 8; ++pl) plane[pl] = s_raw[pl * 256 + tid];
    bs::to_pixels(plane, px);
    uint8_t *ob = reinterpret_cast<uint8_t *>(s_w) + row * uint32_t(S::OUT_STRIDE) + t;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) ob[8 * (4 * k + q)] = uint8_t(px[k] >> (8 * q));
  }
  __syncthreads();
}
#endif  // D2PC_EXPERIMENTS

}  // namespace d2pc
