// d2pc_fusion.hip -- the depth-map fusion inner loop on gfx950
// (SURVEY.md section 8(f) #4): per-pixel fusion rule + combined confidence
// (reference src/depth_map_fusion.cpp:113-123, rules :162-235), 3x3 median of
// the fused image (:124) and the border crop (:130), in ONE pass over HBM:
// 6 bytes read and ~2 written per pixel, no intermediate image.
//
// Register-rolling stencil, no LDS, no barriers, no divergent branches.  The
// unit of work is a WAVE: a strip of 248 columns by R rows.  Lane l holds 4
// adjacent pixels (one dword per plane per row), lanes 0 and 63 are the
// left/right halo columns, so a row of a strip is one 256-byte coalesced load
// per plane.  All arithmetic is done on PAIRS: a dword is split once into its
// even and odd bytes (two 16-bit fields each), after which v_pk_*_u16
// instructions handle two pixels per lane per instruction:
//  * the rule: every comparison a < b is the sign of the 16-bit difference
//    a - b; the signs of one condition chain are AND-ed and expanded to a field
//    mask with one arithmetic shift; the result is picked with bit-selects;
//  * the 3x3 median, separable in the usual way: sort every vertical triple
//    once (shared by three output pixels), fetch the neighbouring lanes' edge
//    columns with DPP wave shifts, take med3(max3(lows), med3(mids),
//    min3(highs)).
// Borders replicate by clamping the LOAD coordinates (the rule is per pixel,
// so that equals replicating the fused image, which is what cv::medianBlur's
// BORDER_REPLICATE sees).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d2pc_launch.hpp"

namespace d2pc {

namespace {

constexpr int kStripCols = 248;  // output columns per wave: 62 lanes x 4 pixels
#ifndef D2PC_FUSE_BLOCK
#define D2PC_FUSE_BLOCK 256
#endif
constexpr int kFuseBlock = D2PC_FUSE_BLOCK;  // waves of a block take horizontally adjacent strips

// ---- two 16-bit fields per dword -------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk(uint32_t c) { return c | (c << 16); }  // both fields = c
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {         // per-field a - b (wraps)
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {         // per-field a + b (wraps)
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_shl2(uint32_t a) {                    // per-field a << 2
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) << 2);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// field mask (0xffff / 0) from the sign bit of each field.  Inline asm: written
// as a vector shift, LLVM turns mask-and-pick into per-field compare + select,
// which has no packed form and costs five instructions instead of one.
__device__ __forceinline__ uint32_t pk_sign_mask(uint32_t a) {
  uint32_t r;
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t pk_half(uint32_t a) {  // per-field a >> 1
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) >> 1);
}
__device__ __forceinline__ uint32_t pick(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ uint32_t pk_med3(uint32_t a, uint32_t b, uint32_t c) {
  return pk_max(pk_min(a, b), pk_min(pk_max(a, b), c));
}
// lane i receives lane i-1's / lane i+1's value (the wave's end lanes keep their own)
__device__ __forceinline__ uint32_t from_left_lane(uint32_t v) {
  return uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t from_right_lane(uint32_t v) {
  return uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// Where a lane finds its pixels x .. x+3 (columns outside [0, w) replicate the
// edge pixel): ONE dword at the in-range column `col`, then two byte shuffles
// (v_perm_b32) that fix up the edge lanes and split the dword into its even
// and odd pixels in one go.  Needs w >= 4; narrower images gather bytes.
struct LaneCols {
  uint32_t col;          // dword load position (w >= 4)
  uint32_t selE, selO;   // v_perm_b32 selectors -> pixels (0,2) / (1,3) as 16-bit fields
  uint32_t xs[4];        // clamped columns (w < 4 only)
};
__device__ __forceinline__ LaneCols lane_cols(int x, int w) {
  LaneCols c;
  c.col = uint32_t(min(max(x, 0), max(w - 4, 0)));
  uint32_t sel[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c.xs[k] = uint32_t(min(max(x + k, 0), w - 1));
    sel[k] = (c.xs[k] - c.col) & 3u;
  }
  constexpr uint32_t kZero = 0x0cu;  // v_perm_b32: constant 0x00
  c.selE = sel[0] | (kZero << 8) | (sel[2] << 16) | (kZero << 24);
  c.selO = sel[1] | (kZero << 8) | (sel[3] << 16) | (kZero << 24);
  return c;
}
// The selected rule (reference src/depth_map_fusion.cpp:162-235) on a pair of
// pixels; every operand field holds an 8-bit value.  lt(a, b) below is "the
// sign bit of a - b", valid while |a - b| < 2^15.
// GRAD_FILTER's float test 0.8 < float(d1)/float(d2) < 1.25 (cpp:224,230) is
// 5*d1 >= 4*d2 && 4*d1 < 5*d2: the quotient is compared as a float against
// DOUBLE literals, float(0.8) > 0.8 so the exact ratio 4/5 passes, 5/4 is exact
// and fails, no other 8-bit ratio is within a float ulp of either bound, and
// d2 == 0 (inf or NaN) fails both ways.
template <int RULE>
__device__ __forceinline__ uint32_t fuse_pair(uint32_t d1, uint32_t d2, uint32_t s1, uint32_t s2) {
  const uint32_t avg = pk_half(d1 + d2);
  switch (RULE) {
    case FUSE_WEIGHTED_AVERAGE: {  // int weights: 1 for score 0, else 0; 0/0 (undefined there) -> 0
      const uint32_t w1 = pk_sign_mask(pk_sub(s1, pk(1))), w2 = pk_sign_mask(pk_sub(s2, pk(1)));
      return pick(w1, pick(w2, avg, d1), w2 & d2);
    }
    case FUSE_MAX_DIST: return pk_min(d1, d2);
    case FUSE_MAX_DIST_UNLESS_BLACK: {
      const uint32_t black = pk_sign_mask(pk_sub(d1, pk(1)) | pk_sub(d2, pk(1)));
      return pick(black, pk_max(d1, d2), pk_min(d1, d2));
    }
    case FUSE_BETTER_SCORE: return pick(pk_sign_mask(pk_sub(s1, s2)), d1, d2);
    case FUSE_ONLY_GOOD_1: return pk_sign_mask(pk_sub(s2, pk(50))) & d2;
    case FUSE_ONLY_GOOD_AVG: return pk_sign_mask(pk_sub(s1, pk(100)) & pk_sub(s2, pk(100))) & avg;
    case FUSE_OVERLAP: {
      const uint32_t a = pk_sign_mask(pk_sub(s1, s2) & pk_sub(s1, pk(20)));
      const uint32_t b = pk_sign_mask(pk_sub(s2, s1) & pk_sub(s2, pk(20)));
      return pick(a, pk(150), b & pk(255));
    }
    case FUSE_BLACK_TO_WHITE: return pk(255) - s1;
    default: {  // FUSE_GRAD_FILTER
      const uint32_t a = pk_sign_mask(pk_sub(s1, s2) & pk_sub(s1, pk(100)) & pk_sub(d1, pk(230)));
      const uint32_t b = pk_sign_mask(pk_sub(s2, s1) & pk_sub(s2, pk(100)) & pk_sub(d2, pk(230)));
      // 5*d1 >= 4*d2 && 4*d1 < 5*d2  <=>  t + d1 >= 0 && t - d2 < 0  with  t = 4*(d1 - d2)
      const uint32_t t = pk_shl2(pk_sub(d1, d2));
      const uint32_t c = pk_sign_mask(~pk_add(t, d1) & pk_sub(t, d2) & pk_sub(s1, pk(125)) & pk_sub(s2, pk(125)));
      return pick(a, d1, pick(b, d2, c & avg));
    }
  }
}

// Stores the bytes of v whose column lies in [lo, hi) at dst[x - lo].
__device__ __forceinline__ void store_px4(uint8_t *__restrict__ dst_row, int x, int lo, int hi, uint32_t v) {
  if (x >= lo && x + 3 < hi) {
    __builtin_memcpy(dst_row + (x - lo), &v, 4);
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (x + k >= lo && x + k < hi) dst_row[x + k - lo] = uint8_t(v >> (8 * k));
}

}  // namespace

// Raw dwords of one image row for one lane.
struct RowRaw {
  uint32_t d1, d2, s1, s2, g1, g2;
};

// One kernel per rule (a run-time rule switch in the inner loop would be paid
// per pixel pair).  NARROW = images less than 4 pixels wide.
//
// A wave walks down its strip two rows per step, holding a rolling window of
// fused rows: with S0,S1 = rows k,k+1 already fused and N2,N3 = rows k+2,k+3
// new, the sorted pair (S1,N2) serves BOTH outputs of the step (rows k+1 and
// k+2 take S0 resp. N3 as the third value), and the raw loads of the next step
// are issued before this step's arithmetic.  Only the first row of a strip is
// fused twice (by this wave and the one above), so strips can be tall without
// costing registers.
template <int RULE, bool NARROW>
__global__ __launch_bounds__(kFuseBlock) void k_fuse_median3(const FuseArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  // A block is four horizontally adjacent strips of one chunk (1 KB of every row: measured 12-15 %
  // faster than four vertically adjacent chunks of one strip, despite the latter's shared halo rows).
  // The divisions run on the VALU; readfirstlane puts the (uniform) results back into SGPRs so that
  // every row pointer below is a scalar base.
  const uint32_t item = blockIdx.x * (kFuseBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (item >= a.items) return;  // wave-uniform
  const uint32_t rest = __builtin_amdgcn_readfirstlane(item / a.strips_x);
  const uint32_t strip = item - rest * a.strips_x;
  const uint32_t frame = __builtin_amdgcn_readfirstlane(rest / a.chunks_y);
  const uint32_t chunk = rest - frame * a.chunks_y;
  const int w = int(a.width), h = int(a.height);
  const int x0 = int(strip) * kStripCols, y0 = int(chunk * a.rows_per_wave);
  const int rows = min(int(a.rows_per_wave), h - y0);  // >= 1
  const int x = x0 - 4 + int(lane) * 4;  // lane 0 = left halo, lane 63 = right halo
  const LaneCols lc = lane_cols(x, w);

  const uint8_t *pl[6];
#pragma unroll
  for (int p = 0; p < 6; ++p) pl[p] = a.in[p] + uint64_t(frame) * a.in_frame_stride[p];
  const bool owner = lane >= 1u && lane <= 62u && x < w;  // halo lanes store nothing
  uint8_t *cf = a.combined ? a.combined + uint64_t(frame) * a.combined_frame_stride : nullptr;
  uint8_t *ff = a.fused + uint64_t(frame) * a.fused_frame_stride;
  const int cx0 = int(a.crop_left), cx1 = cx0 + int(a.out_width);
  const int cy0 = int(a.crop_top), cy1 = cy0 + int(a.out_height);

  // window row k <-> image row y0 - 1 + k, replicated at the image border
  auto fetch = [&](int k) {
    const int y = min(max(y0 - 1 + k, 0), h - 1);
    // Opaque copy of the (loop-invariant) lane column: otherwise LLVM hoists plane + column into a
    // 64-bit VGPR pair and pays a 64-bit vector add per load; this way every load is
    // "scalar row base + 32-bit lane offset" and costs no VALU work.
    uint32_t col = lc.col;
    asm volatile("" : "+v"(col));
    RowRaw r;
    __builtin_memcpy(&r.d1, (pl[0] + uint32_t(y) * a.in_pitch[0]) + col, 4);  // plane extents < 4 GiB (host-checked)
    __builtin_memcpy(&r.d2, (pl[1] + uint32_t(y) * a.in_pitch[1]) + col, 4);
    __builtin_memcpy(&r.s1, (pl[2] + uint32_t(y) * a.in_pitch[2]) + col, 4);
    __builtin_memcpy(&r.s2, (pl[3] + uint32_t(y) * a.in_pitch[3]) + col, 4);
    r.g1 = r.g2 = 0;
    if (cf) {  // launch-uniform
      __builtin_memcpy(&r.g1, (pl[4] + uint32_t(y) * a.in_pitch[4]) + col, 4);
      __builtin_memcpy(&r.g2, (pl[5] + uint32_t(y) * a.in_pitch[5]) + col, 4);
    }
    return r;
  };
  auto fetch_narrow = [&](int k) {  // w < 4: gather the clamped columns
    const int y = min(max(y0 - 1 + k, 0), h - 1);
    auto gather = [&](int p) {
      const uint8_t *row = pl[p] + uint32_t(y) * a.in_pitch[p];
      return uint32_t(row[lc.xs[0]]) | (uint32_t(row[lc.xs[1]]) << 8) | (uint32_t(row[lc.xs[2]]) << 16) |
             (uint32_t(row[lc.xs[3]]) << 24);
    };
    RowRaw r;
    r.d1 = gather(0), r.d2 = gather(1), r.s1 = gather(2), r.s2 = gather(3);
    r.g1 = r.g2 = 0;
    if (cf) r.g1 = gather(4), r.g2 = gather(5);
    return r;
  };
  auto get = [&](int k) { return NARROW ? fetch_narrow(k) : fetch(k); };
  const uint32_t selE = NARROW ? 0x0c020c00u : lc.selE, selO = NARROW ? 0x0c030c01u : lc.selO;

  struct Pair2 {
    uint32_t e, o;
  };
  // fused row (and, for the strip's own rows, the combined confidence of cpp:118-121)
  auto fuse_row = [&](const RowRaw &r, int k) {
    Pair2 f;
    f.e = fuse_pair<RULE>(__builtin_amdgcn_perm(r.d1, r.d1, selE), __builtin_amdgcn_perm(r.d2, r.d2, selE),
                          __builtin_amdgcn_perm(r.s1, r.s1, selE), __builtin_amdgcn_perm(r.s2, r.s2, selE));
    f.o = fuse_pair<RULE>(__builtin_amdgcn_perm(r.d1, r.d1, selO), __builtin_amdgcn_perm(r.d2, r.d2, selO),
                          __builtin_amdgcn_perm(r.s1, r.s1, selO), __builtin_amdgcn_perm(r.s2, r.s2, selO));
    if (cf && k >= 1 && k <= rows) {  // wave-uniform
      const uint32_t m = pk_min(__builtin_amdgcn_perm(r.g1, r.g1, selE), __builtin_amdgcn_perm(r.g2, r.g2, selE)) |
                         (pk_min(__builtin_amdgcn_perm(r.g1, r.g1, selO), __builtin_amdgcn_perm(r.g2, r.g2, selO)) << 8);
      const int y = y0 - 1 + k;
      if (owner) store_px4(cf + uint32_t(y) * a.combined_pitch, x, 0, w, m);
    }
    return f;
  };
  // median of output row j (image row y0 + j) from the sorted pair (mn, mx) of its two other rows and c
  auto emit = [&](int j, uint32_t mnE, uint32_t mxE, uint32_t mnO, uint32_t mxO, const Pair2 &c) {
    // sorted vertical triples of the lane's columns: E = columns (0,2), O = (1,3)
    const uint32_t loE = pk_min(mnE, c.e), hiE = pk_max(mxE, c.e), meE = pk_max(mnE, pk_min(mxE, c.e));
    const uint32_t loO = pk_min(mnO, c.o), hiO = pk_max(mxO, c.o), meO = pk_max(mnO, pk_min(mxO, c.o));
    // columns (-1,1) and (2,4): column -1 is the left lane's column 3, column 4 the right lane's column 0
    const uint32_t loL = __builtin_amdgcn_alignbit(loO, from_left_lane(loO), 16);
    const uint32_t meL = __builtin_amdgcn_alignbit(meO, from_left_lane(meO), 16);
    const uint32_t hiL = __builtin_amdgcn_alignbit(hiO, from_left_lane(hiO), 16);
    const uint32_t loR = __builtin_amdgcn_alignbit(from_right_lane(loE), loE, 16);
    const uint32_t meR = __builtin_amdgcn_alignbit(from_right_lane(meE), meE, 16);
    const uint32_t hiR = __builtin_amdgcn_alignbit(from_right_lane(hiE), hiE, 16);
    // outputs (0,2) see columns L,E,O; outputs (1,3) see columns E,O,R: the E,O part is shared
    const uint32_t loEO = pk_max(loE, loO), hiEO = pk_min(hiE, hiO);
    const uint32_t meMn = pk_min(meE, meO), meMx = pk_max(meE, meO);
    const uint32_t outE = pk_med3(pk_max(loL, loEO), pk_max(meMn, pk_min(meMx, meL)), pk_min(hiL, hiEO));
    const uint32_t outO = pk_med3(pk_max(loR, loEO), pk_max(meMn, pk_min(meMx, meR)), pk_min(hiR, hiEO));
    const int y = y0 + j;
    if (owner && j < rows && y >= cy0 && y < cy1)
      store_px4(ff + uint32_t(y - cy0) * a.fused_pitch, x, cx0, cx1, outE | (outO << 8));
  };

  // two steps per trip, the raw rows alternating between two register sets (no window copies)
  auto step = [&](int k, const RowRaw &c2, const RowRaw &c3, Pair2 &s0, Pair2 &s1) {
    const Pair2 f2 = fuse_row(c2, k + 2), f3 = fuse_row(c3, k + 3);
    const uint32_t mnE = pk_min(s1.e, f2.e), mxE = pk_max(s1.e, f2.e);
    const uint32_t mnO = pk_min(s1.o, f2.o), mxO = pk_max(s1.o, f2.o);
    emit(k, mnE, mxE, mnO, mxO, s0);
    emit(k + 1, mnE, mxE, mnO, mxO, f3);
    s0 = f2;
    s1 = f3;
  };
  Pair2 s0, s1;
  {
    const RowRaw r0 = get(0), r1 = get(1);
    s0 = fuse_row(r0, 0);
    s1 = fuse_row(r1, 1);
  }
  RowRaw a2 = get(2), a3 = get(3), b2 = a2, b3 = a3;
  for (int k = 0; k < rows; k += 4) {  // outputs k .. k+3 from window rows k .. k+5
    if (k + 2 < rows) {  // the next step's rows are in flight during this step's arithmetic
      b2 = get(k + 4);
      b3 = get(k + 5);
    }
    step(k, a2, a3, s0, s1);
    if (k + 2 >= rows) break;
    if (k + 4 < rows) {
      a2 = get(k + 6);
      a3 = get(k + 7);
    }
    step(k + 2, b2, b3, s0, s1);
  }
}

namespace {
template <int RULE>
void launch_rule(const FuseArgs &a, uint32_t grid, hipStream_t stream) {
  if (a.width < 4)
    hipLaunchKernelGGL((k_fuse_median3<RULE, true>), dim3(grid), dim3(kFuseBlock), 0, stream, a);
  else
    hipLaunchKernelGGL((k_fuse_median3<RULE, false>), dim3(grid), dim3(kFuseBlock), 0, stream, a);
}
}  // namespace

// rows_hint: 0 = choose; else rows per wave (tuning; even, 2..1024).
hipError_t launch_fuse(FuseArgs a, hipStream_t stream, int rows_hint) {
  // Tall strips amortise the one re-fused halo row; small launches want more, shorter strips so that
  // every SIMD has waves to switch between.
  const uint32_t strips = (a.width + kStripCols - 1) / kStripCols;
  uint32_t rows = (rows_hint >= 2 && rows_hint <= 1024) ? uint32_t(rows_hint) & ~1u : 0u;
  if (!rows) {
    // tools/fusion_bench.py: 16 rows (8 when the combined plane is written too) beat 4 and 32 on full
    // launches; a single small frame wants more, shorter strips
    rows = a.combined ? 8 : 16;
    while (rows > 4 && uint64_t(strips) * ((a.height + rows - 1) / rows) * a.n_frames < 2048u) rows >>= 1;
  }
  a.rows_per_wave = rows;
  a.strips_x = strips;
  a.chunks_y = (a.height + rows - 1) / rows;
  const uint64_t items = uint64_t(a.strips_x) * a.chunks_y * a.n_frames;
  if (items == 0 || items > 0x7fffffffull) return hipErrorInvalidValue;
  a.items = uint32_t(items);
  const uint32_t grid = (a.items + kFuseBlock / 64 - 1) / (kFuseBlock / 64);
  switch (a.rule) {
    case FUSE_WEIGHTED_AVERAGE: launch_rule<FUSE_WEIGHTED_AVERAGE>(a, grid, stream); break;
    case FUSE_MAX_DIST: launch_rule<FUSE_MAX_DIST>(a, grid, stream); break;
    case FUSE_MAX_DIST_UNLESS_BLACK: launch_rule<FUSE_MAX_DIST_UNLESS_BLACK>(a, grid, stream); break;
    case FUSE_BETTER_SCORE: launch_rule<FUSE_BETTER_SCORE>(a, grid, stream); break;
    case FUSE_ONLY_GOOD_1: launch_rule<FUSE_ONLY_GOOD_1>(a, grid, stream); break;
    case FUSE_ONLY_GOOD_AVG: launch_rule<FUSE_ONLY_GOOD_AVG>(a, grid, stream); break;
    case FUSE_OVERLAP: launch_rule<FUSE_OVERLAP>(a, grid, stream); break;
    case FUSE_BLACK_TO_WHITE: launch_rule<FUSE_BLACK_TO_WHITE>(a, grid, stream); break;
    case FUSE_GRAD_FILTER: launch_rule<FUSE_GRAD_FILTER>(a, grid, stream); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// rotateMat (reference src/depth_map_fusion.cpp:268-273): cv::transpose then
// cv::flip(.., 1) = 90 degrees clockwise; dst has `cols` rows of `rows`
// pixels, dst(i, j) = src(rows-1-j, i).
// 128 x 128 byte tiles through LDS (round 3; rounds 1-2 moved 64 x 64 tiles as dwords and gathered single bytes from
// LDS: every wave instruction touched 64-byte pieces of four rows and the gather cost one LDS read per byte --
// 3.5 TB/s).  Loads: a wave instruction covers two source rows of 128 bytes; the rows land in LDS as they are.
// Stores: thread (rg, cd) reads the dword column cd of the 16 source rows 16 rg .. 16 rg + 15, transposes the four
// 4 x 4 byte blocks in registers (two v_perm levels) and stores 16 consecutive bytes to each of the four destination
// rows 4 cd .. 4 cd + 3; a wave instruction covers eight destination rows of 128 bytes.
// ---------------------------------------------------------------------------
constexpr int kRotTile = 128;
constexpr int kRotStride = 33;  // dwords per staged row: the store phase's column reads of 8 row groups x 8 columns
                                // spread over 32 banks (2-way), the load phase's row writes are conflict-free

__device__ __forceinline__ void rot_transpose_4x4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t (&o)[4]) {
  // o[i] = (byte i of b0, of b1, of b2, of b3), lowest address first
  const uint32_t a0 = __builtin_amdgcn_perm(b1, b0, 0x05010400u), a1 = __builtin_amdgcn_perm(b1, b0, 0x07030602u);
  const uint32_t a2 = __builtin_amdgcn_perm(b3, b2, 0x05010400u), a3 = __builtin_amdgcn_perm(b3, b2, 0x07030602u);
  o[0] = __builtin_amdgcn_perm(a2, a0, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(a2, a0, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(a3, a1, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(a3, a1, 0x07060302u);
}

__global__ __launch_bounds__(kBlock) void k_rotate_cw(const RotateArgs a) {
  __shared__ uint32_t tile[kRotTile * kRotStride];
  uint32_t b = blockIdx.x;
  const uint32_t f = b / (a.tiles_x * a.tiles_y);
  b -= f * a.tiles_x * a.tiles_y;
  const uint32_t ty = b / a.tiles_x, tx = b - ty * a.tiles_x;
  const int c0 = int(tx) * kRotTile, r0 = int(ty) * kRotTile;  // source tile origin
  const uint8_t *src = a.src + uint64_t(f) * a.src_frame_stride;
  uint8_t *dst = a.dst + uint64_t(f) * a.dst_frame_stride;
  const int cols = int(a.cols), rows = int(a.rows);
  {  // rows in: thread (lr, lc) takes dword column lc of the rows lr, lr + 8, ...
    const int lc = int(threadIdx.x & 31u), lr = int(threadIdx.x >> 5);
    const int x = c0 + 4 * lc;
    uint32_t v[kRotTile / 8];
#pragma unroll
    for (int k = 0; k < kRotTile / 8; ++k) {
      const int y = r0 + lr + 8 * k;
      v[k] = 0;
      if (y < rows) {
        const uint8_t *row = src + uint64_t(y) * a.src_pitch;
        if (x + 3 < cols) {
          __builtin_memcpy(&v[k], row + x, 4);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (x + i < cols) v[k] |= uint32_t(row[x + i]) << (8 * i);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kRotTile / 8; ++k) tile[(lr + 8 * k) * kRotStride + lc] = v[k];
  }
  __syncthreads();
  // destination tile: rows c0 .. c0+127 (source columns); source row r0 + l lands in column j0 + 127 - l
  const int j0 = rows - 1 - (r0 + kRotTile - 1);
  const int rg = int(threadIdx.x & 7u), cd = int(threadIdx.x >> 3);
  uint32_t w[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = tile[(16 * rg + k) * kRotStride + cd];
  uint32_t o[4][4];  // o[c][d]: dword d of the 16 bytes of destination row 4 cd + c
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t t[4];
    rot_transpose_4x4(w[15 - 4 * d], w[14 - 4 * d], w[13 - 4 * d], w[12 - 4 * d], t);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c][d] = t[c];
  }
  const int j = j0 + kRotTile - 16 - 16 * rg;  // destination column of the thread's first byte (< 0: past the image end)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int i = c0 + 4 * cd + c;  // destination row = source column
    if (i >= cols) continue;
    uint8_t *row = dst + uint64_t(i) * a.dst_pitch;
    if (j >= 0) {
      __builtin_memcpy(row + j, o[c], 16);
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (j + t >= 0) row[j + t] = uint8_t(o[c][t >> 2] >> (8 * (t & 3)));
    }
  }
}

hipError_t launch_rotate_cw(RotateArgs a, hipStream_t stream) {
  a.tiles_x = (a.cols + kRotTile - 1) / kRotTile;
  a.tiles_y = (a.rows + kRotTile - 1) / kRotTile;
  const uint64_t blocks = uint64_t(a.tiles_x) * a.tiles_y * a.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_rotate_cw, dim3(uint32_t(blocks)), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

}  // namespace d2pc
