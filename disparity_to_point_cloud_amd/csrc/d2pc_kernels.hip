// d2pc_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the
// disparity -> point-cloud path.  See d2pc_device.hpp for the reference
// call sites this replaces.
//
// Design (HBM-bound streaming map, ~1 flop/byte; MFMA does not apply):
//  * Work is indexed by OUTPUT point, flat over the frame's ROI, so every
//    wave-level store is one contiguous, 1-KiB, 16-B-per-lane write of final
//    PointCloud2 bytes; a tile is 256*PXT consecutive ROI pixels.
//  * Disparity is read once with coalesced per-lane dword loads (64 lanes =
//    256 contiguous bytes); ROI rows wrap inside a tile: one exact
//    multiply-high division per thread and tile, then (u,v) are stepped.
//  * Q and the geometry are kernel arguments: they sit in SGPRs for the whole
//    kernel (cheaper than LDS: no ds_read, no bank traffic, no barrier).
//  * Arithmetic follows OpenCV's double-precision evaluation: fp64 FMA chain
//    for the four row products, one IEEE fp64 reciprocal, one cast to fp32.
//    When Q has the structure cv::stereoRectify produces (QK_STEREO) the
//    multiplications by its exact zeros and ones are dropped -- bit-identical.
//  * COMPACT mode: wave ballot + mbcnt ranks, LDS scan over the block's
//    (slot, wave) counts, and a two-level counted prefix across tiles
//    (64-bit {arrivals,sum} group accumulators + tagged per-tile granules)
//    so the output order equals the CPU loop's row-major order bit-for-bit.
#include <cstdio>
#include <cstdlib>

#include "d2pc_device.hpp"
#include "d2pc_launch.hpp"
#include "d2pc_median_bs_tile.hpp"

namespace d2pc {

typedef float v4f __attribute__((ext_vector_type(4)));

// Cache-policy knobs.  tools/ab.py builds the library with other values and
// times all builds interleaved in ONE process on ONE set of buffers (timings
// differ by +-6 % between allocations and ~10 % between devices, so nothing
// else ranks variants).  Measured on MI355X, 16 x 4K frames per launch:
//   loads : plain beats nt by 1-2 % (with a border the 256-B / 1-KiB wave
//           pieces are not line-aligned; nt makes L2 drop the shared edge
//           lines and re-fetch them: FETCH_SIZE 1.31x vs 1.07x algorithmic)
//   stores: PARITY: nt beats plain by ~2 % (full aligned 1-KiB pieces, written once, never re-read);
//           COMPACT single pass: plain -- survivors leave as ragged pieces whose end lines are
//           completed by the neighbouring piece, so L2 should keep them to merge (30 % iid holes +
//           indices: plain -5 % on one device, equal on another; never worse);
//           COMPACT two-pass scatter: nt (-3..-7 % against plain, all cases)
#ifndef D2PC_LOAD_NT
#define D2PC_LOAD_NT 0
#endif
#ifndef D2PC_STORE_NT
#define D2PC_STORE_NT 1
#endif
#ifndef D2PC_ONEPASS_STORE_NT
#define D2PC_ONEPASS_STORE_NT 0
#endif
#ifndef D2PC_SCATTER_STORE_NT
#define D2PC_SCATTER_STORE_NT 1
#endif
// chunked two-pass (compact_algo 4) and the register-resident one-launch form (k_compact_resident_lean): point and index
// stores of their ragged pieces.  PLAIN: a piece of K x 16 bytes starts and ends inside 64-byte lines that the
// neighbouring pieces complete, and L2 must keep those lines to merge them -- tools/membench11.hip, 30 % holes: 5.49 TB/s
// plain against 4.27 nt (rows re-blocked to whole lines: 5.59); moving the survivors to the low lanes changes nothing.
// (the chunked two-pass keeps nt all the same: with plain stores its launches run 9-12 % slower, 568 vs 521 us per 16 x 4K
// with 30 % holes + indices -- the output then competes with the chunk's input for the caches; profiles/r04_ab_store_nt.txt)
#ifndef D2PC_CHUNK_STORE_NT
#define D2PC_CHUNK_STORE_NT 1
#endif
#ifndef D2PC_RESIDENT_STORE_NT
#define D2PC_RESIDENT_STORE_NT 0
#endif
#ifndef D2PC_CHUNK_INDEX_NT
#define D2PC_CHUNK_INDEX_NT 0
#endif
#ifndef D2PC_CLEAR_WITH_MEMSET
#define D2PC_CLEAR_WITH_MEMSET 0
#endif
// A/B switch for the production counters of the single pass (tools/ab.py): 0 removes them
#ifndef D2PC_ONEPASS_STATS
#define D2PC_ONEPASS_STATS 1
#endif
// ... and their index stores: nt as well (pipelined form, 30 % holes + indices: 814 -> 711 us)
#ifndef D2PC_CB_INDEX_NT
#define D2PC_CB_INDEX_NT 1
#endif
// point stores of the tile-fused COMPACT kernels: nt (interleaved, 16 x 4K: one tile per block 908 -> 808 us,
// pipelined 870 -> 732 us all valid; profiles/r03_callback_compact.txt) -- unlike the single pass, whose plain stores won
#ifndef D2PC_CB_STORE_NT
#define D2PC_CB_STORE_NT 1
#endif
template <class T>
__device__ __forceinline__ T ld(const T *p) {
#if D2PC_LOAD_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <bool NT, class T>
__device__ __forceinline__ void st(T *p, T v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// --------------------------------------------------------------------------
// per-pixel pieces
// --------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float load_disparity(const uint8_t *frame, uint32_t byte_off, float scale) {
  // `frame` is wave-uniform, `byte_off` a 32-bit per-lane offset: one
  // global_load with an SGPR base.  Plain (cached) loads on purpose: with a
  // border the 256-B wave segments are not line-aligned, so neighbouring
  // waves share their edge lines; `nt` loads made L2 drop those lines and
  // re-fetch them (measured: FETCH_SIZE 1.31x the algorithmic bytes).
  if constexpr (DT == DT_F32) {
    return ld(reinterpret_cast<const float *>(frame + byte_off));
  } else if constexpr (DT == DT_U8) {
    // cpp:61 convertTo(CV_32FC1, scale): product formed in fp32
    return __fmul_rn(float(ld(frame + byte_off)), scale);
  } else {
    return __fmul_rn(float(ld(reinterpret_cast<const uint16_t *>(frame + byte_off))), scale);
  }
}

template <int DT>
__device__ __forceinline__ uint32_t elem_bytes() {
  return DT == DT_F32 ? 4u : DT == DT_U16 ? 2u : 1u;
}

template <int QK>
struct QArg;
template <>
struct QArg<QK_GENERAL> {
  QMat m;
};
template <>
struct QArg<QK_STEREO> {
  QStereo s;
};
template <>
struct QArg<QK_STEREO_CV24> {
  QStereo s;
  QxSegs seg;
};
template <>
struct QArg<QK_STEREO_CV4> {
  QStereo s;
};

// c with qx(u) = double(u) + c for OpenCV 2.4's running column sum (QxSegs): the segment column u lies in.
__device__ __forceinline__ double qx_offset(const QxSegs &sg, uint32_t u) {
  double c = sg.c[0];
#pragma unroll
  for (int j = 1; j < kQxSegs; ++j) {
    if (uint32_t(j) >= sg.n) break;  // (wave-uniform: a scalar branch)
    if (u >= sg.x[j]) c = sg.c[j];
  }
  return c;
}

// cv::reprojectImageTo3D ends every pixel with `if (fabs(d - minDisparity) <= FLT_EPSILON) Z = bigZ`
// (bigZ = 10000); with handleMissingValues = false (cpp:64) minDisparity stays FLT_MAX, so the test
// holds for d == FLT_MAX only.  One compare + select; X and Y stay as computed.
__device__ __forceinline__ float big_z_rule(float d, float Z) { return d == 3.402823466e+38f ? 10000.0f : Z; }

// cpp:63-64  [X Y Z W] = Q.(u,v,d,1); (X/W, Y/W, Z/W) for a GENERAL Q, in ONE published association, bit for bit:
// OpenCV 3.x/4.x's reprojectImageTo3D (calib3d/calibration.cpp + core/matx.hpp; oracle/d2pc_oracle.c FORM_CV4)
//     Vec4d h = Q * Vec4d(x, y, d, 1)   every row s = 0; s += q_k * b_k, left to right, each product and sum rounded
//     Vec3f p = Vec3d(h.val)            the three numerators cast to float
//     p /= h[3]                         ia = 1./h[3]; p[i] = float(p[i] * ia), the product formed in double
// (no contraction: #pragma clang fp contract(off)).  Round 2 evaluated the rows with fused multiply-adds, which matched
// neither published form where a dense Q makes a numerator cancel (tens of float ulp); that form stays behind the
// test hook "general_q_form" = 1 (d2pc_ext_set_test_hook) for comparison.  cv::stereoRectify's Q takes the specialised path below.
__device__ __forceinline__ void reproject(const QArg<QK_GENERAL> &A, uint32_t u, uint32_t v, float d, float &X,
                                          float &Y, float &Z) {
  const double *q = A.m.q;
  const double du = double(u), dv = double(v), dd = double(d);
  if (A.m.form == 1u) {  // (wave-uniform) round 2's fused multiply-adds
    const double nx = fma(q[2], dd, fma(q[0], du, fma(q[1], dv, q[3])));
    const double ny = fma(q[6], dd, fma(q[4], du, fma(q[5], dv, q[7])));
    const double nz = fma(q[10], dd, fma(q[8], du, fma(q[9], dv, q[11])));
    const double nw = fma(q[14], dd, fma(q[12], du, fma(q[13], dv, q[15])));
    const double iw = 1.0 / nw;
    X = float(nx * iw);
    Y = float(ny * iw);
    Z = big_z_rule(d, float(nz * iw));
    return;
  }
  if (A.m.form == 2u) {
    // OpenCV 2.4's loop (calib3d/calibration.cpp; oracle FORM_CV24), bit for bit, for a Q with exact column increments:
    //   per row    qx = q01*y + q03, qy = q11*y + q13, qz = q21*y + q23, qw = q31*y + q33
    //   per column iW = 1./(qw + q32*d); X = (qx + q02*d)*iW ...; then qx += q00, qy += q10, qz += q20, qw += q30
    // q10 = q20 = q30 = +0 leave qy, qz, qw as the row formed them (but for a -0 there, which the first += turns
    // into +0); qx is the running sum replayed by the host: column u lies in one of n_seg segments in which
    // qx = u + seg_c[j] exactly.
#pragma clang fp contract(off)
    const double qx = du + qx_offset(A.m.seg, u);
    double qy = q[5] * dv + q[7], qz = q[9] * dv + q[11], qw = q[13] * dv + q[15];
    if (u != 0u) qy = qy + q[4], qz = qz + q[8], qw = qw + q[12];
    const double iw = 1.0 / (qw + q[14] * dd);
    X = float((qx + q[2] * dd) * iw);
    Y = float((qy + q[6] * dd) * iw);
    Z = big_z_rule(d, float((qz + q[10] * dd) * iw));
    return;
  }
  double h[4];
  {
    // every product and every sum rounds on its own, as in the x86-64 builds of OpenCV: no fused multiply-add
    // (HIP's __dmul_rn / __dadd_rn are plain operators and would be contracted like them)
#pragma clang fp contract(off)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      h[r] = (((0.0 + q[4 * r] * du) + q[4 * r + 1] * dv) + q[4 * r + 2] * dd) + q[4 * r + 3];  // (q_3 * 1.0 is exact)
  }
  const double ia = 1.0 / h[3];
  X = float(double(float(h[0])) * ia);
  Y = float(double(float(h[1])) * ia);
  Z = big_z_rule(d, float(double(float(h[2])) * ia));
}

// Same evaluation with Q = [1 0 0 cx; 0 1 0 cy; 0 0 0 f; 0 0 a b]: the
// products with +0.0 and 1.0 are exact, so only the sums that can round
// remain.  A non-finite d makes every coordinate NaN in the general form
// (0*inf), reproduced here by poisoning d before W is formed.
//
// W and the numerators of the kind's OpenCV generation (d2pc_device.hpp).
template <int QK>
__device__ __forceinline__ double stereo_w(const QArg<QK> &A, double dd) {
  if constexpr (QK == QK_STEREO) {
    return fma(A.s.a, dd, A.s.b);  // the default kind: one rounding
  } else {
#pragma clang fp contract(off)
    const double t = A.s.a * dd;  // both generations round the product and the sum apart
    return A.s.b + t;
  }
}
template <int QK>
__device__ __forceinline__ double stereo_nx(const QArg<QK> &A, uint32_t u) {
  const double du = double(u);
  if constexpr (QK == QK_STEREO_CV24) {
    return du + qx_offset(A.seg, u);
  } else {
    const double n = du + A.s.cx;
    if constexpr (QK == QK_STEREO_CV4) return double(float(n));
    return n;
  }
}
template <int QK>
__device__ __forceinline__ double stereo_ny(const QArg<QK> &A, uint32_t v) {
  const double n = double(v) + A.s.cy;
  if constexpr (QK == QK_STEREO_CV4) return double(float(n));
  return n;
}
template <int QK, typename std::enable_if<is_stereo(QK), int>::type = 0>
__device__ __forceinline__ void reproject(const QArg<QK> &A, uint32_t u, uint32_t v, float d, float &X, float &Y, float &Z) {
  const float dsel = fabsf(d) < __builtin_huge_valf() ? d : __builtin_nanf("");
  const double nw = stereo_w(A, double(dsel));
  const double iw = 1.0 / nw;
  X = float(stereo_nx(A, u) * iw);
  Y = float(stereo_ny(A, v) * iw);
  Z = big_z_rule(d, float(A.s.f * iw));
}

__device__ __forceinline__ bool point_is_valid(float X, float Y, float Z, float d, float min_disparity) {
  // finite <=> |x| < inf; NaN compares false
  const float inf = __builtin_huge_valf();
  return (int(fabsf(X) < inf) & int(fabsf(Y) < inf) & int(fabsf(Z) < inf) & int(!(d <= min_disparity))) != 0;
}

template <bool NT>
__device__ __forceinline__ void store_point(float4 *frame_out, uint32_t point, float X, float Y, float Z) {
  // pcl::PointXYZ = {x,y,z,1.0f} (cpp:74): one global_store_dwordx4 with an
  // SGPR base and a 32-bit byte offset (host guarantees roi_n <= 2^28).
  const v4f p = {X, Y, Z, 1.0f};
  st<NT>(reinterpret_cast<v4f *>(reinterpret_cast<uint8_t *>(frame_out) + (point << 4)), p);
}

#ifndef D2PC_INDEX_STORE_NT
#define D2PC_INDEX_STORE_NT 0
#endif
#ifndef D2PC_ONEPASS_INDEX_NT
#define D2PC_ONEPASS_INDEX_NT 0
#endif
__device__ __forceinline__ void store_index(uint32_t *frame_idx, uint32_t point, uint32_t pix) {
  // plain store: a wave writes only 256 B of indices (partial lines that L2 must merge with its
  // neighbours' pieces); nt here cost +20 % on the 30 %-holes + index case
  st<D2PC_INDEX_STORE_NT != 0>(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(frame_idx) + (point << 2)), pix);
}

// Exact validity of a STEREO-structured point WITHOUT evaluating it (used by
// the two-pass count kernel, which only needs the predicate):
//   |W| >= w_safe  =>  |num/W| <= 2^126 for every numerator of the frame, so
//                      X, Y, Z are finite floats            -> valid
//   W == 0 or NaN  =>  iW is inf/NaN, Z = f*iW is not finite -> invalid
//   0 < |W| < w_safe (never seen with real calibrations): evaluate fully.
// w_safe = 2^-126 * max|numerator| is formed on the host (QStereo::w_safe).
template <int QK>
__device__ __forceinline__ bool stereo_point_valid(const QArg<QK> &A, uint32_t u, uint32_t v, float d,
                                                   float min_disparity) {
  const float dsel = fabsf(d) < __builtin_huge_valf() ? d : __builtin_nanf("");
  const double nw = stereo_w(A, double(dsel));
  const double aw = fabs(nw);
  bool ok = aw >= A.s.w_safe;  // false for NaN
  if (!ok && aw > 0.0) {       // tiny non-zero W: decide by the real arithmetic
    float X, Y, Z;
    reproject(A, u, v, d, X, Y, Z);
    const float inf = __builtin_huge_valf();
    ok = (fabsf(X) < inf) && (fabsf(Y) < inf) && (fabsf(Z) < inf);
  }
  return ok && !(d <= min_disparity);
}

// ---- pixel <-> (slot, wave, lane) mapping of a tile -------------------------
// A tile is 256*PXT consecutive ROI pixels, cut into batches of 1024; inside a
// batch each WAVE owns 256 consecutive pixels and walks them in 4 slots of 64:
//   pixel(k, wave, lane) = base + (k/4)*1024 + wave*256 + (k%4)*64 + lane
// so a wave's store for slot k is one contiguous 1-KiB piece, and a wave's
// input for a batch is one contiguous 1-KiB piece as well (staged through a
// wave-private LDS strip when it can be fetched 16 B per lane).
__device__ __forceinline__ uint32_t slot_pixel(uint32_t base, uint32_t wave, uint32_t lane, int k) {
  return base + uint32_t(k >> 2) * 1024u + wave * 256u + uint32_t(k & 3) * 64u + lane;
}
// Row-major order of the (slot, wave) cells == pixel order inside the tile.
__device__ __forceinline__ int cell_index(int k, uint32_t wave) { return ((k >> 2) * 4 + int(wave)) * 4 + (k & 3); }

// ROI coordinates of a pixel, advanced by constant pixel counts whose
// (rows, columns) decomposition the host precomputed.
struct Walker {
  uint32_t u, v;  // ROI-relative column / row
  __device__ __forceinline__ Walker(const Geom &g, uint32_t i0) {
    v = fdiv(i0, g.div_roi_w);
    u = i0 - v * g.roi_w;
  }
  __device__ __forceinline__ void step(const Geom &g, uint32_t dv, uint32_t du) {
    u += du;
    v += dv;
    if (u >= g.roi_w) {
      u -= g.roi_w;
      ++v;
    }
  }
};

// Image coordinates of ROI pixel i (the paths that have no stepped coordinates at hand).
__device__ __forceinline__ void pixel_coords(const Geom &g, uint32_t i, uint32_t &uu, uint32_t &vv) {
  const uint32_t v = fdiv(i, g.div_roi_w);
  uu = i - v * g.roi_w + g.border;
  vv = v + g.border;
}

// Disparities + image coordinates of the PXT pixels of one thread.
template <int PXT>
struct TileIn {
  float d[PXT];
  uint32_t uu[PXT], vv[PXT];  // image coordinates (border added)
};

// Image coordinates of the thread's PXT pixels (see slot_pixel).
template <int PXT>
__device__ __forceinline__ void tile_coords(uint32_t (&uu)[PXT], uint32_t (&vv)[PXT], const Geom &g, uint32_t base,
                                            uint32_t wave, uint32_t lane) {
  Walker w(g, base + wave * 256u + lane);
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    uu[k] = w.u + g.border;
    vv[k] = w.v + g.border;
    if ((k & 3) == 3) w.step(g, g.s832_v, g.s832_u);  // to slot 0 of the next batch
    else w.step(g, g.s64_v, g.s64_u);
  }
}

// Disparities of the thread's PXT pixels; loads are issued first (all in
// flight), arithmetic comes later.
//  VEC = false: one dword per lane and slot (any dtype, any alignment).
//  VEC = true : fp32 rows whose 4-pixel groups are 16-B aligned and never
//               straddle a ROI row (host-checked): one 16-B load per lane and
//               batch (a 1-KiB coalesced piece per wave), transposed to the
//               slot layout through the wave's own LDS strip.  LDS is in-order
//               per wave, so no barrier is involved.
template <int DT, int PXT, bool VEC>
__device__ __forceinline__ void tile_load_d(float (&d)[PXT], const uint8_t *fin, const Geom &g, uint32_t base,
                                            uint32_t wave, uint32_t lane, float *wave_strip) {
  if constexpr (!VEC) {
    uint32_t uu[PXT], vv[PXT];
    tile_coords<PXT>(uu, vv, g, base, wave, lane);
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
      // byte offsets grow with the ROI index, so clamping to the last ROI
      // pixel keeps the tail slots of a frame's last tile in bounds without
      // predicating the loads (their results are never stored)
      const uint32_t off = vv[k] * g.row_stride + uu[k] * elem_bytes<DT>();
      d[k] = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
    }
  } else {
    static_assert(!VEC || DT == DT_F32, "16-B row loads are fp32 only");
    v4f q[PXT / 4];
    Walker w4(g, base + wave * 256u + lane * 4u);
#pragma unroll
    for (int j = 0; j < PXT / 4; ++j) {
      const uint32_t off = (w4.v + g.border) * g.row_stride + (w4.u + g.border) * 4u;
      const uint32_t last4 = g.last_off - 12u;  // the frame's last aligned group
      q[j] = ld(reinterpret_cast<const v4f *>(fin + (off < last4 ? off : last4)));
      w4.step(g, g.s1024_v, g.s1024_u);
    }
#pragma unroll
    for (int j = 0; j < PXT / 4; ++j) {
      *reinterpret_cast<v4f *>(wave_strip + lane * 4u) = q[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) d[j * 4 + sl] = wave_strip[uint32_t(sl) * 64u + lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int DT, int PXT, bool VEC>
__device__ __forceinline__ void tile_load(TileIn<PXT> &t, const uint8_t *fin, const Geom &g, uint32_t base,
                                          uint32_t wave, uint32_t lane, float *wave_strip) {
  tile_coords<PXT>(t.uu, t.vv, g, base, wave, lane);
  tile_load_d<DT, PXT, VEC>(t.d, fin, g, base, wave, lane, wave_strip);
}

template <int DT, int QK, int PXT>
struct TileRegs {
  float X[PXT], Y[PXT], Z[PXT], d[PXT];
  uint32_t pix[PXT];  // source pixel index v*W+u (image coordinates)
};

template <int DT, int QK, int PXT, bool VEC>
__device__ __forceinline__ void tile_compute(TileRegs<DT, QK, PXT> &r, const uint8_t *fin, const Geom &g,
                                             const QArg<QK> &Q, uint32_t base, uint32_t wave, uint32_t lane,
                                             float *wave_strip) {
  TileIn<PXT> t;
  tile_load<DT, PXT, VEC>(t, fin, g, base, wave, lane, wave_strip);
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    r.d[k] = t.d[k];
    reproject(Q, t.uu[k], t.vv[k], t.d[k], r.X[k], r.Y[k], r.Z[k]);
    r.pix[k] = t.vv[k] * g.width + t.uu[k];
  }
}

// Wave-private staging strips for the VEC load path (1 KiB per wave).
#define D2PC_DECLARE_STRIPS(VEC, wave)                                  \
  __shared__ float s_strips_[(VEC) ? (kBlock / 64) * 256 : 1];         \
  float *wave_strip = (VEC) ? s_strips_ + (wave) * 256u : nullptr

// --------------------------------------------------------------------------
// K1: PARITY mode -- every ROI pixel, reference order, nothing filtered.
// --------------------------------------------------------------------------
template <int DT, int QK, int PXT, bool VEC>
__global__ __launch_bounds__(kBlock) void k_reproject_pack(const uint8_t *__restrict__ disp,
                                                           float4 *__restrict__ out,
                                                           uint32_t *__restrict__ out_index,
                                                           uint32_t *__restrict__ counts, const Geom g,
                                                           const QArg<QK> Q) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  D2PC_DECLARE_STRIPS(VEC, wave);
  for (uint32_t t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
    float4 *fout = out + uint64_t(f) * g.out_frame_stride;
    uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
    const uint32_t base = lt * uint32_t(kBlock * PXT);
    TileRegs<DT, QK, PXT> r;
    tile_compute<DT, QK, PXT, VEC>(r, fin, g, Q, base, wave, lane, wave_strip);
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
      const uint32_t i = slot_pixel(base, wave, lane, k);
      if (i < g.roi_n) {
        store_point<D2PC_STORE_NT != 0>(fout, i, r.X[k], r.Y[k], r.Z[k]);
        if (fidx) store_index(fidx, i, r.pix[k]);
      }
    }
    if (counts && lt == 0 && tid == 0) counts[f] = g.roi_n;
  }
}

// --------------------------------------------------------------------------
// K1s: PARITY with SMALL one-shot blocks: a block is one tile of 256 * S consecutive ROI pixels (S = 1, 2 or 4),
// thread t takes pixels base + k * 256 + t, and the grid is the tile count -- no loop over tiles.  tools/membench9.hip:
// the same 4 B -> 16 B stream moves at 6.6 TB/s with two pixels per thread and one block per 512 pixels, against 5.7
// TB/s with eight per thread (K1's shape) on the same device: short-lived waves keep more independent requests in
// flight than long ones whose stores queue, in order, behind their own loads.
// --------------------------------------------------------------------------
template <int DT, int QK, int S>
__global__ __launch_bounds__(kBlock) void k_reproject_pack_small(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                                 uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
                                                                 const Geom g, const QArg<QK> Q) {
  const uint32_t t = blockIdx.x;
  const uint32_t f = fdiv(t, g.div_tpf);
  const uint32_t lt = t - f * g.tiles_per_frame;
  const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
  const uint32_t base = lt * uint32_t(kBlock * S) + threadIdx.x;
  float d[S];
  uint32_t uu[S], vv[S];
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const uint32_t i = base + uint32_t(k) * uint32_t(kBlock);
    const uint32_t v = fdiv(i, g.div_roi_w);
    uu[k] = i - v * g.roi_w + g.border;
    vv[k] = v + g.border;
    // (clamped to the frame's last ROI pixel: the tail of a frame's last tile loads in bounds and stores nothing)
    const uint32_t off = vv[k] * g.row_stride + uu[k] * elem_bytes<DT>();
    d[k] = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
  }
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const uint32_t i = base + uint32_t(k) * uint32_t(kBlock);
    float X, Y, Z;
    reproject(Q, uu[k], vv[k], d[k], X, Y, Z);
    if (i < g.roi_n) {
      store_point<D2PC_STORE_NT != 0>(fout, i, X, Y, Z);
      if (fidx) store_index(fidx, i, vv[k] * g.width + uu[k]);
    }
  }
  if (counts && lt == 0 && threadIdx.x == 0) counts[f] = g.roi_n;
}

// --------------------------------------------------------------------------
// K1g: the callback body TILE BY TILE -- bit-sliced k x k median of a 256 x 32 tile of the inset ROI
// (d2pc_median_bs_tile.hpp, cpp:55-57) and, from the filtered bytes still in LDS, the tile's points
// (cpp:60-85, PARITY).  No hand-off between blocks and no filtered image in memory: the VALU-bound filter
// and the store stream of the reprojection overlap because the chip's ~770 resident blocks are at
// different stages at any time.  (An earlier form -- one persistent kernel whose blocks switched between
// filter tiles and reprojection tiles, handing frames over through sc1 stores and loads -- was 1.4x SLOWER
// than two launches and has been removed: DESIGN.md section 9.)
//  * 8-bit input has 256 disparities, and with stereoRectify's Q (QK_STEREO) W = a*d + b does not depend on
//    the pixel: every block evaluates 1/W and Z once per byte value (one division per THREAD) into LDS, and a
//    pixel costs two fp64 additions, two multiplications and two casts -- the same operations on the same
//    operands as reproject(), so the points are bit-identical to k_reproject_pack's.
//  * a wave stores 64 consecutive points per instruction (1 KiB), like the PARITY kernel.
// --------------------------------------------------------------------------
template <int KS, int QK>
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_callback_bs(
    const uint8_t *__restrict__ src, float4 *__restrict__ out, uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
    const MedianArgs ma, const Geom g, const QArg<QK> Q) {
  using S = MedianBsShape<KS>;
  static_assert(S::THREADS == 256, "one thread per byte value fills the table");
  __shared__ __attribute__((aligned(16))) uint32_t s_w[S::W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_raw[S::RAW_WORDS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t b = blockIdx.x;
  const uint32_t f = b / (ma.tiles_x * ma.tiles_y);
  b -= f * ma.tiles_x * ma.tiles_y;
  const uint32_t ty = b / ma.tiles_x, tx = b - ty * ma.tiles_x;
  const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);  // first output pixel
  median_bs_tile<KS>(src + uint64_t(f) * ma.src_frame_stride, ma, int(x0), int(y0), s_w, s_raw, tid);

  double *lut_iw = reinterpret_cast<double *>(s_raw);              // [256]
  float *lut_z = reinterpret_cast<float *>(s_raw) + 2 * 256;       // [256]
  static_assert(S::RAW_WORDS >= 3 * 256, "the table fits where the staged rows were");
  if constexpr (is_stereo(QK)) {
    const float d = __fmul_rn(float(tid), g.scale);  // cpp:61, as load_disparity<DT_U8>
    const float dsel = fabsf(d) < __builtin_huge_valf() ? d : __builtin_nanf("");
    const double iw = 1.0 / stereo_w(Q, double(dsel));
    lut_iw[tid] = iw;
    lut_z[tid] = big_z_rule(d, float(Q.s.f * iw));
    __syncthreads();
  }
  const uint8_t *ob = reinterpret_cast<const uint8_t *>(s_w);
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
  const uint32_t x_end = ma.out_x0 + ma.out_w, y_end = ma.out_y0 + ma.out_h;
  // a lane's four columns do not change from row to row: (u + cx) is formed once (QK_STEREO)
  double xs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xs[q] = 0.0;
    if constexpr (is_stereo(QK)) xs[q] = stereo_nx(Q, x0 + 64u * uint32_t(q) + lane);
  }
#pragma unroll 1
  for (uint32_t r = wave; r < uint32_t(S::TH); r += uint32_t(S::THREADS / 64)) {  // a wave takes every fourth row
    const uint32_t y = y0 + r;
    if (y >= y_end) break;
    const uint32_t row_point = (y - g.border) * g.roi_w - g.border;  // + x = the point's index (wraps for x < border: never used)
    uint32_t raw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[q] = ob[r * uint32_t(S::OUT_STRIDE) + 64u * uint32_t(q) + lane];
    double ys = 0.0;
    if constexpr (is_stereo(QK)) ys = stereo_ny(Q, y);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t x = x0 + 64u * uint32_t(q) + lane;
      float X, Y, Z;
      if constexpr (is_stereo(QK)) {
        const double iw = lut_iw[raw[q]];
        X = float(xs[q] * iw);
        Y = float(ys * iw);
        Z = lut_z[raw[q]];
      } else {
        reproject(Q, x, y, __fmul_rn(float(raw[q]), g.scale), X, Y, Z);
      }
      if (x < x_end) {
        store_point<D2PC_STORE_NT != 0>(fout, row_point + x, X, Y, Z);
        if (fidx) st<D2PC_CB_INDEX_NT != 0>(fidx + (row_point + x), y * g.width + x);  // (nt: 647 -> 626 us with indices, 16 x 4K)
      }
    }
  }
  if (counts && b == 0 && tid == 0) counts[f] = g.roi_n;
}

// --------------------------------------------------------------------------
// COMPACT mode building blocks
// --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
  // number of set bits of `mask` in lanes below this one
  return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// Validity ballots of a computed tile, one 64-bit wave mask per slot.
template <int DT, int QK, int PXT>
__device__ __forceinline__ void tile_ballots(const TileRegs<DT, QK, PXT> &r, const Geom &g, uint32_t base,
                                             uint32_t wave, uint32_t lane, uint64_t (&mask)[PXT]) {
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const uint32_t i = slot_pixel(base, wave, lane, k);
    const bool ok = (i < g.roi_n) && point_is_valid(r.X[k], r.Y[k], r.Z[k], r.d[k], g.min_disparity);
    mask[k] = __ballot(ok);
  }
}

// Exclusive offsets of every (slot, wave) cell of a block in row-major
// (slot-major, wave-minor) order == pixel order inside the tile.
// s_cnt[cell_index(k, w)] holds wave w's popcount for slot k.  Returns the
// exclusive scan in lanes 0..CELLS-1 and the tile total in `total`.
template <int CELLS>
__device__ __forceinline__ uint32_t scan_cells(const uint32_t *s_cnt, uint32_t lane, uint32_t &total) {
  static_assert(CELLS <= 64, "one wave scans all cells");
  const uint32_t c = lane < CELLS ? s_cnt[lane] : 0u;
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t n = __shfl_up(incl, o, 64);
    if (lane >= uint32_t(o)) incl += n;
  }
  total = __builtin_amdgcn_readlane(incl, 63);
  return incl - c;
}

struct FrameState {
  uint32_t *ticket;
  uint64_t *group_acc;
  uint64_t *granules;
  __device__ __forceinline__ FrameState(uint8_t *state, const Geom &g, uint32_t f) {
    uint8_t *fs = state + sizeof(StateHeader) + uint64_t(f) * g.frame_state_stride;
    ticket = reinterpret_cast<uint32_t *>(fs);
    group_acc = reinterpret_cast<uint64_t *>(fs + kFrameTicketBytes);
    granules = reinterpret_cast<uint64_t *>(fs + kFrameTicketBytes + uint64_t(g.groups_per_frame) * kGroupAccStride);
  }
  // two-pass view of the granule area: 4 x uint32 per tile (per-wave counts,
  // then the tile's exclusive prefix in word 0)
  __device__ __forceinline__ uint32_t *partials() const { return reinterpret_cast<uint32_t *>(granules); }
  // every group accumulator sits on a line of its own: each is hit by 64
  // atomics and by the polls of every later tile of the frame
  __device__ __forceinline__ uint64_t *group_word(uint32_t grp) const {
    return reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(group_acc) + uint64_t(grp) * kGroupAccStride);
  }
};

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding global
// stores (vmcnt(0)), which between an epilogue's store burst and the next tile's work is exactly what must overlap.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void backoff(uint32_t spins) {
  // 64 .. ~2000 clocks between polls; pollers must not crowd the memory
  // channel the publishers' atomics go through
  const uint32_t n = spins < 5 ? (1u << spins) : 32u;
  for (uint32_t j = 0; j < n; ++j) __builtin_amdgcn_s_sleep(1);
}

// What a control wave's waits cost, summed over the block's tiles.  The sums live in LDS (three words of the
// block), not in registers: the single pass has no scalar register to spare -- kept in registers, these two
// counters cost 9 % (16 x 4K) to 25 % (32 x 1080p) of the kernel's time through the spills they caused in the
// workers' loop (profiles/r03_ab_counters.txt).
struct PollStats {
  uint32_t *lds;  // [0] tiles served, [1] failed polls, [2] 100 MHz ticks spent in waits that needed more than one look
};

// Waits (WAIT) until the 64-bit word at p satisfies `ready`, and returns it.
// First look is a normal cached load: a word that already carries its
// completion mark (all 64 arrivals / the granule tag) is final, so a cached
// copy of it is as good as memory; only words not yet complete are re-read
// with agent-scope (coherent) loads, with back-off.
template <bool WAIT, class Ready>
__device__ __forceinline__ uint64_t read_counted(const uint64_t *p, bool on, StateHeader *hdr, uint32_t lane,
                                                 PollStats &ps, uint32_t spin_ticks, Ready ready) {
  using gu64 = __attribute__((address_space(1))) const uint64_t;
  uint64_t v = 0;
  if constexpr (!WAIT) {
    if (on) v = *p;
    return v;
  } else {
    if (on) v = __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool ok = !on || ready(v);
    uint32_t spins = 0;
    uint64_t t0 = 0;
    while (!__all(ok)) {
      if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
      backoff(spins);
      if (!ok) {
        v = __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ready(v);
      }
      // bounded by time: give up once the budget is spent, and as soon as ANY wave of the launch has
      // given up (sticky flag), so a broken launch drains at once instead of timing out tile by tile
      ++spins;
      if ((spins & 15u) == 0 &&
          (__builtin_amdgcn_s_memrealtime() - t0 > uint64_t(spin_ticks) ||
           __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0 && __hip_atomic_exchange(&hdr->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
          atomicAdd(&hdr->stats->timeouts, 1ull);  // the launch's first give-up counts it (d2pc_compact_stats)
        break;
      }
    }
#if D2PC_ONEPASS_STATS
    if (spins && lane == 0) {  // production counters (d2pc_compact_stats): failed polls and the time they took
      ps.lds[1] += spins;
      ps.lds[2] += uint32_t(__builtin_amdgcn_s_memrealtime() - t0);
    }
#endif
    return v;
  }
}

// Sum of the point counts of all tiles of the frame that precede local tile
// lt: complete groups via the group accumulators, the own (partial) group via
// tile granules.  WAIT = true (single pass): bounded wait until every
// predecessor has published; WAIT = false (two-pass): values are final.
// What a block already knows about its frame: groups [0, groups) are complete
// and their counts add up to `sum`.  A block's successive tiles are less than
// a group apart, so each prefix needs ~one new group word, not all of them.
struct KnownGroups {
  uint32_t groups = 0, sum = 0;
};

template <bool WAIT>
__device__ __forceinline__ uint32_t prefix_before(const FrameState &fs, StateHeader *hdr, uint32_t lt,
                                                  uint32_t lane, PollStats &ps, KnownGroups &known,
                                                  uint32_t spin_ticks) {
  const uint32_t grp = lt / kGroupTiles;
  uint32_t sum = 0;
  for (uint32_t g0 = known.groups; g0 < grp; g0 += 64) {  // groups below grp hold kGroupTiles tiles each
    const uint32_t gi = g0 + lane;
    const bool on = gi < grp;
    const uint64_t v = read_counted<WAIT>(fs.group_word(gi), on, hdr, lane, ps, spin_ticks,
                                          [](uint64_t x) { return uint32_t(x >> 32) == uint32_t(kGroupTiles); });
    sum += on ? uint32_t(v) : 0u;
  }
  if (grp > known.groups) {  // wave-uniform
    known.sum += wave_sum(sum);
    known.groups = grp;
  }
  sum = 0;
  {  // tiles grp*64 .. lt-1 of the own group (< 64 of them)
    const uint32_t ti = grp * kGroupTiles + lane;
    const bool on = ti < lt;
    const uint64_t v = read_counted<WAIT>(fs.granules + 2u * ti, on, hdr, lane, ps, spin_ticks,
                                          [](uint64_t x) { return (x & kGranuleTag) != 0; });
    sum += on ? uint32_t(v) : 0u;
  }
  return known.sum + wave_sum(sum);
}

template <int DT, int QK, int PXT>
__device__ __forceinline__ void tile_scatter(const TileRegs<DT, QK, PXT> &r, const uint64_t (&mask)[PXT],
                                             float4 *fout, uint32_t *fidx, uint32_t tile_prefix,
                                             uint32_t cell_excl, uint32_t wave, uint32_t lane, uint32_t roi_n) {
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const uint32_t cell = __builtin_amdgcn_readlane(cell_excl, cell_index(k, wave));
    const uint32_t pos = tile_prefix + cell + mbcnt64(mask[k]);
    // pos < roi_n always holds for a correct prefix; the guard keeps a stale
    // or timed-out prefix from ever becoming an out-of-bounds store
    if (((mask[k] >> lane) & 1) && pos < roi_n) {
      store_point<D2PC_SCATTER_STORE_NT != 0>(fout, pos, r.X[k], r.Y[k], r.Z[k]);
      if (fidx) store_index(fidx, pos, r.pix[k]);
    }
  }
}

// --------------------------------------------------------------------------
// K2a/K2b: two-pass compaction (count -> scatter).  No in-launch hand-off.
// --------------------------------------------------------------------------
template <int DT, int QK, int PXT, bool VEC>
__global__ __launch_bounds__(kBlock) void k_compact_count(const uint8_t *__restrict__ disp, uint8_t *state,
                                                          const Geom g, const QArg<QK> Q) {
  // Counting needs no pixel order inside a tile, so there is no LDS, no
  // barrier and no cross-wave reduction here: every WAVE leaves its own
  // partial count (4 per tile); the scan kernel adds them up.
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (uint32_t t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    const uint32_t base = lt * uint32_t(kBlock * PXT);
    const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
    uint32_t c = 0;
    if constexpr (is_stereo(QK) && VEC) {
      // 16 B per lane straight from the rows; the exact predicate needs only d
      // (~4 fp64 operations per pixel), so the pass stays read-bound
      v4f q[PXT / 4];
      uint32_t uu[PXT / 4], vv[PXT / 4];
      Walker w4(g, base + wave * 256u + lane * 4u);
#pragma unroll
      for (int j = 0; j < PXT / 4; ++j) {
        uu[j] = w4.u + g.border;
        vv[j] = w4.v + g.border;
        const uint32_t off = vv[j] * g.row_stride + uu[j] * 4u;
        const uint32_t last4 = g.last_off - 12u;
        q[j] = ld(reinterpret_cast<const v4f *>(fin + (off < last4 ? off : last4)));
        w4.step(g, g.s1024_v, g.s1024_u);
      }
#pragma unroll
      for (int j = 0; j < PXT / 4; ++j) {
        const uint32_t i0 = base + uint32_t(j) * 1024u + wave * 256u + lane * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (i0 + uint32_t(e) < g.roi_n) &&
                          stereo_point_valid(Q, uu[j] + uint32_t(e), vv[j], q[j][e], g.min_disparity);
          c += uint32_t(__popcll(__ballot(ok)));
        }
      }
    } else {
      float *wave_strip = nullptr;  // VEC staging is only worth it for the ordered passes
      if constexpr (is_stereo(QK)) {
        TileIn<PXT> in;
        tile_load<DT, PXT, false>(in, fin, g, base, wave, lane, wave_strip);
#pragma unroll
        for (int k = 0; k < PXT; ++k) {
          const uint32_t i = slot_pixel(base, wave, lane, k);
          const bool ok = (i < g.roi_n) && stereo_point_valid(Q, in.uu[k], in.vv[k], in.d[k], g.min_disparity);
          c += uint32_t(__popcll(__ballot(ok)));
        }
      } else {
        TileRegs<DT, QK, PXT> r;
        uint64_t mask[PXT];
        tile_compute<DT, QK, PXT, false>(r, fin, g, Q, base, wave, lane, wave_strip);
        tile_ballots<DT, QK, PXT>(r, g, base, wave, lane, mask);
#pragma unroll
        for (int k = 0; k < PXT; ++k) c += uint32_t(__popcll(mask[k]));
      }
    }
    if (lane == 0) {
      const FrameState fs(state, g, f);
      fs.partials()[lt * 4u + wave] = c;  // plain store: read by k_compact_scan after the kernel boundary
    }
  }
}

// K2a': per-frame exclusive scan of the tile counts (4 wave partials each),
// one block of 1024 threads per frame.  Every thread owns a run of consecutive
// tiles, so the block synchronises once whatever the frame size; up to
// kScanBatch tiles per thread (8192 tiles: 16.7 Mpixel frames at 2048-pixel
// tiles) are fetched with independent loads issued together and stay in
// registers for the write-back -- a loop of dependent-looking loads made this
// kernel 8 us for one 4K frame, a third of the scatter it feeds.
// Leaves the exclusive prefix of tile i in partials[4*i].
constexpr int kScanThreads = 1024, kScanBatch = 8;
constexpr uint32_t kSelfScanTiles = 1024;  // frames up to this many tiles: the scatter kernel sums the counts itself
__global__ __launch_bounds__(kScanThreads) void k_compact_scan(uint8_t *state, uint32_t *__restrict__ counts,
                                                               const Geom g) {
  __shared__ uint32_t s_w[kScanThreads / 64];
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
  const FrameState fs(state, g, blockIdx.x);
  const uint4 *part = reinterpret_cast<const uint4 *>(fs.partials());
  const uint32_t per = (g.tiles_per_frame + kScanThreads - 1) / kScanThreads;
  const uint32_t t0 = tid * per, t1 = t0 + per < g.tiles_per_frame ? t0 + per : g.tiles_per_frame;
  const bool batched = per <= uint32_t(kScanBatch);  // block-uniform
  uint32_t tot[kScanBatch];
  uint32_t mine = 0;
  if (batched) {
#pragma unroll
    for (int k = 0; k < kScanBatch; ++k) {
      const uint32_t i = t0 + uint32_t(k);
      uint4 p = {0u, 0u, 0u, 0u};
      if (i < t1) p = part[i];
      tot[k] = p.x + p.y + p.z + p.w;
      mine += tot[k];
    }
  } else {
    for (uint32_t i = t0; i < t1; ++i) {
      const uint4 p = part[i];
      mine += p.x + p.y + p.z + p.w;
    }
  }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t n = __shfl_up(incl, o, 64);
    if (lane >= uint32_t(o)) incl += n;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const uint32_t x = s_w[w];
    before += uint32_t(w) < wave ? x : 0u;
    total += x;
  }
  uint32_t run = before + incl - mine;  // exclusive prefix of this thread's first tile
  if (batched) {
#pragma unroll
    for (int k = 0; k < kScanBatch; ++k) {
      const uint32_t i = t0 + uint32_t(k);
      if (i < t1) fs.partials()[4u * i] = run;
      run += tot[k];
    }
  } else {
    for (uint32_t i = t0; i < t1; ++i) {
      const uint4 p = part[i];
      fs.partials()[4u * i] = run;
      run += p.x + p.y + p.z + p.w;
    }
  }
  if (tid == 0) counts[blockIdx.x] = total;
}

template <int DT, int QK, int PXT, bool VEC>
__global__ __launch_bounds__(kBlock) void k_compact_scatter(const uint8_t *__restrict__ disp,
                                                            float4 *__restrict__ out,
                                                            uint32_t *__restrict__ out_index,
                                                            uint32_t *__restrict__ counts, uint8_t *state,
                                                            const Geom g, const QArg<QK> Q, const uint32_t selfscan) {
  constexpr int CELLS = PXT * (kBlock / 64);
  __shared__ uint32_t s_cnt[CELLS];
  __shared__ uint32_t s_red[kBlock / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  D2PC_DECLARE_STRIPS(VEC, wave);
  for (uint32_t t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    const uint32_t base = lt * uint32_t(kBlock * PXT);
    const FrameState fs(state, g, f);
    // selfscan (frames of <= kSelfScanTiles tiles: single camera frames): the block adds up the counts of the tiles
    // before its own itself -- <= 16 KB from L2, requested ahead of the disparity loads -- and the scan kernel with
    // its launch gap (a quarter of a 1080p frame's compaction time) is not launched at all
    uint32_t before = 0;
    if (selfscan) {
      const uint4 *part = reinterpret_cast<const uint4 *>(fs.partials());
      for (uint32_t i = tid; i < lt; i += uint32_t(kBlock)) {
        const uint4 p = part[i];
        before += p.x + p.y + p.z + p.w;
      }
    }
    TileRegs<DT, QK, PXT> r;
    uint64_t mask[PXT];
    tile_compute<DT, QK, PXT, VEC>(r, disp + uint64_t(f) * g.in_frame_stride, g, Q, base, wave, lane, wave_strip);
    tile_ballots<DT, QK, PXT>(r, g, base, wave, lane, mask);
    if (selfscan) before = wave_sum(before);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < PXT; ++k) s_cnt[cell_index(k, wave)] = uint32_t(__popcll(mask[k]));
      s_red[wave] = before;
    }
    __syncthreads();
    uint32_t total;
    const uint32_t excl = scan_cells<CELLS>(s_cnt, lane, total);
    uint32_t prefix;
    if (selfscan) {
      prefix = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) prefix += s_red[w];
      if (counts && lt == g.tiles_per_frame - 1 && tid == 0) counts[f] = prefix + total;
    } else {
      prefix = fs.partials()[4u * lt];  // exclusive prefix left by k_compact_scan (uniform load)
    }
    float4 *fout = out + uint64_t(f) * g.out_frame_stride;
    uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
    tile_scatter<DT, QK, PXT>(r, mask, fout, fidx, prefix, excl, wave, lane, g.roi_n);
    __syncthreads();
  }
}

// --------------------------------------------------------------------------
// K2r: COMPACT for camera-size launches in ONE launch (compact_algo 3): one block per tile, every block RESIDENT.
// The two-pass form costs a single frame two or three launches and two reads of the input (one 1080p frame 16 us
// against 6.4 us PARITY); the persistent single pass serialises a lone frame on its ticket word.  Here every block
// computes its tile once, publishes its survivor count as an 8-byte granule {epoch, count} and adds up the granules
// of all tiles before it in the frame (<= 1023: all requested together), then scatters.
//  * No zeroing launch: the granule carries the launch's EPOCH (a per-context counter in [2^30, 2^31): no count and no
//    other kernel's state word looks like one), so whatever an earlier launch left in the buffer reads "not yet".
//    An epoch is a kernel argument and freezes inside a captured graph: captures use the two-pass form.
//  * No deadlock as long as the grid is resident at once (the host admits at most 4 blocks per CU: <= 128 VGPRs,
//    hardly any LDS): a block waits only for blocks of the same launch, which are running.  Should the device
//    be shared with something that keeps blocks from starting, the wait is bounded by time like the single pass's
//    (0xFFFFFFFF in d_counts; the synchronous entry points rerun the frame with the two-pass form).
// --------------------------------------------------------------------------
template <int DT, int QK, int PXT, bool VEC>
__global__ __launch_bounds__(kBlock) void k_compact_resident(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                             uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
                                                             uint8_t *state, CompactStats *stats, const Geom g, const QArg<QK> Q,
                                                             const uint32_t epoch) {
  using gu64 = __attribute__((address_space(1))) uint64_t;
  constexpr int CELLS = PXT * (kBlock / 64);
  __shared__ uint32_t s_cnt[CELLS];
  __shared__ uint32_t s_prefix;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  D2PC_DECLARE_STRIPS(VEC, wave);
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  const uint32_t t = blockIdx.x;
  const uint32_t f = fdiv(t, g.div_tpf);
  const uint32_t lt = t - f * g.tiles_per_frame;
  const uint32_t base = lt * uint32_t(kBlock * PXT);
  const FrameState fs(state, g, f);
  TileRegs<DT, QK, PXT> r;
  uint64_t mask[PXT];
  // (Publishing the count from the cheap validity predicate BEFORE computing the points -- so that the arithmetic
  // would run while the counts travel -- was slower: 9.6 -> 10.8 us at 752x480, 13.5 -> 14.3 us at 1080p.)
  tile_compute<DT, QK, PXT, VEC>(r, disp + uint64_t(f) * g.in_frame_stride, g, Q, base, wave, lane, wave_strip);
  tile_ballots<DT, QK, PXT>(r, g, base, wave, lane, mask);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < PXT; ++k) s_cnt[cell_index(k, wave)] = uint32_t(__popcll(mask[k]));
  }
  __syncthreads();
  uint32_t total;
  const uint32_t excl = scan_cells<CELLS>(s_cnt, lane, total);
  if (wave == 0) {
    if (lane == 0)
      __hip_atomic_store((gu64 *)(fs.granules + 2u * lt), (uint64_t(epoch) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t sum = 0, spins = 0;
    uint64_t w0 = 0;
    bool gave_up = false;
    for (;;) {
      bool ok = true;
      sum = 0;
      for (uint32_t i0 = 0; i0 < lt; i0 += 512u) {  // eight granules per lane and step, requested together
        uint64_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t i = i0 + uint32_t(j) * 64u + lane;
          v[j] = i < lt ? __hip_atomic_load((gu64 *)(fs.granules + 2u * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : (uint64_t(epoch) << 32);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ok = ok && uint32_t(v[j] >> 32) == epoch;
          sum += uint32_t(v[j]);
        }
      }
      if (__all(ok)) break;
      if (spins == 0) w0 = __builtin_amdgcn_s_memrealtime();
      backoff(spins);
      ++spins;
      if ((spins & 7u) == 0 && (__builtin_amdgcn_s_memrealtime() - w0 > uint64_t(g.spin_ticks) ||
                                __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)) {
        // the header's flag carries the epoch here (nothing zeroes it between launches)
        if (lane == 0 && __hip_atomic_exchange(&hdr->timeout, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
          atomicAdd(&stats->timeouts, 1ull);
        gave_up = true;
        break;
      }
    }
    sum = wave_sum(sum);
    if (lane == 0) {
      s_prefix = sum;
      if (lt == g.tiles_per_frame - 1u) {
        // a tile of this launch that gave up earlier (and scattered with a partial prefix) must not be papered over by
        // a last tile whose own timer had not run out yet: the flag carries the epoch of the launch that broke
        const bool broken = gave_up || __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
        __hip_atomic_store(counts + f, broken ? kCountTimedOut : sum + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (gave_up)
        __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
      if (spins) {
        CompactStats::Slot *sl = stats->slot + (blockIdx.x % uint32_t(kStatSlots));
        atomicAdd(&sl->failed_polls, (unsigned long long)spins);
        atomicAdd(&sl->wait_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - w0));
      }
      if (t == 0) {
        atomicAdd(&stats->launches, 1ull);
        atomicAdd(&stats->slot[0].tiles, (unsigned long long)g.total_tiles);
      }
#endif
    }
  }
  __syncthreads();
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
  tile_scatter<DT, QK, PXT>(r, mask, fout, fidx, s_prefix, excl, wave, lane, g.roi_n);
}

// W = a*d + b of a stereoRectify-structured Q decides validity without the point:
//   finite and |W| >= w_safe            => every coordinate is a finite float          -> valid
//   W zero, infinite or NaN (d = +-inf gives +-inf or NaN; no poisoning of d needed)    -> invalid
//   0 < |W| < w_safe, the "sliver"      => only the real arithmetic can tell (never seen with a real
//                                          calibration; a tile that holds one takes the exact path)
template <int QK>
__device__ __forceinline__ double stereo_nw(const QArg<QK> &A, float d) { return stereo_w(A, double(d)); }
__device__ __forceinline__ bool finite_nonzero(double x) {
  return __builtin_isfpclass(x, 0x0008 | 0x0010 | 0x0080 | 0x0100);  // -normal, -subnormal, +subnormal, +normal
}

// --------------------------------------------------------------------------
// K2R: the same one-launch form for frames of more than 1,024 ordinary tiles (one or two 4K frames): a block takes R
// pixels per thread -- 8,192 (R = 32) or 16,384 (R = 64) consecutive ROI pixels -- so that a 4K frame is 955 / 478 blocks
// and the whole launch is still resident at once (one 4K frame in COMPACT mode used to take two launches and two reads of
// its input: 41 us against 23 us PARITY).  The block's pixels stay in REGISTERS between the count and the scatter: only
// the disparities (R dwords per lane, one coalesced 256-byte piece per wave and load, all requested before the first is
// looked at); survivors are counted with the exact predicate (W = a*d + b for stereoRectify's Q, the real arithmetic for a
// wave that meets a sliver or a general Q), the block publishes ONE epoch-tagged granule, sums those of its
// predecessors in the frame (<= 1,023: all requested together), and then forms the points and stores them in order --
// a wave owns 64 * R consecutive pixels, so every store instruction is still one contiguous piece of <= 1 KiB.
// Epochs, time-outs and the residency rule are k_compact_resident's.
// --------------------------------------------------------------------------
// A value the compiler cannot relate to its source (no instruction is emitted): breaks common-subexpression reuse
// where recomputing is cheaper than keeping.
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ float opaque(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int DT, int QK, int R>
__global__ __launch_bounds__(kBlock) void k_compact_resident_lean(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                                  uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
                                                                  uint8_t *state, CompactStats *stats, const Geom g, const QArg<QK> Q,
                                                                  const uint32_t epoch) {
  using gu64 = __attribute__((address_space(1))) uint64_t;
  __shared__ uint32_t s_red[kBlock / 64];
  __shared__ uint32_t s_prefix;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  const uint32_t t = blockIdx.x;
  const uint32_t f = fdiv(t, g.div_tpf);
  const uint32_t lt = t - f * g.tiles_per_frame;
  const FrameState fs(state, g, f);
  const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
  const uint32_t i0 = lt * uint32_t(kBlock * R) + wave * uint32_t(64 * R) + lane;  // pixel k of the lane: i0 + 64 k
  // A RAMPED start: every block of the launch is resident and would ask for its pixels at once; the whole launch's input then
  // arrives together (~5.5 us for a 4K frame), everybody counts, publishes and looks back together (~3 us) and only then does the
  // first store leave -- read phase, bubble, write phase.  Block t waits t x (its bytes / the read rate) instead, so the data
  // arrive in block order at the rate memory delivers them anyway, the first blocks are storing while the last ones still load,
  // and the bubble is hidden (one 4K frame: profiles/r04_ab_resident.txt).
  for (uint32_t n = (t * g.stagger) >> 10; n > 0; --n) __builtin_amdgcn_s_sleep(1);
  float d[R];
  {
    // coordinates stepped from slot to slot (rows wrap inside the run): one division per thread
    Walker w(g, i0);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint32_t off = (w.v + g.border) * g.row_stride + (w.u + g.border) * elem_bytes<DT>();
      d[k] = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
      w.step(g, g.s64_v, g.s64_u);
      __builtin_amdgcn_sched_barrier(0);  // (compiler fence only: addresses are formed one load at a time, not R at once)
    }
  }
  // A frame's last block may reach past the ROI: those slots (their loads were clamped into the frame) become NaN, which
  // every predicate below drops -- one block-uniform branch instead of a range test per slot in both phases (tests that
  // depend on the lane only are hoisted and kept: 2 R scalar registers, spilled)
  if (lt == g.tiles_per_frame - 1u) {
#pragma unroll
    for (int k = 0; k < R; ++k) d[k] = i0 + uint32_t(k) * 64u < g.roi_n ? d[k] : __builtin_nanf("");
  }
  // ---- count ---- (per LANE, summed over the wave once: a ballot + popcount per slot left R masks waiting in scalar
  // registers -- ~400 of them spilled at R = 64)
  uint32_t cnt = 0;
  bool exact = !is_stereo(QK);
  if constexpr (is_stereo(QK)) {
    uint32_t sl = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double nw = stereo_nw(Q, d[k]);
      const bool fin_ = finite_nonzero(nw), big = fabs(nw) >= Q.s.w_safe;
      const bool keep = !(d[k] <= g.min_disparity);
      cnt += uint32_t(bool(fin_ & big & keep));  // (& on bools: no short-circuit branches)
      cnt = opaque(cnt);  // (the sum must advance slot by slot: reassociated into a tree, every slot's masks wait for the end)
      sl = opaque(sl | uint32_t(bool(fin_ & !big)));
    }
    exact = __ballot(sl != 0u) != 0;
  }
  if (exact) {  // (wave-uniform) general Q, or a sliver: the real arithmetic decides, as the scatter below does
    cnt = 0;
    Walker w(g, opaque(i0));
#pragma unroll
    for (int k = 0; k < R; ++k) {
      float X, Y, Z;
      reproject(Q, w.u + g.border, w.v + g.border, d[k], X, Y, Z);
      cnt += uint32_t(point_is_valid(X, Y, Z, d[k], g.min_disparity));
      cnt = opaque(cnt);
      w.step(g, g.s64_v, g.s64_u);
    }
  }
  const uint32_t total = wave_sum(cnt);  // wave-uniform
  if (lane == 0) s_red[wave] = total;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < uint32_t(kBlock / 64); ++w) {
    const uint32_t x = s_red[w];
    before += w < wave ? x : 0u;
    all += x;
  }
  // ---- publish, and the counts of the frame's blocks before this one ----
  if (wave == 0) {
    if (lane == 0)
      __hip_atomic_store((gu64 *)(fs.granules + 2u * lt), (uint64_t(epoch) << 32) | all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t sum = 0, spins = 0;
    uint64_t w0 = 0;
    bool gave_up = false;
    for (;;) {
      bool ok = true;
      sum = 0;
      for (uint32_t b0 = 0; b0 < lt; b0 += 512u) {  // eight granules per lane and step, requested together
        uint64_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t i = b0 + uint32_t(j) * 64u + lane;
          v[j] = i < lt ? __hip_atomic_load((gu64 *)(fs.granules + 2u * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : (uint64_t(epoch) << 32);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ok = ok && uint32_t(v[j] >> 32) == epoch;
          sum += uint32_t(v[j]);
        }
      }
      if (__all(ok)) break;
      if (spins == 0) w0 = __builtin_amdgcn_s_memrealtime();
      backoff(spins);
      ++spins;
      if ((spins & 7u) == 0 && (__builtin_amdgcn_s_memrealtime() - w0 > uint64_t(g.spin_ticks) ||
                                __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)) {
        if (lane == 0 && __hip_atomic_exchange(&hdr->timeout, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
          atomicAdd(&stats->timeouts, 1ull);
        gave_up = true;
        break;
      }
    }
    sum = wave_sum(sum);
    if (lane == 0) {
      s_prefix = sum;
      if (lt == g.tiles_per_frame - 1u) {
        const bool broken = gave_up || __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
        __hip_atomic_store(counts + f, broken ? kCountTimedOut : sum + all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (gave_up) {
        __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#if D2PC_ONEPASS_STATS
      if (spins) {
        CompactStats::Slot *sl = stats->slot + (blockIdx.x % uint32_t(kStatSlots));
        atomicAdd(&sl->failed_polls, (unsigned long long)spins);
        atomicAdd(&sl->wait_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - w0));
      }
      if (t == 0) {
        atomicAdd(&stats->launches, 1ull);
        atomicAdd(&stats->slot[0].tiles, (unsigned long long)g.total_tiles);
      }
#endif
    }
  }
  __syncthreads();
  // ---- the points, in order ----
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
  uint32_t pos = s_prefix + before;
  // (opaque: the coordinates are stepped AGAIN here; left to itself the compiler keeps the R coordinate pairs of the load
  // loop alive across the whole kernel instead -- 3 R registers per lane, 222 VGPRs at R = 64)
  // (the same for the disparities: sub-masks of the count's predicate -- d <= min_disparity, pixel < roi_n -- would be kept
  // for every slot: 2 R scalar register pairs, spilled)
  Walker w(g, opaque(i0));
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t uu = w.u + g.border, vv = w.v + g.border;
    const float dk = opaque(d[k]);
    float X, Y, Z;
    reproject(Q, uu, vv, dk, X, Y, Z);
    const bool ok = point_is_valid(X, Y, Z, dk, g.min_disparity);
    const uint64_t m = __ballot(ok);
    const uint32_t p = pos + mbcnt64(m);
    // p < roi_n always holds for a correct prefix; the guard keeps a timed-out prefix from becoming an out-of-bounds store
    if (ok && p < g.roi_n) {
      store_point<D2PC_RESIDENT_STORE_NT != 0>(fout, p, X, Y, Z);
      if (fidx) store_index(fidx, p, vv * g.width + uu);
    }
    pos += uint32_t(__popcll(m));
    w.step(g, g.s64_v, g.s64_u);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// --------------------------------------------------------------------------
// K2c: CHUNKED two-pass compaction (compact_algo 4) for big batches: one-shot blocks only, nothing waits inside a launch.
//
// The batch is cut into chunks of whole frames whose input fits the 256 MiB Infinity Cache (the host aims at <= ~100 MB),
// and launch i does two things at once, in ONE grid of short-lived blocks:
//   * SCATTER blocks (one per 512 ROI pixels of chunk i-1; the PARITY headline kernel's shape): every WAVE owns a run of
//     128 consecutive pixels, two per lane, and is on its own -- no LDS, no barrier.  Where its survivors go is known
//     when the wave starts: two scalar loads (the exclusive prefix of its group in the frame + of its run in the
//     group, left by launch i-1), then loads, points, two ballots, ranks, stores.  No ticket, no poll, no long-lived
//     block: the two things DESIGN section 9 blames for the single pass's distance from PARITY.  The disparities were
//     read by launch i-1's count blocks one launch ago and come from the Infinity Cache, not from HBM.
//   * COUNT blocks (one per group of 16,384 pixels = 128 runs of chunk i, every `period`-th block): the exact validity
//     predicate read 16 B per lane; per run the exclusive prefix inside the group, per group the total.  The block that
//     finishes a frame's LAST group (a counter per frame, bumped once per block; it resets itself) scans the frame's
//     group totals into exclusive prefixes and writes the frame's count.  Nothing in the same launch reads any of it,
//     so nothing ever waits: no deadlock whatever the dispatch order or residency, no time-outs.
//     Their HBM reads are the only reads of the launch that go to HBM: a launch moves the bytes of a PARITY launch.
// Launch 0 only counts (chunk 0, kept short by the host: one frame of a 4K stream), the last launch only scatters.
// Capturable: no epochs, no per-call zeroing: the frame counters and the group totals' "empty" marks are put back by the
// scanning block; k_chunk_clear sets them when a state buffer is taken over from another algorithm or batch shape.
// --------------------------------------------------------------------------
constexpr int kChunkS = 2;                                    // pixels per thread of a scatter block
constexpr uint32_t kChunkTile = uint32_t(kBlock) * kChunkS;   // 512 pixels per scatter block
constexpr uint32_t kChunkRun = 128;                           // pixels per scatter WAVE (a run)
constexpr uint32_t kChunkGroupShift = 7, kChunkGroupRuns = 1u << kChunkGroupShift;  // runs per count block (16,384 pixels)
constexpr uint32_t kChunkGroupTiles = kChunkGroupRuns * kChunkRun / kChunkTile;     // = 32 scatter tiles
constexpr uint32_t kChunkHdrWords = 4;                        // [0] = groups of the frame counted so far
constexpr uint32_t kChunkEmpty = 0xffffffffu;                 // a group total that has not been stored yet (a total is <= 16,384)

struct ChunkFrameState {
  uint32_t *done, *gsum, *gpre, *rp;  // frame counter; group totals; their exclusive prefixes; run prefixes inside the group
  __device__ __forceinline__ ChunkFrameState(uint8_t *state, const Geom &g, const ChunkArgs &c, uint32_t f) {
    done = reinterpret_cast<uint32_t *>(state + sizeof(StateHeader) + uint64_t(f) * g.frame_state_stride);
    gsum = done + kChunkHdrWords;
    gpre = gsum + c.gsum_words;
    rp = gpre + c.gsum_words;
  }
};

// validity of the pixel (image coordinates uu, vv; disparity d) exactly as the scatter blocks decide it
template <int QK>
__device__ __forceinline__ bool chunk_pixel_valid(const QArg<QK> &Q, const Geom &g, uint32_t uu, uint32_t vv, float d) {
  float X, Y, Z;
  reproject(Q, uu, vv, d, X, Y, Z);
  return point_is_valid(X, Y, Z, d, g.min_disparity);
}

// A wave's share of a count block: 32 runs (4,096 pixels).  Returns the wave's survivors; lane r < 32 leaves with the
// exclusive prefix of run r inside the wave.
template <int DT, int QK, bool VEC>
__device__ __forceinline__ uint32_t chunk_count_wave(const uint8_t *fin, const Geom &g, const QArg<QK> &Q, uint32_t run0,
                                                     uint32_t lane, uint32_t &mine) {
  constexpr uint32_t kWaveRuns = kChunkGroupRuns / 4u;  // 32
  uint32_t total = 0;                                   // wave-uniform
  mine = 0;
  // the real arithmetic pixel by pixel, `nruns` runs from run r0 on, in a rolled loop that reads the pixels itself: the
  // general Q's path, and the path of a wave that met a sliver (it keeps no register of the fast path alive)
  auto exact_runs = [&](uint32_t r0, uint32_t nruns) {
    for (uint32_t r = 0; r < nruns; ++r) {
      uint32_t ct = 0;
      for (uint32_t h = 0; h < 2u; ++h) {
        const uint32_t i = (run0 + r0 + r) * kChunkRun + h * 64u + lane;
        uint32_t uu, vv;
        pixel_coords(g, i, uu, vv);
        const uint32_t off = vv * g.row_stride + uu * elem_bytes<DT>();
        const float d = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
        ct += uint32_t(__popcll(__ballot(i < g.roi_n && chunk_pixel_valid<QK>(Q, g, uu, vv, d))));
      }
      mine = lane == r0 + r ? total : mine;
      total += ct;
    }
  };
  if constexpr (VEC && is_stereo(QK)) {
    // 1-KiB pieces, 16 B per lane (lanes 0-31: one run, lanes 32-63: the next): eight pieces = 16 runs requested
    // together, twice -- a real loop: one copy of the code, the pieces of one half in registers and nothing else across the
    // wait.  The kernel must keep the scatter blocks' register budget (<= 64 VGPRs: 8 waves per SIMD); forms of this loop
    // that kept coordinates or all the ballot masks alive took 127-139 VGPRs and spilled ~250 scalar registers, for
    // EVERY block of the launch.
    constexpr int NP = 8;                     // pieces per half
    const uint32_t last4 = g.last_off - 12u;  // the frame's last aligned group (tails load in bounds and count nothing)
#pragma nounroll
    for (uint32_t half = 0; half < 2u; ++half) {
      const uint32_t r0 = half * (kWaveRuns / 2u);
      const uint32_t i00 = (run0 + r0) * kChunkRun + lane * 4u;
      v4f q[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const uint32_t i0 = i00 + uint32_t(j) * 256u;
        const uint32_t v = fdiv(i0, g.div_roi_w);
        const uint32_t off = (v + g.border) * g.row_stride + (i0 - v * g.roi_w + g.border) * 4u;
        q[j] = ld(reinterpret_cast<const v4f *>(fin + (off < last4 ? off : last4)));
      }
      // W = a*d + b decides without the point (as tile_count does for the single pass); a sliver sends the half through
      // the real arithmetic instead.  Runs are accounted for piece by piece (no array of counts waits in scalar registers).
      const uint32_t total0 = total, mine0 = mine;
      uint64_t sliver = 0;
      // the frame's last group may reach past the ROI: those pixels (their loads were clamped into the frame) become NaN,
      // which the predicate drops -- a range test per pixel would be hoisted and its 32 masks kept in scalar registers
      if ((run0 + r0 + kWaveRuns / 2u) * kChunkRun > g.roi_n) {  // (wave-uniform)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int32_t left = int32_t(g.roi_n - (i00 + uint32_t(j) * 256u));  // the frame's pixels from this lane's group on (<= 0: none)
#pragma unroll
          for (int e = 0; e < 4; ++e) q[j][e] = e < left ? q[j][e] : __builtin_nanf("");
        }
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = q[j][e];
          const double nw = stereo_nw(Q, d);
          const bool fin_ = finite_nonzero(nw), big = fabs(nw) >= Q.s.w_safe;
          const bool keep = !(d <= g.min_disparity);
          const uint64_t m = __ballot(bool(fin_ & big & keep));  // (& on bools: no short-circuit branches)
          lo += uint32_t(__builtin_popcount(uint32_t(m)));
          hi += uint32_t(__builtin_popcount(uint32_t(m >> 32)));
          sliver |= __ballot(bool(fin_ & !big));
        }
        mine = lane == r0 + 2u * uint32_t(j) ? total : mine;
        total += lo;
        mine = lane == r0 + 2u * uint32_t(j) + 1u ? total : mine;
        total += hi;
        __builtin_amdgcn_sched_barrier(0);  // keeps the masks of one piece from piling up behind those of the next
      }
      if (sliver != 0) {  // (wave-uniform; never taken with a real calibration)
        total = total0;
        mine = mine0;
        exact_runs(r0, kWaveRuns / 2u);
      }
    }
  } else {
    exact_runs(0, kWaveRuns);
  }
  return total;
}

template <int DT, int QK, bool VEC>
__device__ __forceinline__ void chunk_count_block(const uint8_t *__restrict__ disp, uint32_t *__restrict__ counts,
                                                  uint8_t *state, const Geom &g, const QArg<QK> &Q, const ChunkArgs &c,
                                                  uint32_t cb, uint32_t *s_red) {
  using gu32 = __attribute__((address_space(1))) uint32_t;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t fl = fdiv(cb, c.div_gpf);
  const uint32_t grp = cb - fl * c.groups_per_frame;
  const uint32_t f = c.count_f0 + fl;
  const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
  const ChunkFrameState fs(state, g, c, f);
  constexpr uint32_t kWaveRuns = kChunkGroupRuns / 4u;
  const uint32_t run0 = (grp << kChunkGroupShift) + wave * kWaveRuns;
  uint32_t mine;
  const uint32_t total = chunk_count_wave<DT, QK, VEC>(fin, g, Q, run0, lane, mine);
  if (lane == 0) s_red[wave] = total;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4u; ++w) {
    const uint32_t x = s_red[w];
    before += w < wave ? x : 0u;
    all += x;
  }
  if (lane < kWaveRuns) fs.rp[run0 + lane] = before + mine;  // (the run area is padded to whole groups)
  if (tid == 0) {
    // The group's total, then the frame's counter: whoever counts the frame's last group scans the totals.  Relaxed
    // agent-scope atomics only (the other groups' blocks ran on other XCDs, whose L2s do not see each other's plain
    // stores in flight) -- NOT release/acquire: at agent scope those write back and invalidate the XCD's whole L2, once
    // per count block, under the scatter blocks' feet (a launch of 3 + 3 frames took 200 us instead of 118).  Instead
    // the data is its own flag: a total is never kChunkEmpty, the scanner puts kChunkEmpty back behind it.
    __hip_atomic_store((gu32 *)(fs.gsum + grp), all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t arrived = __hip_atomic_fetch_add((gu32 *)fs.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_red[4] = arrived == c.groups_per_frame - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (s_red[4] && wave == 0) {  // (block-uniform flag) one wave scans the frame's group totals
    // Every other block of the frame has ISSUED its total (it bumped the counter behind it); a total not visible yet
    // is a matter of the memory system's latency: looked at again, never waited for in any scheduling sense.
    uint32_t running = 0;
    bool lost = false;  // a total that never became visible (cannot happen unless a count block died): bounded, reported in-band
    for (uint32_t base = 0; base < c.groups_per_frame; base += 512u) {  // eight totals per lane, requested together
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t h = base + uint32_t(k) * 64u + lane;
        v[k] = h < c.groups_per_frame ? __hip_atomic_load((gu32 *)(fs.gsum + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t h = base + uint32_t(k) * 64u + lane;
        for (uint32_t tries = 0; v[k] == kChunkEmpty; ++tries) {
          if (tries == (1u << 22)) {  // (~1 s of looking: every launch must end)
            lost = true;
            v[k] = 0u;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          v[k] = __hip_atomic_load((gu32 *)(fs.gsum + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t incl = v[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t n = __shfl_up(incl, o, 64);
          if (lane >= uint32_t(o)) incl += n;
        }
        if (h < c.groups_per_frame) {
          fs.gpre[h] = running + incl - v[k];  // plain store: read by the NEXT launch
          __hip_atomic_store((gu32 *)(fs.gsum + h), kChunkEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
        }
        running += __builtin_amdgcn_readlane(incl, 63);
      }
    }
    lost = __ballot(lost) != 0;
    if (lane == 0) {
      if (counts) counts[f] = lost ? kCountTimedOut : running;
      __hip_atomic_store((gu32 *)fs.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One wave of a scatter block: a run of 128 consecutive ROI pixels, two per lane, on its own.
template <int DT, int QK>
__device__ __forceinline__ void chunk_scatter_wave(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                   uint32_t *__restrict__ out_index, uint8_t *state, const Geom &g,
                                                   const QArg<QK> &Q, const ChunkArgs &c, uint32_t tile) {
  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t fl = fdiv(tile, g.div_tpf);
  const uint32_t lt = tile - fl * g.tiles_per_frame;
  const uint32_t f = c.scatter_f0 + fl;
  const uint32_t run = lt * (kChunkTile / kChunkRun) + wave;  // (wave-uniform: scalar registers)
  const uint32_t base = run * kChunkRun + lane;
  if (run * kChunkRun >= g.roi_n) return;  // a frame's last block may have waves past the ROI
  const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
  const ChunkFrameState fs(state, g, c, f);
  // where the run's survivors go: two scalar loads, requested ahead of the disparities
  const uint32_t prefix = fs.gpre[run >> kChunkGroupShift] + fs.rp[run];
  float d[kChunkS];
  uint32_t uu[kChunkS], vv[kChunkS];
#pragma unroll
  for (int k = 0; k < kChunkS; ++k) {
    pixel_coords(g, base + uint32_t(k) * 64u, uu[k], vv[k]);
    const uint32_t off = vv[k] * g.row_stride + uu[k] * elem_bytes<DT>();
    d[k] = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
  }
  uint32_t pos = prefix;
#pragma unroll
  for (int k = 0; k < kChunkS; ++k) {
    float X, Y, Z;
    reproject(Q, uu[k], vv[k], d[k], X, Y, Z);
    const bool ok = base + uint32_t(k) * 64u < g.roi_n && point_is_valid(X, Y, Z, d[k], g.min_disparity);
    const uint64_t m = __ballot(ok);
    const uint32_t p = pos + mbcnt64(m);
    // p < roi_n always holds; the guard keeps a count that is not this call's from becoming an out-of-bounds store
    if (ok && p < g.roi_n) {
      store_point<D2PC_CHUNK_STORE_NT != 0>(fout, p, X, Y, Z);
      if (fidx) st<D2PC_CHUNK_INDEX_NT != 0>(fidx + p, vv[k] * g.width + uu[k]);
    }
    pos += uint32_t(__popcll(m));
  }
}

template <int DT, int QK, bool VEC>
__global__ __launch_bounds__(kBlock) void k_compact_chunk(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                          uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
                                                          uint8_t *state, const Geom g, const QArg<QK> Q, const ChunkArgs c) {
  __shared__ uint32_t s_red[5];
  const uint32_t b = blockIdx.x;
  const uint32_t q = fdiv(b, c.div_period), rem = b - q * c.period;
  if (rem == 0 && q < c.count_blocks) {  // (block-uniform)
    chunk_count_block<DT, QK, VEC>(disp, counts, state, g, Q, c, q, s_red);
    return;
  }
  uint32_t ahead = rem ? q + 1u : q;  // count blocks at positions below b
  if (ahead > c.count_blocks) ahead = c.count_blocks;
  chunk_scatter_wave<DT, QK>(disp, out, out_index, state, g, Q, c, b - ahead);
}

// --------------------------------------------------------------------------
// K2: single-pass compaction (each disparity is read once).
//  * A block serves ONE frame at a time (frame = blockIdx % n_frames) and
//    takes that frame's tiles from the frame's own ticket counter: every
//    predecessor of a tile is already running (or done) when the tile starts,
//    so waiting for predecessors' COUNTS cannot deadlock whatever the
//    dispatch order or residency.
//  * A tile publishes its count as soon as it is known and only needs the
//    counts of its predecessors -- no scan ripples through the frame.
//  * 5 waves per block: four WORKER waves stream pixels; one CONTROL wave
//    owns the protocol (ticket atomics, publishing, polling), so the workers
//    never sit behind an atomic's round trip.
//  * Software pipeline over a block's tiles, holding only DISPARITIES in
//    registers: iteration i COUNTS tile t (exact validity predicate, a few
//    operations per pixel for a stereoRectify-structured Q), the control wave
//    publishes t, fetches the ticket of t+1 and waits for the prefix of t-1
//    (published an iteration ago, so normally ready at the first look); then
//    t+1's loads are issued and tile t-1 is reprojected and scattered.
// --------------------------------------------------------------------------
constexpr uint32_t kNoTile = 0xffffffffu;

// ---- single-pass building blocks: validity and points of one wave's share of a tile -------------------

// Exact validity of pixel i by the real arithmetic (general Q, and tiles with a sliver).
template <int QK>
__device__ __forceinline__ bool pixel_valid_exact(const QArg<QK> &Q, const Geom &g, uint32_t i, float d) {
  uint32_t uu, vv;
  pixel_coords(g, i, uu, vv);
  float X, Y, Z;
  reproject(Q, uu, vv, d, X, Y, Z);
  return point_is_valid(X, Y, Z, d, g.min_disparity);
}

// Count phase: per-slot survivor counts of this wave's pixels of the tile at `base`.  Returns whether the
// tile needs the exact path (the scatter phase two iterations later must then take it as well, so that
// both phases decide every pixel identically).
template <int QK, int PXT>
__device__ __forceinline__ bool tile_count(const QArg<QK> &Q, const Geom &g, const float (&d)[PXT], uint32_t base,
                                           uint32_t wave, uint32_t lane, uint32_t (&cnt)[PXT]) {
  const uint32_t i0 = base + wave * 256u + lane;
  const uint32_t lim = base + uint32_t(kBlock * PXT) > g.roi_n ? g.roi_n : 0xffffffffu;  // ragged: a frame's last tile
  bool exact = !is_stereo(QK);
  if constexpr (is_stereo(QK)) {
    uint64_t sliver = 0;
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
      const uint32_t i = i0 + uint32_t(k >> 2) * 1024u + uint32_t(k & 3) * 64u;
      const double nw = stereo_nw(Q, d[k]);
      const bool fin = finite_nonzero(nw), big = fabs(nw) >= Q.s.w_safe;
      cnt[k] = uint32_t(__popcll(__ballot(int(fin) & int(big) & int(!(d[k] <= g.min_disparity)) & int(i < lim))));
      sliver |= __ballot(int(fin) & int(!big));
    }
    exact = sliver != 0;
  }
  if (exact) {
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
      const uint32_t i = i0 + uint32_t(k >> 2) * 1024u + uint32_t(k & 3) * 64u;
      cnt[k] = uint32_t(__popcll(__ballot(pixel_valid_exact<QK>(Q, g, i, d[k]) && i < lim)));
    }
  }
  return exact;
}

// Scatter phase: the same decisions, the points, and their ordered stores.  prefix + cell_excl[cell] = output
// position of the first survivor of a slot (frame prefix + the cell's exclusive offset inside the tile).
// A wave-uniform pointer moved into vector registers: the single-pass kernel runs out of scalar registers,
// and a spilled scalar base costs a v_readlane pair before every store.  With the base in VGPRs the store
// address is one v_lshl_add_u64.
__device__ __forceinline__ uint64_t vgpr_pointer(const void *p) {
  const uint64_t x = reinterpret_cast<uint64_t>(p);
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=v"(hi) : "s"(uint32_t(x)), "s"(uint32_t(x >> 32)));
  return (uint64_t(hi) << 32) | lo;
}

template <int QK, int PXT, bool EXACT, bool IDX>
__device__ __forceinline__ void tile_scatter_lean(const QArg<QK> &Q, const Geom &g, const float (&d)[PXT],
                                                  const uint32_t *cell_excl, uint32_t prefix, uint32_t base,
                                                  uint32_t wave, uint32_t lane, uint64_t fout, uint64_t fidx) {
  const uint32_t i0 = base + wave * 256u + lane;
  const uint32_t lim = base + uint32_t(kBlock * PXT) > g.roi_n ? g.roi_n : 0xffffffffu;
  // image coordinates of the wave's slots first (stepped from slot to slot; rows wrap inside a tile): the
  // stepping constants are then dead in the arithmetic below, which is short of scalar registers
  uint32_t uus[PXT], vvs[PXT];
  tile_coords<PXT>(uus, vvs, g, base, wave, lane);
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const uint32_t uu = uus[k], vv = vvs[k];
    const uint32_t i = i0 + uint32_t(k >> 2) * 1024u + uint32_t(k & 3) * 64u;
    bool ok;
    double nw = 0.0;
    float X, Y, Z;
    if constexpr (is_stereo(QK) && !EXACT) {
      nw = stereo_nw(Q, d[k]);
      ok = int(finite_nonzero(nw)) & int(fabs(nw) >= Q.s.w_safe) & int(!(d[k] <= g.min_disparity)) & int(i < lim);
    } else {
      reproject(Q, uu, vv, d[k], X, Y, Z);
      ok = point_is_valid(X, Y, Z, d[k], g.min_disparity) && i < lim;
    }
    const uint64_t m = __ballot(ok);
    if (m != 0) {  // whole slots of holes (blocky invalid regions) skip the arithmetic and the stores
      if constexpr (is_stereo(QK) && !EXACT) {
        const double iw = 1.0 / nw;
        X = float(stereo_nx(Q, uu) * iw);
        Y = float(stereo_ny(Q, vv) * iw);
        Z = big_z_rule(d[k], float(Q.s.f * iw));
      }
      // rank among the slot's survivors, accumulated onto the cell's base in the same two instructions
      const uint32_t cell_base = prefix + cell_excl[cell_index(k, wave)];  // (LDS broadcast read)
      const uint32_t pos = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), cell_base));
      // pos < roi_n always holds for a correct prefix; the guard keeps a stale or timed-out prefix from
      // ever becoming an out-of-bounds store
      if (ok && pos < g.roi_n) {
        using gv4f = __attribute__((address_space(1))) v4f;
        using gu32 = __attribute__((address_space(1))) uint32_t;
        const v4f p = {X, Y, Z, 1.0f};
        if (D2PC_ONEPASS_STORE_NT) __builtin_nontemporal_store(p, (gv4f *)(fout + (uint64_t(pos) << 4)));
        else *(gv4f *)(fout + (uint64_t(pos) << 4)) = p;
        if constexpr (IDX) {
          if (D2PC_ONEPASS_INDEX_NT) __builtin_nontemporal_store(vv * g.width + uu, (gu32 *)(fidx + (uint64_t(pos) << 2)));
          else *(gu32 *)(fidx + (uint64_t(pos) << 2)) = vv * g.width + uu;
        }
      }
    }
  }
}

// A worker wave's disparities of one tile in flight: the raw 16-byte row pieces (VEC) or the decoded
// slot values.  Issue and finish are separate so that the loads fly across the scatter of an older tile
// and the block barriers; finish() turns the pieces into the slot layout through the wave's LDS strip.
template <int DT, int PXT, bool VEC>
struct TileFetch {
  v4f q[VEC ? PXT / 4 : 1];
  float d[VEC ? 1 : PXT];
  __device__ __forceinline__ void issue(const uint8_t *fin, const Geom &g, uint32_t base, uint32_t wave, uint32_t lane) {
    if constexpr (VEC) {
      Walker w4(g, base + wave * 256u + lane * 4u);
#pragma unroll
      for (int j = 0; j < PXT / 4; ++j) {
        const uint32_t off = (w4.v + g.border) * g.row_stride + (w4.u + g.border) * 4u;
        const uint32_t last4 = g.last_off - 12u;  // the frame's last aligned group
        q[j] = ld(reinterpret_cast<const v4f *>(fin + (off < last4 ? off : last4)));
        w4.step(g, g.s1024_v, g.s1024_u);
      }
    } else {
      tile_load_d<DT, PXT, false>(d, fin, g, base, wave, lane, nullptr);
    }
  }
  // Lands the tile in the wave's own part of an LDS stage, pixel-linear: PXT/4 pieces of 256 floats.  The
  // stage IS the pipeline storage: count and scatter phases read their slot values from it (slot k of lane L =
  // piece k/4, element (k%4)*64 + L), so no tile lives in registers across iterations.
  __device__ __forceinline__ void finish(float *stage, uint32_t lane) const {
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < PXT / 4; ++j) *reinterpret_cast<v4f *>(stage + uint32_t(j) * 256u + lane * 4u) = q[j];
    } else {
#pragma unroll
      for (int k = 0; k < PXT; ++k) stage[uint32_t(k >> 2) * 256u + uint32_t(k & 3) * 64u + lane] = d[k];
    }
  }
};

// A wave's slot values of a staged tile (LDS is in-order per wave: the wave's own earlier writes are visible).
template <int PXT>
__device__ __forceinline__ void stage_read(float (&d)[PXT], const float *stage, uint32_t lane) {
#pragma unroll
  for (int k = 0; k < PXT; ++k) d[k] = stage[uint32_t(k >> 2) * 256u + uint32_t(k & 3) * 64u + lane];
}

template <int DT, int QK, int PXT, bool VEC>
__global__ __launch_bounds__(kBlock + 64) void k_compact_onepass(const uint8_t *__restrict__ disp,
                                                                 float4 *__restrict__ out,
                                                                 uint32_t *__restrict__ out_index,
                                                                 uint32_t *__restrict__ counts, uint8_t *state,
                                                                 const Geom g, const QArg<QK> Q) {
  using gu64 = __attribute__((address_space(1))) uint64_t;
  constexpr int CELLS = PXT * (kBlock / 64);
  constexpr uint32_t TILE = uint32_t(kBlock * PXT);
  __shared__ uint32_t s_cnt[CELLS];
  // per-cell exclusive offsets and totals of the last three counted tiles: a tile counted in iteration
  // `it` is scattered in iteration it + 2, so its offsets stay in LDS instead of in registers
  __shared__ uint32_t s_excl[3][CELLS];
  __shared__ uint32_t s_total[3], s_next[2], s_prefix[2];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool ctl = wave == kBlock / 64;  // the fifth wave
  // the disparities of the four tiles a block has in flight (being fetched / counted / waiting / scattered):
  // 4 stages x 4 worker waves x PXT/4 pieces x 1 KiB; every wave touches its own part only (no barrier)
  constexpr uint32_t kWaveStage = uint32_t(PXT / 4) * 256u, kStage = (kBlock / 64) * kWaveStage;
  __shared__ float s_tile[4 * kStage];
  float *const my_tile = s_tile + (wave < kBlock / 64 ? wave : 0u) * kWaveStage;
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  __shared__ uint32_t s_stat[3];  // control wave, lane 0: tiles served, failed polls, wait ticks (PollStats)
  PollStats polls{s_stat};
  if (tid < 3) s_stat[tid] = 0;  // (ordered before the control wave's first use by the barrier below)
#ifdef D2PC_DIAG
  // phase timers (shader clock), lane 0 of worker wave 0 and of the control wave; named
  // scalars on purpose: a runtime-indexed array would live in scratch and distort the run
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, nIt = 0;
#define D2PC_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define D2PC_STAMP(x)
#endif

  {  // a block serves ONE frame (the launcher sizes the grid to a multiple of n_frames)
    const uint32_t f = blockIdx.x % g.n_frames;
    const FrameState fs(state, g, f);
    const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
    float4 *fout = out + uint64_t(f) * g.out_frame_stride;
    uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;

    if (ctl && lane == 0) s_next[1] = atomicAdd(fs.ticket, 1u);
    __syncthreads();
    uint32_t cur = s_next[1];
    if (cur >= g.tiles_per_frame) cur = kNoTile;
    uint32_t prev = kNoTile;   // counted in the previous iteration
    uint32_t prev2 = kNoTile;  // counted two iterations ago: scattered now.  The lag gives every
                               // predecessor a whole extra iteration to publish before it is polled
    KnownGroups known;  // control wave: prefix of the frame's complete groups seen so far

    // A worker's whole pipeline state: the disparities of the three tiles in flight.  Validity is
    // re-derived in the scatter phase by the same arithmetic (no wave masks kept in scalar registers --
    // 48 of them spilled in the round-1 form), cell offsets wait in LDS.
    TileFetch<DT, PXT, VEC> fetch;     // tile `next`: issued as soon as its ticket is known (after barrier 1),
                                       // landed in its LDS stage after barrier 2
    bool cexact = false, pexact = false, qexact = false;  // does the tile take the exact path (see tile_count)
    const uint64_t vout = vgpr_pointer(fout), vidx = vgpr_pointer(fidx);
    if (!ctl && cur != kNoTile) {
      fetch.issue(fin, g, cur * TILE, wave, lane);
      fetch.finish(my_tile, lane);  // iteration 0 counts stage 0
    }

    for (uint32_t it = 0; cur != kNoTile || prev != kNoTile || prev2 != kNoTile; ++it) {
      const uint32_t slot = it & 1u;
      const uint32_t ring = it % 3u, ring2 = (it + 1u) % 3u;  // this iteration's tile / the tile two iterations back
      D2PC_STAMP(c0);
      if (ctl) {
        // ticket of the tile after `cur` and the prefix of `prev2`: the atomic's round trip (2-3 us under a
        // saturating write stream) runs under the polls, its result is only needed at the barrier
        uint32_t tk = 0;
        if (cur != kNoTile && lane == 0) tk = atomicAdd(fs.ticket, 1u);
        if (prev2 != kNoTile) {
          const uint32_t p = prefix_before<true>(fs, hdr, prev2, lane, polls, known, g.spin_ticks);
          if (lane == 0) s_prefix[slot] = p;
        }
        if (cur != kNoTile && lane == 0) s_next[slot] = tk;
#if D2PC_ONEPASS_STATS
        if (cur != kNoTile && lane == 0) s_stat[0] += 1u;
#endif
      } else if (cur != kNoTile) {
        float dc[PXT];
        stage_read<PXT>(dc, my_tile + (it & 3u) * kStage, lane);
        uint32_t cnt[PXT];
        cexact = tile_count<QK, PXT>(Q, g, dc, cur * TILE, wave, lane, cnt);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < PXT; ++k) s_cnt[cell_index(k, wave)] = cnt[k];
        }
      }
      D2PC_STAMP(c1);
      __syncthreads();
      D2PC_STAMP(c2);
      uint32_t next = kNoTile;
      if (cur != kNoTile) {
        next = s_next[slot];
        if (next >= g.tiles_per_frame) next = kNoTile;
      }
      // the next tile's loads go out the moment its ticket is known; they fly while the control wave scans
      // and publishes
      if (!ctl && next != kNoTile) fetch.issue(fin, g, next * TILE, wave, lane);
      if (ctl && cur != kNoTile) {
        uint32_t total;
        const uint32_t excl = scan_cells<CELLS>(s_cnt, lane, total);
        if (lane < uint32_t(CELLS)) s_excl[ring][lane] = excl;
        if (lane == 0) {
          s_total[ring] = total;
          // publish: tagged granule (the data is the flag) + group accumulator
          __hip_atomic_store((gu64 *)(fs.granules + 2u * cur), kGranuleTag | total, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gu64 *)fs.group_word(cur / kGroupTiles), (uint64_t(1) << 32) | total,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      D2PC_STAMP(c3);
      if (!ctl) {
        // the tile counted in iteration it + 1 lands in stage (it + 1) % 4, whose previous tenant (counted in
        // iteration it - 3) was scattered an iteration ago by this same wave
        if (next != kNoTile) fetch.finish(my_tile + ((it + 1u) & 3u) * kStage, lane);
        if (prev2 != kNoTile) {
          float dq[PXT];
          stage_read<PXT>(dq, my_tile + ((it + 2u) & 3u) * kStage, lane);  // counted in iteration it - 2
          const uint32_t prefix = s_prefix[slot];
          const uint32_t *cell_base = s_excl[ring2];
          if (qexact) {  // (a tile with a sliver, or a general Q: rare / not the calibrated case -- one code copy)
            if (fidx) tile_scatter_lean<QK, PXT, true, true>(Q, g, dq, cell_base, prefix, prev2 * TILE, wave, lane, vout, vidx);
            else tile_scatter_lean<QK, PXT, true, false>(Q, g, dq, cell_base, prefix, prev2 * TILE, wave, lane, vout, vidx);
          } else if (fidx) {
            tile_scatter_lean<QK, PXT, false, true>(Q, g, dq, cell_base, prefix, prev2 * TILE, wave, lane, vout, vidx);
          } else {
            tile_scatter_lean<QK, PXT, false, false>(Q, g, dq, cell_base, prefix, prev2 * TILE, wave, lane, vout, vidx);
          }
          if (counts && prev2 == g.tiles_per_frame - 1 && tid == 0) {
            // a frame whose hand-off broke reports kCountTimedOut instead of a count: visible in-band
            const bool bad = __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            __hip_atomic_store(counts + f, bad ? kCountTimedOut : prefix + s_total[ring2], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        qexact = pexact;
        pexact = cexact;
      }
#ifdef D2PC_DIAG
      {
        D2PC_STAMP(c4);
        tA += c1 - c0;  // worker: count phase            | control: ticket + prefix
        tB += c2 - c1;  // waiting at barrier 1 for the other side
        tC += c3 - c2;  // worker: waits for scan/publish | control: scan + publish (+ barrier 2)
        tD += c4 - c3;  // worker: next loads + reproject + scatter
        ++nIt;
      }
#endif
      prev2 = prev;
      prev = cur;
      cur = next;
    }
    __syncthreads();
    // a block that saw the launch break marks the frame it was serving (it may have scattered with a
    // prefix it never obtained), whether or not the frame's last tile has reported its count already
    if (counts && tid == 0 && __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
    if (ctl && lane == 0) {  // the block's share of the context's counters: no-return atomics, once per block, on the
                             // block's slot (one word for all blocks serialised the launch's end: d2pc_device.hpp)
      CompactStats::Slot *sl = hdr->stats->slot + (blockIdx.x % uint32_t(kStatSlots));
      atomicAdd(&sl->tiles, (unsigned long long)s_stat[0]);
      if (s_stat[1]) {
        atomicAdd(&sl->failed_polls, (unsigned long long)s_stat[1]);
        atomicAdd(&sl->wait_ticks, (unsigned long long)s_stat[2]);
      }
    }
#endif
  }
#ifdef D2PC_DIAG
  if (lane == 0 && wave == 0) {
    atomicAdd(&hdr->diag[0], nIt);
    atomicAdd(&hdr->diag[2], tA);
    atomicAdd(&hdr->diag[3], tB);
    atomicAdd(&hdr->diag[4], tC);
    atomicAdd(&hdr->diag[5], tD);
  }
  if (lane == 0 && ctl) {
    atomicAdd(&hdr->diag[1], (unsigned long long)s_stat[1]);
    atomicAdd(&hdr->diag[6], tA);  // control: ticket + prefix
  }
#endif
#undef D2PC_STAMP
}

// --------------------------------------------------------------------------
// K1c: the callback body TILE BY TILE in COMPACT mode -- bit-sliced median of a 256 x 32 tile, then the tile's
// SURVIVING points, in the CPU loop's row-major order (cpp:70-76 + the north-star's validity compaction), in one
// kernel.  The two-launch form (filter launch, filtered frames through memory, compaction launch) stays as the
// fallback and as the device-side oracle.
//
// Order.  A tile holds 32 rows of 256 columns; in the output, row y of tile (band ty, column tx) follows row y of
// the tile to its left and precedes row y of the tile to its right, so the position of the first survivor of a row is
//     S(ty)                     survivors of all bands above            (band accumulators, counted like the
//                                                                        single pass's group accumulators)
//   + sum of the band's rows above y over ALL its tiles                  (the 32 row counts every tile of the
//   + sum of row y over the band's tiles to the left                      band publishes: 64 bytes per tile)
// A tile therefore needs every tile of ITS BAND (left and right) and the totals of all bands above.
//
// Hand-off.  Tiles are handed out by a per-frame ticket, band by band, left to right (a block serves frame
// blockIdx % n_frames; exactly tiles_per_frame blocks per frame), so the tiles of a band hold consecutive tickets.
// A block publishes its row counts as soon as the filter is done -- sixteen tagged dwords (the data is the flag:
// two 9-bit counts and a tag bit each, one sc1 store instruction, nothing to drain) plus ONE agent-scope add of
// {1, tile total} to the band's accumulator -- and waits until every tile of its band has published and all bands
// above are complete; the two kinds of words are polled in the same pass, so a wait that finds everything ready
// costs one memory round trip.  No
// deadlock at any residency >= tiles_x blocks (the host refuses wider frames): every ticket below the highest one
// issued is held by a running block; a band whose tickets are all issued completes because its tiles wait only for
// bands that are all issued (induction from band 0); the blocks that retire then take the remaining tickets of
// the one band that may be partly issued.  Waits are bounded by time like the single pass's (sticky flag,
// 0xFFFFFFFF in d_counts).
// --------------------------------------------------------------------------
constexpr uint32_t kCbRowTag = 1u << 31;
struct CbCompactState {
  uint32_t *ticket;
  uint64_t *band_acc;  // (tiles arrived << 32) | survivors, one per band, packed
  uint32_t *row_cnt;   // [tile][16]: dword p = kCbRowTag | survivors of the tile's rows 2p | 2p+1 << 9 (a count is <= 256)
  __device__ __forceinline__ CbCompactState(uint8_t *state, const Geom &g, uint32_t f, uint32_t tiles_y) {
    uint8_t *fs = state + sizeof(StateHeader) + uint64_t(f) * g.frame_state_stride;
    ticket = reinterpret_cast<uint32_t *>(fs);
    band_acc = reinterpret_cast<uint64_t *>(fs + kCbTicketBytes);
    row_cnt = reinterpret_cast<uint32_t *>(fs + kCbTicketBytes + cb_band_acc_bytes(tiles_y));
  }
};

template <int KS, int QK>
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_callback_bs_compact(
    const uint8_t *__restrict__ src, float4 *__restrict__ out, uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
    uint8_t *state, const MedianArgs ma, const Geom g, const QArg<QK> Q) {
  using S = MedianBsShape<KS>;
  using gu32 = __attribute__((address_space(1))) uint32_t;
  using gu64 = __attribute__((address_space(1))) uint64_t;
  static_assert(S::THREADS == 256 && S::TH == 32, "one thread per byte value fills the table; 32 row counts per tile");
  __shared__ __attribute__((aligned(16))) uint32_t s_w[S::W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_raw[S::RAW_WORDS];
  __shared__ uint32_t s_tile, s_exact;
  __shared__ uint32_t s_cnt[32], s_base[32];
  __shared__ uint32_t s_stat[3];
  // per byte value: 1/W, Z and the validity class of the point (0 dropped, 1 kept, 2 = only the arithmetic can tell).
  // Tables of their own (not in the staged rows' space as in k_callback_bs): they are filled while the ticket's
  // atomic is in flight
  __shared__ double lut_iw[256];
  __shared__ float lut_z[256];
  __shared__ uint8_t lut_cls[256];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  const uint32_t f = blockIdx.x % g.n_frames;  // the grid is tiles_per_frame * n_frames: every frame gets its tiles' worth of blocks
  const CbCompactState cs(state, g, f, ma.tiles_y);
  uint32_t tk = 0;
  if (tid == 0) {
    tk = atomicAdd(cs.ticket, 1u);  // its round trip (2-3 us under load) runs under the table's divisions
    s_exact = !is_stereo(QK) ? 1u : 0u;
  }
  if (tid < 3) s_stat[tid] = 0;
  if constexpr (is_stereo(QK)) {
    const float d = __fmul_rn(float(tid), g.scale);  // cpp:61, as load_disparity<DT_U8>
    const float dsel = fabsf(d) < __builtin_huge_valf() ? d : __builtin_nanf("");
    const double nw = stereo_w(Q, double(dsel));
    const double iw = 1.0 / nw;
    lut_iw[tid] = iw;
    lut_z[tid] = big_z_rule(d, float(Q.s.f * iw));
    // the single pass's predicate (tile_count): finite, non-zero W of at least w_safe => every coordinate finite
    const bool fin = finite_nonzero(nw), big = fabs(nw) >= Q.s.w_safe, keep = !(d <= g.min_disparity);
    const uint32_t cls = fin && keep ? (big ? 1u : 2u) : 0u;
    lut_cls[tid] = uint8_t(cls);
    __syncthreads();                  // (s_exact's initial value is in place)
    if (cls == 2u) s_exact = 1u;      // (benign race: every writer stores 1)
  }
  if (tid == 0) s_tile = tk;
  __syncthreads();
  const uint32_t lt = s_tile;  // < tiles_x * tiles_y: as many tickets as blocks
  const uint32_t ty = lt / ma.tiles_x, tx = lt - ty * ma.tiles_x;
  const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);
  median_bs_tile<KS>(src + uint64_t(f) * ma.src_frame_stride, ma, int(x0), int(y0), s_w, s_raw, tid);
  const bool exact = s_exact != 0;  // block-uniform: both phases below decide every pixel the same way

  const uint8_t *ob = reinterpret_cast<const uint8_t *>(s_w);
  const uint32_t x_end = ma.out_x0 + ma.out_w, y_end = ma.out_y0 + ma.out_h;
  double xs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xs[q] = 0.0;
    if constexpr (is_stereo(QK)) xs[q] = stereo_nx(Q, x0 + 64u * uint32_t(q) + lane);
  }
  // one pixel: its point (when wanted) and whether it survives
  auto pixel = [&](uint32_t x, uint32_t y, uint32_t raw, int q, double ys, bool want_point, float &X, float &Y, float &Z) -> bool {
    if constexpr (is_stereo(QK)) {
      if (!exact && !want_point) return lut_cls[raw] == 1u;
      const double iw = lut_iw[raw];
      X = float(xs[q] * iw);
      Y = float(ys * iw);
      Z = lut_z[raw];
      if (!exact) return lut_cls[raw] == 1u;
      return point_is_valid(X, Y, Z, __fmul_rn(float(raw), g.scale), g.min_disparity);
    } else {
      const float d = __fmul_rn(float(raw), g.scale);
      reproject(Q, x, y, d, X, Y, Z);
      return point_is_valid(X, Y, Z, d, g.min_disparity);
    }
  };

  // ---- count: survivors per row of the tile (a wave takes every fourth row, 64 columns per step) ----------
#pragma unroll 1
  for (uint32_t r = wave; r < uint32_t(S::TH); r += uint32_t(S::THREADS / 64)) {
    const uint32_t y = y0 + r;
    uint32_t cnt = 0;
    if (y < y_end) {
      double ys = 0.0;
      if constexpr (is_stereo(QK)) ys = stereo_ny(Q, y);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t x = x0 + 64u * uint32_t(q) + lane;
        const uint32_t raw = ob[r * uint32_t(S::OUT_STRIDE) + 64u * uint32_t(q) + lane];
        float X, Y, Z;
        const bool ok = pixel(x, y, raw, q, ys, false, X, Y, Z) && x < x_end;
        cnt += uint32_t(__popcll(__ballot(ok)));
      }
    }
    if (lane == 0) s_cnt[r] = cnt;
  }
  __syncthreads();

  // ---- hand-off (wave 0): publish the 32 row counts, wait for the band and the bands above, place the rows ----
  if (wave == 0) {
    const uint32_t mine = lane < 32u ? s_cnt[lane] : 0u;
    const uint32_t tile_total = wave_sum(mine);
    if (lane < 16u)
      __hip_atomic_store((gu32 *)(cs.row_cnt + lt * 16u + lane), kCbRowTag | s_cnt[2u * lane] | (s_cnt[2u * lane + 1u] << 9),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0)
      __hip_atomic_fetch_add((gu64 *)(cs.band_acc + ty), (uint64_t(1) << 32) | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // One pass = the accumulators of the bands above (64 per step) and the band's row-count words (lane = (j, p):
    // tiles j, j + 4, ..., row pair p), all requested together; a pass in which every word is complete ends the wait.
    const uint32_t j = lane >> 4, p = lane & 15u;
    uint32_t above = 0, t0 = 0, t1 = 0, l0 = 0, l1 = 0, spins = 0;
    uint64_t w0 = 0;
    for (;;) {
      bool ok = true;
      above = 0;
      for (uint32_t b0 = 0; b0 < ty; b0 += 64u) {
        const uint32_t bi = b0 + lane;
        if (bi < ty) {
          const uint64_t v = __hip_atomic_load((gu64 *)(cs.band_acc + bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && uint32_t(v >> 32) == ma.tiles_x;
          above += uint32_t(v);
        }
      }
      t0 = t1 = l0 = l1 = 0;
      for (uint32_t k = j; k < ma.tiles_x; k += 4u) {
        const uint32_t v = __hip_atomic_load((gu32 *)(cs.row_cnt + (ty * ma.tiles_x + k) * 16u + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (v & kCbRowTag) != 0u;
        const uint32_t c0 = v & 0x1ffu, c1 = (v >> 9) & 0x1ffu;
        t0 += c0, t1 += c1;
        if (k < tx) l0 += c0, l1 += c1;
      }
      if (__all(ok)) break;
      if (spins == 0) w0 = __builtin_amdgcn_s_memrealtime();
      backoff(spins);
      ++spins;
      // bounded by time, and over as soon as ANY wave of the launch has given up (sticky flag)
      if ((spins & 7u) == 0 && (__builtin_amdgcn_s_memrealtime() - w0 > uint64_t(g.spin_ticks) ||
                                __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0 && __hip_atomic_exchange(&hdr->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
          atomicAdd(&hdr->stats->timeouts, 1ull);
        break;
      }
    }
#if D2PC_ONEPASS_STATS
    if (spins && lane == 0) {
      s_stat[1] = spins;
      s_stat[2] = uint32_t(__builtin_amdgcn_s_memrealtime() - w0);
    }
#endif
    above = wave_sum(above);
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      t0 += __shfl_xor(t0, o, 64), t1 += __shfl_xor(t1, o, 64);
      l0 += __shfl_xor(l0, o, 64), l1 += __shfl_xor(l1, o, 64);
    }
    // lanes 0..15 hold the row pairs; spread to one row per lane: row r = 2p + h comes from lane p
    const uint32_t src_lane = (lane & 31u) >> 1;
    const uint32_t tt0 = __shfl(t0, src_lane, 64), tt1 = __shfl(t1, src_lane, 64);
    const uint32_t ll0 = __shfl(l0, src_lane, 64), ll1 = __shfl(l1, src_lane, 64);
    const uint32_t row_total = lane < 32u ? ((lane & 1u) ? tt1 : tt0) : 0u;
    const uint32_t row_left = (lane & 1u) ? ll1 : ll0;
    uint32_t incl = row_total;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t n = __shfl_up(incl, o, 64);
      if (lane >= uint32_t(o)) incl += n;
    }
    if (lane < 32u) s_base[lane] = above + incl - row_total + row_left;
    const uint32_t band_total = __builtin_amdgcn_readlane(incl, 31);
    if (counts && ty == ma.tiles_y - 1u && tx == 0u && lane == 0u) {
      const bool bad = __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      __hip_atomic_store(counts + f, bad ? kCountTimedOut : above + band_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();

  if (tid == 0) {
    // a block that saw the launch break marks its frame, whether or not the frame's last tile has reported already.
    // (Here, not behind the stores: a barrier after them would hold every wave until its stores have drained,
    // and the block could not make room for the next one while they do.)
    if (counts && __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
    CompactStats::Slot *sl = hdr->stats->slot + (blockIdx.x % uint32_t(kStatSlots));
    atomicAdd(&sl->tiles, 1ull);
    if (s_stat[1]) {
      atomicAdd(&sl->failed_polls, (unsigned long long)s_stat[1]);
      atomicAdd(&sl->wait_ticks, (unsigned long long)s_stat[2]);
    }
#endif
  }
  // ---- scatter: the same decisions, the points, their ordered stores ------------------------------------
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
#pragma unroll 1
  for (uint32_t r = wave; r < uint32_t(S::TH); r += uint32_t(S::THREADS / 64)) {
    const uint32_t y = y0 + r;
    if (y >= y_end) break;
    uint32_t row_pos = s_base[r];
    double ys = 0.0;
    if constexpr (is_stereo(QK)) ys = stereo_ny(Q, y);
    uint32_t raw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[q] = ob[r * uint32_t(S::OUT_STRIDE) + 64u * uint32_t(q) + lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t x = x0 + 64u * uint32_t(q) + lane;
      float X, Y, Z;
      const bool ok = pixel(x, y, raw[q], q, ys, true, X, Y, Z) && x < x_end;
      const uint64_t m = __ballot(ok);
      const uint32_t pos = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), row_pos));
      // pos < roi_n always holds for a correct prefix; the guard keeps a timed-out prefix from becoming an out-of-bounds store
      if (ok && pos < g.roi_n) {
        store_point<D2PC_CB_STORE_NT != 0>(fout, pos, X, Y, Z);
        if (fidx) st<D2PC_CB_INDEX_NT != 0>(fidx + pos, y * g.width + x);
      }
      row_pos += uint32_t(__popcll(m));
    }
  }
}

// --------------------------------------------------------------------------
// K1d: the same, SOFTWARE-PIPELINED over a block's tiles.  In K1c a block idles from the moment it has published
// its row counts until the slowest tile of its band has published too (5-7 us of 42 per tile in failed polls, plus
// the ticket's and the poll's round trips: profiles/r03_callback_compact.txt).  Here a block keeps taking tiles of its
// frame (ticket fetched under the previous tile's filter) and scatters tile i - 1 -- whose filtered bytes wait in a
// third LDS buffer -- AFTER the filter of tile i: by then its band published a whole tile ago and the wait is one
// look.  Hand-off words and output as in K1c.
// No deadlock with more than tiles_x blocks per frame (host-checked): a block waits only after it has published
// its current tile, so if every block of a frame waited, every issued tile would be published and each block's
// previous tile would lie in the one band that still has unissued tiles -- more blocks than that band has tiles.
// --------------------------------------------------------------------------
template <int KS, int QK>
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_callback_bs_compact_pipe(
    const uint8_t *__restrict__ src, float4 *__restrict__ out, uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
    uint8_t *state, const MedianArgs ma, const Geom g, const QArg<QK> Q) {
  using S = MedianBsShape<KS>;
  using gu32 = __attribute__((address_space(1))) uint32_t;
  using gu64 = __attribute__((address_space(1))) uint64_t;
  static_assert(S::THREADS == 256 && S::TH == 32, "one thread per byte value fills the table; 32 row counts per tile");
  constexpr uint32_t KEEP_WORDS = uint32_t(S::OUT_STRIDE * S::TH / 4);
  __shared__ __attribute__((aligned(16))) uint32_t s_w[S::W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_raw[S::RAW_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_keep[KEEP_WORDS];  // the filtered bytes of the tile waiting for its scatter
  __shared__ uint32_t s_next, s_exact;
  __shared__ uint32_t s_cnt[32], s_base[32];
  __shared__ uint32_t s_stat[3];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  const uint32_t f = blockIdx.x % g.n_frames;
  const CbCompactState cs(state, g, f, ma.tiles_y);
  const uint32_t tpf = ma.tiles_x * ma.tiles_y;
  const uint8_t *fsrc = src + uint64_t(f) * ma.src_frame_stride;
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
  const uint32_t x_end = ma.out_x0 + ma.out_w, y_end = ma.out_y0 + ma.out_h;
  double *lut_iw = reinterpret_cast<double *>(s_raw);               // [256]   (the staged rows' space, free between filters)
  float *lut_z = reinterpret_cast<float *>(s_raw) + 2 * 256;        // [256]
  uint8_t *lut_cls = reinterpret_cast<uint8_t *>(s_raw + 3 * 256);  // [256]
  static_assert(S::RAW_WORDS >= 3 * 256 + 64, "the tables fit where the staged rows were");

  if (tid == 0) {
    s_next = atomicAdd(cs.ticket, 1u);
    s_exact = !is_stereo(QK) ? 1u : 0u;
  }
  if (tid < 3) s_stat[tid] = 0;
  __syncthreads();
  uint32_t cur = s_next, prev = kNoTile;
  if (cur >= tpf) cur = kNoTile;
  bool exact = !is_stereo(QK);

  // one pixel: its point (when wanted) and whether it survives
  auto pixel = [&](uint32_t x, uint32_t y, uint32_t raw, double xs, double ys, bool want_point, float &X, float &Y, float &Z) -> bool {
    if constexpr (is_stereo(QK)) {
      if (!exact && !want_point) return lut_cls[raw] == 1u;
      const double iw = lut_iw[raw];
      X = float(xs * iw);
      Y = float(ys * iw);
      Z = lut_z[raw];
      if (!exact) return lut_cls[raw] == 1u;
      return point_is_valid(X, Y, Z, __fmul_rn(float(raw), g.scale), g.min_disparity);
    } else {
      const float d = __fmul_rn(float(raw), g.scale);
      reproject(Q, x, y, d, X, Y, Z);
      return point_is_valid(X, Y, Z, d, g.min_disparity);
    }
  };

  while (cur != kNoTile || prev != kNoTile) {
    uint32_t tk = 0, nxt = kNoTile;
    if (cur != kNoTile) {
      // the ticket of the tile after `cur`: its round trip runs under the filter
      if (tid == 0) tk = atomicAdd(cs.ticket, 1u);
      const uint32_t ty = cur / ma.tiles_x, tx = cur - ty * ma.tiles_x;
      const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);
      // (Requesting the NEXT tile's rows one tile ahead, so that they would complete before this iteration's store
      // burst, was built and is slower -- 722 -> 788 us: the 12 registers it keeps across the phases spill.)
      median_bs_tile<KS>(fsrc, ma, int(x0), int(y0), s_w, s_raw, tid);
      if (tid == 0) s_next = tk;
      if constexpr (is_stereo(QK)) {  // per byte value: 1/W, Z and the validity class (see K1c)
        const float d = __fmul_rn(float(tid), g.scale);
        const float dsel = fabsf(d) < __builtin_huge_valf() ? d : __builtin_nanf("");
        const double nw = stereo_w(Q, double(dsel));
        const double iw = 1.0 / nw;
        lut_iw[tid] = iw;
        lut_z[tid] = big_z_rule(d, float(Q.s.f * iw));
        const bool fin = finite_nonzero(nw), big = fabs(nw) >= Q.s.w_safe, keep = !(d <= g.min_disparity);
        const uint32_t cls = fin && keep ? (big ? 1u : 2u) : 0u;
        lut_cls[tid] = uint8_t(cls);
        if (cls == 2u) s_exact = 1u;  // (benign race: every writer stores 1; never reset)
      }
      __syncthreads();
      exact = s_exact != 0;
      nxt = s_next < tpf ? s_next : kNoTile;
      // ---- count: survivors per row of `cur` -------------------------------------------------------------
      const uint8_t *ob = reinterpret_cast<const uint8_t *>(s_w);
#pragma unroll 1
      for (uint32_t r = wave; r < uint32_t(S::TH); r += uint32_t(S::THREADS / 64)) {
        const uint32_t y = y0 + r;
        uint32_t cnt = 0;
        if (y < y_end) {
          double ys = 0.0;
          if constexpr (is_stereo(QK)) ys = stereo_ny(Q, y);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t x = x0 + 64u * uint32_t(q) + lane;
            const uint32_t raw = ob[r * uint32_t(S::OUT_STRIDE) + 64u * uint32_t(q) + lane];
            double xs = 0.0;
            if constexpr (is_stereo(QK)) xs = stereo_nx(Q, x);
            float X, Y, Z;
            const bool ok = pixel(x, y, raw, xs, ys, false, X, Y, Z) && x < x_end;
            cnt += uint32_t(__popcll(__ballot(ok)));
          }
        }
        if (lane == 0) s_cnt[r] = cnt;
      }
      __syncthreads();
      // ---- publish `cur` (wave 0): sixteen tagged dwords + the band's accumulator; nothing to wait for ------
      if (wave == 0) {
        const uint32_t mine = lane < 32u ? s_cnt[lane] : 0u;
        const uint32_t tile_total = wave_sum(mine);
        if (lane < 16u)
          __hip_atomic_store((gu32 *)(cs.row_cnt + cur * 16u + lane), kCbRowTag | s_cnt[2u * lane] | (s_cnt[2u * lane + 1u] << 9),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
          __hip_atomic_fetch_add((gu64 *)(cs.band_acc + ty), (uint64_t(1) << 32) | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
          s_stat[0] += 1u;
#endif
        }
      }
    }
    if (prev != kNoTile) {
      // ---- place `prev` (wave 0): its band published a whole filter ago -----------------------------------
      const uint32_t ty = prev / ma.tiles_x, tx = prev - ty * ma.tiles_x;
      const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);
      if (wave == 0) {
        const uint32_t j = lane >> 4, p = lane & 15u;
        uint32_t above = 0, t0 = 0, t1 = 0, l0 = 0, l1 = 0, spins = 0;
        uint64_t w0 = 0;
        for (;;) {
          bool ok = true;
          above = 0;
          for (uint32_t b0 = 0; b0 < ty; b0 += 64u) {
            const uint32_t bi = b0 + lane;
            if (bi < ty) {
              const uint64_t v = __hip_atomic_load((gu64 *)(cs.band_acc + bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && uint32_t(v >> 32) == ma.tiles_x;
              above += uint32_t(v);
            }
          }
          t0 = t1 = l0 = l1 = 0;
          for (uint32_t k = j; k < ma.tiles_x; k += 4u) {
            const uint32_t v = __hip_atomic_load((gu32 *)(cs.row_cnt + (ty * ma.tiles_x + k) * 16u + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (v & kCbRowTag) != 0u;
            const uint32_t c0 = v & 0x1ffu, c1 = (v >> 9) & 0x1ffu;
            t0 += c0, t1 += c1;
            if (k < tx) l0 += c0, l1 += c1;
          }
          if (__all(ok)) break;
          if (spins == 0) w0 = __builtin_amdgcn_s_memrealtime();
          backoff(spins);
          ++spins;
          if ((spins & 7u) == 0 && (__builtin_amdgcn_s_memrealtime() - w0 > uint64_t(g.spin_ticks) ||
                                    __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            if (lane == 0 && __hip_atomic_exchange(&hdr->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
              atomicAdd(&hdr->stats->timeouts, 1ull);
            break;
          }
        }
#if D2PC_ONEPASS_STATS
        if (spins && lane == 0) {
          s_stat[1] += spins;
          s_stat[2] += uint32_t(__builtin_amdgcn_s_memrealtime() - w0);
        }
#endif
        above = wave_sum(above);
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          t0 += __shfl_xor(t0, o, 64), t1 += __shfl_xor(t1, o, 64);
          l0 += __shfl_xor(l0, o, 64), l1 += __shfl_xor(l1, o, 64);
        }
        const uint32_t src_lane = (lane & 31u) >> 1;
        const uint32_t tt0 = __shfl(t0, src_lane, 64), tt1 = __shfl(t1, src_lane, 64);
        const uint32_t ll0 = __shfl(l0, src_lane, 64), ll1 = __shfl(l1, src_lane, 64);
        const uint32_t row_total = lane < 32u ? ((lane & 1u) ? tt1 : tt0) : 0u;
        const uint32_t row_left = (lane & 1u) ? ll1 : ll0;
        uint32_t incl = row_total;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          const uint32_t n = __shfl_up(incl, o, 64);
          if (lane >= uint32_t(o)) incl += n;
        }
        if (lane < 32u) s_base[lane] = above + incl - row_total + row_left;
        const uint32_t band_total = __builtin_amdgcn_readlane(incl, 31);
        if (counts && ty == ma.tiles_y - 1u && tx == 0u && lane == 0u) {
          const bool bad = __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
          __hip_atomic_store(counts + f, bad ? kCountTimedOut : above + band_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      lds_barrier();
      // ---- scatter `prev` from the kept bytes: the same decisions, the points, their ordered stores ----------
      const uint8_t *kb = reinterpret_cast<const uint8_t *>(s_keep);
#pragma unroll 1
      for (uint32_t r = wave; r < uint32_t(S::TH); r += uint32_t(S::THREADS / 64)) {
        const uint32_t y = y0 + r;
        if (y >= y_end) break;
        uint32_t row_pos = s_base[r];
        double ys = 0.0;
        if constexpr (is_stereo(QK)) ys = stereo_ny(Q, y);
        uint32_t raw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = kb[r * uint32_t(S::OUT_STRIDE) + 64u * uint32_t(q) + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t x = x0 + 64u * uint32_t(q) + lane;
          double xs = 0.0;
          if constexpr (is_stereo(QK)) xs = stereo_nx(Q, x);
          float X, Y, Z;
          const bool ok = pixel(x, y, raw[q], xs, ys, true, X, Y, Z) && x < x_end;
          const uint64_t m = __ballot(ok);
          const uint32_t pos = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), row_pos));
          if (ok && pos < g.roi_n) {  // (the guard keeps a timed-out prefix from becoming an out-of-bounds store)
            store_point<D2PC_CB_STORE_NT != 0>(fout, pos, X, Y, Z);
            if (fidx) st<D2PC_CB_INDEX_NT != 0>(fidx + pos, y * g.width + x);
          }
          row_pos += uint32_t(__popcll(m));
        }
      }
    }
    // (LDS-only barriers from here to the filter: a full one would hold every wave until its stores have drained)
    lds_barrier();  // the scatter has read s_keep and the tables; the count has read s_w
    if (cur != kNoTile) {
      // `cur` becomes the tile in waiting: its bytes move out of the filter's way
      for (uint32_t i = tid; i < KEEP_WORDS / 4u; i += uint32_t(S::THREADS))
        reinterpret_cast<uint4 *>(s_keep)[i] = reinterpret_cast<const uint4 *>(s_w)[i];
    }
    lds_barrier();
    prev = cur;
    cur = nxt;
  }
  if (tid == 0) {
    // a block that saw the launch break marks its frame, whether or not the frame's last tile has reported already
    if (counts && __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
    CompactStats::Slot *sl = hdr->stats->slot + (blockIdx.x % uint32_t(kStatSlots));
    atomicAdd(&sl->tiles, (unsigned long long)s_stat[0]);
    if (s_stat[1]) {
      atomicAdd(&sl->failed_polls, (unsigned long long)s_stat[1]);
      atomicAdd(&sl->wait_ticks, (unsigned long long)s_stat[2]);
    }
#endif
  }
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
template <int QK>
static QArg<QK> make_qarg(const LaunchArgs &a);
template <>
QArg<QK_GENERAL> make_qarg<QK_GENERAL>(const LaunchArgs &a) {
  QArg<QK_GENERAL> r;
  r.m = a.q;  // (form and, for form 2, the segment table are filled by the host: d2pc_capi.hip)
  return r;
}
template <>
QArg<QK_STEREO> make_qarg<QK_STEREO>(const LaunchArgs &a) {
  QArg<QK_STEREO> r;
  r.s = a.qs;
  return r;
}
template <>
QArg<QK_STEREO_CV24> make_qarg<QK_STEREO_CV24>(const LaunchArgs &a) {
  QArg<QK_STEREO_CV24> r;
  r.s = a.qs;
  r.seg = a.q.seg;  // the running column sum as the host replayed it (fill_q in d2pc_capi.hip)
  return r;
}
template <>
QArg<QK_STEREO_CV4> make_qarg<QK_STEREO_CV4>(const LaunchArgs &a) {
  QArg<QK_STEREO_CV4> r;
  r.s = a.qs;  // (f already rounded to float by the host)
  return r;
}

template <int DT, int QK, int PXT, bool VEC>
static hipError_t launch_parity_t(const LaunchArgs &a) {
  hipLaunchKernelGGL((k_reproject_pack<DT, QK, PXT, VEC>), dim3(a.grid), dim3(kBlock), 0, a.stream,
                     static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index,
                     a.counts, a.geom, make_qarg<QK>(a));
  return hipGetLastError();
}

// Zeroes the compaction state ahead of a single-pass launch and starts its header (the pointer to the context's
// counters; one launch counted).  A kernel of our own rather than hipMemsetAsync: captured into a hipGraph, the
// runtime's memset node left the state UNTOUCHED on replays when the graph was launched on another stream than it
// was captured on and the host had synchronised in between (the kernel node behind it found all of it dirty:
// profiles/r03_graph_memset.txt); a plain kernel node has exactly the ordering of the kernels around it.
__global__ __launch_bounds__(256) void k_state_clear(uint4 *__restrict__ p, uint32_t n16, CompactStats *stats) {
  constexpr uint32_t kHdr16 = uint32_t(sizeof(StateHeader) / 16);
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) {
    StateHeader fresh{};
    fresh.stats = stats;
    *reinterpret_cast<StateHeader *>(p) = fresh;
    atomicAdd(&stats->launches, 1ull);
  } else if (i >= kHdr16 && i < n16) {
    p[i] = uint4{0u, 0u, 0u, 0u};
  }
}

#if D2PC_CLEAR_WITH_MEMSET
__global__ __launch_bounds__(256) void k_state_verify(uint4 *__restrict__ p, uint32_t n16, CompactStats *stats) {
  using gu32 = __attribute__((address_space(1))) const uint32_t;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) atomicAdd(&stats->dbg[1], 1ull);
  if (i < n16) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p + i);
    uint32_t any = 0;
    for (int k = 0; k < 4; ++k) any |= __hip_atomic_load((gu32 *)(w + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (any) atomicAdd(&stats->dbg[0], 1ull);
  }
  // the memset wiped the header's pointer to the context's counters, which k_state_clear would have written: without it
  // the single pass behind this kernel adds its counters through a null pointer (round 4: a memory access fault at
  // 0x1000 on the first launch of this experiment build).  Written by the thread that looked at piece 0, after it looked.
  if (i == 0) reinterpret_cast<StateHeader *>(p)->stats = stats;
}
#endif

template <int DT, int QK, int PXT, bool VEC>
static hipError_t launch_compact_t(const LaunchArgs &a) {
  const uint8_t *disp = static_cast<const uint8_t *>(a.disp);
  float4 *out = static_cast<float4 *>(a.out_points);
  uint8_t *state = static_cast<uint8_t *>(a.state);
  if (a.compact_algo == 1) {  // count -> scan -> scatter: every state word is written before it is read
    hipLaunchKernelGGL((k_compact_count<DT, QK, PXT, VEC>), dim3(a.grid), dim3(kBlock), 0, a.stream, disp, state, a.geom,
                       make_qarg<QK>(a));
    const uint32_t selfscan = a.geom.tiles_per_frame <= kSelfScanTiles ? 1u : 0u;
    if (!selfscan)
      hipLaunchKernelGGL(k_compact_scan, dim3(a.geom.n_frames), dim3(kScanThreads), 0, a.stream, state, a.counts, a.geom);
    hipLaunchKernelGGL((k_compact_scatter<DT, QK, PXT, VEC>), dim3(a.grid), dim3(kBlock), 0, a.stream, disp, out,
                       a.out_index, a.counts, state, a.geom, make_qarg<QK>(a), selfscan);
  } else if (a.compact_algo == 3) {  // one launch, one resident block per tile (the host checked the grid against the residency)
    if (a.grid != a.geom.total_tiles || !a.stats || a.epoch < kEpochBase) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_compact_resident<DT, QK, PXT, VEC>), dim3(a.grid), dim3(kBlock), 0, a.stream, disp, out, a.out_index,
                       a.counts, state, static_cast<CompactStats *>(a.stats), a.geom, make_qarg<QK>(a), a.epoch);
  } else {
    const uint32_t n16 = uint32_t((a.state_bytes + 15) / 16);  // buffers are allocated in whole MiB
#if D2PC_CLEAR_WITH_MEMSET  // experiment only (tools/graph_memset_probe.py): round 1's hipMemsetAsync instead of the kernel
    if (getenv("D2PC_TRACE_MEMSET")) fprintf(stderr, "d2pc: hipMemsetAsync(%p, 0, %zu) on stream %p\n", a.state, a.state_bytes, (void *)a.stream);
    if (hipError_t e = hipMemsetAsync(a.state, 0, a.state_bytes, a.stream); e != hipSuccess) return e;
    // ... followed by a kernel that counts the 16-byte pieces of the state that are NOT zero when it runs
    // (stats->pad[0]) and the launches it looked at (pad[1]): d2pc_debug_read_stats
    hipLaunchKernelGGL(k_state_verify, dim3((n16 + 255) / 256), dim3(256), 0, a.stream, static_cast<uint4 *>(a.state), n16,
                       static_cast<CompactStats *>(a.stats));
#else
    hipLaunchKernelGGL(k_state_clear, dim3((n16 + 255) / 256), dim3(256), 0, a.stream, static_cast<uint4 *>(a.state), n16,
                       static_cast<CompactStats *>(a.stats));
#endif
    // frame-static assignment: a block serves frame blockIdx % n_frames, so the grid is a multiple of
    // n_frames (the C ABI falls back to the two-pass form when there are more frames than blocks)
    uint32_t grid = a.grid;
    if (grid < a.geom.n_frames) return hipErrorInvalidValue;
    grid -= grid % a.geom.n_frames;
    hipLaunchKernelGGL((k_compact_onepass<DT, QK, PXT, VEC>), dim3(grid), dim3(kBlock + 64), 0, a.stream, disp, out,
                       a.out_index, a.counts, state, a.geom, make_qarg<QK>(a));
  }
  return hipGetLastError();
}

// The frame counters of the chunked two-pass must read zero and the group totals "empty" when a call starts.  The scanning
// blocks leave them so; this runs only when a state buffer is taken over from another algorithm or another batch shape
// (the host keeps track).
__global__ __launch_bounds__(256) void k_chunk_clear(uint8_t *state, uint32_t stride, uint32_t n_frames, uint32_t gsum_words) {
  const uint32_t per = kChunkHdrWords + gsum_words;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t f = i / per, w = i - f * per;
  if (f < n_frames)
    reinterpret_cast<uint32_t *>(state + sizeof(StateHeader) + uint64_t(f) * stride)[w] = w < kChunkHdrWords ? 0u : kChunkEmpty;
}

// compact_algo 4: launch i scatters chunk i-1 and counts chunk i (launch 0 only counts, the last only scatters)
template <int DT, int QK, bool VEC>
static hipError_t launch_compact_chunked_t(const LaunchArgs &a) {
  const Geom &g = a.geom;
  uint32_t gw0 = 0;
  (void)chunk_frame_state_stride(g.tiles_per_frame, &gw0);
  if (a.chunk_clear) {
    const uint64_t words = uint64_t(g.n_frames) * (kChunkHdrWords + gw0);
    hipLaunchKernelGGL(k_chunk_clear, dim3(uint32_t((words + 255u) / 256u)), dim3(256), 0, a.stream, static_cast<uint8_t *>(a.state),
                       g.frame_state_stride, g.n_frames, gw0);
  }
  if (g.pxt != uint32_t(kChunkS) || a.chunk_frames == 0 || a.chunk_first == 0 || g.tiles_per_frame == 0) return hipErrorInvalidValue;
  ChunkArgs c{};
  c.groups_per_frame = (g.tiles_per_frame + kChunkGroupTiles - 1u) / kChunkGroupTiles;
  uint32_t gw = 0;
  if (chunk_frame_state_stride(g.tiles_per_frame, &gw) != g.frame_state_stride) return hipErrorInvalidValue;
  c.gsum_words = gw;
  c.div_gpf = make_fastdiv(c.groups_per_frame);
  uint32_t prev0 = 0, prevn = 0;  // the chunk counted by the previous launch
  uint32_t next0 = 0;
  for (;;) {
    uint32_t nextn = next0 == 0 ? a.chunk_first : a.chunk_frames;
    if (nextn > g.n_frames - next0) nextn = g.n_frames - next0;
    c.scatter_f0 = prev0;
    c.scatter_tiles = prevn * g.tiles_per_frame;
    c.count_f0 = next0;
    c.count_blocks = nextn * c.groups_per_frame;
    const uint64_t grid = uint64_t(c.scatter_tiles) + c.count_blocks;
    if (grid == 0) break;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    // ODD: workgroups go to the eight XCDs round-robin by index, and an even period put every count block -- the long
    // blocks of the launch -- on one or two XCDs (period 32: all 1,434 on XCD 0, 490 us for a launch that takes 105)
    c.period = c.count_blocks ? uint32_t(grid / c.count_blocks) : 1u;
    if (c.period > 1u && (c.period & 1u) == 0u) c.period -= 1u;
    c.div_period = make_fastdiv(c.period);
    hipLaunchKernelGGL((k_compact_chunk<DT, QK, VEC>), dim3(uint32_t(grid)), dim3(kBlock), 0, a.stream,
                       static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts,
                       static_cast<uint8_t *>(a.state), g, make_qarg<QK>(a), c);
    prev0 = next0;
    prevn = nextn;
    next0 += nextn;
  }
  return hipGetLastError();
}
template <int QK>
static hipError_t launch_compact_chunked_q(const LaunchArgs &a) {
  switch (a.dtype) {
    case DT_F32: return a.vec_rows ? launch_compact_chunked_t<DT_F32, QK, true>(a) : launch_compact_chunked_t<DT_F32, QK, false>(a);
    case DT_U8: return launch_compact_chunked_t<DT_U8, QK, false>(a);
    case DT_U16: return launch_compact_chunked_t<DT_U16, QK, false>(a);
  }
  return hipErrorInvalidValue;
}
static hipError_t launch_compact_chunked(const LaunchArgs &a) {
  switch (a.q_kind) {
    case QK_STEREO: return launch_compact_chunked_q<QK_STEREO>(a);
    case QK_STEREO_CV24: return launch_compact_chunked_q<QK_STEREO_CV24>(a);
    case QK_STEREO_CV4: return launch_compact_chunked_q<QK_STEREO_CV4>(a);
    case QK_GENERAL: return launch_compact_chunked_q<QK_GENERAL>(a);
  }
  return hipErrorInvalidValue;
}

template <int DT, int QK, int S>
static hipError_t launch_parity_small_t(const LaunchArgs &a) {
  hipLaunchKernelGGL((k_reproject_pack_small<DT, QK, S>), dim3(a.geom.total_tiles), dim3(kBlock), 0, a.stream,
                     static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts, a.geom,
                     make_qarg<QK>(a));
  return hipGetLastError();
}
template <int QK, int S>
static hipError_t dispatch_small(const LaunchArgs &a) {
  switch (a.dtype) {
    case DT_F32: return launch_parity_small_t<DT_F32, QK, S>(a);
    case DT_U8: return launch_parity_small_t<DT_U8, QK, S>(a);
    case DT_U16: return launch_parity_small_t<DT_U16, QK, S>(a);
  }
  return hipErrorInvalidValue;
}

template <int QK, int PXT>
static hipError_t dispatch_dtype(const LaunchArgs &a, bool compact) {
  switch (a.dtype) {
    case DT_F32:
      if (a.vec_rows) return compact ? launch_compact_t<DT_F32, QK, PXT, true>(a) : launch_parity_t<DT_F32, QK, PXT, true>(a);
      return compact ? launch_compact_t<DT_F32, QK, PXT, false>(a) : launch_parity_t<DT_F32, QK, PXT, false>(a);
    case DT_U8: return compact ? launch_compact_t<DT_U8, QK, PXT, false>(a) : launch_parity_t<DT_U8, QK, PXT, false>(a);
    case DT_U16: return compact ? launch_compact_t<DT_U16, QK, PXT, false>(a) : launch_parity_t<DT_U16, QK, PXT, false>(a);
  }
  return hipErrorInvalidValue;
}

template <int PXT>
static hipError_t dispatch_q(const LaunchArgs &a, bool compact) {
  switch (a.q_kind) {
    case QK_STEREO: return dispatch_dtype<QK_STEREO, PXT>(a, compact);
    case QK_STEREO_CV24: return dispatch_dtype<QK_STEREO_CV24, PXT>(a, compact);
    case QK_STEREO_CV4: return dispatch_dtype<QK_STEREO_CV4, PXT>(a, compact);
    case QK_GENERAL: return dispatch_dtype<QK_GENERAL, PXT>(a, compact);
  }
  return hipErrorInvalidValue;
}

bool tile_shape_supported(int pxt) { return pxt == 4 || pxt == 8 || pxt == 16; }

uint32_t frame_state_stride(uint32_t tiles_per_frame) {
  const uint32_t groups = (tiles_per_frame + kGroupTiles - 1) / kGroupTiles;
  const uint64_t b = kFrameTicketBytes + uint64_t(groups) * kGroupAccStride + uint64_t(tiles_per_frame) * 16;
  return uint32_t((b + 255) & ~uint64_t(255));
}

uint32_t chunk_frame_state_stride(uint32_t tiles_per_frame, uint32_t *gsum_words) {
  // [counter, 4 words][group totals][their exclusive prefixes][run prefixes, whole groups]
  const uint32_t groups = (tiles_per_frame + kChunkGroupTiles - 1u) / kChunkGroupTiles;
  const uint32_t gw = (groups + 3u) & ~3u;
  if (gsum_words) *gsum_words = gw;
  const uint64_t b = 4ull * (uint64_t(kChunkHdrWords) + 2ull * gw + uint64_t(groups) * kChunkGroupRuns);
  return uint32_t((b + 255) & ~uint64_t(255));
}

size_t compact_state_bytes(const Geom &g) {
  return sizeof(StateHeader) + size_t(g.n_frames) * g.frame_state_stride;
}

static hipError_t dispatch(const LaunchArgs &a, bool compact) {
#ifdef D2PC_FOCUS  // `make asm-focus`: only the kernels under development are instantiated (seconds instead of minutes)
  (void)a; (void)compact;
  return hipErrorNotSupported;
#else
  if (!compact && a.parity_small) {  // the small one-shot tiles (PARITY only)
#define D2PC_SMALL(S)                                                          \
  switch (a.q_kind) {                                                          \
    case QK_STEREO: return dispatch_small<QK_STEREO, S>(a);                    \
    case QK_STEREO_CV24: return dispatch_small<QK_STEREO_CV24, S>(a);          \
    case QK_STEREO_CV4: return dispatch_small<QK_STEREO_CV4, S>(a);            \
    case QK_GENERAL: return dispatch_small<QK_GENERAL, S>(a);                  \
  }                                                                            \
  return hipErrorInvalidValue
    switch (a.pxt) {
      case 1: D2PC_SMALL(1);
      case 2: D2PC_SMALL(2);
      case 4: D2PC_SMALL(4);
    }
#undef D2PC_SMALL
    return hipErrorInvalidValue;
  }
  switch (a.pxt) {
    case 4: return dispatch_q<4>(a, compact);
    case 8: return dispatch_q<8>(a, compact);
    case 16: return dispatch_q<16>(a, compact);
  }
  return hipErrorInvalidValue;
#endif
}

hipError_t launch_callback_bs(const LaunchArgs &a, MedianArgs m, const void *src, int ksize) {
#ifdef D2PC_FOCUS
  (void)a; (void)m; (void)src; (void)ksize;
  return hipErrorNotSupported;
#else
  if (!median_ksize_supported(ksize) || m.out_w == 0 || m.out_h == 0) return hipErrorInvalidValue;
  if (m.out_x0 != a.geom.border || m.out_y0 != a.geom.border || m.out_w != a.geom.roi_w ||
      uint64_t(m.out_w) * m.out_h != a.geom.roi_n)
    return hipErrorInvalidValue;  // the filter's output rectangle must be the reprojection's ROI
  using S = MedianBsShape<11>;  // the tile shape does not depend on k
  m.tiles_x = (m.out_w + S::TW - 1) / S::TW;
  m.tiles_y = (m.out_h + S::TH - 1) / S::TH;
  const uint64_t blocks = uint64_t(m.tiles_x) * m.tiles_y * m.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const uint8_t *s8 = static_cast<const uint8_t *>(src);
  float4 *o = static_cast<float4 *>(a.out_points);
#define D2PC_CB_BS(KS, QK)                                                                                              \
  hipLaunchKernelGGL((k_callback_bs<KS, QK>), dim3(uint32_t(blocks)), dim3(S::THREADS), 0, a.stream, s8, o, a.out_index, \
                     a.counts, m, a.geom, make_qarg<QK>(a))
  if (a.q_kind < QK_GENERAL || a.q_kind > QK_STEREO_CV4) return hipErrorInvalidValue;
  switch (ksize * 4 + a.q_kind) {
    case 12: D2PC_CB_BS(3, QK_GENERAL); break;
    case 13: D2PC_CB_BS(3, QK_STEREO); break;
    case 14: D2PC_CB_BS(3, QK_STEREO_CV24); break;
    case 15: D2PC_CB_BS(3, QK_STEREO_CV4); break;
    case 20: D2PC_CB_BS(5, QK_GENERAL); break;
    case 21: D2PC_CB_BS(5, QK_STEREO); break;
    case 22: D2PC_CB_BS(5, QK_STEREO_CV24); break;
    case 23: D2PC_CB_BS(5, QK_STEREO_CV4); break;
    case 28: D2PC_CB_BS(7, QK_GENERAL); break;
    case 29: D2PC_CB_BS(7, QK_STEREO); break;
    case 30: D2PC_CB_BS(7, QK_STEREO_CV24); break;
    case 31: D2PC_CB_BS(7, QK_STEREO_CV4); break;
    case 36: D2PC_CB_BS(9, QK_GENERAL); break;
    case 37: D2PC_CB_BS(9, QK_STEREO); break;
    case 38: D2PC_CB_BS(9, QK_STEREO_CV24); break;
    case 39: D2PC_CB_BS(9, QK_STEREO_CV4); break;
    case 44: D2PC_CB_BS(11, QK_GENERAL); break;
    case 45: D2PC_CB_BS(11, QK_STEREO); break;
    case 46: D2PC_CB_BS(11, QK_STEREO_CV24); break;
    case 47: D2PC_CB_BS(11, QK_STEREO_CV4); break;
    default: return hipErrorInvalidValue;
  }
#undef D2PC_CB_BS
  return hipGetLastError();
#endif
}

size_t callback_compact_state_bytes(uint32_t tiles_x, uint32_t tiles_y, uint32_t n_frames, uint32_t *frame_stride) {
  const uint64_t b = uint64_t(kCbTicketBytes) + cb_band_acc_bytes(tiles_y) + uint64_t(tiles_x) * tiles_y * 64u;
  const uint32_t stride = uint32_t((b + 255) & ~uint64_t(255));
  if (frame_stride) *frame_stride = stride;
  return sizeof(StateHeader) + size_t(n_frames) * stride;
}

hipError_t launch_callback_bs_compact(const LaunchArgs &a, MedianArgs m, const void *src, int ksize) {
#ifdef D2PC_FOCUS
  (void)a; (void)m; (void)src; (void)ksize;
  return hipErrorNotSupported;
#else
  if (!median_ksize_supported(ksize) || m.out_w == 0 || m.out_h == 0 || !a.state || !a.stats || !a.counts) return hipErrorInvalidValue;
  if (m.out_x0 != a.geom.border || m.out_y0 != a.geom.border || m.out_w != a.geom.roi_w ||
      uint64_t(m.out_w) * m.out_h != a.geom.roi_n)
    return hipErrorInvalidValue;  // the filter's output rectangle must be the reprojection's ROI
  using S = MedianBsShape<11>;  // the tile shape does not depend on k
  m.tiles_x = (m.out_w + S::TW - 1) / S::TW;
  m.tiles_y = (m.out_h + S::TH - 1) / S::TH;
  if (m.tiles_x > kCbMaxTilesX) return hipErrorInvalidValue;  // a band must fit the resident blocks (see the kernel)
  const uint64_t blocks = uint64_t(m.tiles_x) * m.tiles_y * m.n_frames;
  if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
  uint32_t stride = 0;
  if (callback_compact_state_bytes(m.tiles_x, m.tiles_y, m.n_frames, &stride) != a.state_bytes || stride != a.geom.frame_state_stride ||
      a.geom.n_frames != m.n_frames)
    return hipErrorInvalidValue;
  const uint32_t n16 = uint32_t((a.state_bytes + 15) / 16);
  hipLaunchKernelGGL(k_state_clear, dim3((n16 + 255) / 256), dim3(256), 0, a.stream, static_cast<uint4 *>(a.state), n16,
                     static_cast<CompactStats *>(a.stats));
  const uint8_t *s8 = static_cast<const uint8_t *>(src);
  float4 *o = static_cast<float4 *>(a.out_points);
  uint8_t *state = static_cast<uint8_t *>(a.state);
  // a.compact_algo 2: the pipelined form, a.grid persistent blocks (a multiple of n_frames, more than tiles_x per frame)
  const bool pipe = a.compact_algo == 2;
  if (pipe && (a.grid % m.n_frames != 0 || (a.grid / m.n_frames <= m.tiles_x && a.grid / m.n_frames < m.tiles_x * m.tiles_y)))
    return hipErrorInvalidValue;
  const uint32_t grid = pipe ? a.grid : uint32_t(blocks);
#define D2PC_CB_BSC(KS, QK)                                                                                                  \
  if (pipe)                                                                                                                  \
    hipLaunchKernelGGL((k_callback_bs_compact_pipe<KS, QK>), dim3(grid), dim3(S::THREADS), 0, a.stream, s8, o, a.out_index,  \
                       a.counts, state, m, a.geom, make_qarg<QK>(a));                                                       \
  else                                                                                                                       \
    hipLaunchKernelGGL((k_callback_bs_compact<KS, QK>), dim3(grid), dim3(S::THREADS), 0, a.stream, s8, o, a.out_index,      \
                       a.counts, state, m, a.geom, make_qarg<QK>(a))
  if (a.q_kind < QK_GENERAL || a.q_kind > QK_STEREO_CV4) return hipErrorInvalidValue;
  switch (ksize * 4 + a.q_kind) {
    case 12: D2PC_CB_BSC(3, QK_GENERAL); break;
    case 13: D2PC_CB_BSC(3, QK_STEREO); break;
    case 14: D2PC_CB_BSC(3, QK_STEREO_CV24); break;
    case 15: D2PC_CB_BSC(3, QK_STEREO_CV4); break;
    case 20: D2PC_CB_BSC(5, QK_GENERAL); break;
    case 21: D2PC_CB_BSC(5, QK_STEREO); break;
    case 22: D2PC_CB_BSC(5, QK_STEREO_CV24); break;
    case 23: D2PC_CB_BSC(5, QK_STEREO_CV4); break;
    case 28: D2PC_CB_BSC(7, QK_GENERAL); break;
    case 29: D2PC_CB_BSC(7, QK_STEREO); break;
    case 30: D2PC_CB_BSC(7, QK_STEREO_CV24); break;
    case 31: D2PC_CB_BSC(7, QK_STEREO_CV4); break;
    case 36: D2PC_CB_BSC(9, QK_GENERAL); break;
    case 37: D2PC_CB_BSC(9, QK_STEREO); break;
    case 38: D2PC_CB_BSC(9, QK_STEREO_CV24); break;
    case 39: D2PC_CB_BSC(9, QK_STEREO_CV4); break;
    case 44: D2PC_CB_BSC(11, QK_GENERAL); break;
    case 45: D2PC_CB_BSC(11, QK_STEREO); break;
    case 46: D2PC_CB_BSC(11, QK_STEREO_CV24); break;
    case 47: D2PC_CB_BSC(11, QK_STEREO_CV4); break;
    default: return hipErrorInvalidValue;
  }
#undef D2PC_CB_BSC
  return hipGetLastError();
#endif
}

hipError_t launch_parity(const LaunchArgs &a) { return dispatch(a, false); }
template <int QK, int R>
static hipError_t launch_resident_lean_q(const LaunchArgs &a) {
#define D2PC_RL(DT)                                                                                                             \
  hipLaunchKernelGGL((k_compact_resident_lean<DT, QK, R>), dim3(a.grid), dim3(kBlock), 0, a.stream,                              \
                     static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts,           \
                     static_cast<uint8_t *>(a.state), static_cast<CompactStats *>(a.stats), a.geom, make_qarg<QK>(a), a.epoch)
  switch (a.dtype) {
    case DT_F32: D2PC_RL(DT_F32); break;
    case DT_U8: D2PC_RL(DT_U8); break;
    case DT_U16: D2PC_RL(DT_U16); break;
    default: return hipErrorInvalidValue;
  }
#undef D2PC_RL
  return hipGetLastError();
}
template <int R>
static hipError_t launch_resident_lean(const LaunchArgs &a) {
  if (a.grid != a.geom.total_tiles || !a.stats || a.epoch < kEpochBase || a.geom.pxt != uint32_t(R)) return hipErrorInvalidValue;
  switch (a.q_kind) {
    case QK_STEREO: return launch_resident_lean_q<QK_STEREO, R>(a);
    case QK_STEREO_CV24: return launch_resident_lean_q<QK_STEREO_CV24, R>(a);
    case QK_STEREO_CV4: return launch_resident_lean_q<QK_STEREO_CV4, R>(a);
    case QK_GENERAL: return launch_resident_lean_q<QK_GENERAL, R>(a);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_compact(const LaunchArgs &a) {
  if (a.compact_algo == 4) return launch_compact_chunked(a);
  if (a.compact_algo == 3 && a.pxt == 32) return launch_resident_lean<32>(a);
  if (a.compact_algo == 3 && a.pxt == 64) return launch_resident_lean<64>(a);
  return dispatch(a, true);
}

}  // namespace d2pc
