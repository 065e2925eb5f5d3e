// d2pc_pixel.hpp -- per-pixel device pieces shared by every reprojection kernel (gfx950, wave64): input decode
// (cpp:60-61), cv::reprojectImageTo3D's arithmetic (cpp:63-64) in its published associations, the 16-byte
// pcl::PointXYZ store (cpp:74, cpp:84-85), and the tile <-> pixel mapping.  See d2pc_device.hpp for the
// reference call sites this replaces.
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
#pragma once

#include <type_traits>

#include "d2pc_device.hpp"
#include "d2pc_launch.hpp"

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
// neither published form where a dense Q makes a numerator cancel (tens of float ulp); that form exists in the
// experiment build only (test hook "general_q_form" = 1, d2pc_ext_set_test_hook), for comparison.  cv::stereoRectify's Q takes the specialised path below.
__device__ __forceinline__ void reproject(const QArg<QK_GENERAL> &A, uint32_t u, uint32_t v, float d, float &X,
                                          float &Y, float &Z) {
  const double *q = A.m.q;
  const double du = double(u), dv = double(v), dd = double(d);
#if D2PC_EXPERIMENTS
  if (A.m.form == 1u) {  // (wave-uniform) round 2's fused multiply-adds: experiment build only
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
#endif
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


// ---- kernel-argument forms of Q ----
template <int QK>
inline QArg<QK> make_qarg(const LaunchArgs &a);
template <>
inline QArg<QK_GENERAL> make_qarg<QK_GENERAL>(const LaunchArgs &a) {
  QArg<QK_GENERAL> r;
  r.m = a.q;  // (form and, for form 2, the segment table are filled by the host: fill_q in d2pc_capi_context.hip)
  return r;
}
template <>
inline QArg<QK_STEREO> make_qarg<QK_STEREO>(const LaunchArgs &a) {
  QArg<QK_STEREO> r;
  r.s = a.qs;
  return r;
}
template <>
inline QArg<QK_STEREO_CV24> make_qarg<QK_STEREO_CV24>(const LaunchArgs &a) {
  QArg<QK_STEREO_CV24> r;
  r.s = a.qs;
  r.seg = a.q.seg;  // the running column sum as the host replayed it (fill_q in d2pc_capi_context.hip)
  return r;
}
template <>
inline QArg<QK_STEREO_CV4> make_qarg<QK_STEREO_CV4>(const LaunchArgs &a) {
  QArg<QK_STEREO_CV4> r;
  r.s = a.qs;  // (f already rounded to float by the host)
  return r;
}

// ---- host-side dispatch over the kernels' template parameters ----
// f(std::integral_constant<int, QK>{}) for the launch's Q kind
template <class F>
inline hipError_t for_q_kind(int q_kind, F &&f) {
  switch (q_kind) {
    case QK_STEREO: return f(std::integral_constant<int, QK_STEREO>{});
    case QK_STEREO_CV24: return f(std::integral_constant<int, QK_STEREO_CV24>{});
    case QK_STEREO_CV4: return f(std::integral_constant<int, QK_STEREO_CV4>{});
    case QK_GENERAL: return f(std::integral_constant<int, QK_GENERAL>{});
  }
  return hipErrorInvalidValue;
}
// f(dtype constant) for the launch's sample type
template <class F>
inline hipError_t for_dtype(int dtype, F &&f) {
  switch (dtype) {
    case DT_F32: return f(std::integral_constant<int, DT_F32>{});
    case DT_U8: return f(std::integral_constant<int, DT_U8>{});
    case DT_U16: return f(std::integral_constant<int, DT_U16>{});
  }
  return hipErrorInvalidValue;
}
// f(dtype constant, std::bool_constant<VEC>{}): 16-byte row loads exist for fp32 rows only
template <class F>
inline hipError_t for_dtype_vec(const LaunchArgs &a, F &&f) {
  switch (a.dtype) {
    case DT_F32:
      if (a.vec_rows) return f(std::integral_constant<int, DT_F32>{}, std::true_type{});
      return f(std::integral_constant<int, DT_F32>{}, std::false_type{});
    case DT_U8: return f(std::integral_constant<int, DT_U8>{}, std::false_type{});
    case DT_U16: return f(std::integral_constant<int, DT_U16>{}, std::false_type{});
  }
  return hipErrorInvalidValue;
}

}  // namespace d2pc
