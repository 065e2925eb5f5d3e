// d2pc_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the
// disparity -> point-cloud path.  See d2pc_device.hpp for the reference
// call sites this replaces.
//
// Design (HBM-bound streaming map, ~1 flop/byte; MFMA does not apply):
//  * Work is indexed by OUTPUT point, flat over the frame's ROI, so every
//    wave-level store is one contiguous, 1-KiB, 16-B-per-lane write of final
//    PointCloud2 bytes; a tile is BLOCK*PXT consecutive ROI pixels.
//  * Disparity is read once with coalesced per-lane dword loads (64 lanes =
//    256 contiguous bytes; ROI rows wrap inside a tile through an exact
//    multiply-high division, no per-row tails).
//  * Q and the geometry are kernel arguments: they sit in SGPRs for the whole
//    kernel (cheaper than LDS: no ds_read, no bank traffic, no barrier).
//  * Arithmetic follows OpenCV's double-precision evaluation: fp64 FMA chain
//    for the four row products, one IEEE fp64 reciprocal, one cast to fp32.
//  * COMPACT mode: wave ballot + mbcnt ranks, LDS scan over the block's
//    (slot, wave) counts, and a two-level counted prefix across tiles
//    (64-bit {arrivals,sum} group accumulators + tagged per-tile granules)
//    so the output order equals the CPU loop's row-major order bit-for-bit.
#include "d2pc_device.hpp"
#include "d2pc_launch.hpp"

namespace d2pc {

typedef float v4f __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------
// per-pixel pieces
// --------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float load_disparity(const uint8_t *frame, const Geom &g, uint32_t v, uint32_t u) {
  const uint8_t *row = frame + uint64_t(v) * g.row_stride;
  if constexpr (DT == DT_F32) {
    return __builtin_nontemporal_load(reinterpret_cast<const float *>(row) + u);
  } else if constexpr (DT == DT_U8) {
    // cpp:61 convertTo(CV_32FC1, scale): product formed in fp32
    return __fmul_rn(float(__builtin_nontemporal_load(row + u)), g.scale);
  } else {
    return __fmul_rn(float(__builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(row) + u)), g.scale);
  }
}

// cpp:63-64  [X Y Z W] = Q.(u,v,d,1); (X/W, Y/W, Z/W) evaluated in fp64 with
// the association of OpenCV 2.4's loop: (row term + u*q_0) + d*q_2, then
// iW = 1./W and num*iW, one cast to fp32 at the end.
__device__ __forceinline__ void reproject(const QMat &Q, uint32_t u, uint32_t v, float d, float &X, float &Y,
                                          float &Z) {
  const double du = double(u), dv = double(v), dd = double(d);
  const double nx = fma(Q.q[2], dd, fma(Q.q[0], du, fma(Q.q[1], dv, Q.q[3])));
  const double ny = fma(Q.q[6], dd, fma(Q.q[4], du, fma(Q.q[5], dv, Q.q[7])));
  const double nz = fma(Q.q[10], dd, fma(Q.q[8], du, fma(Q.q[9], dv, Q.q[11])));
  const double nw = fma(Q.q[14], dd, fma(Q.q[12], du, fma(Q.q[13], dv, Q.q[15])));
  const double iw = 1.0 / nw;
  X = float(nx * iw);
  Y = float(ny * iw);
  Z = float(nz * iw);
}

__device__ __forceinline__ bool point_is_valid(float X, float Y, float Z, float d, float min_disparity) {
  // finite <=> |x| < inf; NaN compares false
  const float inf = __builtin_huge_valf();
  return (int(fabsf(X) < inf) & int(fabsf(Y) < inf) & int(fabsf(Z) < inf) & int(!(d <= min_disparity))) != 0;
}

__device__ __forceinline__ void store_point(float4 *dst, float X, float Y, float Z) {
  // pcl::PointXYZ = {x,y,z,1.0f} (cpp:74); written once, never re-read here
  const v4f p = {X, Y, Z, 1.0f};
  __builtin_nontemporal_store(p, reinterpret_cast<v4f *>(dst));  // one global_store_dwordx4 nt
}

// --------------------------------------------------------------------------
// K1: PARITY mode -- every ROI pixel, reference order, nothing filtered.
// --------------------------------------------------------------------------
template <int DT, int BLOCK, int PXT>
__global__ __launch_bounds__(BLOCK) void k_reproject_pack(const uint8_t *__restrict__ disp,
                                                          float4 *__restrict__ out,
                                                          uint32_t *__restrict__ out_index,
                                                          uint32_t *__restrict__ counts, const Geom g,
                                                          const QMat Q) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
    float4 *fout = out + uint64_t(f) * g.out_frame_stride;
    const uint32_t base = lt * uint32_t(BLOCK * PXT);

    float d[PXT];
    uint32_t uu[PXT], vv[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
      const uint32_t i = base + uint32_t(k * BLOCK) + tid;
      const uint32_t ic = i < g.roi_n ? i : g.roi_n - 1;  // keep tail loads in bounds
      const uint32_t rv = fdiv(ic, g.div_roi_w);
      uu[k] = ic - rv * g.roi_w + g.border;
      vv[k] = rv + g.border;
      d[k] = load_disparity<DT>(fin, g, vv[k], uu[k]);
    }
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
      const uint32_t i = base + uint32_t(k * BLOCK) + tid;
      float X, Y, Z;
      reproject(Q, uu[k], vv[k], d[k], X, Y, Z);
      if (i < g.roi_n) {
        store_point(fout + i, X, Y, Z);
        if (out_index) __builtin_nontemporal_store(vv[k] * g.width + uu[k], out_index + uint64_t(f) * g.out_frame_stride + i);
      }
    }
    if (counts && lt == 0 && tid == 0) counts[f] = g.roi_n;
  }
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

// Exclusive offsets of every (slot, wave) cell of a block in row-major
// (slot-major, wave-minor) order == pixel order inside the tile.
// s_cnt[k*WAVES + w] holds the wave's popcount for slot k.  Returns the
// exclusive scan in lanes 0..PXT*WAVES-1 and the tile total in `total`.
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

// Sum of the point counts of all tiles of frame f that precede local tile lt:
// full groups via the group accumulators, the partial group via tile granules.
// WAIT = true (single pass): spin, bounded, until every predecessor has
// published; WAIT = false (two-pass scatter): values are final already.
template <bool WAIT>
__device__ __forceinline__ uint32_t prefix_before(const uint64_t *group_acc, const uint64_t *granules,
                                                  CompactHeader *hdr, const Geom &g, uint32_t f, uint32_t lt,
                                                  uint32_t lane) {
  using gu64 = __attribute__((address_space(1))) const uint64_t;
  const uint32_t grp = lt / kGroupTiles;
  const uint64_t *ga = group_acc + uint64_t(f) * g.groups_per_frame;
  const uint64_t *tg = granules + uint64_t(f) * g.tiles_per_frame;
  uint32_t sum = 0;
  // groups 0..grp-1 are complete groups of kGroupTiles tiles each
  for (uint32_t g0 = 0; g0 < grp; g0 += 64) {
    const uint32_t gi = g0 + lane;
    const bool on = gi < grp;
    uint64_t v = 0;
    if constexpr (WAIT) {
      uint32_t spins = 0;
      for (;;) {
        if (on) v = __hip_atomic_load((gu64 *)(ga + gi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = !on || uint32_t(v >> 32) == uint32_t(kGroupTiles);
        if (__all(ready)) break;
        if (++spins > kSpinLimit) {
          if (lane == 0) __hip_atomic_store(&hdr->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    } else {
      if (on) v = ga[gi];
    }
    sum += on ? uint32_t(v) : 0u;
  }
  // tiles grp*64 .. lt-1 of the own group (< 64 of them)
  {
    const uint32_t ti = grp * kGroupTiles + lane;
    const bool on = ti < lt;
    uint64_t v = 0;
    if constexpr (WAIT) {
      uint32_t spins = 0;
      for (;;) {
        if (on) v = __hip_atomic_load((gu64 *)(tg + ti), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = !on || (v & kGranuleTag) != 0;
        if (__all(ready)) break;
        if (++spins > kSpinLimit) {
          if (lane == 0) __hip_atomic_store(&hdr->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    } else {
      if (on) v = tg[ti];
    }
    sum += on ? uint32_t(v) : 0u;
  }
  return wave_sum(sum);
}

// One tile of COMPACT work held in registers.
template <int DT, int BLOCK, int PXT>
struct TileRegs {
  float X[PXT], Y[PXT], Z[PXT];
  uint32_t pix[PXT];   // source pixel index v*W+u
  uint64_t mask[PXT];  // wave ballot of validity per slot
};

template <int DT, int BLOCK, int PXT>
__device__ __forceinline__ void tile_compute(TileRegs<DT, BLOCK, PXT> &r, const uint8_t *fin, const Geom &g,
                                             const QMat &Q, uint32_t base, uint32_t tid) {
  float d[PXT];
  uint32_t uu[PXT], vv[PXT];
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const uint32_t i = base + uint32_t(k * BLOCK) + tid;
    const uint32_t ic = i < g.roi_n ? i : g.roi_n - 1;
    const uint32_t rv = fdiv(ic, g.div_roi_w);
    uu[k] = ic - rv * g.roi_w + g.border;
    vv[k] = rv + g.border;
    d[k] = load_disparity<DT>(fin, g, vv[k], uu[k]);
  }
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const uint32_t i = base + uint32_t(k * BLOCK) + tid;
    reproject(Q, uu[k], vv[k], d[k], r.X[k], r.Y[k], r.Z[k]);
    r.pix[k] = vv[k] * g.width + uu[k];
    const bool ok = (i < g.roi_n) && point_is_valid(r.X[k], r.Y[k], r.Z[k], d[k], g.min_disparity);
    r.mask[k] = __ballot(ok);
  }
}

template <int DT, int BLOCK, int PXT>
__device__ __forceinline__ void tile_scatter(const TileRegs<DT, BLOCK, PXT> &r, float4 *fout, uint32_t *fidx,
                                             uint32_t tile_prefix, uint32_t cell_excl, uint32_t wave,
                                             uint32_t lane) {
  constexpr int WAVES = BLOCK / 64;
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const uint32_t cell = __builtin_amdgcn_readlane(cell_excl, k * WAVES + int(wave));
    const uint32_t pos = tile_prefix + cell + mbcnt64(r.mask[k]);
    if ((r.mask[k] >> lane) & 1) {
      store_point(fout + pos, r.X[k], r.Y[k], r.Z[k]);
      if (fidx) __builtin_nontemporal_store(r.pix[k], fidx + pos);
    }
  }
}

// --------------------------------------------------------------------------
// K2a/K2b: two-pass compaction (count -> scatter).  No in-launch hand-off.
// --------------------------------------------------------------------------
template <int DT, int BLOCK, int PXT>
__global__ __launch_bounds__(BLOCK) void k_compact_count(const uint8_t *__restrict__ disp, uint64_t *group_acc,
                                                         uint64_t *granules, const Geom g, const QMat Q) {
  constexpr int WAVES = BLOCK / 64;
  __shared__ uint32_t s_w[WAVES];
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (uint32_t t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    TileRegs<DT, BLOCK, PXT> r;
    tile_compute<DT, BLOCK, PXT>(r, disp + uint64_t(f) * g.in_frame_stride, g, Q, lt * uint32_t(BLOCK * PXT), tid);
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < PXT; ++k) c += uint32_t(__popcll(r.mask[k]));
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (tid == 0) {
      uint32_t tot = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += s_w[w];
      granules[uint64_t(f) * g.tiles_per_frame + lt] = tot;
      atomicAdd(reinterpret_cast<unsigned long long *>(group_acc + uint64_t(f) * g.groups_per_frame + lt / kGroupTiles),
                (unsigned long long)tot);
    }
    __syncthreads();
  }
}

template <int DT, int BLOCK, int PXT>
__global__ __launch_bounds__(BLOCK) void k_compact_scatter(const uint8_t *__restrict__ disp,
                                                           float4 *__restrict__ out,
                                                           uint32_t *__restrict__ out_index,
                                                           uint32_t *__restrict__ counts,
                                                           const uint64_t *__restrict__ group_acc,
                                                           const uint64_t *__restrict__ granules, const Geom g,
                                                           const QMat Q) {
  constexpr int WAVES = BLOCK / 64;
  constexpr int CELLS = PXT * WAVES;
  __shared__ uint32_t s_cnt[CELLS];
  __shared__ uint32_t s_prefix;
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (uint32_t t = blockIdx.x; t < g.total_tiles; t += gridDim.x) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    TileRegs<DT, BLOCK, PXT> r;
    tile_compute<DT, BLOCK, PXT>(r, disp + uint64_t(f) * g.in_frame_stride, g, Q, lt * uint32_t(BLOCK * PXT), tid);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < PXT; ++k) s_cnt[k * WAVES + wave] = uint32_t(__popcll(r.mask[k]));
    }
    if (wave == 0) {
      const uint32_t p = prefix_before<false>(group_acc, granules, nullptr, g, f, lt, lane);
      if (lane == 0) s_prefix = p;
    }
    __syncthreads();
    uint32_t total;
    const uint32_t excl = scan_cells<CELLS>(s_cnt, lane, total);
    const uint32_t prefix = s_prefix;
    float4 *fout = out + uint64_t(f) * g.out_frame_stride;
    uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
    tile_scatter<DT, BLOCK, PXT>(r, fout, fidx, prefix, excl, wave, lane);
    if (counts && lt == g.tiles_per_frame - 1 && tid == 0) counts[f] = prefix + total;
    __syncthreads();
  }
}

// --------------------------------------------------------------------------
// K2: single-pass compaction.  Tiles are handed out by a ticket counter, so
// every predecessor of a tile is already running (or done) when the tile
// starts: waiting on predecessors' COUNTS cannot deadlock whatever the
// dispatch order or residency.  A tile publishes its count right after its
// own loads (it never waits before publishing), so there is no serial chain:
// the wait is for the slowest predecessor's load, not for a scan to ripple.
// --------------------------------------------------------------------------
template <int DT, int BLOCK, int PXT>
__global__ __launch_bounds__(BLOCK) void k_compact_onepass(const uint8_t *__restrict__ disp,
                                                           float4 *__restrict__ out,
                                                           uint32_t *__restrict__ out_index,
                                                           uint32_t *__restrict__ counts, CompactHeader *hdr,
                                                           uint64_t *group_acc, uint64_t *granules, const Geom g,
                                                           const QMat Q) {
  using gu64 = __attribute__((address_space(1))) uint64_t;
  constexpr int WAVES = BLOCK / 64;
  constexpr int CELLS = PXT * WAVES;
  __shared__ uint32_t s_cnt[CELLS];
  __shared__ uint32_t s_prefix;
  __shared__ uint32_t s_ticket[2];
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (tid == 0) s_ticket[1] = atomicAdd(&hdr->ticket, 1u);
  __syncthreads();
  uint32_t t = s_ticket[1];
  uint32_t it = 0;
  while (t < g.total_tiles) {
    const uint32_t f = fdiv(t, g.div_tpf);
    const uint32_t lt = t - f * g.tiles_per_frame;
    TileRegs<DT, BLOCK, PXT> r;
    tile_compute<DT, BLOCK, PXT>(r, disp + uint64_t(f) * g.in_frame_stride, g, Q, lt * uint32_t(BLOCK * PXT), tid);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < PXT; ++k) s_cnt[k * WAVES + wave] = uint32_t(__popcll(r.mask[k]));
    }
    __syncthreads();
    uint32_t total;
    const uint32_t excl = scan_cells<CELLS>(s_cnt, lane, total);
    uint32_t next_t = 0;
    if (wave == 0) {
      if (lane == 0) {
        // publish: tagged granule (the data is the flag) + group accumulator
        __hip_atomic_store((gu64 *)(granules + uint64_t(f) * g.tiles_per_frame + lt), kGranuleTag | total,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add((gu64 *)(group_acc + uint64_t(f) * g.groups_per_frame + lt / kGroupTiles),
                               (uint64_t(1) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // take the NEXT ticket now: its latency hides under the wait below
        next_t = atomicAdd(&hdr->ticket, 1u);
      }
      const uint32_t p = prefix_before<true>(group_acc, granules, hdr, g, f, lt, lane);
      if (lane == 0) {
        s_prefix = p;
        s_ticket[it & 1] = next_t;
      }
    }
    __syncthreads();
    const uint32_t prefix = s_prefix;
    t = s_ticket[it & 1];
    ++it;
    float4 *fout = out + uint64_t(f) * g.out_frame_stride;
    uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;
    tile_scatter<DT, BLOCK, PXT>(r, fout, fidx, prefix, excl, wave, lane);
    if (counts && lt == g.tiles_per_frame - 1 && tid == 0) counts[f] = prefix + total;
  }
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
template <int DT, int BLOCK, int PXT>
static hipError_t launch_parity_t(const LaunchArgs &a) {
  hipLaunchKernelGGL((k_reproject_pack<DT, BLOCK, PXT>), dim3(a.grid), dim3(BLOCK), 0, a.stream,
                     static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index,
                     a.counts, a.geom, a.q);
  return hipGetLastError();
}

template <int DT, int BLOCK, int PXT>
static hipError_t launch_compact_t(const LaunchArgs &a) {
  const uint8_t *disp = static_cast<const uint8_t *>(a.disp);
  float4 *out = static_cast<float4 *>(a.out_points);
  hipError_t e = hipMemsetAsync(a.state, 0, a.state_bytes, a.stream);
  if (e != hipSuccess) return e;
  CompactHeader *hdr = static_cast<CompactHeader *>(a.state);
  uint64_t *group_acc = reinterpret_cast<uint64_t *>(hdr + 1);
  uint64_t *granules = group_acc + uint64_t(a.geom.n_frames) * a.geom.groups_per_frame;
  if (a.compact_algo == 1) {
    hipLaunchKernelGGL((k_compact_count<DT, BLOCK, PXT>), dim3(a.grid), dim3(BLOCK), 0, a.stream, disp, group_acc,
                       granules, a.geom, a.q);
    hipLaunchKernelGGL((k_compact_scatter<DT, BLOCK, PXT>), dim3(a.grid), dim3(BLOCK), 0, a.stream, disp, out,
                       a.out_index, a.counts, group_acc, granules, a.geom, a.q);
  } else {
    hipLaunchKernelGGL((k_compact_onepass<DT, BLOCK, PXT>), dim3(a.grid), dim3(BLOCK), 0, a.stream, disp, out,
                       a.out_index, a.counts, hdr, group_acc, granules, a.geom, a.q);
  }
  return hipGetLastError();
}

template <int BLOCK, int PXT>
static hipError_t dispatch_dtype(const LaunchArgs &a, bool compact) {
  switch (a.dtype) {
    case DT_F32: return compact ? launch_compact_t<DT_F32, BLOCK, PXT>(a) : launch_parity_t<DT_F32, BLOCK, PXT>(a);
    case DT_U8: return compact ? launch_compact_t<DT_U8, BLOCK, PXT>(a) : launch_parity_t<DT_U8, BLOCK, PXT>(a);
    case DT_U16: return compact ? launch_compact_t<DT_U16, BLOCK, PXT>(a) : launch_parity_t<DT_U16, BLOCK, PXT>(a);
  }
  return hipErrorInvalidValue;
}

bool tile_shape_supported(int pxt) { return pxt == 4 || pxt == 8 || pxt == 16; }

size_t compact_state_bytes(const Geom &g) {
  size_t b = sizeof(CompactHeader) + (uint64_t(g.n_frames) * g.groups_per_frame + uint64_t(g.n_frames) * g.tiles_per_frame) * 8;
  return (b + 15) & ~size_t(15);
}

hipError_t launch_parity(const LaunchArgs &a) {
  switch (a.pxt) {
    case 4: return dispatch_dtype<256, 4>(a, false);
    case 8: return dispatch_dtype<256, 8>(a, false);
    case 16: return dispatch_dtype<256, 16>(a, false);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_compact(const LaunchArgs &a) {
  switch (a.pxt) {
    case 4: return dispatch_dtype<256, 4>(a, true);
    case 8: return dispatch_dtype<256, 8>(a, true);
    case 16: return dispatch_dtype<256, 16>(a, true);
  }
  return hipErrorInvalidValue;
}

}  // namespace d2pc
