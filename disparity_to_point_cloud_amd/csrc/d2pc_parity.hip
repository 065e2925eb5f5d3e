// d2pc_parity.hip -- PARITY mode (what the reference publishes: every ROI pixel, cpp:63-85): the reprojection +
// PointCloud2 pack kernels and their launcher.
#include "d2pc_pixel.hpp"

namespace d2pc {

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

#if D2PC_EXPERIMENTS
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

#endif  // D2PC_EXPERIMENTS

template <int S>
static hipError_t launch_small(const LaunchArgs &a) {
  return for_q_kind(a.q_kind, [&](auto qk) {
    return for_dtype(a.dtype, [&](auto dt) {
      constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
      hipLaunchKernelGGL((k_reproject_pack_small<DT, QK, S>), dim3(a.geom.total_tiles), dim3(kBlock), 0, a.stream,
                         static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts, a.geom,
                         make_qarg<QK>(a));
      return hipGetLastError();
    });
  });
}

#if D2PC_EXPERIMENTS
template <int PXT>
static hipError_t launch_walking(const LaunchArgs &a) {
  return for_q_kind(a.q_kind, [&](auto qk) {
    return for_dtype_vec(a, [&](auto dt, auto vec) {
      constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
      constexpr bool VEC = decltype(vec)::value;
      hipLaunchKernelGGL((k_reproject_pack<DT, QK, PXT, VEC>), dim3(a.grid), dim3(kBlock), 0, a.stream,
                         static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts, a.geom,
                         make_qarg<QK>(a));
      return hipGetLastError();
    });
  });
}
#endif

hipError_t launch_parity(const LaunchArgs &a) {
  if (a.parity_small) {  // one-shot blocks of 256 * pxt pixels
    switch (a.pxt) {
      case 1: return launch_small<1>(a);
      case 2: return launch_small<2>(a);
#if D2PC_EXPERIMENTS
      case 4: return launch_small<4>(a);
#endif
    }
    return hipErrorInvalidValue;
  }
#if D2PC_EXPERIMENTS
  switch (a.pxt) {  // tiles walked by fewer, longer-lived blocks (rounds 1-2)
    case 4: return launch_walking<4>(a);
    case 8: return launch_walking<8>(a);
    case 16: return launch_walking<16>(a);
  }
#endif
  return hipErrorInvalidValue;
}

}  // namespace d2pc
