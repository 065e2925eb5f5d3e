// d2pc_callback.hip -- the callback body (cpp:55-85) TILE BY TILE in one kernel: bit-sliced k x k median of a tile of
// the inset ROI, then the tile's points from the filtered bytes still in LDS -- PARITY, and COMPACT in two forms.
#include "d2pc_compact_common.hpp"
#include "d2pc_median_bs_tile.hpp"

// OCCUPANCY PROBE (make variant NAME=occ2 DEFS=-DD2PC_CB_LDS_PAD=20000): unused dynamic LDS per block of the PARITY body, so that
// two blocks (or one) fit a CU instead of three -- how does a kernel at the socket power cap answer to fewer waves?
// (profiles/r06_energy_probe.txt)
#ifndef D2PC_CB_LDS_PAD
#define D2PC_CB_LDS_PAD 0
#endif



namespace d2pc {

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
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(D2PC_BS_WAVES))) void k_callback_bs(
    const uint8_t *__restrict__ src, float4 *__restrict__ out, uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
    const MedianArgs ma, const Geom g, const QArg<QK> Q) {
  using S = MedianBsShape<KS>;
  static_assert(S::THREADS == 256, "one thread per byte value fills the table");
  __shared__ __attribute__((aligned(16))) uint32_t s_w[S::W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_raw[S::RAW_WORDS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef D2PC_DIAG
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  uint32_t b = blockIdx.x;
  const uint32_t f = b / (ma.tiles_x * ma.tiles_y);
  b -= f * ma.tiles_x * ma.tiles_y;
  const uint32_t ty = b / ma.tiles_x, tx = b - ty * ma.tiles_x;
  const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);  // first output pixel
  median_bs_tile<KS>(src + uint64_t(f) * ma.src_frame_stride, ma, int(x0), int(y0), s_w, s_raw, tid);
  D2PC_BS_STAMP(e0);

  // (round 6: the table in 3 KB of its own, filled under the tile's loads instead of here behind a barrier: no difference --
  //  562.7 against 562.5 us, profiles/r06_ab_callback_prio.txt; under the power cap a shorter stage buys nothing by itself)
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
  const uint32_t r_end = y_end - y0 < uint32_t(S::TH) ? y_end - y0 : uint32_t(S::TH);  // rows of the tile inside the ROI
  // (two or four rows per trip, so that their chains LDS byte -> table entry -> fp64 products -> store overlap: no difference,
  // 578.9 / 578.8 / 579.4 us per 16 x 4K; the epilogue is 13 % of a block's cycles and store-issue-bound: profiles/r05_callback_phases.txt.
  // Round 6, by hand what the compiler does not do -- it emits four serial chains per row, each under an exec mask of its own: the
  // next row's bytes requested a trip ahead, the row's eight table reads in flight together, only the stores under the edge test:
  // 584.5 against 583.2 us, nothing either: profiles/r06_ab_callback_prio.txt)
#pragma unroll 1
  for (uint32_t r = wave; r < r_end; r += uint32_t(S::THREADS / 64)) {  // a wave takes every fourth row
    const uint32_t y = y0 + r;
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
#ifdef D2PC_DIAG
  {
    const unsigned long long e1 = __builtin_amdgcn_s_memtime();
    D2PC_BS_ADD(0, 1);
    D2PC_BS_ADD(5, e1 - e0);
    D2PC_BS_ADD(6, e1 - diag_t0);
    if (tid == 0) atomicAdd(&g_bs_diag[blockIdx.x & 255u][7], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - diag_r0));
  }
#endif
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
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(D2PC_BS_WAVES))) void k_callback_bs_compact(
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
// The first look at the words that place a tile (wave 0 of k_callback_bs_compact_pipe): the band accumulators above the
// tile's band (up to 128 bands: two per lane) and the row counts of the band's tiles (lane = 16 j + p reads dword p of
// tiles j, j + 4, ...: up to 4 per lane = 16 tiles across), requested in one place and examined in another.
struct CbEarlyPlace {
  using gu32 = __attribute__((address_space(1))) uint32_t;
  using gu64 = __attribute__((address_space(1))) uint64_t;
  static constexpr int kBands = 2, kTiles = 4;  // (images up to 4,096 x 4,096 output pixels; larger ones poll as before)
  uint64_t band[kBands];
  uint32_t row[kTiles];
  bool fits = false;
  __device__ __forceinline__ void request(const CbCompactState &cs, const MedianArgs &ma, uint32_t tile, uint32_t lane) {
    const uint32_t ty = tile / ma.tiles_x;
    fits = ty <= 64u * kBands && ma.tiles_x <= 4u * kTiles;
    if (!fits) return;
    const uint32_t j = lane >> 4, p = lane & 15u;
#pragma unroll
    for (int b = 0; b < kBands; ++b) {
      const uint32_t bi = 64u * uint32_t(b) + lane;
      band[b] = bi < ty ? __hip_atomic_load((gu64 *)(cs.band_acc + bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
#pragma unroll
    for (int i = 0; i < kTiles; ++i) {
      const uint32_t k = j + 4u * uint32_t(i);
      row[i] = k < ma.tiles_x ? __hip_atomic_load((gu32 *)(cs.row_cnt + (ty * ma.tiles_x + k) * 16u + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
  }
  // the same sums the polling loop forms; false when any word was not final yet (or nothing was requested)
  __device__ __forceinline__ bool take(const MedianArgs &ma, uint32_t ty, uint32_t tx, uint32_t lane, uint32_t &above, uint32_t &t0,
                                       uint32_t &t1, uint32_t &l0, uint32_t &l1) const {
    if (!fits) return false;
    const uint32_t j = lane >> 4;
    bool ok = true;
    above = 0;
#pragma unroll
    for (int b = 0; b < kBands; ++b) {
      const uint32_t bi = 64u * uint32_t(b) + lane;
      if (bi < ty) {
        ok = ok && uint32_t(band[b] >> 32) == ma.tiles_x;
        above += uint32_t(band[b]);
      }
    }
    t0 = t1 = l0 = l1 = 0;
#pragma unroll
    for (int i = 0; i < kTiles; ++i) {
      const uint32_t k = j + 4u * uint32_t(i);
      if (k < ma.tiles_x) {
        const uint32_t v = row[i];
        ok = ok && (v & kCbRowTag) != 0u;
        const uint32_t c0 = v & 0x1ffu, c1 = (v >> 9) & 0x1ffu;
        t0 += c0, t1 += c1;
        if (k < tx) l0 += c0, l1 += c1;
      }
    }
    return ok;
  }
};

template <int KS, int QK>
__global__ __launch_bounds__(MedianBsShape<KS>::THREADS) __attribute__((amdgpu_waves_per_eu(D2PC_BS_WAVES))) void k_callback_bs_compact_pipe(
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
  const uint32_t tid0 = threadIdx.x;
  const uint32_t tid = tid0, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

#ifdef D2PC_DIAG  // stage timers: tools/diag_callback.py
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long dA = 0, dB = 0, dC = 0, dD = 0, dE = 0, dN = 0;
#endif
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
    // Everything derived from the thread index is formed AGAIN in every iteration (opaque: the compiler cannot relate this
    // `tid` to the loop-invariant one).  Left to itself it hoists those values out of the loop -- the staging offsets of the
    // row loads and the thread's table entry 1/W, Z, class -- and, because the bit-sliced select needs every register of
    // the 168 that three waves per SIMD leave, parks them in scratch: 23-29 spilled VGPRs, ~25 scratch reloads per tile, each
    // a dependent round trip in front of the loads and table stores it feeds.  One fp64 division per thread and TILE is
    // cheaper: 16 x 4K with 30 % zero pixels + indices 766.8 -> 723.5 us (blocky), 813.9 -> 772.5 (iid), interleaved on one
    // device (profiles/r05_ab_callback_nohoist.txt); 0 spilled VGPRs, 0 bytes of scratch (round 4's verdict, item 3a).
    const uint32_t tid = opaque(tid0), lane = tid & 63u;
    uint32_t tk = 0, nxt = kNoTile;
    CbEarlyPlace early{};  // (born in the iteration: nothing of it is alive across the filter)
#ifdef D2PC_DIAG
    const unsigned long long f1 = __builtin_amdgcn_s_memtime();
    unsigned long long f0 = f1, f2 = f1, f3 = f1;
#endif
    if (cur != kNoTile) {
      // the ticket of the tile after `cur`: its round trip runs under the filter
      if (tid == 0) tk = atomicAdd(cs.ticket, 1u);
      const uint32_t ty = cur / ma.tiles_x, tx = cur - ty * ma.tiles_x;
      const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);
      // (Requesting the NEXT tile's rows one tile ahead, so that they would complete before this iteration's store
      // burst, was built and is slower -- 722 -> 788 us: the 12 registers it keeps across the phases spill.)
      median_bs_tile<KS>(fsrc, ma, int(x0), int(y0), s_w, s_raw, tid);
#ifdef D2PC_DIAG
      f0 = __builtin_amdgcn_s_memtime();
      ++dN;
#endif
      if (tid == 0) s_next = tk;
      // the words that place `prev` are requested HERE by wave 0 and looked at behind the count: their round trip runs
      // under the table, the count and two (LDS-only) barriers instead of in front of the scatter with three waves waiting
      if (wave == 0 && prev != kNoTile) early.request(cs, ma, prev, lane);
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
      lds_barrier();
#ifdef D2PC_DIAG
      f2 = __builtin_amdgcn_s_memtime();
#endif
      exact = s_exact != 0;
      nxt = s_next < tpf ? s_next : kNoTile;
      // ---- count: survivors per row of `cur` -------------------------------------------------------------
      const uint8_t *ob = reinterpret_cast<const uint8_t *>(s_w);
      if (is_stereo(QK) && !exact) {
        // a count needs no order: a lane takes 4 consecutive bytes of each of its wave's 8 rows (one word each, all eight
        // requested before the first is used) and the class of each byte from the table.  Row by row, column by column
        // (LDS byte -> table byte -> ballot, 32 dependent round trips per wave) this phase took 8,400 cycles per tile, now
        // 3,600 (profiles/r05_callback_phases.txt)
        constexpr int RPW = S::TH / (S::THREADS / 64);
        uint32_t wrd[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
          wrd[i] = reinterpret_cast<const uint32_t *>(ob + (wave + uint32_t(i) * uint32_t(S::THREADS / 64)) * uint32_t(S::OUT_STRIDE))[lane];
        const uint32_t xl = x0 + 4u * lane;
        // (the fences are scheduling boundaries: left to itself the compiler, short of registers for the select, issues
        //  one table read, waits for it, uses it, and so on -- 40 dependent LDS round trips per wave)
        asm volatile("" ::: "memory");
        uint32_t cl[RPW][4];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
          for (int b = 0; b < 4; ++b) cl[i][b] = lut_cls[(wrd[i] >> (8 * b)) & 0xffu];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const uint32_t r = wave + uint32_t(i) * uint32_t(S::THREADS / 64);
          uint32_t cnt = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) cnt += uint32_t(__popcll(__ballot(cl[i][b] == 1u && xl + uint32_t(b) < x_end)));
          if (lane == 0) s_cnt[r] = y0 + r < y_end ? cnt : 0u;
        }
      } else {
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
      }
      lds_barrier();
#ifdef D2PC_DIAG
      f3 = __builtin_amdgcn_s_memtime();
      dA += f2 - f0;
      dB += f3 - f2;
#endif
      // ---- publish `cur` (wave 1): sixteen tagged dwords + the band's accumulator; nothing to wait for.  Not wave 0: it
      // places `prev` at the same time, and a wave's vector memory operations retire in order -- the words it requested
      // for that would wait for these stores' acknowledgement ------
      if (wave == 1u) {
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
#ifdef D2PC_DIAG
    unsigned long long f5 = __builtin_amdgcn_s_memtime();
#endif
    if (prev != kNoTile) {
      // ---- place `prev` (wave 0): its band published a whole filter ago -----------------------------------
      const uint32_t ty = prev / ma.tiles_x, tx = prev - ty * ma.tiles_x;
      const uint32_t x0 = ma.out_x0 + tx * uint32_t(S::TW), y0 = ma.out_y0 + ty * uint32_t(S::TH);
      if (wave == 0) {
        const uint32_t j = lane >> 4, p = lane & 15u;
        uint32_t above = 0, t0 = 0, t1 = 0, l0 = 0, l1 = 0, spins = 0;
        uint64_t w0 = 0;
        bool first = cur != kNoTile;  // (the last tile of a block has no filter in front of its placing: nothing was requested)
        for (;;) {
          bool ok = true;
          above = 0;
          if (first) {
            first = false;
            ok = early.take(ma, ty, tx, lane, above, t0, t1, l0, l1);
            if (__all(ok)) break;
            goto look_again;
          }
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
        look_again:
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
#ifdef D2PC_DIAG
      f5 = __builtin_amdgcn_s_memtime();
#endif
      // ---- scatter `prev` from the kept bytes: the same decisions, the points, their ordered stores ----------
      const uint8_t *kb = reinterpret_cast<const uint8_t *>(s_keep);
      // a lane's four columns do not change from row to row: (u + cx) is formed once per tile (two fp64-rate instructions
      // fewer per pixel)
      [[maybe_unused]] double xs4[4] = {0.0, 0.0, 0.0, 0.0};
      if constexpr (is_stereo(QK)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xs4[q] = stereo_nx(Q, x0 + 64u * uint32_t(q) + lane);
      }
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
        // the usual case (a calibrated Q, no sliver): the row's twelve table reads in flight together behind scheduling fences
        // (see the count phase), the products only where a point is stored: -1 % of the launch (r05_ab_callback_stages.txt)
        if constexpr (is_stereo(QK)) if (!exact) {
          double t_iw[4];
          float t_z[4];
          uint32_t t_cls[4];
          asm volatile("" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            t_iw[q] = lut_iw[raw[q]];
            t_z[q] = lut_z[raw[q]];
            t_cls[q] = lut_cls[raw[q]];
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t x = x0 + 64u * uint32_t(q) + lane;
            const float X = float(xs4[q] * t_iw[q]), Y = float(ys * t_iw[q]), Z = t_z[q];
            const bool ok = t_cls[q] == 1u && x < x_end;
            const uint64_t m = __ballot(ok);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), row_pos));
            if (ok && pos < g.roi_n) {
              store_point<D2PC_CB_STORE_NT != 0>(fout, pos, X, Y, Z);
              if (fidx) st<D2PC_CB_INDEX_NT != 0>(fidx + pos, y * g.width + x);
            }
            row_pos += uint32_t(__popcll(m));
          }
          continue;
        }
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
#ifdef D2PC_DIAG
    const unsigned long long f6 = __builtin_amdgcn_s_memtime();
#endif
    // (LDS-only barriers from here to the filter: a full one would hold every wave until its stores have drained)
    lds_barrier();  // the scatter has read s_keep and the tables; the count has read s_w
    if (cur != kNoTile) {
      // `cur` becomes the tile in waiting: its bytes move out of the filter's way
      for (uint32_t i = tid; i < KEEP_WORDS / 4u; i += uint32_t(S::THREADS))
        reinterpret_cast<uint4 *>(s_keep)[i] = reinterpret_cast<const uint4 *>(s_w)[i];
    }
    lds_barrier();
#ifdef D2PC_DIAG
    dC += f5 - f3;
    dD += f6 - f5;
    dE += __builtin_amdgcn_s_memtime() - f6;
#endif
    prev = cur;
    cur = nxt;
  }
#ifdef D2PC_DIAG
  {
    const uint32_t tid = tid0;
    D2PC_BS_ADD(0, dN);
    D2PC_BS_ADD(5, dA);
    D2PC_BS_ADD(8, dB);
    D2PC_BS_ADD(9, dC);
    D2PC_BS_ADD(10, dD);
    D2PC_BS_ADD(11, dE);
    D2PC_BS_ADD(6, __builtin_amdgcn_s_memtime() - diag_t0);
    if (tid == 0) atomicAdd(&g_bs_diag[blockIdx.x & 255u][7], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - diag_r0));
  }
#endif
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
hipError_t launch_callback_bs(const LaunchArgs &a, MedianArgs m, const void *src, int ksize) {
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
  hipLaunchKernelGGL((k_callback_bs<KS, QK>), dim3(uint32_t(blocks)), dim3(S::THREADS), D2PC_CB_LDS_PAD, a.stream, s8, o, a.out_index, \
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
}

size_t callback_compact_state_bytes(uint32_t tiles_x, uint32_t tiles_y, uint32_t n_frames, uint32_t *frame_stride) {
  const uint64_t b = uint64_t(kCbTicketBytes) + cb_band_acc_bytes(tiles_y) + uint64_t(tiles_x) * tiles_y * 64u;
  const uint32_t stride = uint32_t((b + 255) & ~uint64_t(255));
  if (frame_stride) *frame_stride = stride;
  return sizeof(StateHeader) + size_t(n_frames) * stride;
}

hipError_t launch_callback_bs_compact(const LaunchArgs &a, MedianArgs m, const void *src, int ksize) {
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
  if (hipError_t e = launch_state_clear(a.state, a.state_bytes, a.stats, a.stream, a.keep_timeout); e != hipSuccess) return e;
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
}


}  // namespace d2pc

#ifdef D2PC_DIAG
// diagnostic build only (tools/diag_callback.py): read and reset the stage timers of the tile body
extern "C" int d2pc_debug_read_bs_diag(unsigned long long *out32) {
  static unsigned long long all[256][32];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(d2pc::g_bs_diag), sizeof all) != hipSuccess) return 6;
  for (int i = 0; i < 32; ++i) {
    out32[i] = 0;
    for (int s = 0; s < 256; ++s) out32[i] += all[s][i];
  }
  for (auto &row : all)
    for (auto &x : row) x = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(d2pc::g_bs_diag), all, sizeof all) != hipSuccess) return 6;
  return 0;
}
#endif
