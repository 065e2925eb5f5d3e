// d2pc_compact.hip -- COMPACT mode (the north-star's extension: order-preserving removal of points with a non-finite
// coordinate or d <= min_disparity; output order == the CPU loop's, cpp:70-76): the two-pass form, the one-launch
// resident forms, and the dispatcher over every COMPACT algorithm.  The single pass lives in d2pc_onepass.hip.
#include "d2pc_compact_common.hpp"

namespace d2pc {

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
      // (a budget of ZERO ticks -- only the test hook "handoff_spin_ticks_first" sets it -- gives up at the first look that fails)
      if (((spins & 7u) == 0 || g.spin_ticks == 0u) && (__builtin_amdgcn_s_memrealtime() - w0 > uint64_t(g.spin_ticks) ||
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
      // (a budget of ZERO ticks -- only the test hook "handoff_spin_ticks_first" sets it -- gives up at the first look that fails)
      if (((spins & 7u) == 0 || g.spin_ticks == 0u) && (__builtin_amdgcn_s_memrealtime() - w0 > uint64_t(g.spin_ticks) ||
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
// launchers
// --------------------------------------------------------------------------
template <int PXT>
static hipError_t launch_tiles(const LaunchArgs &a) {
  return for_q_kind(a.q_kind, [&](auto qk) {
    return for_dtype_vec(a, [&](auto dt, auto vec) {
      constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
      constexpr bool VEC = decltype(vec)::value;
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
        return hipErrorInvalidValue;
      }
      return hipGetLastError();
    });
  });
}

template <int R>
static hipError_t launch_resident_lean(const LaunchArgs &a) {
  if (a.grid != a.geom.total_tiles || !a.stats || a.epoch < kEpochBase || a.geom.pxt != uint32_t(R)) return hipErrorInvalidValue;
  return for_q_kind(a.q_kind, [&](auto qk) {
    return for_dtype(a.dtype, [&](auto dt) {
      constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
      hipLaunchKernelGGL((k_compact_resident_lean<DT, QK, R>), dim3(a.grid), dim3(kBlock), 0, a.stream,
                         static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts,
                         static_cast<uint8_t *>(a.state), static_cast<CompactStats *>(a.stats), a.geom, make_qarg<QK>(a), a.epoch);
      return hipGetLastError();
    });
  });
}

// Tile shapes of the two-pass form, the resident form and the single pass: 2,048 pixels (8 per thread) in the product;
// the experiment build also has 1,024 and 4,096 (swept in rounds 1-3: never better on 4K or 1080p frames).
bool tile_shape_supported(int pxt) { return pxt == 8 || (D2PC_EXPERIMENTS && (pxt == 4 || pxt == 16)); }

uint32_t frame_state_stride(uint32_t tiles_per_frame) {
  const uint32_t groups = (tiles_per_frame + kGroupTiles - 1) / kGroupTiles;
  const uint64_t b = kFrameTicketBytes + uint64_t(groups) * kGroupAccStride + uint64_t(tiles_per_frame) * 16;
  return uint32_t((b + 255) & ~uint64_t(255));
}

size_t compact_state_bytes(const Geom &g) {
  return sizeof(StateHeader) + size_t(g.n_frames) * g.frame_state_stride;
}

hipError_t launch_compact(const LaunchArgs &a) {
#if D2PC_EXPERIMENTS
  if (a.compact_algo == 4) return launch_compact_chunked(a);
#endif
  if (a.compact_algo == 3 && a.pxt == 32) return launch_resident_lean<32>(a);
  if (a.compact_algo == 3 && a.pxt == 64) return launch_resident_lean<64>(a);
  if (a.compact_algo == 2) return launch_onepass(a);
  if (a.compact_algo != 1 && a.compact_algo != 3) return hipErrorInvalidValue;
  switch (a.pxt) {
    case 8: return launch_tiles<8>(a);
#if D2PC_EXPERIMENTS
    case 4: return launch_tiles<4>(a);
    case 16: return launch_tiles<16>(a);
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace d2pc
