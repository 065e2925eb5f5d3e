// d2pc_onepass.hip -- COMPACT mode in ONE pass over the input (compact_algo 2: the default for big batches):
// persistent 5-wave blocks, software-pipelined over their tiles, counts handed over between tiles inside the launch.
#include "d2pc_compact_common.hpp"

namespace d2pc {

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


hipError_t launch_state_clear(void *state, size_t state_bytes, void *stats, hipStream_t stream) {
  const uint32_t n16 = uint32_t((state_bytes + 15) / 16);  // buffers are allocated in whole MiB
  hipLaunchKernelGGL(k_state_clear, dim3((n16 + 255) / 256), dim3(256), 0, stream, static_cast<uint4 *>(state), n16,
                     static_cast<CompactStats *>(stats));
  return hipGetLastError();
}

template <int PXT>
static hipError_t launch_onepass_tiles(const LaunchArgs &a, uint32_t grid) {
  return for_q_kind(a.q_kind, [&](auto qk) {
    return for_dtype_vec(a, [&](auto dt, auto vec) {
      constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
      constexpr bool VEC = decltype(vec)::value;
      hipLaunchKernelGGL((k_compact_onepass<DT, QK, PXT, VEC>), dim3(grid), dim3(kBlock + 64), 0, a.stream,
                         static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts,
                         static_cast<uint8_t *>(a.state), a.geom, make_qarg<QK>(a));
      return hipGetLastError();
    });
  });
}

hipError_t launch_onepass(const LaunchArgs &a) {
  if (hipError_t e = launch_state_clear(a.state, a.state_bytes, a.stats, a.stream); e != hipSuccess) return e;
  // frame-static assignment: a block serves frame blockIdx % n_frames, so the grid is a multiple of
  // n_frames (the C ABI falls back to the two-pass form when there are more frames than blocks)
  uint32_t grid = a.grid;
  if (grid < a.geom.n_frames) return hipErrorInvalidValue;
  grid -= grid % a.geom.n_frames;
  switch (a.pxt) {
    case 8: return launch_onepass_tiles<8>(a, grid);
#if D2PC_EXPERIMENTS
    case 4: return launch_onepass_tiles<4>(a, grid);
    case 16: return launch_onepass_tiles<16>(a, grid);
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace d2pc
