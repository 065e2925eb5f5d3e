// d2pc_compact_common.hpp -- building blocks of the COMPACT kernels (order-preserving validity compaction: wave
// ballot + mbcnt ranks, LDS scan of a block's (slot, wave) counts, counted prefixes across tiles so that the output
// order equals the CPU loop's row-major order, cpp:70-76, bit for bit).
#pragma once

#include "d2pc_pixel.hpp"

namespace d2pc {

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

constexpr uint32_t kNoTile = 0xffffffffu;

}  // namespace d2pc
