// d2pc_onepass.hip -- COMPACT mode in ONE pass over the input (compact_algo 2: the default for big batches):
// persistent 5-wave blocks, software-pipelined over their tiles, counts handed over between tiles inside the launch.
//
//   k_compact_onepass_dense<.., NW = 4, ..>   THE single pass of the product (round 5): the count phase packs each run's
//                                             survivors in LDS, the scatter phase runs dense; the launch also zeroes the
//                                             state of its successor
//   experiment build only (-DD2PC_EXPERIMENTS=1; recorded in profiles/r05_ab_onepass_forms_*.txt):
//   k_compact_onepass                         rounds 2-4: raw tiles in LDS, every pixel decided in both phases (form 1)
//   k_compact_onepass_dense<.., NW = 8, ..>   the dense pass on 4,096-pixel tiles with 8 worker waves (form 3: 25 % slower)
//   k_compact_onepass_lw                      the dense pass with the control wave as the block's loader (form 4: 5-10 % slower)
#include "d2pc_compact_common.hpp"

// A/B switches of the dense single pass (tools/ab.py; make variant NAME=x DEFS=-D...)
#ifndef D2PC_DENSE_ONE_BARRIER
#define D2PC_DENSE_ONE_BARRIER 0
#endif
#ifndef D2PC_DENSE_LAND_LATE
#define D2PC_DENSE_LAND_LATE 0
#endif
#ifndef D2PC_ONEPASS_CTL_PRIO
#define D2PC_ONEPASS_CTL_PRIO 0
#endif

namespace d2pc {
// --------------------------------------------------------------------------
// K2: single-pass compaction (each disparity is read once) -- the PROTOCOL and the PIPELINE every form below shares.
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
//  * Software pipeline over a block's tiles; the tiles in flight live in a
//    4-stage LDS ring, not in registers: iteration i COUNTS tile t (exact
//    validity predicate, a few operations per pixel for a stereoRectify-
//    structured Q), the control wave publishes t, fetches the ticket of t+1
//    and waits for the prefix of t-2 (published two iterations ago, so
//    normally ready at the first look); then t+1's loads are issued and
//    tile t-2 is reprojected and scattered.
// The product's kernel is K2d (k_compact_onepass_dense) below; "K2" alone names
// round 4's form of the two phases (k_compact_onepass, experiment build).
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

// A wave-uniform pointer moved into vector registers: the single-pass kernel runs out of scalar registers,
// and a spilled scalar base costs a v_readlane pair before every store.  With the base in VGPRs the store
// address is one v_lshl_add_u64.
__device__ __forceinline__ uint64_t vgpr_pointer(const void *p) {
  const uint64_t x = reinterpret_cast<uint64_t>(p);
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=v"(hi) : "s"(uint32_t(x)), "s"(uint32_t(x >> 32)));
  return (uint64_t(hi) << 32) | lo;
}

// --------------------------------------------------------------------------
// K2d: the single pass with INPUT-SIDE compaction (round 5; the product's).  Same protocol, same pipeline, same bytes out
// as round 4's K2; what differs is what waits in LDS between a tile's count and its scatter.  K2 kept the tile's raw disparities and
// decided every pixel a second time in the scatter phase: all 8 slots of a wave ran the fp64 division and issued a
// store instruction whose lanes were ragged -- with 30 % of the pixels invalid the launch executed the instructions and
// the L1 -> L2 write requests of an all-valid one (profiles/r04_compact_counters.json), with 90 % invalid it still took
// 340 us per 16 x 4K (the scatter phase alone 5,100 of 10,000 cycles per iteration: profiles/r05_onepass_phases.txt).
// Here the COUNT phase -- which knows every pixel's validity anyway -- packs the survivors of each run of 256 consecutive
// pixels to the front of the run's own LDS slice (4 bytes of disparity + 1 byte of offset inside the run: round 2
// compacted the 16-byte POINTS, four times the LDS traffic, and lost).  The scatter phase then walks ceil(c / 64) DENSE
// slots per run instead of 4: every lane holds a survivor, the arithmetic runs for survivors only, and a wave's store is
// one contiguous piece of 64 points (1 KiB) at prefix + j -- no ragged pieces, no ballot, no rank, no second decision.
//   tile  = NW worker waves x 2 runs x 256 pixels (NW = 4: the 2,048-pixel tile of K2; NW = 8: 4,096 pixels, i.e. the
//           control chain -- ticket, polls, two barriers -- amortised over twice the pixels at the same registers per lane)
//   run r = b * NW + wave (b = 0, 1) holds pixels [tile + 256 r, tile + 256 r + 256): pixel order == run order
// The host geometry is the one of 256 * (2 NW) pixels per tile (pxt = 2 NW).
// --------------------------------------------------------------------------
// inclusive scan of up to 16 cells held one per lane (lanes 0 .. CELLS-1 of a DPP row): three or four row shifts, no LDS
template <int CELLS>
__device__ __forceinline__ uint32_t scan_row(uint32_t c, uint32_t &total) {
  static_assert(CELLS <= 16, "one DPP row");
  uint32_t incl = c;
  incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x111, 0xf, 0xf, false));  // row_shr:1
  incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x112, 0xf, 0xf, false));  // row_shr:2
  incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x114, 0xf, 0xf, false));  // row_shr:4
  if (CELLS > 8) incl += uint32_t(__builtin_amdgcn_update_dpp(0, int(incl), 0x118, 0xf, 0xf, false));  // row_shr:8
  total = uint32_t(__builtin_amdgcn_readlane(int(incl), CELLS - 1));
  return incl;
}

// Where a dense single-pass launch leaves clean state for the next launch of its buffer (see the kernel's first lines).
struct SelfClean {
  uint4 *other;          // the buffer's other half (nullptr: nothing to clean -- a captured launch, which replays on ONE half)
  uint32_t n16;          // its size in 16-byte pieces
  uint32_t count_launch; // this launch was not preceded by k_state_clear: count it here
  CompactStats *stats;
};

template <int DT, int NW, bool VEC, int RPW = 2>
struct RunFetch {
  v4f q[VEC ? RPW : 1];
  float d[VEC ? 1 : 4 * RPW];
  // requests the wave's two runs of tile `base` (all loads in flight; nothing is waited for here)
  __device__ __forceinline__ void issue(const uint8_t *fin, const Geom &g, uint32_t base, uint32_t wave, uint32_t lane) {
#pragma unroll
    for (int b = 0; b < RPW; ++b) {
      const uint32_t run_base = base + (uint32_t(b) * uint32_t(NW) + wave) * 256u;
      if constexpr (VEC) {
        const uint32_t i = run_base + lane * 4u;
        const uint32_t v = fdiv(i, g.div_roi_w), u = i - v * g.roi_w;
        const uint32_t off = (v + g.border) * g.row_stride + (u + g.border) * 4u;
        const uint32_t last4 = g.last_off - 12u;  // the frame's last aligned group: tails load in bounds and count nothing
        q[b] = ld(reinterpret_cast<const v4f *>(fin + (off < last4 ? off : last4)));
      } else {
        Walker w(g, run_base + lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t off = (w.v + g.border) * g.row_stride + (w.u + g.border) * elem_bytes<DT>();
          d[b * 4 + k] = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
          w.step(g, g.s64_v, g.s64_u);
        }
      }
    }
  }
  // lands them, pixel-linear, in the wave's two run slices of an LDS stage
  __device__ __forceinline__ void finish(float *stage, uint32_t wave, uint32_t lane) const {
#pragma unroll
    for (int b = 0; b < RPW; ++b) {
      float *run = stage + (uint32_t(b) * uint32_t(NW) + wave) * 256u;
      if constexpr (VEC) {
        *reinterpret_cast<v4f *>(run + lane * 4u) = q[b];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) run[uint32_t(k) * 64u + lane] = d[b * 4 + k];
      }
    }
  }
};

// Count phase of one run: decides its 256 pixels (the exact predicate of tile_count; the real arithmetic for a general Q
// or a run that holds a sliver), packs the survivors' disparities to the front of the run's slice IN PLACE (all four
// slots are read before the first is written, and a survivor's rank never exceeds its pixel's offset; LDS is in-order
// per wave) with their offsets inside the run beside them, and returns the survivors' number (wave-uniform).
template <int QK>
__device__ __forceinline__ uint32_t run_count_pack(const QArg<QK> &Q, const Geom &g, float *run, uint8_t *off, uint32_t run_base,
                                                   uint32_t lane) {
  float d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = run[uint32_t(k) * 64u + lane];
  const uint32_t left = run_base < g.roi_n ? g.roi_n - run_base : 0u;  // pixels of the frame from this run on (ragged: a frame's last tile)
  bool ok[4];
  bool exact = !is_stereo(QK);
  if constexpr (is_stereo(QK)) {
    uint64_t sliver = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double nw = stereo_nw(Q, d[k]);
      const bool fin = finite_nonzero(nw), big = fabs(nw) >= Q.s.w_safe;
      ok[k] = (int(fin) & int(big) & int(!(d[k] <= g.min_disparity)) & int(uint32_t(k) * 64u + lane < left)) != 0;
      sliver |= __ballot(int(fin) & int(!big));
    }
    exact = sliver != 0;
  }
  if (exact) {  // (wave-uniform)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ok[k] = pixel_valid_exact<QK>(Q, g, run_base + uint32_t(k) * 64u + lane, d[k]) && uint32_t(k) * 64u + lane < left;
  }
  uint32_t c = 0;  // (scalar: survivors of the slots before k)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint64_t m = __ballot(ok[k]);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), c));
    if (ok[k]) {
      run[rank] = d[k];
      off[rank] = uint8_t(uint32_t(k) * 64u + lane);
    }
    c += uint32_t(__popcll(m));
  }
  return c;
}

// Scatter phase of one run: its c survivors, dense, to out[pos0 ...) in order.
template <int QK, bool IDX>
__device__ __forceinline__ void run_scatter_dense(const QArg<QK> &Q, const Geom &g, const float *run, const uint8_t *off, uint32_t c,
                                                  uint32_t pos0, uint32_t run_base, uint32_t lane, uint64_t fout, uint64_t fidx) {
  // the run's first pixel (wave-uniform: scalar arithmetic); a run of 256 pixels wraps at most once when the ROI is at
  // least 256 pixels wide, and takes the division per pixel otherwise
  const uint32_t v0 = fdiv(run_base, g.div_roi_w), u0 = run_base - v0 * g.roi_w;
  const bool wide = g.roi_w >= 256u;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (uint32_t(s) * 64u >= c) break;  // (wave-uniform)
    // (starting every store instruction on a 64-byte line of the output -- the run's first pos0 & 3 lanes idle, 16 whole
    // lines per instruction instead of 17 partial ones -- changes nothing: L2 merges the shared end lines of plain stores;
    // profiles/r05_ab_dense_variants.txt)
    const uint32_t j = uint32_t(s) * 64u + lane;
    const float d = run[j];       // (lanes past c read what the pack left there: computed, never stored)
    const uint32_t o = off[j];
    uint32_t u, v;
    if (wide) {
      u = u0 + o;
      v = v0;
      if (u >= g.roi_w) u -= g.roi_w, v += 1u;
    } else {
      const uint32_t i = run_base + o;
      v = fdiv(i, g.div_roi_w);
      u = i - v * g.roi_w;
    }
    const uint32_t uu = u + g.border, vv = v + g.border;
    float X, Y, Z;
    if constexpr (is_stereo(QK)) {  // (a survivor's d is finite: reproject()'s poisoning of a non-finite d has nothing to do)
      const double iw = 1.0 / stereo_nw(Q, d);
      X = float(stereo_nx(Q, uu) * iw);
      Y = float(stereo_ny(Q, vv) * iw);
      Z = big_z_rule(d, float(Q.s.f * iw));
    } else {
      reproject(Q, uu, vv, d, X, Y, Z);
    }
    const uint32_t pos = pos0 + j;
    // pos < roi_n always holds for a correct prefix; the guard keeps a stale or timed-out prefix from ever becoming an
    // out-of-bounds store
    if (j < c && pos < g.roi_n) {
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

// RPW: runs of 256 pixels per worker wave and tile (2: tiles of 512 NW pixels).  DEFER: a tile's loads fly for a whole
// iteration (its ticket is taken two tiles ahead of its count; one barrier per iteration).
template <int DT, int QK, int NW, bool VEC, int RPW = 2, bool DEFER = false>
__global__ __launch_bounds__(64 * (NW + 1)) void k_compact_onepass_dense(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                                         uint32_t *__restrict__ out_index,
                                                                         uint32_t *__restrict__ counts, uint8_t *state, const Geom g,
                                                                         const QArg<QK> Q, const SelfClean sc) {
  using gu64 = __attribute__((address_space(1))) uint64_t;
  constexpr int RUNS = RPW * NW;
  constexpr uint32_t TILE = uint32_t(RUNS) * 256u;
  static_assert(RUNS <= 64, "one wave scans a tile's run counts");
  __shared__ uint32_t s_cnt2[2][RUNS];  // (two buffers: with ONE barrier per iteration the control wave scans tile t's counts while the workers count t + 1)
  // exclusive offsets, survivor counts and totals of the last three counted tiles (counted in iteration it, scattered in it + 2)
  __shared__ uint32_t s_excl[3][RUNS], s_rcnt[3][RUNS];
  __shared__ uint32_t s_total[3], s_next[2], s_prefix[2];
  // four tiles in flight (being fetched / counted / waiting / scattered): raw pixels until the count phase has packed them
  __shared__ __attribute__((aligned(16))) float s_tile[4 * RUNS * 256];
  __shared__ uint8_t s_off[3 * RUNS * 256];  // the survivors' offsets inside their runs, for the three counted tiles
  __shared__ uint32_t s_stat[3];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool ctl = wave == uint32_t(NW);  // the last wave
#if D2PC_ONEPASS_CTL_PRIO
  if (ctl) __builtin_amdgcn_s_setprio(3);  // (A/B, round 6: the control wave's few instructions ahead of the workers' on its SIMD)
#endif
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  PollStats polls{s_stat};
  if (tid < 3) s_stat[tid] = 0;
  // The state of the NEXT launch on this buffer (its other half, which nothing reads or writes during this launch) is
  // zeroed here, a few 16-byte stores per block, under the first ticket's round trip -- instead of by a kernel of its own
  // in front of every launch (k_state_clear: ~4.3 us + a kernel boundary, 1.7 % of a 16 x 4K launch, 3 % of 32 x 1080p)
  if (sc.other) {
    constexpr uint32_t kHdr16 = uint32_t(sizeof(StateHeader) / 16);
    for (uint32_t i = blockIdx.x * blockDim.x + tid; i < sc.n16; i += gridDim.x * blockDim.x) {
      if (i >= kHdr16) sc.other[i] = uint4{0u, 0u, 0u, 0u};
    }
    if (blockIdx.x == 0 && tid == 0) {
      StateHeader fresh{};
      fresh.stats = sc.stats;
      *reinterpret_cast<StateHeader *>(sc.other) = fresh;
      if (sc.count_launch) atomicAdd(&sc.stats->launches, 1ull);  // (k_state_clear counts it when it runs)
    }
  }
#ifdef D2PC_DIAG  // phase timers (shader clock), as in k_compact_onepass: tools/diag_onepass.py
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, nIt = 0;
  [[maybe_unused]] unsigned long long tS0 = 0, tS1 = 0, tS2 = 0;
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#define D2PC_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define D2PC_STAMP(x)
#endif

  // a block serves ONE frame (the launcher sizes the grid to a multiple of n_frames).  Blocks are dealt to the 8 XCDs
  // round-robin, so with a multiple of 8 frames all the blocks of a frame share an XCD and its L2: their hand-off words
  // are then seen 0.2-0.6 us after the store instead of 0.7-1.0 us across XCDs (profiles/r05_xcd_handoff.txt)
  const uint32_t f = blockIdx.x % g.n_frames;
  const FrameState fs(state, g, f);
  const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;

  if (ctl && lane == 0) {
    s_next[1] = atomicAdd(fs.ticket, 1u);
    if constexpr (DEFER) s_next[0] = atomicAdd(fs.ticket, 1u);
  }
  __syncthreads();
  uint32_t cur = s_next[1];
  if (cur >= g.tiles_per_frame) cur = kNoTile;
  uint32_t prev = kNoTile, prev2 = kNoTile;  // counted one / two iterations ago; prev2 is scattered now
  KnownGroups known;
  RunFetch<DT, NW, VEC, RPW> fetch;
  const uint64_t vout = vgpr_pointer(fout), vidx = vgpr_pointer(fidx);
  if (!ctl && cur != kNoTile) {
    fetch.issue(fin, g, cur * TILE, wave, lane);
    fetch.finish(s_tile, wave, lane);  // iteration 0 counts stage 0
  }
  // `nx` is counted in the NEXT iteration: its loads were issued one iteration ago and land behind this iteration's
  // barrier, so they have had a whole count phase + barrier to arrive and the wait for them is static (vmcnt(0) with
  // nothing younger outstanding but stores that are a scatter old)
  uint32_t nx = kNoTile;
  if constexpr (DEFER) {
    if (cur != kNoTile) nx = s_next[0];
    if (nx >= g.tiles_per_frame) nx = kNoTile;
    __syncthreads();  // (s_next[0] is written again in iteration 0)
    if (!ctl && nx != kNoTile) fetch.issue(fin, g, nx * TILE, wave, lane);
  }

  for (uint32_t it = 0; cur != kNoTile || prev != kNoTile || prev2 != kNoTile; ++it) {
    const uint32_t slot = it & 1u;
    const uint32_t ring = it % 3u, ring2 = (it + 1u) % 3u;  // this iteration's tile / the tile two iterations back
    D2PC_STAMP(c0);
    if (ctl) {
      uint32_t tk = 0;
      const bool more = (DEFER ? nx : cur) != kNoTile;
#if defined(D2PC_DIAG) && defined(D2PC_DIAG_CTL_SPLIT)
      // the control wave's three round trips one after the other (NOT the product's order): what is left of the publishing
      // stores' acknowledgement, the prefix poll alone, the ticket atomic alone (profiles/r05_ctl_split.txt)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long s0 = __builtin_amdgcn_s_memtime();
      if (prev2 != kNoTile) {
        const uint32_t p = prefix_before<true>(fs, hdr, prev2, lane, polls, known, g.spin_ticks);
        if (lane == 0) s_prefix[slot] = p;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long s1 = __builtin_amdgcn_s_memtime();
      if (more && lane == 0) tk = atomicAdd(fs.ticket, 1u);
      if (more && lane == 0) s_next[slot] = tk;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long s2 = __builtin_amdgcn_s_memtime();
      tS0 += s0 - c0, tS1 += s1 - s0, tS2 += s2 - s1;
#else
      // (the compiler's wave-aggregated atomicAdd reads its result back where it is issued -- s_waitcnt vmcnt(0) --, so the
      //  ticket's round trip, ~1,600 cycles, does NOT run under the polls: profiles/r05_ctl_split.txt)
      if (more && lane == 0) tk = atomicAdd(fs.ticket, 1u);
      if (prev2 != kNoTile) {
        const uint32_t p = prefix_before<true>(fs, hdr, prev2, lane, polls, known, g.spin_ticks);
        if (lane == 0) s_prefix[slot] = p;
      }
      if (more && lane == 0) s_next[slot] = tk;
#endif
#if D2PC_ONEPASS_STATS
      if (cur != kNoTile && lane == 0) s_stat[0] += 1u;
#endif
    } else if (cur != kNoTile) {
      float *stage = s_tile + (it & 3u) * uint32_t(RUNS * 256);
      uint8_t *offs = s_off + ring * uint32_t(RUNS * 256);
#pragma unroll
      for (int b = 0; b < RPW; ++b) {
        const uint32_t r = uint32_t(b) * uint32_t(NW) + wave;
        const uint32_t c = run_count_pack<QK>(Q, g, stage + r * 256u, offs + r * 256u, cur * TILE + r * 256u, lane);
        if (lane == 0) s_cnt2[slot][r] = c;
      }
    }
    D2PC_STAMP(c1);
    __syncthreads();
    D2PC_STAMP(c2);
    uint32_t next = kNoTile;
    if ((DEFER ? nx : cur) != kNoTile) {
      next = s_next[slot];
      if (next >= g.tiles_per_frame) next = kNoTile;
    }
    if constexpr (DEFER) {
      if (!ctl) {
        if (nx != kNoTile) fetch.finish(s_tile + ((it + 1u) & 3u) * uint32_t(RUNS * 256), wave, lane);  // issued an iteration ago
        if (next != kNoTile) fetch.issue(fin, g, next * TILE, wave, lane);                              // lands an iteration from now
      }
    } else {
      if (!ctl && next != kNoTile) fetch.issue(fin, g, next * TILE, wave, lane);  // fly while the control wave scans and publishes
    }
    if (ctl && cur != kNoTile) {
      const uint32_t c = lane < uint32_t(RUNS) ? s_cnt2[slot][lane] : 0u;
      uint32_t total;
      const uint32_t incl = scan_row<RUNS>(c, total);  // (three DPP row shifts; six ds_bpermute steps in scan_cells: ~700 cycles during which the workers wait)
      if (lane < uint32_t(RUNS)) {
        s_excl[ring][lane] = incl - c;
        s_rcnt[ring][lane] = c;
      }
      if (lane == 0) {
        s_total[ring] = total;
        __hip_atomic_store((gu64 *)(fs.granules + 2u * cur), kGranuleTag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add((gu64 *)fs.group_word(cur / kGroupTiles), (uint64_t(1) << 32) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#if !D2PC_DENSE_ONE_BARRIER
    if constexpr (!DEFER) __syncthreads();
#endif
    D2PC_STAMP(c3);
    if (!ctl) {
#if !D2PC_DENSE_LAND_LATE
      if (!DEFER && next != kNoTile) fetch.finish(s_tile + ((it + 1u) & 3u) * uint32_t(RUNS * 256), wave, lane);
#endif
#ifdef D2PC_DIAG
      {
        D2PC_STAMP(c3b);
        tE += c3b - c3;
      }
#endif
      if (prev2 != kNoTile) {
        const float *stage = s_tile + ((it + 2u) & 3u) * uint32_t(RUNS * 256);  // counted (and packed) in iteration it - 2
        const uint8_t *offs = s_off + ring2 * uint32_t(RUNS * 256);
        const uint32_t prefix = s_prefix[slot];
#pragma unroll
        for (int b = 0; b < RPW; ++b) {
          const uint32_t r = uint32_t(b) * uint32_t(NW) + wave;
          const uint32_t c = __builtin_amdgcn_readfirstlane(s_rcnt[ring2][r]);
          const uint32_t pos0 = prefix + __builtin_amdgcn_readfirstlane(s_excl[ring2][r]);
          if (fidx) run_scatter_dense<QK, true>(Q, g, stage + r * 256u, offs + r * 256u, c, pos0, prev2 * TILE + r * 256u, lane, vout, vidx);
          else run_scatter_dense<QK, false>(Q, g, stage + r * 256u, offs + r * 256u, c, pos0, prev2 * TILE + r * 256u, lane, vout, vidx);
        }
        if (counts && prev2 == g.tiles_per_frame - 1 && tid == 0) {
          // a frame whose hand-off broke reports kCountTimedOut instead of a count: visible in-band
          const bool bad = __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
          __hip_atomic_store(counts + f, bad ? kCountTimedOut : prefix + s_total[ring2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#if D2PC_DENSE_LAND_LATE
      // the next tile lands BEHIND the scatter: its loads have had the whole scatter to arrive (landed in front of it they
      // cost ~1,800 exposed cycles per iteration, profiles/r05_onepass_phases.txt); the wait now also covers this
      // iteration's stores (vector memory operations complete in order), which are the shorter round trip
      if (next != kNoTile) fetch.finish(s_tile + ((it + 1u) & 3u) * uint32_t(RUNS * 256), wave, lane);
#endif
    }
#ifdef D2PC_DIAG
    {
      D2PC_STAMP(c4);
      tA += c1 - c0;
      tB += c2 - c1;
      tC += c3 - c2;
      tD += c4 - c3;
      ++nIt;
    }
#endif
    prev2 = prev;
    prev = cur;
    if constexpr (DEFER) {
      cur = nx;
      nx = next;
    } else {
      cur = next;
    }
  }
  __syncthreads();
  if (counts && tid == 0 && __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
  if (ctl && lane == 0) {
    CompactStats::Slot *sl = hdr->stats->slot + (blockIdx.x % uint32_t(kStatSlots));
    atomicAdd(&sl->tiles, (unsigned long long)s_stat[0]);
    if (s_stat[1]) {
      atomicAdd(&sl->failed_polls, (unsigned long long)s_stat[1]);
      atomicAdd(&sl->wait_ticks, (unsigned long long)s_stat[2]);
    }
  }
#endif
#ifdef D2PC_DIAG
  if (lane == 0 && wave == 0) {
    atomicAdd(&hdr->diag[0], nIt);
    atomicAdd(&hdr->diag[2], tA);
    atomicAdd(&hdr->diag[3], tB);
    atomicAdd(&hdr->diag[4], tC);
    atomicAdd(&hdr->diag[5], tD);
    atomicAdd(&hdr->pad2[0], tE);
    atomicAdd(&hdr->pad2[1], (unsigned long long)(__builtin_amdgcn_s_memtime() - diag_t0));
    atomicAdd(&hdr->pad2[2], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - diag_r0));
    atomicAdd(&hdr->pad2[3], 1ull);
  }
  if (lane == 0 && ctl) {
    atomicAdd(&hdr->diag[1], (unsigned long long)s_stat[1]);
    atomicAdd(&hdr->diag[6], tA);
#ifdef D2PC_DIAG_CTL_SPLIT  // (in place of the control wave's other three phases: tools/diag_onepass.py prints them under those names)
    atomicAdd(&hdr->pad2[4], tS0);
    atomicAdd(&hdr->pad2[5], tS1);
    atomicAdd(&hdr->pad2[6], tS2);
#else
    atomicAdd(&hdr->pad2[4], tB);
    atomicAdd(&hdr->pad2[5], tC);
    atomicAdd(&hdr->pad2[6], tD);
#endif
  }
#endif
#undef D2PC_STAMP
}

#if D2PC_EXPERIMENTS
// ---- form 1 (rounds 2-4): raw tiles in LDS; both phases decide every pixel -------------------------------------------
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
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, nIt = 0;
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
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
#ifdef D2PC_DIAG
        {
          D2PC_STAMP(c3b);
          tE += c3b - c3;  // worker: the next tile's loads land (the wait for their data) and go to LDS
        }
#endif
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
        tD += c4 - c3;  // worker: next loads landing (tE of it) + reproject + scatter | control: idle until the next iteration
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
    atomicAdd(&hdr->pad2[0], tE);
    atomicAdd(&hdr->pad2[1], (unsigned long long)(__builtin_amdgcn_s_memtime() - diag_t0));      // block lifetime, shader cycles
    atomicAdd(&hdr->pad2[2], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - diag_r0));  // ... and 100 MHz ticks
    atomicAdd(&hdr->pad2[3], 1ull);                                                                // blocks
  }
  if (lane == 0 && ctl) {
    atomicAdd(&hdr->diag[1], (unsigned long long)s_stat[1]);
    atomicAdd(&hdr->diag[6], tA);  // control: ticket + prefix
    atomicAdd(&hdr->pad2[4], tB);  // control: waiting at barrier 1 for the workers' counts
    atomicAdd(&hdr->pad2[5], tC);  // control: scan + publish + barrier 2
    atomicAdd(&hdr->pad2[6], tD);  // control: nothing to do until the workers finish their scatter
  }
#endif
#undef D2PC_STAMP
}

// --------------------------------------------------------------------------
// K2l: the dense single pass with a LOADER wave (round 5).  In K2 / K2d a worker wave requests its share of the next
// tile the moment the ticket is known (behind barrier 1) and needs it for the next iteration's count: the load's round
// trip (~3,200 cycles under the launch's own write stream) is only partly covered -- the workers of K2d sit ~1,400 cycles
// at barrier 2 and ~1,800 more waiting for their loads in every iteration of ~11,000 (profiles/r05_onepass_phases.txt) --
// and it cannot be moved behind the scatter either: vector memory operations of a wave complete IN ORDER, so a wait for
// loads issued in front of the scatter's stores is a wait for those stores as well (variant "ll": slower).
// Here the CONTROL wave, which knows the ticket first and otherwise waits, is the block's loader: it takes the ticket
// one iteration earlier, requests the whole tile (NW x 2 runs) right behind its polls, does its scan and publish while the
// loads fly and writes them to the LDS stage before barrier 2.  Its own in-order queue holds loads and one atomic, no
// stores; the workers' queues hold stores only, which nobody ever waits for.  A worker iteration is count + scatter.
//   ticket (it) -> loads (it + 1) -> count (it + 2) -> scatter (it + 4)
// Protocol, state layout and output: K2's.  Tile = 2,048 pixels (NW = 4).
// --------------------------------------------------------------------------
template <int DT, int NW, bool VEC>
struct TileFetchAll {
  static constexpr int RUNS = 2 * NW;
  v4f q[VEC ? RUNS : 1];
  float d[VEC ? 1 : RUNS * 4];
  __device__ __forceinline__ void issue(const uint8_t *fin, const Geom &g, uint32_t base, uint32_t lane) {
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
      const uint32_t run_base = base + uint32_t(r) * 256u;
      if constexpr (VEC) {
        const uint32_t i = run_base + lane * 4u;
        const uint32_t v = fdiv(i, g.div_roi_w), u = i - v * g.roi_w;
        const uint32_t off = (v + g.border) * g.row_stride + (u + g.border) * 4u;
        const uint32_t last4 = g.last_off - 12u;
        q[r] = ld(reinterpret_cast<const v4f *>(fin + (off < last4 ? off : last4)));
      } else {
        Walker w(g, run_base + lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t off = (w.v + g.border) * g.row_stride + (w.u + g.border) * elem_bytes<DT>();
          d[r * 4 + k] = load_disparity<DT>(fin, off < g.last_off ? off : g.last_off, g.scale);
          w.step(g, g.s64_v, g.s64_u);
        }
      }
    }
  }
  __device__ __forceinline__ void finish(float *stage, uint32_t lane) const {
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
      float *run = stage + uint32_t(r) * 256u;
      if constexpr (VEC) {
        *reinterpret_cast<v4f *>(run + lane * 4u) = q[r];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) run[uint32_t(k) * 64u + lane] = d[r * 4 + k];
      }
    }
  }
};

template <int DT, int QK, int NW, bool VEC>
__global__ __launch_bounds__(64 * (NW + 1)) void k_compact_onepass_lw(const uint8_t *__restrict__ disp, float4 *__restrict__ out,
                                                                      uint32_t *__restrict__ out_index, uint32_t *__restrict__ counts,
                                                                      uint8_t *state, const Geom g, const QArg<QK> Q) {
  using gu64 = __attribute__((address_space(1))) uint64_t;
  constexpr int RUNS = 2 * NW;
  constexpr uint32_t TILE = uint32_t(RUNS) * 256u;
  __shared__ uint32_t s_cnt2[2][RUNS];
  __shared__ uint32_t s_excl[3][RUNS], s_rcnt[3][RUNS];
  __shared__ uint32_t s_total[3], s_next[2], s_prefix[2];
  __shared__ __attribute__((aligned(16))) float s_tile[4 * RUNS * 256];
  __shared__ uint8_t s_off[3 * RUNS * 256];
  __shared__ uint32_t s_stat[3];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool ctl = wave == uint32_t(NW);  // the last wave: tickets, polls, publishing AND the block's loads
  StateHeader *hdr = reinterpret_cast<StateHeader *>(state);
  PollStats polls{s_stat};
  if (tid < 3) s_stat[tid] = 0;

  const uint32_t f = blockIdx.x % g.n_frames;
  const FrameState fs(state, g, f);
  const uint8_t *fin = disp + uint64_t(f) * g.in_frame_stride;
  float4 *fout = out + uint64_t(f) * g.out_frame_stride;
  uint32_t *fidx = out_index ? out_index + uint64_t(f) * g.out_frame_stride : nullptr;

  if (ctl && lane == 0) {
    s_next[0] = atomicAdd(fs.ticket, 1u);  // the block's first two tiles
    s_next[1] = atomicAdd(fs.ticket, 1u);
  }
  __syncthreads();
  uint32_t cur = s_next[0], nxt = s_next[1];
  if (cur >= g.tiles_per_frame) cur = kNoTile;
  if (nxt >= g.tiles_per_frame) nxt = kNoTile;
  uint32_t prev = kNoTile, prev2 = kNoTile;
  KnownGroups known;
  TileFetchAll<DT, NW, VEC> fetch;
  const uint64_t vout = vgpr_pointer(fout), vidx = vgpr_pointer(fidx);
  if (ctl && cur != kNoTile) {  // (the one load nobody can hide: the block's first tile)
    fetch.issue(fin, g, cur * TILE, lane);
    fetch.finish(s_tile, lane);
  }
  __syncthreads();

  for (uint32_t it = 0; cur != kNoTile || prev != kNoTile || prev2 != kNoTile; ++it) {
    const uint32_t slot = it & 1u;
    const uint32_t ring = it % 3u, ring2 = (it + 1u) % 3u;
    uint32_t tk = 0;
    if (ctl) {
      // polls first (their answers must not queue behind the tile's loads), then the loads of the NEXT tile, then the atomic
      // for the ticket of the tile after it: the loads complete in front of the atomic, whose round trip (2-3 us under load)
      // has until the end of the iteration
      if (prev2 != kNoTile) {
        const uint32_t p = prefix_before<true>(fs, hdr, prev2, lane, polls, known, g.spin_ticks);
        if (lane == 0) s_prefix[slot] = p;
      }
      if (nxt != kNoTile) {
        fetch.issue(fin, g, nxt * TILE, lane);
        if (lane == 0) tk = atomicAdd(fs.ticket, 1u);
      }
#if D2PC_ONEPASS_STATS
      if (cur != kNoTile && lane == 0) s_stat[0] += 1u;
#endif
    } else if (cur != kNoTile) {
      float *stage = s_tile + (it & 3u) * uint32_t(RUNS * 256);
      uint8_t *offs = s_off + ring * uint32_t(RUNS * 256);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t r = uint32_t(b) * uint32_t(NW) + wave;
        const uint32_t c = run_count_pack<QK>(Q, g, stage + r * 256u, offs + r * 256u, cur * TILE + r * 256u, lane);
        if (lane == 0) s_cnt2[slot][r] = c;
      }
    }
    __syncthreads();  // 1: the tile's run counts are in LDS; the prefix of the tile scattered now is known
    if (ctl) {
      if (cur != kNoTile) {
        const uint32_t c = lane < uint32_t(RUNS) ? s_cnt2[slot][lane] : 0u;
        uint32_t total;
        const uint32_t incl = scan_row<RUNS>(c, total);
        if (lane < uint32_t(RUNS)) {
          s_excl[ring][lane] = incl - c;
          s_rcnt[ring][lane] = c;
        }
        if (lane == 0) {
          s_total[ring] = total;
          __hip_atomic_store((gu64 *)(fs.granules + 2u * cur), kGranuleTag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gu64 *)fs.group_word(cur / kGroupTiles), (uint64_t(1) << 32) | total, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (nxt != kNoTile) {
        fetch.finish(s_tile + ((it + 1u) & 3u) * uint32_t(RUNS * 256), lane);  // counted in the next iteration
        if (lane == 0) s_next[slot] = tk;
      }
    } else if (prev2 != kNoTile) {
      const float *stage = s_tile + ((it + 2u) & 3u) * uint32_t(RUNS * 256);  // counted (and packed) in iteration it - 2
      const uint8_t *offs = s_off + ring2 * uint32_t(RUNS * 256);
      const uint32_t prefix = s_prefix[slot];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t r = uint32_t(b) * uint32_t(NW) + wave;
        const uint32_t c = __builtin_amdgcn_readfirstlane(s_rcnt[ring2][r]);
        const uint32_t pos0 = prefix + __builtin_amdgcn_readfirstlane(s_excl[ring2][r]);
        if (fidx) run_scatter_dense<QK, true>(Q, g, stage + r * 256u, offs + r * 256u, c, pos0, prev2 * TILE + r * 256u, lane, vout, vidx);
        else run_scatter_dense<QK, false>(Q, g, stage + r * 256u, offs + r * 256u, c, pos0, prev2 * TILE + r * 256u, lane, vout, vidx);
      }
      if (counts && prev2 == g.tiles_per_frame - 1 && tid == 0) {
        const bool bad = __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        __hip_atomic_store(counts + f, bad ? kCountTimedOut : prefix + s_total[ring2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();  // 2: the next tile has landed; its successor's ticket is in LDS
    uint32_t nn = kNoTile;
    if (nxt != kNoTile) {
      nn = s_next[slot];
      if (nn >= g.tiles_per_frame) nn = kNoTile;
    }
    prev2 = prev;
    prev = cur;
    cur = nxt;
    nxt = nn;
  }
  if (counts && tid == 0 && __hip_atomic_load(&hdr->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    __hip_atomic_store(counts + f, kCountTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if D2PC_ONEPASS_STATS
  if (ctl && lane == 0) {
    CompactStats::Slot *sl = hdr->stats->slot + (blockIdx.x % uint32_t(kStatSlots));
    atomicAdd(&sl->tiles, (unsigned long long)s_stat[0]);
    if (s_stat[1]) {
      atomicAdd(&sl->failed_polls, (unsigned long long)s_stat[1]);
      atomicAdd(&sl->wait_ticks, (unsigned long long)s_stat[2]);
    }
  }
#endif
}

#endif  // D2PC_EXPERIMENTS

// Zeroes the compaction state ahead of a single-pass launch and starts its header (the pointer to the context's
// counters; one launch counted).  A kernel of our own rather than hipMemsetAsync: captured into a hipGraph, the
// runtime's memset node left the state UNTOUCHED on replays when the graph was launched on another stream than it
// was captured on and the host had synchronised in between (the kernel node behind it found all of it dirty:
// profiles/r03_graph_memset.txt); a plain kernel node has exactly the ordering of the kernels around it.
__global__ __launch_bounds__(256) void k_state_clear(uint4 *__restrict__ p, uint32_t n16, CompactStats *stats, uint32_t keep_timeout) {
  constexpr uint32_t kHdr16 = uint32_t(sizeof(StateHeader) / 16);
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) {
    StateHeader fresh{};
    fresh.stats = stats;
    // (sub-batches of one d2pc_process_mono_device call: the flag covers the call, not its last sub-batch)
    if (keep_timeout) fresh.timeout = reinterpret_cast<const StateHeader *>(p)->timeout;
    *reinterpret_cast<StateHeader *>(p) = fresh;
    atomicAdd(&stats->launches, 1ull);
  } else if (i >= kHdr16 && i < n16) {
    p[i] = uint4{0u, 0u, 0u, 0u};
  }
}


hipError_t launch_state_clear(void *state, size_t state_bytes, void *stats, hipStream_t stream, bool keep_timeout) {
  const uint32_t n16 = uint32_t((state_bytes + 15) / 16);  // buffers are allocated in whole MiB
  hipLaunchKernelGGL(k_state_clear, dim3((n16 + 255) / 256), dim3(256), 0, stream, static_cast<uint4 *>(state), n16,
                     static_cast<CompactStats *>(stats), keep_timeout ? 1u : 0u);
  return hipGetLastError();
}

#if D2PC_EXPERIMENTS
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

#endif  // D2PC_EXPERIMENTS

template <int NW, int RPW = 2, bool DEFER = false>
static hipError_t launch_onepass_dense(const LaunchArgs &a, uint32_t grid) {
  if (a.geom.pxt != uint32_t(RPW * NW)) return hipErrorInvalidValue;  // the geometry of 256 RPW NW pixels per tile
  return for_q_kind(a.q_kind, [&](auto qk) {
    return for_dtype_vec(a, [&](auto dt, auto vec) {
      constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
      constexpr bool VEC = decltype(vec)::value;
      const SelfClean sc{static_cast<uint4 *>(a.state_other), uint32_t((a.state_bytes + 15) / 16), a.state_is_clean ? 1u : 0u,
                         static_cast<CompactStats *>(a.stats)};
      hipLaunchKernelGGL((k_compact_onepass_dense<DT, QK, NW, VEC, RPW, DEFER>), dim3(grid), dim3(64 * (NW + 1)), 0, a.stream,
                         static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts,
                         static_cast<uint8_t *>(a.state), a.geom, make_qarg<QK>(a), sc);
      return hipGetLastError();
    });
  });
}

hipError_t launch_onepass(const LaunchArgs &a) {
  // (a.state_is_clean: the previous dense launch on this buffer zeroed this half inside its own launch)
  if (!a.state_is_clean)
    if (hipError_t e = launch_state_clear(a.state, a.state_bytes, a.stats, a.stream); e != hipSuccess) return e;
  // frame-static assignment: a block serves frame blockIdx % n_frames, so the grid is a multiple of
  // n_frames (the C ABI falls back to the two-pass form when there are more frames than blocks)
  uint32_t grid = a.grid;
  if (grid < a.geom.n_frames) return hipErrorInvalidValue;
  grid -= grid % a.geom.n_frames;
  if (a.onepass_form == 2) return launch_onepass_dense<4>(a, grid);
#if D2PC_EXPERIMENTS
  if (a.onepass_form == 3) return launch_onepass_dense<8>(a, grid);
  if (a.onepass_form == 5) return launch_onepass_dense<4, 4>(a, grid);        // 4 workers x 4 runs: tiles of 4,096 pixels
  if (a.onepass_form == 6) return launch_onepass_dense<4, 2, true>(a, grid);  // deferred landing
  if (a.onepass_form == 7) return launch_onepass_dense<4, 4, true>(a, grid);  // both
  if (a.onepass_form == 4) {
    if (a.geom.pxt != 8u) return hipErrorInvalidValue;
    return for_q_kind(a.q_kind, [&](auto qk) {
      return for_dtype_vec(a, [&](auto dt, auto vec) {
        constexpr int QK = decltype(qk)::value, DT = decltype(dt)::value;
        constexpr bool VEC = decltype(vec)::value;
        hipLaunchKernelGGL((k_compact_onepass_lw<DT, QK, 4, VEC>), dim3(grid), dim3(64 * 5), 0, a.stream,
                           static_cast<const uint8_t *>(a.disp), static_cast<float4 *>(a.out_points), a.out_index, a.counts,
                           static_cast<uint8_t *>(a.state), a.geom, make_qarg<QK>(a));
        return hipGetLastError();
      });
    });
  }
  if (a.onepass_form == 1) {
    switch (a.pxt) {
      case 8: return launch_onepass_tiles<8>(a, grid);
      case 4: return launch_onepass_tiles<4>(a, grid);
      case 16: return launch_onepass_tiles<16>(a, grid);
    }
  }
#endif
  return hipErrorInvalidValue;
}

}  // namespace d2pc
