// d2pc_chunk.hip -- EXPERIMENT BUILD ONLY (-DD2PC_EXPERIMENTS=1): the chunked two-pass compaction of round 4
// (compact_algo 4).  Bit-identical to the other COMPACT forms and 15-25 % slower than the single pass on every big
// batch measured (profiles/r04_ab_chunked.txt); kept as a recorded negative that tests and tools can still run.
#include "d2pc_compact_common.hpp"

#if D2PC_EXPERIMENTS
namespace d2pc {

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
hipError_t launch_compact_chunked(const LaunchArgs &a) {
  switch (a.q_kind) {
    case QK_STEREO: return launch_compact_chunked_q<QK_STEREO>(a);
    case QK_STEREO_CV24: return launch_compact_chunked_q<QK_STEREO_CV24>(a);
    case QK_STEREO_CV4: return launch_compact_chunked_q<QK_STEREO_CV4>(a);
    case QK_GENERAL: return launch_compact_chunked_q<QK_GENERAL>(a);
  }
  return hipErrorInvalidValue;
}

uint32_t chunk_frame_state_stride(uint32_t tiles_per_frame, uint32_t *gsum_words) {
  // [counter, 4 words][group totals][their exclusive prefixes][run prefixes, whole groups]
  const uint32_t groups = (tiles_per_frame + kChunkGroupTiles - 1u) / kChunkGroupTiles;
  const uint32_t gw = (groups + 3u) & ~3u;
  if (gsum_words) *gsum_words = gw;
  const uint64_t b = 4ull * (uint64_t(kChunkHdrWords) + 2ull * gw + uint64_t(groups) * kChunkGroupRuns);
  return uint32_t((b + 255) & ~uint64_t(255));
}


}  // namespace d2pc
#endif  // D2PC_EXPERIMENTS
