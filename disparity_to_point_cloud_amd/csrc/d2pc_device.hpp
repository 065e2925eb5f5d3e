// d2pc_device.hpp -- shared host/device definitions for the gfx950 kernels.
//
// Replaces, on the device, the three scalar CPU loops of
// Disparity2PCloud::DisparityCb (reference src/disparity_to_point_cloud.cpp):
//   :63-64  cv::reprojectImageTo3D      (Q . (u,v,d,1), perspective divide)
//   :70-76  ROI inset push_back loop    (row-major gather, 12 B -> 16 B)
//   :84-85  pcl::toROSMsg               (memcpy into PointCloud2.data)
// One pass: 4 B read, 16 B written per ROI pixel, straight into the final
// PointCloud2 byte layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The shipped library holds the product's kernels only.  -DD2PC_EXPERIMENTS=1 (libd2pc_exp.so; `make exp`) adds the
// laboratory: tile shapes other than the defaults, the tile-walking PARITY kernel of rounds 1-2, the chunked two-pass
// (compact_algo 4), the register-resident form over more blocks than are resident, round 2's fused general-Q form.
// tests/ and tools/ that exercise those load the experiment build; nothing in include/d2pc.h refers to them.
#ifndef D2PC_EXPERIMENTS
#define D2PC_EXPERIMENTS 0
#endif

namespace d2pc {

constexpr int kBlock = 256;  // threads per workgroup (4 waves of 64)

// ---- exact unsigned division by a launch-time constant --------------------
// Granlund-Montgomery round-up method: q = (t + ((n - t) >> s1)) >> s2 with
// t = mulhi(m, n); exact for every 32-bit n and every 1 <= d < 2^32.
struct FastDiv {
  uint32_t m, s1, s2;
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  uint32_t l = 0;
  while (l < 32 && (uint64_t(1) << l) < d) ++l;  // l = ceil(log2 d)
  f.m = uint32_t(((uint64_t(1) << 32) * ((uint64_t(1) << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l == 0 ? 0 : l - 1;
  return f;
}

__host__ __device__ inline uint32_t fdiv(uint32_t n, const FastDiv &f) {
  const uint32_t t = uint32_t((uint64_t(f.m) * n) >> 32);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// ---- launch geometry (kernarg; lives in SGPRs) ----------------------------
struct Geom {
  uint32_t width, border;
  uint32_t roi_w, roi_n;              // ROI width; roi_n = ROI points per frame
  uint32_t tiles_per_frame, n_frames, total_tiles;
  uint32_t groups_per_frame;          // compaction: ceil(tiles_per_frame/64)
  uint32_t row_stride;                // bytes between image rows (< 2^32)
  uint32_t last_off;                  // byte offset of the frame's last ROI pixel
  uint32_t s64_v, s64_u;              // 64   = s64_v*roi_w + s64_u   (next slot)
  uint32_t s832_v, s832_u;            // 832  pixels: slot 3 -> slot 0 of the next batch
  uint32_t s1024_v, s1024_u;          // 1024 pixels: one batch (16-B row loads)
  uint32_t frame_state_stride;        // compaction: bytes of state per frame
  FastDiv div_roi_w, div_tpf;
  uint64_t in_frame_stride;           // bytes between frames
  uint64_t out_frame_stride;          // points between frames' outputs
  float scale;                        // U8/U16 decode: d = (float)raw*scale
  float min_disparity;                // compact predicate: drop d <= this
  uint32_t spin_ticks;                // single pass: hand-off wait budget in s_memrealtime ticks (100 MHz)
  uint32_t pxt;                       // ROI pixels per thread the tile counts above were formed with (host side)
  uint32_t stagger;                   // k_compact_resident_lean: block t starts its loads (t * stagger) >> 10 sleeps of 64 cycles late
};

// Binades a 2.4-style running sum may cross inside one image row: one per doubling of the column for a small principal
// point (q03 = -0.1: 2^-1 .. 2^15 over 2^15 columns), 7 for the usual |cx| of a few hundred.  The kernels stop at the
// table's length `n` (wave-uniform), so the usual case pays for its own segments only.
constexpr int kQxSegs = 18;

// One launch of the chunked two-pass compaction (k_compact_chunk, compact_algo 4): which frames its scatter blocks
// serve, which frames its count blocks serve, and how the two kinds are interleaved in the grid.
struct ChunkArgs {
  uint32_t scatter_f0, scatter_tiles;  // frames from scatter_f0 on: scatter_tiles = frames x tiles_per_frame one-shot blocks
  uint32_t count_f0, count_blocks;     // frames from count_f0 on: count_blocks = frames x groups_per_frame
  uint32_t period;                     // blocks 0, P, 2P, ... are count blocks while they last (grid = scatter_tiles + count_blocks)
  uint32_t groups_per_frame;           // ceil(tiles_per_frame / 32): groups of 128 runs = 16,384 pixels
  uint32_t gsum_words;                 // words the group totals take in a frame's state (padded; their prefixes take as many)
  FastDiv div_gpf, div_period;
};

// OpenCV 2.4's running column sum qx (per row qx = q01*y + q03, then qx += q00 per column, one rounding per step)
// as the host replayed it for one (Q, width): columns [x[j], x[j+1]) have qx(u) = double(u) + c[j] EXACTLY.
struct QxSegs {
  uint32_t n;           // segments in use (>= 1)
  uint32_t x[kQxSegs];  // x[0] = 0
  double c[kQxSegs];
};

// Q_ (reference hpp:72): row-major 4x4 doubles.  Kernarg => scalar registers,
// i.e. one copy per wave with no LDS or vector-memory traffic at all.
struct QMat {
  double q[16];
  // how the general kernel evaluates (d2pc_set_reproject_form; test hook "general_q_form" of d2pc_ext.h):
  //   0  OpenCV 3/4's association, bit for bit (default)
  //   1  fused multiply-adds (round 2's form)
  //   2  OpenCV 2.4's loop, bit for bit -- for a Q whose column increments are exact (q00 = 1, q01 = q10 = q20 = q30 = +0,
  //      as cv::stereoRectify's): 2.4 forms qx by adding q00 once per column, which rounds whenever the running sum
  //      crosses a binade; the host replays that recurrence once per (Q, width) and hands over the <= kQxSegs segments
  //      of columns in which qx - x is one constant
  uint32_t form;
  QxSegs seg;                     // form 2
};

// The structure cv::stereoRectify always produces (hpp:104):
//   [1 0 0 cx; 0 1 0 cy; 0 0 0 f; 0 0 a b]   (zeros are +0.0, ones are 1.0)
// lets the kernel drop nine multiply-adds per pixel with bit-identical
// results: X = (u + cx)/W, Y = (v + cy)/W, Z = f/W, W = a*d + b.
//
// One OpenCV generation bit for bit (d2pc_set_reproject_form) costs this structure little.  With the zeros and ones
// above, either generation's evaluation collapses to W = b + RN(a*d) (product and sum rounded apart, where the default
// kind fuses them), and
//   QK_STEREO_CV24  (2.4)  X = qx(u)*iW with the running column sum qx of QxSegs, Y = (v + cy)*iW, Z = f*iW
//   QK_STEREO_CV4   (3/4)  the numerators pass through float: X = float(u + cx)*ia, Y = float(v + cy)*ia,
//                          Z = float(f)*ia (f arrives already rounded to float from the host)
// They are kernel KINDS (template instances), not a run-time switch: the default kind's kernels -- short of scalar
// registers as they are -- stay exactly as they were.  Derivation: DESIGN.md section 2; each is checked bit for bit
// against the general kernel in the same form and against the oracle.
struct QStereo {
  double cx, cy, f, a, b;  // cx = q03 + 0.0, cy = q13 + 0.0, f = q23 + 0.0, b = q33 + 0.0
  double w_safe;           // |W| >= w_safe  =>  every coordinate is a finite float (per launch)
};

enum : int { DT_F32 = 0, DT_U8 = 1, DT_U16 = 2 };
enum : int { QK_GENERAL = 0, QK_STEREO = 1, QK_STEREO_CV24 = 2, QK_STEREO_CV4 = 3 };
constexpr bool is_stereo(int qk) { return qk != QK_GENERAL; }

// ---- compaction state (zeroed by a memset node before every launch) -------
//   [StateHeader 128 B][frame 0 state][frame 1 state]...
//   frame state (frame_state_stride bytes, 256-B aligned):
//     [ticket u32 alone in 256 B][group_acc u64, one per 128-B line][16 B x tiles]
//   the 16 B per tile hold one u64 granule (single pass) or 4 x u32 wave counts (two-pass)
//   group_acc = (tiles arrived << 32) | sum of their point counts
//   granule   = kGranuleTag | point count of one tile
struct CompactStats;
struct StateHeader {
  uint32_t timeout;       // set to 1 if a bounded spin expired
  uint32_t pad;
  CompactStats *stats;    // the context's production counters (written by k_state_clear for the launch behind it)
  // diagnostic build only (-DD2PC_DIAG): shader-clock sums over all tiles
  unsigned long long diag[7];  // iterations, spins, t_compute, t_ticket, t_wait, t_scatter, t_total
  unsigned long long pad2[7];
};
static_assert(sizeof(StateHeader) == 128, "the header is one 128-byte line");
// Production counters of a context's single-pass launches (device memory, never cleared by a launch).  Every
// block adds its share once, at its exit, to slot blockIdx % kStatSlots: 768-1024 blocks finishing together on ONE
// word serialise at the memory side and cost 40-60 us per launch (profiles/r03_ab_counters.txt); spread over
// 64 lines they cost nothing.  The host sums the slots.
constexpr int kStatSlots = 64;
struct CompactStats {
  struct Slot {
    unsigned long long tiles;         // tiles served
    unsigned long long failed_polls;  // polls of a predecessor's count that found it not yet published
    unsigned long long wait_ticks;    // time control waves spent in such waits, 100 MHz ticks, summed over blocks
    unsigned long long pad[13];       // a slot is one 128-byte line
  } slot[kStatSlots];
  unsigned long long launches, timeouts, dbg[2];
};
// tile-fused COMPACT callback kernel (k_callback_bs_compact): per frame [ticket, 128 B][band accumulators, 8 B
// each, rounded up to 128 B][64 B of row counts per tile]
constexpr uint32_t kCbTicketBytes = 128;
constexpr uint32_t kCbMaxTilesX = 128;  // tiles per band: far below the blocks resident at once (3 per CU)
__host__ __device__ inline uint32_t cb_band_acc_bytes(uint32_t tiles_y) { return (tiles_y * 8u + 127u) & ~127u; }
// compact_algo 3 (k_compact_resident): the launch epochs its granules carry instead of being zeroed
constexpr uint32_t kEpochBase = 1u << 30, kEpochEnd = 1u << 31;
constexpr int kResidentBlocksPerCu = 4;  // what k_compact_resident's grid may be at most, per CU (<= 128 VGPRs: admitted)
constexpr uint32_t kFrameTicketBytes = 256;   // the ticket word has a 256-B block to itself
constexpr uint32_t kGroupAccStride = 128;     // one group accumulator per 128-B line
constexpr uint64_t kGranuleTag = uint64_t(1) << 63;
constexpr int kGroupTiles = 64;         // tiles per counting group
// Hand-off waits are bounded by TIME (constant 100 MHz clock), not by a poll count: a block holding an
// earlier ticket can be descheduled for long (several processes on one GPU, CWSR preemption, a debugger),
// and a healthy launch must not be declared broken for it.  Default 4 s; d2pc_set_tuning("spin_timeout_ms").
constexpr uint32_t kSpinTicksPerMs = 100000u;
constexpr uint32_t kDefaultSpinMs = 4000u;
// counts[f] of a frame whose hand-off timed out: visible in-band, not only through d2pc_check_async_error
constexpr uint32_t kCountTimedOut = 0xffffffffu;

}  // namespace d2pc
