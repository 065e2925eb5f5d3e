// d2pc_launch.hpp -- host-side launch interface between the C ABI
// (d2pc_capi_*.hip, d2pc_ctx.hpp) and the kernels (d2pc_parity / _compact / _onepass / _callback / _median* / _fusion .hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "d2pc_device.hpp"

namespace d2pc {

struct LaunchArgs {
  const void *disp = nullptr;     // device: n_frames disparity images
  void *out_points = nullptr;     // device: 16-byte points
  uint32_t *out_index = nullptr;  // device, nullable
  uint32_t *counts = nullptr;     // device, nullable (required in COMPACT)
  void *state = nullptr;          // device: compaction state (COMPACT)
  size_t state_bytes = 0;
  void *state_other = nullptr;    // dense single pass: the buffer's other half, zeroed by this launch for the next one (nullable)
  bool state_is_clean = false;    // ... and `state` was left clean by the previous launch: no k_state_clear in front
  void *stats = nullptr;          // device: CompactStats of the context (single pass)
  int dtype = DT_F32;
  int pxt = 4;                    // ROI pixels per thread (tile = 256*pxt)
  int compact_algo = 1;           // 1 two-pass, 2 single-pass, 3 one launch of resident blocks (k_compact_resident[_lean]);
                                  // experiment build: 4 chunked two-pass of one-shot blocks (k_compact_chunk; geom in 512-pixel tiles)
  uint32_t chunk_frames = 0;      // algo 4: frames per chunk, and frames of the first chunk (only counted, nothing
  uint32_t chunk_first = 0;       //         to overlap with: kept short)
  bool chunk_clear = false;       // algo 4: the state buffer was last used otherwise: zero its frame counters first
  uint32_t epoch = 0;             // algo 3: this launch's epoch (kEpochBase <= epoch < kEpochEnd)
  int onepass_form = 1;           // algo 2: 1 = raw tiles in LDS, every pixel decided twice (k_compact_onepass); 2 / 3 = survivors packed by
                                  // the count phase, dense scatter, 4 / 8 worker waves (k_compact_onepass_dense; geom.pxt = 8 / 16)
  bool keep_timeout = false;      // tile-fused COMPACT callback kernels: the state clear keeps the header's timeout flag (a
                                  // later sub-batch of ONE call shares the buffer and must not wipe an earlier one's give-up)
  bool parity_small = false;      // PARITY: one-shot blocks of 256 * pxt pixels (pxt 1, 2 or 4), k_reproject_pack_small
  bool vec_rows = false;          // fp32 rows fetchable 16 B per lane (alignment checked by the host)
  uint32_t grid = 1;
  hipStream_t stream = nullptr;
  Geom geom{};
  int q_kind = QK_GENERAL;        // QK_STEREO when Q has stereoRectify's structure
  QMat q{};
  QStereo qs{};
};

// k x k median of 8-bit frames (d2pc_median.hip)
struct MedianArgs {
  uint32_t width = 0, height = 0, n_frames = 1;
  uint32_t src_row_stride = 0, dst_row_stride = 0;  // bytes
  uint64_t src_frame_stride = 0, dst_frame_stride = 0;
  // Output rectangle (median only): pixels [out_x0, out_x0+out_w) x [out_y0, out_y0+out_h) of dst are
  // written, the rest of dst is left untouched; the window still reads the WHOLE image (replicated only
  // at the true image edges).  out_w == 0 means the whole image.  The fused callback entry points pass the
  // inset ROI (cpp:70,72 read nothing else of the filtered image).
  uint32_t out_x0 = 0, out_y0 = 0, out_w = 0, out_h = 0;
  uint32_t tiles_x = 0, tiles_y = 0;                // filled by launch_median
  // 0: choose per launch; 1: one pixel per thread (d2pc_median.hip); 2: bit-sliced across pixels (d2pc_median_bs.hip)
  int algo = 0;
};
bool median_ksize_supported(int k);
uint64_t median_bs_tiles(const MedianArgs &a);  // a.out_* resolved
bool median_uses_bs(const MedianArgs &a, int ksize);  // a.out_* resolved: would launch_median take the bit-sliced kernel?
hipError_t launch_median_bs(const void *src, void *dst, const MedianArgs &a, int ksize, hipStream_t stream);
hipError_t launch_median(const void *src, void *dst, const MedianArgs &a, int ksize, hipStream_t stream);
// cv_bridge mono16 -> mono8 (d2pc_median.hip); strides in bytes, src rows hold uint16
hipError_t launch_mono16_to_mono8(const void *src, void *dst, const MedianArgs &a, hipStream_t stream);

// Depth-map fusion inner loop (d2pc_fusion.hip): rule + combined confidence +
// 3x3 median + crop.  Rules are numbered as in include/d2pc.h (source order of
// reference src/depth_map_fusion.cpp:162-235).
enum {
  FUSE_WEIGHTED_AVERAGE = 0, FUSE_MAX_DIST = 1, FUSE_MAX_DIST_UNLESS_BLACK = 2, FUSE_BETTER_SCORE = 3,
  FUSE_ONLY_GOOD_1 = 4, FUSE_ONLY_GOOD_AVG = 5, FUSE_OVERLAP = 6, FUSE_BLACK_TO_WHITE = 7, FUSE_GRAD_FILTER = 8,
  FUSE_RULE_COUNT = 9
};
struct FuseArgs {
  const uint8_t *in[6] = {};            // depth1, depth2, score1, score2, grad1, grad2 (grads unused without `combined`)
  uint64_t in_frame_stride[6] = {};
  uint32_t in_pitch[6] = {};
  uint8_t *fused = nullptr;             // out_width x out_height
  uint8_t *combined = nullptr;          // width x height, nullable
  uint64_t fused_frame_stride = 0, combined_frame_stride = 0;
  uint32_t fused_pitch = 0, combined_pitch = 0;
  uint32_t width = 0, height = 0, n_frames = 1;
  int rule = FUSE_GRAD_FILTER;
  uint32_t crop_left = 0, crop_top = 0, out_width = 0, out_height = 0;
  uint32_t strips_x = 0, chunks_y = 0, items = 0, rows_per_wave = 0;  // filled by launch_fuse
};
hipError_t launch_fuse(FuseArgs a, hipStream_t stream, int rows_hint = 0);

// rotateMat (d2pc_fusion.hip): dst(i, j) = src(rows-1-j, i), dst is cols rows of `rows` pixels
struct RotateArgs {
  const uint8_t *src = nullptr;
  uint8_t *dst = nullptr;
  uint64_t src_frame_stride = 0, dst_frame_stride = 0;
  uint32_t src_pitch = 0, dst_pitch = 0;
  uint32_t cols = 0, rows = 0, n_frames = 1;
  uint32_t tiles_x = 0, tiles_y = 0;  // filled by launch_rotate_cw
};
hipError_t launch_rotate_cw(RotateArgs a, hipStream_t stream);

bool tile_shape_supported(int pxt);
uint32_t frame_state_stride(uint32_t tiles_per_frame);
size_t compact_state_bytes(const Geom &g);
hipError_t launch_parity(const LaunchArgs &a);          // d2pc_parity.hip
hipError_t launch_onepass(const LaunchArgs &a);         // d2pc_onepass.hip (compact_algo 2: state clear + single pass)
// zeroes a compaction state buffer and starts its header (d2pc_onepass.hip); also ahead of the tile-fused COMPACT callback kernels
hipError_t launch_state_clear(void *state, size_t state_bytes, void *stats, hipStream_t stream, bool keep_timeout = false);
// experiment build only (d2pc_chunk.hip, compact_algo 4): bytes of state per frame for frames of `tiles_per_frame`
// 512-pixel tiles, and the words its group totals take
uint32_t chunk_frame_state_stride(uint32_t tiles_per_frame, uint32_t *gsum_words);
hipError_t launch_compact_chunked(const LaunchArgs &a);
// tile-fused callback body (bit-sliced k x k median + PARITY reprojection of the tile from LDS;
// k_callback_bs): `m` carries the filter's geometry with the output rectangle = the ROI of a.geom; a.out_points,
// a.out_index (nullable), a.counts (nullable), a.q*, a.stream as in launch_parity
hipError_t launch_callback_bs(const LaunchArgs &a, MedianArgs m, const void *src, int ksize);
hipError_t launch_compact(const LaunchArgs &a);
// the same in COMPACT mode (k_callback_bs_compact): a.state / a.state_bytes (callback_compact_state_bytes; the
// frame stride it returns goes into a.geom.frame_state_stride), a.stats and a.counts are required
size_t callback_compact_state_bytes(uint32_t tiles_x, uint32_t tiles_y, uint32_t n_frames, uint32_t *frame_stride);
hipError_t launch_callback_bs_compact(const LaunchArgs &a, MedianArgs m, const void *src, int ksize);

// Device calibration for bench.py (d2pc_membench.hip): a plain dwordx4 fill and a dwordx4 copy, the two
// streaming shapes the roofline fraction of the reprojection kernel is read against on the SAME device
hipError_t launch_clock_probe(void *out16, uint32_t min_us, hipStream_t stream);
hipError_t launch_membench_fill(void *dst, size_t bytes, uint32_t blocks, int unroll, bool nt, hipStream_t stream);
hipError_t launch_membench_copy(const void *src, void *dst, size_t bytes, uint32_t blocks, int unroll, bool nt, hipStream_t stream);

}  // namespace d2pc
