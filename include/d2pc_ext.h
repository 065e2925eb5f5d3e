/*
 * d2pc_ext.h -- UNSTABLE companion of d2pc.h: benchmarks, tests, tools.
 *
 * Nothing here is part of the drop-in boundary (the reference surface being replaced is one constructor and one
 * callback, include/disparity_to_point_cloud/disparity_to_point_cloud.hpp:60-109; d2pc.h is what replaces it).
 * These entry points exist so that bench.py, tools/ and tests/ can calibrate the device, read the kernels' own
 * counters, pick launch shapes and prepare hipGraph captures.  They may change or disappear between builds without
 * D2PC_ABI_VERSION moving; D2PC_EXT_REVISION below is bumped whenever they do.  The ROS adaptor and the host mirror
 * include this header only to print stage times (the reference's printf breadcrumbs, cpp:47-91).
 */
#ifndef D2PC_EXT_H
#define D2PC_EXT_H

#include "d2pc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define D2PC_EXT_REVISION 6   /* round 6: d2pc_clock_probe_device, test hook "handoff_spin_ticks_first"; 5: the laboratory (experiment build) split from the product; 4: split off d2pc.h */
int d2pc_ext_revision(void);

/* ---- hipGraph capture plumbing ------------------------------------------------------------------------------ */
/* Guarantee one free compaction-state buffer for frames up to width x height
 * and batches up to n_frames (and make that the minimum size of any allocated
 * later).  Needed before a capture; optional otherwise (buffers grow on demand). */
int d2pc_reserve(d2pc_ctx *ctx, int width, int height, int n_frames);

/* d2pc_process_mono_device's counterpart of d2pc_reserve: guarantees one free scratch buffer for the two-launch
 * form of a batch of n_frames frames of width x height (dtype D2PC_DTYPE_U8 or D2PC_DTYPE_MONO16) and, in COMPACT
 * mode, the compaction state as d2pc_reserve does -- so that the call can be captured without having run once. */
int d2pc_reserve_mono(d2pc_ctx *ctx, int dtype, int width, int height, int n_frames);

/* Buffers that a stream capture baked into a graph (compaction state, callback scratch) belong to that graph and are
 * never reused.  Call this once every graph captured from this context's launches has been destroyed: the buffers
 * return to the context's pools.  Buffers of graphs never take away from the 8 per pool that eager launches use. */
int d2pc_release_graph_buffers(d2pc_ctx *ctx);

/* ---- the kernels' own counters ------------------------------------------------------------------------------ */
/*
 * Counters of the in-launch hand-offs, summed over the context's launches since creation or d2pc_compact_stats_reset:
 * EVERY kernel that hands counts over inside a launch adds to them -- the single pass (compact_algo 2), the resident
 * one-launch forms (compact_algo 3: k_compact_resident, k_compact_resident_lean) and the tile-fused callback kernels of
 * d2pc_process_mono_device (advisor, round 3: the header used to name the single pass only).  The production build's
 * view of the hand-offs (the reference has only printf breadcrumbs, cpp:47-91).  Call after synchronising the streams
 * that carried the launches.
 *   launches           launches of those kernels
 *   tiles              tiles they served (each takes one ticket and needs the counts of its predecessors)
 *   failed_polls       looks at a predecessor's count that found it unpublished; failed_polls / tiles is the
 *                      hand-off's health: ~0.1 on an idle device, more when predecessors are delayed
 *   wait_us            time the control waves spent in such waits, summed over all blocks (divide by the number
 *                      of resident blocks for wall time)
 *   timeouts           launches in which a wait ran out of its budget (d2pc_check_async_error)
 *   twopass_fallbacks  synchronous host calls (d2pc_process*) that reran such a launch with the two-pass form
 */
typedef struct d2pc_compact_stats_t {
  uint32_t struct_size;   /* = sizeof(d2pc_compact_stats_t), set by the caller */
  uint32_t reserved;
  uint64_t launches, tiles, failed_polls, wait_us, timeouts, twopass_fallbacks;
} d2pc_compact_stats_t;
int d2pc_compact_stats(d2pc_ctx *ctx, d2pc_compact_stats_t *out);
int d2pc_compact_stats_reset(d2pc_ctx *ctx);

/*
 * Device calibration for benchmarks: a plain fill of `bytes` bytes and a plain copy (16 bytes per lane, 1 KiB per
 * wave instruction), asynchronous on `stream`.  bench.py times them in the same run as the reprojection kernel,
 * so its fraction of the 8 TB/s specification can also be read against what THIS device gives a kernel that only
 * streams.  Buffers 16-byte aligned, `bytes` a multiple of 16, source and destination disjoint.
 * Launch shape by tuning keys (d2pc_set_tuning): persistent grid-stride blocks (default; the single-pass kernels'
 * shape) or one-shot blocks (the headline kernel's shape, the fastest store stream found on the chip), plain or
 * non-temporal stores.
 */
int d2pc_membench_fill(d2pc_ctx *ctx, void *d_dst, size_t bytes, void *stream);
int d2pc_membench_copy(d2pc_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, void *stream);

/*
 * The shader clock while other work runs: eight one-wave blocks (one per XCD, as the dispatcher deals them) sleep for
 * `min_us` microseconds of the constant 100 MHz counter and store, per block, {shader cycles passed, 100-MHz ticks passed}
 * into d_out16 (16 x uint64, device memory).  GHz = cycles / ticks x 0.1.  Launch it on ANOTHER stream than the kernel
 * whose clock is wanted, while that kernel is queued many times over; asynchronous; min_us 1..2,000,000.
 * (bench.py: callback_*_clock_GHz -- the bit-sliced select is held near 1.7 GHz, the rest of the chip's kernels at 2.1-2.4.)
 */
int d2pc_clock_probe_device(d2pc_ctx *ctx, void *d_out16, uint32_t min_us, void *stream);

/*
 * Per-stage timing of the synchronous host entry points (d2pc_process,
 * d2pc_process_mono8/16), the counterpart of the reference's printf
 * breadcrumbs (cpp:47-91).  Off by default; d2pc_set_tuning(ctx,
 * "stage_timing", 1) makes every call record HIP events on its stream.
 * d2pc_last_stage_times returns the times of the last such call:
 *   h2d_ms    upload of the frame
 *   prep_ms   mono16 rescale + median (0 when neither runs)
 *   kernel_ms reprojection (+ compaction)
 *   d2h_ms    count readback + download of the points (and indices)
 */
typedef struct d2pc_stage_times {
  float h2d_ms, prep_ms, kernel_ms, d2h_ms, total_ms;
} d2pc_stage_times;
int d2pc_last_stage_times(d2pc_ctx *ctx, d2pc_stage_times *times);

/* Launch-shape tuning hook (no counterpart in the reference).  Results NEVER depend on it: every key below picks how
 * the same bytes are produced.  Keys of the product library (libd2pc.so): "pxt_parity" (ROI pixels per thread of the
 * one-shot PARITY blocks: 1 or 2; 0 = choose per launch), "pxt_compact" (8), "blocks_per_cu" (grid = blocks_per_cu x CUs,
 * capped by the tile count; 1..4096), "onepass_blocks_per_cu" (persistent blocks per CU of the single pass; 0 = choose:
 * 3 for 4K-class frames, 4 below), "resident_pxt" (shape of the one-launch
 * resident forms), "no_vec_rows", "fuse_rows" (rows per wave of d2pc_fuse_device: 0 = choose, else even 2..1024),
 * "stage_timing" (0/1, see d2pc_last_stage_times), "spin_timeout_ms" (1..40000: time budget of the in-launch hand-off
 * waits), "callback_chunks" (0..64 pipeline chunks of d2pc_process_mono_device; <= 1 = no overlap),
 * "callback_fused" (0/1, see d2pc_process_mono_device; default 1), "callback_fused_compact" (COMPACT mode: 0 = two
 * launches, 1 = one tile per block, 2 = persistent blocks that scatter one tile while filtering the next; default 2),
 * "callback_pipe_blocks_per_cu" (1..8, default 3), "membench_blocks_per_cu" (d2pc_membench_*: persistent blocks per
 * CU, default 8; 0 = one block per "membench_unroll" x 4 KiB), "membench_unroll" (1, 2 or 4 16-byte accesses per thread),
 * "membench_nt" (0/1), "host_direct_read" (0/1, default 1: a pinned fp32 / 8-bit frame handed to d2pc_process /
 * d2pc_process_mono8 without a median is read by the reprojection in place, PARITY mode), "median_algo" (0 = choose
 * per launch, 1 = one pixel per thread, 2 = 32 pixels per thread, bit-sliced; the two give identical bytes),
 * "onepass_form" (0 = choose, 2 = the dense single pass: the only form of the product).
 * (Closed in round 6 and moved to the experiment build: "resident_pair" -- two 4K-class COMPACT frames in one call: 0 = one
 * launch each (the product), 1 = round 4's one launch of blocks twice the size -- and "resident_stagger_pct", the scale of
 * the resident blocks' ramped start: profiles/r04_ab_stagger.txt, r05_ab_pair.txt.)
 *
 * EXPERIMENT BUILD (libd2pc_exp.so = the same sources with -DD2PC_EXPERIMENTS=1, `make -C csrc exp`; what tests/ and
 * tools/ load to re-run recorded negatives; never shipped, never what INTEGRATION.md links).  It adds, and only it
 * accepts: d2pc_config.compact_algo = 4 (round 4's two-pass over Infinity-Cache-sized pieces of the batch, 15-25 % slower
 * than the single pass: docs/HISTORY.md) with its keys "big_batch_algo" (2 / 4: what compact_algo 0 takes for big
 * batches) and the two that size its pieces; "resident_unbounded" (0/1: the resident form over more blocks than are
 * resident); "pxt_parity" 4 / 8 / 16 (the tile-walking PARITY kernel of rounds 1-2) and "parity_small"; "pxt_compact" 4 / 16;
 * "onepass_form" 1 (rounds 2-4's single pass) and 3-7 (round 5's other forms: 8 worker waves, loader wave, 4 runs per
 * wave, deferred landing, both -- profiles/r05_ab_onepass_forms_*.txt, r05_ab_forms567.txt); "median_algo" 3 (the select
 * split over a lane pair); test hook "general_q_form" = 1 (round 2's fused evaluation of a general Q). */
int d2pc_set_tuning(d2pc_ctx *ctx, const char *key, int value);


/* TEST hooks that DO change the arithmetic, kept apart from the tuning keys for that reason (tests compare the two
 * routes to the same bytes with them; nothing else should call this):
 *   "force_general_q"  0/1: a cv::stereoRectify-structured Q goes through the general kernel as well
 *   "handoff_spin_ticks_first"  -1 = off; >= 0: the wait budget, in 100-MHz ticks, of the FIRST of the two launches a
 *                      two-frame 4K-class COMPACT call is made of (0 = give up at the first look that fails): lets a test
 *                      make frame 0's hand-off time out while frame 1's does not (d2pc_check_async_error must report it)
 *   "general_q_form"   experiment build only: 1 = round 2's fused multiply-adds instead of OpenCV 3/4's association */
int d2pc_ext_set_test_hook(d2pc_ctx *ctx, const char *key, int value);

#ifdef __cplusplus
}
#endif
#endif /* D2PC_EXT_H */
