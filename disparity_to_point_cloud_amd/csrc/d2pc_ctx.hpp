// d2pc_ctx.hpp -- the device context behind include/d2pc.h and what the d2pc_capi_*.hip translation units share.
// Host code only (no kernel here): the C ABI was one 2,200-line file until round 6; it is now cut along its seams --
//   d2pc_capi_context.hip  status strings, calibration (Q, blob, closed forms), create / destroy, fill_q / classify_q
//   d2pc_capi_state.hip    the pool of per-stream state buffers (lazy events, captures, self-cleaning halves), d2pc_reserve*,
//                          d2pc_check_async_error, d2pc_compact_stats
//   d2pc_capi_route.hip    launch geometry and the routing of one device-resident call (enqueue), d2pc_process_device
//   d2pc_capi_host.hip     the synchronous host entry points (d2pc_process*) and the pipelined host path (d2pc_pipeline_*)
//   d2pc_capi_mono.hip     cv_bridge rescale, median, the device-resident callback body (d2pc_process_mono_device)
//   d2pc_capi_fusion.hip   depth-map fusion inner loop, rotate, crop
//   d2pc_capi_ext.hip      include/d2pc_ext.h: tuning keys, test hooks, device calibration kernels
// Every exported symbol is unchanged (tests/test_abi_cpu.py::test_library_exports_every_declared_symbol).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "../../include/d2pc.h"
#include "../../include/d2pc_ext.h"
#include "d2pc_device.hpp"
#include "d2pc_launch.hpp"

using namespace d2pc;  // (host translation units of the library only)

// A device buffer that ONE stream's work owns at a time.  Two pools of them per context: the compaction state
// (a COMPACT launch owns its buffer from the zeroing kernel to its last store) and the scratch of
// d2pc_process_mono_device's two-launch form (filtered frames in `p`, rescaled mono16 frames in `p2`).  Launches
// that may overlap (different streams, a captured graph being replayed) never share one.
struct StateBuf {
  void *p = nullptr;   size_t cap = 0;
  void *p2 = nullptr;  size_t cap2 = 0;  // callback scratch only
  hipStream_t stream = nullptr;  // stream of the last launch that used it (valid when `bound`)
  bool bound = false;
  hipEvent_t done = nullptr;     // recorded behind that launch (not while capturing)
  bool pending = false;          // `done` was recorded and has not been seen complete yet
  bool dirty = false;            // work was enqueued on `stream` since `done` was last recorded: the record is made LAZILY,
                                 // when another stream asks for the buffer (settle below).  Recording behind every launch put
                                 // a marker packet between back-to-back COMPACT launches: 5.5 us of idle device per call -- a
                                 // sixth of a single 4K frame's time (kernel 26.5 us, launch period 32.4; PARITY, which records
                                 // nothing: 22.2 / 22.2)
  int algo = 0;                  // algorithm of that launch: 2 = single pass (its header holds the timeout flag),
                                 // 3 = resident blocks (the flag holds the launch's epoch)
  uint32_t epoch = 0;            // algo 3: that launch's epoch
  uint32_t epoch_first = 0;      // algo 3: epoch of the FIRST launch of the call that launch belonged to (a two-frame call is two
                                 // launches with consecutive epochs on this buffer: a give-up in either stores ITS epoch, and
                                 // d2pc_check_async_error reports a flag anywhere in [epoch_first, epoch]; advisor, round 5)
  // dense single pass (algo 2): the buffer holds TWO states of pp_half bytes; a launch runs on one and zeroes the other
  // inside its own launch, for the next launch on this buffer (no k_state_clear kernel in front of every call)
  size_t pp_half = 0;            // bytes per half as the last such launch used them (0: none yet)
  size_t pp_bytes = 0;           // state bytes that launch's kernel ZEROED in the idle half: the next launch may trust the half only
                                 // if it needs no more than that (the kernel zeroes state_bytes, not the half rounded up to 256)
  int pp_next = 0;               // the half the next launch takes, clean iff pp_clean
  bool pp_clean = false;         // reset by every other use of the buffer (other algorithms, captures, reallocation)
  size_t hdr_off = 0;            // where the header of the LAST launch lives (d2pc_check_async_error)
  uint64_t chunk_sig = 0;        // algo 4: tiles per frame and frames of that launch (its frame counters and "empty" marks sit
                                 // where the next launch of the same shape expects them)
  bool captured = false;         // a stream capture baked the pointer into a graph: never freed, moved or shared
                                 // until d2pc_release_graph_buffers
  unsigned long long capture_id = 0;
};
// Buffers that do not belong to a captured graph: at most this many per pool (a ninth stream waits for one);
// buffers owned by graphs come on top, so captures can never starve the eager launches of a context.
constexpr int kMaxEagerBufs = 8;
struct BufPool {
  std::vector<StateBuf *> bufs;  // pointers: a buffer's address is stable while the vector grows
  size_t reserve = 0, reserve2 = 0;  // d2pc_reserve / d2pc_reserve_mono: every buffer is at least this large
  const char *what = "";
};

// One frame in flight on the pipelined host path (d2pc_pipeline_*).
struct PipeSlot {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  void *h_in = nullptr;      size_t h_in_cap = 0;    // pinned; the caller fills it
  void *h_out = nullptr;     size_t h_out_cap = 0;   // pinned; points (+ index behind them)
  uint32_t *h_count = nullptr;                       // pinned
  void *d_in = nullptr;      size_t d_in_cap = 0;
  void *d_med = nullptr;     size_t d_med_cap = 0;
  void *d_cvt = nullptr;     size_t d_cvt_cap = 0;   // MONO16 frames rescaled to 8 bits
  void *d_out = nullptr;     size_t d_out_cap = 0;
  void *d_idx = nullptr;     size_t d_idx_cap = 0;
  StateBuf st;                                       // the slot's own compaction state
  uint32_t *d_count = nullptr;
  d2pc_frame_desc desc{};
  size_t roi_n = 0, idx_off = 0;
  int state = 0;             // 0 free, 1 acquired, 2 submitted, 3 collected (until release)
  uint64_t seq = 0;
};

struct d2pc_ctx {
  d2pc_config cfg{};
  int device = 0;
  int cu_count = 256;
  hipStream_t stream = nullptr;
  bool have_q = false;
  double q[16] = {0};
  int q_kind = QK_GENERAL;
  QStereo qs{};
  // tuning (d2pc_set_tuning); defaults from tools/ab.py sweeps on MI355X
  // (fast and slow devices agree on 2048-pixel tiles and 2-4 tiles per block)
  int pxt_parity = 0, pxt_compact = 8;  // ROI pixels per thread; parity 0 = choose per launch (parity_pxt below)
  int parity_small = 0;          // PARITY kernel form: 0 = choose, 1 = one-shot blocks of 256 * pxt pixels (pxt 1, 2, 4), 2 = tiles walked by fewer blocks
  int blocks_per_cu = 128;
  int onepass_blocks_per_cu = 0;   // resident 5-wave blocks per CU (73 VGPRs, 33 KB LDS each: 4 fit); 0 = choose per launch
  int onepass_form = 0;            // which single-pass kernel: 0 = choose (kDefaultOnepassForm), 1 / 2 / 3: see enqueue
#if D2PC_EXPERIMENTS
  int big_batch_algo = 2;          // COMPACT launches of >= 4 frames and >= 20,480 tiles: 2 = single pass (default: faster), 4 = chunked two-pass of one-shot blocks
  int chunk_mb = 96;               // algo 4: input bytes per chunk (MiB); the chunk must stay in the 256 MiB Infinity Cache for one launch
  int chunk_first_frames = 0;      // algo 4: frames of the first chunk (0 = an eighth of a chunk)
  int resident_unbounded = 0;      // algo 3: admit launches of more blocks than are resident at once (see enqueue)
  int general_q_form = 0;          // 0: OpenCV 3/4's association bit for bit; 1: fused multiply-adds (round 2's form)
  // closed experiments (round 6: behind the experiment build; the product runs their measured defaults)
  int resident_stagger_pct = -1;   // algo 3, register-resident form: scale of the ramped start in % (0 = every block loads at once;
                                   // -1 = choose: 50 for one frame that fills the device, else 0)
  int resident_pair = 0;           // algo 3: 1 = two 4K-class frames in ONE launch of 16,384-pixel blocks (0: two launches of 8,192-pixel blocks)
#else
  static constexpr int big_batch_algo = 2, resident_unbounded = 0, general_q_form = 0;  // (the product: the single pass; bounded; OpenCV's association)
  static constexpr int resident_stagger_pct = -1, resident_pair = 0;  // (ramp for one device-filling frame only; two 4K frames = two launches)
#endif
  int resident_pxt = 0;            // algo 3: pixels per thread of its blocks (0 = choose: the ordinary tile if the launch fits, else 32, else 64)
  int spin_timeout_ms = int(kDefaultSpinMs);  // single pass: hand-off wait budget
  int force_general_q = 0;
  int spin_ticks_first = -1;     // TEST hook "handoff_spin_ticks_first": >= 0 = wait budget (100 MHz ticks) of the FIRST launch of a
                                 // two-frame COMPACT call, so a test can make frame 0's hand-off give up and frame 1's not
  int reproject_form = 0;        // 0: per Q kind (specialised stereoRectify kernel / general kernel in OpenCV 3/4's form);
                                 // 24: OpenCV 2.4's loop bit for bit (Q with exact column increments); 4: OpenCV 3/4's form for every Q
  uint32_t qx_width = 0;         // reproject_form 24: columns the cached segment table below covers (0 = none)
  QxSegs qx_seg{};
  int no_vec_rows = 0;
  int stage_timing = 0;          // record per-stage HIP events in the synchronous host entry points
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  d2pc_stage_times times{};
  bool have_times = false;
  int fuse_rows = 0;             // d2pc_fuse_device rows per wave: 0 = choose, else 2..1024
  int host_direct_read = 1;      // synchronous host entry points: a PINNED input frame is read by the first kernel in place
  int median_algo = 0;           // MedianArgs::algo: 0 choose per launch, 1 per-pixel select, 2 bit-sliced (k = 9, 11)
  // device scratch
  BufPool states;                  // compaction state, one buffer per stream with COMPACT work in flight
  BufPool cb_scratch;              // d2pc_process_mono_device, two-launch form: one scratch per stream in flight
  // production counters (d2pc_compact_stats): reset by d2pc_compact_stats_reset
  uint32_t resident_epoch = kEpochBase;  // compact_algo 3: the next launch's epoch
  uint64_t n_twopass_fallbacks = 0;  // synchronous host calls that reran a timed-out single pass with the two-pass form
  void *d_in = nullptr;      size_t in_cap = 0;
  void *d_out = nullptr;     size_t out_cap = 0;
  void *d_idx = nullptr;     size_t idx_cap = 0;
  void *d_med = nullptr;     size_t med_cap = 0;
  void *d_cvt = nullptr;     size_t cvt_cap = 0;   // mono16 -> mono8 (cpp:50)
  uint32_t *d_counts = nullptr;
  uint32_t *h_counts = nullptr;  // pinned
  CompactStats *d_stats = nullptr;  // single-pass counters, added to by the launches' blocks (d2pc_compact_stats)
  int membench_blocks_per_cu = 8;  // 0: one-shot blocks (one per membench_unroll x 4 KiB)
  int membench_unroll = 4;         // 16-byte accesses per thread and step: 1, 2 or 4
  int membench_nt = 0;             // non-temporal stores
  // d2pc_process_mono_device: two internal streams + scratch for the filtered frames
  int cb_fused_compact = 2;      // ... and the COMPACT form of that kernel: 2 = persistent blocks, software-pipelined over their
                                 // tiles (k_callback_bs_compact_pipe; the default: 16 x 4K with 30 % holes + indices 711 us
                                 // against 758 us for form 1 and 963 us as two launches, profiles/r03_callback_compact.txt);
                                 // 1 = one tile per block (k_callback_bs_compact); 0 = two launches in COMPACT mode
  int cb_pipe_blocks_per_cu = 3; // the pipelined form's persistent blocks per CU (LDS and registers admit 3)
  int cb_fused = 1;              // d2pc_process_mono_device, PARITY: median + points in one kernel, tile by tile (k_callback_bs:
                                 // bit-sliced median, the tile's points from LDS) when the launch is large enough for the
                                 // bit-sliced filter; 0 = always the filter launch followed by the reprojection launch
  int cb_chunks = 1;             // pipeline chunks per call (<= 1: everything in order on the caller's stream;
                                 // overlapping did not pay reliably: profiles/r02_callback_overlap.txt)
  hipStream_t cb_stream_m = nullptr, cb_stream_r = nullptr;
  std::vector<hipEvent_t> cb_events;
  hipEvent_t cb_overlap_done = nullptr;  // the chunked-overlap form shares the two streams above: calls are serialised
  bool cb_overlap_pending = false;
  // pipelined host path
  PipeSlot slots[8];
  int pipe_depth = 0;
  int pipe_direct = 0;           // kernels write straight into pinned host memory
  uint64_t pipe_seq = 0;
  char err[256] = {0};
};

namespace d2pc {
namespace host {

constexpr int kDefaultOnepassForm = 2;  // profiles/r05_ab_onepass_forms_*.txt: never slower than form 1, 5-6 % faster with 30 % holes, 29 % with 90 %

// records the message in the context, returns `status` (d2pc_last_error)
int fail(d2pc_ctx *ctx, int status, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

#define D2PC_HIP(ctx, call)                                                                   \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(ctx, e_ == hipErrorOutOfMemory ? D2PC_ERR_OUT_OF_MEMORY : D2PC_ERR_DEVICE,  \
                  "%s failed: %s", #call, hipGetErrorString(e_));                             \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

inline size_t elem_size(int dtype) { return dtype == D2PC_DTYPE_F32 ? 4 : dtype == D2PC_DTYPE_U16 ? 2 : 1; }

// The synchronous host entry points must not return (even with an error) while
// work that reads the caller's input or writes the caller's output is in flight.
struct SyncOnExit {
  hipStream_t s;
  bool armed = true;
  explicit SyncOnExit(hipStream_t stream) : s(stream) {}
  ~SyncOnExit() {
    if (armed) (void)hipStreamSynchronize(s);
  }
};

// ---- d2pc_capi_route.hip ---------------------------------------------------------------------------------------------
int grow(d2pc_ctx *ctx, void **p, size_t *cap, size_t need);
int parity_pxt(const d2pc_ctx *ctx, int width, int height, int n_frames, int dtype = D2PC_DTYPE_F32);
int make_geom(d2pc_ctx *ctx, int dtype, float scale, int width, int height, size_t row_stride, size_t in_frame_stride,
              int n_frames, size_t out_frame_stride, int pxt, Geom *g);
void retile(Geom *g, int pxt);
double w_safe_for(const d2pc_ctx *ctx, const Geom &g);
int enqueue(d2pc_ctx *ctx, const Geom &g, const void *d_disp, int dtype, void *d_out, uint32_t *d_idx, uint32_t *d_counts,
            hipStream_t stream, StateBuf *fixed_state = nullptr, int force_algo = 0, uint32_t call_epoch_first = 0);

// ---- d2pc_capi_state.hip ---------------------------------------------------------------------------------------------
bool capture_info(hipStream_t s, unsigned long long *id);
bool settle(StateBuf &b);
bool state_idle(StateBuf &b);
int state_alloc(d2pc_ctx *ctx, const BufPool *pool, StateBuf &b, size_t need, size_t need2 = 0);
int eager_bufs(const BufPool &pool);
int acquire_buf(d2pc_ctx *ctx, BufPool &pool, hipStream_t stream, size_t need, size_t need2, StateBuf *fixed, StateBuf **out);
int reserve_buf(d2pc_ctx *ctx, BufPool &pool, size_t need, size_t need2);
void free_pool(BufPool &pool);

// ---- d2pc_capi_context.hip -------------------------------------------------------------------------------------------
int fill_q(d2pc_ctx *ctx, LaunchArgs &a, int width);
void classify_q(d2pc_ctx *ctx);

// ---- d2pc_capi_host.hip / d2pc_capi_mono.hip ---------------------------------------------------------------------------
void *pinned_device_view(const void *p, size_t bytes);
void median_roi_only(MedianArgs &m, const Geom &g, int height);

}  // namespace host
}  // namespace d2pc
