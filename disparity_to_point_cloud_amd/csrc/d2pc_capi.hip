// d2pc_capi.hip -- implementation of include/d2pc.h (the C-ABI drop-in
// boundary for Disparity2PCloud::DisparityCb's cpp:63-85).  Owns the device
// context: stream, staging buffers, compaction state.  No CPU compute path
// exists here: every d2pc_process* call runs the HIP kernels or fails.
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

using namespace d2pc;

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
  // dense single pass (algo 2): the buffer holds TWO states of pp_half bytes; a launch runs on one and zeroes the other
  // inside its own launch, for the next launch on this buffer (no k_state_clear kernel in front of every call)
  size_t pp_half = 0;            // bytes per half as the last such launch used them (0: none yet)
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
#else
  static constexpr int big_batch_algo = 2, resident_unbounded = 0, general_q_form = 0;  // (the product: the single pass; bounded; OpenCV's association)
#endif
  int resident_stagger_pct = -1;   // algo 3, register-resident form: scale of the ramped start in % (0 = every block loads at once;
                                   // -1 = choose: 50 for one frame that fills the device, else 0)
  int resident_pair = 0;           // algo 3: 1 = two 4K-class frames in ONE launch of 16,384-pixel blocks (0: two launches of 8,192-pixel blocks)
  int resident_pxt = 0;            // algo 3: pixels per thread of its blocks (0 = choose: the ordinary tile if the launch fits, else 32, else 64)
  int spin_timeout_ms = int(kDefaultSpinMs);  // single pass: hand-off wait budget
  int force_general_q = 0;
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

namespace {

constexpr int kDefaultOnepassForm = 2;  // profiles/r05_ab_onepass_forms_*.txt: never slower than form 1, 5-6 % faster with 30 % holes, 29 % with 90 %

int fail(d2pc_ctx *ctx, int status, const char *fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return status;
}

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

size_t elem_size(int dtype) { return dtype == D2PC_DTYPE_F32 ? 4 : dtype == D2PC_DTYPE_U16 ? 2 : 1; }

int grow(d2pc_ctx *ctx, void **p, size_t *cap, size_t need) {
  if (need <= *cap) return D2PC_OK;
  if (*p) {
    D2PC_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
  }
  size_t want = (need + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  D2PC_HIP(ctx, hipMalloc(p, want));
  *cap = want;
  return D2PC_OK;
}

// PARITY tile shape.  Default (round 3): ONE-SHOT blocks of 512 pixels, two per thread (k_reproject_pack_small) -- against
// the tile-walking kernel with 8 pixels per thread, interleaved on one device: 16 x 4K 427 -> 394 us (border 40), 457 -> 407 us
// (border 0); one 4K frame 24.9 -> 23.0 us; 64 x 752x480 51.2 -> 49.6 us; never slower (profiles/r03_sweep_parity_small.txt).
// One pixel per thread is as good for fp32 launches that fit the caches and 15 % worse for the big fp32 batch.
// 8- and 16-bit input -- what the reference's callback really holds (cpp:60-61) -- swept in round 4
// (profiles/r04_sweep_parity_small_u8.txt): two pixels per thread up to ~8 x 4K (2 x 4K: 38.1 against 45.5 us), ONE pixel per
// thread beyond (16 x 4K: u8 281.6 against 313.0 us, u16 298.2 against 325.7; equal at 8 x 4K).
// pxt_parity 4 / 8 / 16 select the tile-walking kernel (1024-pixel tiles were its best for launches of <= 32 Mpixel).
int parity_pxt(const d2pc_ctx *ctx, int width, int height, int n_frames, int dtype = D2PC_DTYPE_F32) {
  if (ctx->pxt_parity) return ctx->pxt_parity;
  if (dtype == D2PC_DTYPE_F32) return 2;
  const long long b = ctx->cfg.border, rw = (long long)width - 2 * b, rh = (long long)height - 2 * b;
  const long long px = rw > 0 && rh > 0 ? rw * rh * (long long)n_frames : 0;
  return px >= 96000000ll ? 1 : 2;
}

// Validates the frame description and fills the launch geometry.
int make_geom(d2pc_ctx *ctx, int dtype, float scale, int width, int height, size_t row_stride,
              size_t in_frame_stride, int n_frames, size_t out_frame_stride, int pxt, Geom *g) {
  if (dtype != D2PC_DTYPE_F32 && dtype != D2PC_DTYPE_U8 && dtype != D2PC_DTYPE_U16)
    return fail(ctx, D2PC_ERR_BAD_DTYPE, "dtype %d is not F32/U8/U16", dtype);
  if (width <= 0 || height <= 0) return fail(ctx, D2PC_ERR_BAD_SIZE, "bad image size %dx%d", width, height);
  if (uint64_t(width) * uint64_t(height) > (uint64_t(1) << 31))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "image %dx%d exceeds 2^31 pixels", width, height);
  const size_t es = elem_size(dtype);
  if (row_stride < size_t(width) * es || row_stride % es != 0)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "row stride %zu invalid for width %d (elem %zu B)", row_stride, width, es);
  if (n_frames <= 0 || n_frames > 65535) return fail(ctx, D2PC_ERR_BAD_SIZE, "bad frame count %d", n_frames);
  if (n_frames > 1 && (in_frame_stride < size_t(height) * row_stride || in_frame_stride % es != 0))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "input frame stride %zu too small", in_frame_stride);
  // 32-bit byte offsets inside a frame, with room for the tail slots of the
  // last tile (up to 16*256 pixels past the ROI end)
  if ((uint64_t(height) + 4097) * row_stride > 0xffffffffull)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "frame of %d rows x %zu bytes exceeds 32-bit addressing", height, row_stride);
  const int b = ctx->cfg.border;
  memset(g, 0, sizeof *g);
  g->width = uint32_t(width);
  g->border = uint32_t(b);
  g->roi_w = width > 2 * b ? uint32_t(width - 2 * b) : 0u;
  const uint32_t roi_h = height > 2 * b ? uint32_t(height - 2 * b) : 0u;
  if (uint64_t(g->roi_w) * roi_h > (uint64_t(1) << 28))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "ROI of %u x %u exceeds 2^28 points", g->roi_w, roi_h);
  g->roi_n = g->roi_w * roi_h;
  if (n_frames > 1 && out_frame_stride < g->roi_n)
    return fail(ctx, D2PC_ERR_CAPACITY, "output frame stride %zu < %u ROI points", out_frame_stride, g->roi_n);
  const uint32_t tile_px = uint32_t(kBlock) * uint32_t(pxt);
  g->tiles_per_frame = (g->roi_n + tile_px - 1) / tile_px;
  g->n_frames = uint32_t(n_frames);
  const uint64_t total = uint64_t(g->tiles_per_frame) * g->n_frames;
  if (total > 0x7fffffffull) return fail(ctx, D2PC_ERR_BAD_SIZE, "batch too large (%llu tiles)", (unsigned long long)total);
  g->total_tiles = uint32_t(total);
  g->groups_per_frame = (g->tiles_per_frame + kGroupTiles - 1) / kGroupTiles;
  g->frame_state_stride = frame_state_stride(g->tiles_per_frame);
  const uint32_t rw = g->roi_w ? g->roi_w : 1u;
  g->s64_v = 64u / rw;
  g->s64_u = 64u % rw;
  g->s832_v = 832u / rw;
  g->s832_u = 832u % rw;
  g->s1024_v = 1024u / rw;
  g->s1024_u = 1024u % rw;
  g->div_roi_w = make_fastdiv(rw);
  g->div_tpf = make_fastdiv(g->tiles_per_frame ? g->tiles_per_frame : 1u);
  g->row_stride = uint32_t(row_stride);
  g->last_off = g->roi_n ? uint32_t((uint64_t(b) + roi_h - 1) * row_stride + (uint64_t(b) + g->roi_w - 1) * es) : 0u;
  g->in_frame_stride = in_frame_stride;
  g->out_frame_stride = out_frame_stride;
  g->scale = scale;
  g->min_disparity = ctx->cfg.min_disparity;
  g->spin_ticks = uint32_t(ctx->spin_timeout_ms) * kSpinTicksPerMs;
  g->pxt = uint32_t(pxt);
  return D2PC_OK;
}

// The same frames cut into tiles of 256 * pxt ROI pixels.
void retile(Geom *g, int pxt) {
  const uint32_t tile_px = uint32_t(kBlock) * uint32_t(pxt);
  g->tiles_per_frame = (g->roi_n + tile_px - 1) / tile_px;
  g->total_tiles = uint32_t(uint64_t(g->tiles_per_frame) * g->n_frames);
  g->groups_per_frame = (g->tiles_per_frame + kGroupTiles - 1) / kGroupTiles;
  g->frame_state_stride = frame_state_stride(g->tiles_per_frame);
  g->div_tpf = make_fastdiv(g->tiles_per_frame ? g->tiles_per_frame : 1u);
  g->pxt = uint32_t(pxt);
}

// Is `s` capturing, and if so which capture?
bool capture_info(hipStream_t s, unsigned long long *id) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  if (hipStreamGetCaptureInfo(s, &st, &cid) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (id) *id = cid;
  return st != hipStreamCaptureStatusNone;
}

// Makes b.done cover everything enqueued on b.stream so far.  False if that cannot be done now (the buffer's stream is
// inside a capture: a record there would become a node of somebody's graph).
bool settle(StateBuf &b) {
  if (!b.dirty) return true;
  if (capture_info(b.stream, nullptr)) return false;
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;  // (another stream of this thread may be capturing: see state_idle)
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  // has the stream drained?  Then nothing of the buffer's is in flight and no event is needed (an event recorded NOW would
  // read "not ready" for the microseconds its marker takes, and a capture looking for an idle buffer would find none)
  hipError_t e = hipStreamQuery(b.stream);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    e = b.done ? hipEventRecord(b.done, b.stream) : hipErrorInvalidHandle;
    if (e == hipSuccess) {
      (void)hipThreadExchangeStreamCaptureMode(&mode);
      b.dirty = false;
      b.pending = true;
      return true;
    }
  }
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  b.dirty = false;
  if (e != hipSuccess) {
    // the stream is gone (destroyed by its owner: its work completes regardless): wait for the device instead
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
  }
  b.pending = false;
  return true;
}

bool state_idle(StateBuf &b) {
  if (!settle(b)) return false;
  if (!b.pending) return true;
  // another stream of this thread may be capturing (that is when a captured launch looks for a free
  // buffer): an event query is "unsafe" under the global/thread-local capture modes and would
  // invalidate the capture, so it runs under the relaxed mode; `done` is never part of a capture
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  const hipError_t e = hipEventQuery(b.done);
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (e == hipSuccess) {
    b.pending = false;
    return true;
  }
  (void)hipGetLastError();  // hipErrorNotReady is not an error here
  return false;
}

int state_alloc(d2pc_ctx *ctx, const BufPool *pool, StateBuf &b, size_t need, size_t need2 = 0) {
  if (pool && need < pool->reserve) need = pool->reserve;
  if (pool && need2 < pool->reserve2) need2 = pool->reserve2;
  if (!b.done) D2PC_HIP(ctx, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
  if (b.cap >= need && b.cap2 >= need2) return D2PC_OK;
  if ((b.p || b.p2) && (b.pending || b.dirty)) {
    if (!settle(b)) return fail(ctx, D2PC_ERR_DEVICE, "a buffer that must grow is in use by a stream that is being captured");
    if (b.pending) D2PC_HIP(ctx, hipEventSynchronize(b.done));  // its last launch still reads and writes it
    b.pending = false;
  }
  if (b.cap < need) {
    // the buffer moves: whatever its last launch left in it is gone (the chunked two-pass would otherwise skip the clear of
    // a buffer it believes still holds its counters and "empty" marks; advisor, round 4)
    b.algo = 0;
    b.chunk_sig = 0;
    b.epoch = 0;
    b.pp_clean = false;
    b.hdr_off = 0;
  }
  int st = grow(ctx, &b.p, &b.cap, need);
  // fresh memory starts zeroed: k_compact_resident tells "published by THIS launch" from anything older by the epoch
  // in the word, and an uninitialised word could hold any value
  // (hipMemset of device memory may return before the fill has run, and the launches that follow go to streams
  // that do not wait for the NULL stream: a late fill wiped a running launch's header -- its pointer to the counters
  // included.  So: fill, then wait for it.)
  if (st == D2PC_OK && b.p) {
    D2PC_HIP(ctx, hipMemsetAsync(b.p, 0, b.cap, nullptr));
    D2PC_HIP(ctx, hipStreamSynchronize(nullptr));
  }
  if (st == D2PC_OK && need2) st = grow(ctx, &b.p2, &b.cap2, need2);
  return st;
}

int eager_bufs(const BufPool &pool) {
  int n = 0;
  for (const StateBuf *b : pool.bufs) n += b->captured ? 0 : 1;
  return n;
}

// A buffer of the pool for work of `need` (+ `need2`) bytes on `stream`.
//  * launches on ONE stream are ordered, so a stream keeps reusing its buffer;
//  * a launch on another stream takes a buffer whose last launch has completed, or a new one -- two
//    launches that may overlap never share tickets / partial counts / granules / filtered frames;
//  * during stream capture nothing can be allocated, and the pointer is baked into the graph: the buffer
//    must exist already (d2pc_reserve / d2pc_reserve_mono) and from then on belongs to that capture alone --
//    it is never freed, grown or handed to another launch until d2pc_release_graph_buffers, so replaying the
//    graph stays valid whatever is called later.  Launches of one capture share a buffer only when they are
//    captured on the SAME stream (ordered inside the graph); a forked capture stream gets its own.
int acquire_buf(d2pc_ctx *ctx, BufPool &pool, hipStream_t stream, size_t need, size_t need2, StateBuf *fixed,
                StateBuf **out) {
  unsigned long long cid = 0;
  const bool capturing = capture_info(stream, &cid);
  if (fixed) {  // pipeline slot: the slot's stream orders everything that touches its buffer
    if (capturing) return fail(ctx, D2PC_ERR_INVALID_ARG, "pipeline streams cannot be captured");
    int st = state_alloc(ctx, nullptr, *fixed, need);
    if (st != D2PC_OK) return st;
    fixed->stream = stream;
    fixed->bound = true;
    *out = fixed;
    return D2PC_OK;
  }
  auto fits = [&](const StateBuf *b) { return b->cap >= need && b->cap2 >= need2; };
  StateBuf *pick = nullptr;
  if (capturing) {
    for (StateBuf *b : pool.bufs)  // an earlier launch of the same capture on the same stream: ordered inside the graph
      if (b->captured && b->capture_id == cid && b->bound && b->stream == stream && fits(b)) pick = b;
    if (!pick)
      for (StateBuf *b : pool.bufs)
        // the capturing stream's OWN eager buffer (the flow d2pc.h documents: run the largest batch once, then capture on
        // the same stream): whatever that run left in flight is ordered before the capture by the caller's stream, exactly
        // as for the output buffers -- settle() could never prove it idle, because the stream it would ask is capturing now
        // (advisor, round 4: the captured launch failed with "reserve it" although the buffer was there)
        if ((b->p || b->p2) && !b->captured && b->bound && b->stream == stream && fits(b)) {
          pick = b;
          b->dirty = b->pending = false;
        }
    if (!pick)
      for (StateBuf *b : pool.bufs)
        if ((b->p || b->p2) && !b->captured && fits(b) && state_idle(*b) && (!pick || b->cap < pick->cap)) pick = b;
    if (!pick)
      return fail(ctx, D2PC_ERR_OUT_OF_MEMORY,
                  "no free %s of %zu bytes for a captured launch: reserve it (d2pc_reserve / d2pc_reserve_mono for the "
                  "largest batch) before every capture", pool.what, need + need2);
    pick->captured = true;
    pick->capture_id = cid;
  } else {
    for (StateBuf *b : pool.bufs)
      if (!b->captured && b->bound && b->stream == stream) pick = b;
    if (!pick)  // the smallest idle buffer that fits, else any idle one (it is grown), else a new one
      for (StateBuf *b : pool.bufs)
        if (!b->captured && state_idle(*b) && fits(b) && (!pick || b->cap < pick->cap)) pick = b;
    if (!pick)
      for (StateBuf *b : pool.bufs)
        if (!b->captured && state_idle(*b)) pick = b;
    if (!pick && eager_bufs(pool) < kMaxEagerBufs) {
      pick = new (std::nothrow) StateBuf();
      if (!pick) return fail(ctx, D2PC_ERR_OUT_OF_MEMORY, "out of host memory");
      pool.bufs.push_back(pick);
    }
    if (!pick)  // every eager buffer is busy on some other stream: wait for one
      for (StateBuf *b : pool.bufs)
        if (!b->captured && !pick) pick = b;
    int st = state_alloc(ctx, &pool, *pick, need, need2);  // waits for the buffer's last launch before it frees anything
    if (st != D2PC_OK) return st;
    if ((pick->pending || pick->dirty) && !(pick->bound && pick->stream == stream)) {
      // taken over from another stream while busy (only when all buffers were busy): order behind it
      if (!settle(*pick)) return fail(ctx, D2PC_ERR_DEVICE, "every buffer is in use and one belongs to a stream that is being captured");
      if (pick->pending) D2PC_HIP(ctx, hipStreamWaitEvent(stream, pick->done, 0));
    }
  }
  pick->stream = stream;
  pick->bound = true;
  *out = pick;
  return D2PC_OK;
}

// d2pc_reserve / d2pc_reserve_mono: ONE idle buffer of the pool that no graph owns, of at least this size
int reserve_buf(d2pc_ctx *ctx, BufPool &pool, size_t need, size_t need2) {
  if (need > pool.reserve) pool.reserve = need;
  if (need2 > pool.reserve2) pool.reserve2 = need2;
  need = pool.reserve;
  need2 = pool.reserve2;
  StateBuf *pick = nullptr;
  for (StateBuf *b : pool.bufs)
    if (!b->captured && state_idle(*b) && b->cap >= need && b->cap2 >= need2) return D2PC_OK;
  for (StateBuf *b : pool.bufs)  // an idle one that is too small is grown
    if (!b->captured && state_idle(*b) && !pick) pick = b;
  if (!pick && eager_bufs(pool) < kMaxEagerBufs) {
    pick = new (std::nothrow) StateBuf();
    if (!pick) return fail(ctx, D2PC_ERR_OUT_OF_MEMORY, "out of host memory");
    pool.bufs.push_back(pick);
  }
  if (!pick)
    for (StateBuf *b : pool.bufs)
      if (!b->captured && !pick) pick = b;  // all busy: state_alloc waits for this one
  return state_alloc(ctx, &pool, *pick, need, need2);
}

void free_pool(BufPool &pool) {
  for (StateBuf *b : pool.bufs) {
    if (b->p) (void)hipFree(b->p);
    if (b->p2) (void)hipFree(b->p2);
    if (b->done) (void)hipEventDestroy(b->done);
    delete b;
  }
  pool.bufs.clear();
}

// Q for a launch over frames `width` columns wide: which kernel (specialised / general) and, for the general one, in
// which arithmetic form (QMat::form); for OpenCV 2.4's form the table of its running column sum (cached per width).
int fill_q(d2pc_ctx *ctx, LaunchArgs &a, int width) {
  memcpy(a.q.q, ctx->q, sizeof a.q.q);
  a.qs = ctx->qs;
  a.q_kind = ctx->force_general_q ? QK_GENERAL : ctx->q_kind;
  a.q.form = ctx->general_q_form == 1 ? 1u : 0u;  // (1: experiment build only)
  a.q.seg = QxSegs{};
  a.q.seg.n = 1;
  if (ctx->reproject_form == 0) return D2PC_OK;
  // One OpenCV generation bit for bit.  cv::stereoRectify's Q keeps specialised kernels (QK_STEREO_CV24 / _CV4: the
  // generation's roundings of W and of the numerators, d2pc_device.hpp); any other Q -- and tuning
  // "force_general_q", which lets the tests compare the two routes -- goes through the general kernel.
  const bool stereo = a.q_kind == QK_STEREO;
  if (ctx->reproject_form == 4) {
    if (stereo) {
      a.q_kind = QK_STEREO_CV4;
      a.qs.f = double(float(a.qs.f));  // Vec3f p = Vec3d(h.val): Z's numerator is the float of f
    } else {
      a.q_kind = QK_GENERAL;
      a.q.form = 0;
    }
    return D2PC_OK;
  }
  const double *q = ctx->q;
  auto pz = [](double x) { uint64_t b; memcpy(&b, &x, 8); return b == 0; };
  if (!(q[0] == 1.0 && pz(q[1]) && pz(q[4]) && pz(q[8]) && pz(q[12])))
    return fail(ctx, D2PC_ERR_INVALID_ARG,
                "reproject_form 24 (OpenCV 2.4's loop bit for bit) needs a Q whose column increments are exact "
                "(q00 = 1, q01 = q10 = q20 = q30 = +0, as cv::stereoRectify's): its x-recurrence has no parallel form otherwise");
  if (ctx->qx_width < uint32_t(width)) {  // replay qx = q01*y + q03, then += q00 per column (one rounding per step)
    volatile double s = 0.0 + q[3];       // (+0)*y = +0 for every row
    QxSegs sg{};
    sg.n = 1;
    sg.x[0] = 0;
    sg.c[0] = s;
    // (only the columns this launch has: a small non-dyadic principal point crosses one binade per doubling of the
    // column, and a table replayed over 4096 columns whatever the width refused Qs that a narrow frame can serve)
    const uint32_t cols = uint32_t(width);
    for (uint32_t x = 1; x < cols; ++x) {
      s = s + q[0];
      const double c = s - double(x);
      if (double(x) + c != s) return fail(ctx, D2PC_ERR_INTERNAL, "2.4-form column sum not representable as x + c at column %u", x);
      if (c != sg.c[sg.n - 1]) {
        if (sg.n == uint32_t(kQxSegs))
          return fail(ctx, D2PC_ERR_BAD_SIZE, "reproject_form 24: the 2.4-form column sum changes its rounding more than %d times "
                      "within %u columns for this principal point", kQxSegs, cols);
        sg.x[sg.n] = x;
        sg.c[sg.n] = c;
        ++sg.n;
      }
    }
    ctx->qx_seg = sg;
    ctx->qx_width = cols;
  }
  a.q.seg = ctx->qx_seg;
  if (stereo) {
    a.q_kind = QK_STEREO_CV24;
  } else {
    a.q_kind = QK_GENERAL;
    a.q.form = 2;
  }
  return D2PC_OK;
}

// bound on |u + cx|, |v + cy|, |f| over the frame, scaled by 2^-126: any |W| at least this large keeps every
// quotient below 2^126 < FLT_MAX (QStereo::w_safe: the exact validity predicate of the COMPACT kernels)
double w_safe_for(const d2pc_ctx *ctx, const Geom &g) {
  const double height = double((g.last_off / (g.row_stride ? g.row_stride : 1u)) + 1u);
  const double mx = std::fmax(std::fabs(ctx->qs.cx), std::fabs(ctx->qs.cx + double(g.width)));
  const double my = std::fmax(std::fabs(ctx->qs.cy), std::fabs(ctx->qs.cy + height));
  const double m = std::fmax(std::fabs(ctx->qs.f), std::fmax(mx, my));
  return std::isfinite(m) ? std::ldexp(m, -126) : std::numeric_limits<double>::infinity();
}

int enqueue(d2pc_ctx *ctx, const Geom &g, const void *d_disp, int dtype, void *d_out, uint32_t *d_idx,
            uint32_t *d_counts, hipStream_t stream, StateBuf *fixed_state = nullptr, int force_algo = 0) {
  LaunchArgs a;
  a.disp = d_disp;
  a.out_points = d_out;
  a.out_index = d_idx;
  a.counts = d_counts;
  a.dtype = dtype;
  a.stream = stream;
  a.geom = g;
  {
    int stq = fill_q(ctx, a, int(g.width));
    if (stq != D2PC_OK) return stq;
  }
  // 16-B row loads need every aligned group of four ROI pixels to sit in one
  // row at a 16-B aligned address
  a.vec_rows = !ctx->no_vec_rows && dtype == D2PC_DTYPE_F32 && g.roi_w % 4 == 0 && g.border % 4 == 0 &&
               g.row_stride % 16 == 0 && g.in_frame_stride % 16 == 0 && reinterpret_cast<uintptr_t>(d_disp) % 16 == 0;
  a.qs.w_safe = w_safe_for(ctx, g);
  // grid: many more blocks than fit (the dispatcher keeps the CUs fed as blocks
  // retire), each walking a few tiles: min(T, max(CUs*blocks_per_cu, T/4))
  uint32_t want = uint32_t(ctx->cu_count) * uint32_t(ctx->blocks_per_cu);
  const uint32_t quarter = (g.total_tiles + 3) / 4;
  if (want < quarter) want = quarter;
  a.grid = g.total_tiles < want ? g.total_tiles : want;
  if (a.grid == 0) a.grid = 1;
  if (ctx->cfg.mode == D2PC_MODE_PARITY) {
    a.pxt = int(g.pxt);
    a.parity_small = g.pxt <= 2 || (ctx->parity_small == 1 && g.pxt == 4);
    D2PC_HIP(ctx, launch_parity(a));
    return D2PC_OK;
  }
  if (!d_counts) return fail(ctx, D2PC_ERR_INVALID_ARG, "COMPACT mode needs a d_counts buffer");
  a.pxt = int(g.pxt);
  // TWO frames that fit the resident blocks only as one launch of 16,384-pixel blocks (two 4K frames): two launches of
  // 8,192-pixel blocks, back to back on the stream, instead.  Measured on the driver's device in round 4: 65.6 us for the
  // pair in one launch (every block twice as long in its read-then-write chain, no ramped start: 9 us of waiting per block)
  // against 2 x 27.0 us; the back-to-back launches leave no gap since the buffers' events are recorded lazily.
  // Tuning "resident_pair" = 1 keeps the one launch (tools/ab_resident.sh).
  if (g.n_frames == 2 && !force_algo && (ctx->cfg.compact_algo == 0 || ctx->cfg.compact_algo == 3) && !ctx->resident_pxt &&
      !ctx->resident_pair && !capture_info(stream, nullptr)) {
    const uint32_t cap = uint32_t(ctx->cu_count * kResidentBlocksPerCu);
    const uint32_t tpf32 = (g.roi_n + uint32_t(kBlock * 32) - 1u) / uint32_t(kBlock * 32);
    if (2u * g.tiles_per_frame > cap && 2u * tpf32 > cap && tpf32 <= cap) {
      for (uint32_t f = 0; f < 2u; ++f) {
        Geom g1 = g;
        g1.n_frames = 1;
        g1.total_tiles = g1.tiles_per_frame;
        int st1 = enqueue(ctx, g1, static_cast<const uint8_t *>(d_disp) + uint64_t(f) * g.in_frame_stride, dtype,
                          static_cast<uint8_t *>(d_out) + uint64_t(f) * g.out_frame_stride * 16u,
                          d_idx ? d_idx + uint64_t(f) * g.out_frame_stride : nullptr, d_counts + f, stream, fixed_state, 0);
        if (st1 != D2PC_OK) return st1;
      }
      return D2PC_OK;
    }
  }
  // default (0): the single pass (one read of the input) wins once a launch is big enough to amortise
  // its pipeline fill -- measured crossover ~25k tiles (16 x 4K: 449 vs 495 us; 32 x 1080p: 196 vs 207;
  // 256 x 752x480: 278 vs 290; but 8 x 1080p: 68 vs 61) -- and needs a few frames in flight, because a
  // frame's ticket word serialises at ~18 ns per tile (one 4K frame: 72 vs 35 us)
  // (round 4, profiles/r04_ab_midrange.txt: the crossover is where the input stops fitting the Infinity Cache between the two-pass
  // form's two reads, ~160 MB = ~20k tiles of fp32 -- 6 x 4K (22.9k tiles): single pass 166 us, two-pass 184; 4 x 4K (15.3k): 117 / 114;
  // 16 x 1080p (14.4k): 112 / 103.  The threshold was 24,576 before, which sent 6 x 4K the slower way.)
  const bool big_batch = g.n_frames >= 4 && g.total_tiles >= 20480;
  // camera-size launches whose tiles are all resident at once take ONE launch (k_compact_resident) unless the call
  // is being captured (its epoch argument would freeze in the graph); one 1080p frame 16 -> ~8 us
  // ... in the ordinary tiles (k_compact_resident), or -- one or two 4K frames -- in blocks of 32 / 64 pixels per thread
  // that keep their disparities in registers between count and scatter (k_compact_resident_lean)
  const uint32_t resident_cap = uint32_t(ctx->cu_count * kResidentBlocksPerCu);
  const bool not_captured = !capture_info(stream, nullptr);
  int resident_pxt = 0;
  for (int r : {int(g.pxt), 32, 64}) {
    const uint32_t tpf = (g.roi_n + uint32_t(kBlock * r) - 1u) / uint32_t(kBlock * r);
    if (!resident_pxt && uint64_t(tpf) * g.n_frames <= resident_cap && tpf <= 1024u) resident_pxt = r;
  }
  if (ctx->resident_pxt) {  // (tuning: force one of the shapes where it fits)
    const int r = ctx->resident_pxt;
    const uint32_t tpf = (g.roi_n + uint32_t(kBlock * r) - 1u) / uint32_t(kBlock * r);
    // EXPERIMENT "resident_unbounded": more blocks than fit at once.  A block waits for lower-numbered blocks of its frame
    // only, so this is safe exactly if every XCD starts its share of the grid in index order (then the lowest unfinished
    // block always runs); the time-out turns a violation into kCountTimedOut, not a hang
    const bool fits = uint64_t(tpf) * g.n_frames <= resident_cap || (ctx->resident_unbounded && r >= 32 && uint64_t(tpf) * g.n_frames <= 0x7fffffffull);
    resident_pxt = (fits && (tpf <= 1024u || ctx->resident_unbounded)) ? r : 0;
  }
  const bool resident_ok = resident_pxt != 0 && not_captured;
  const int dflt = big_batch ? ctx->big_batch_algo : resident_ok ? 3 : 1;
  a.compact_algo = force_algo ? force_algo : ctx->cfg.compact_algo ? ctx->cfg.compact_algo : dflt;
  if (a.compact_algo == 3 && !resident_ok) a.compact_algo = big_batch ? ctx->big_batch_algo : 1;  // (asked for, not possible here)
  if (a.compact_algo == 3 && resident_pxt != int(g.pxt)) {
    Geom gr = g;
    retile(&gr, resident_pxt);
    // the ramped start of k_compact_resident_lean: a block's input bytes at ~6 TB/s and ~2.4 GHz, in 64-cycle sleeps x 1024
    // (R = 32, fp32: 32 KiB per block = 13 cycles = 0.2 sleeps per block index); tuning "resident_stagger_pct" scales it
    // Measured (profiles/r04_ab_stagger.txt): half that ramp is worth 1.4-1.8 us on ONE 4K frame (31.2 -> 29.8 us with 30 % holes
    // + indices, 32.2 -> 30.4 all valid) and nothing or less on two frames and on smaller ones, whose blocks do not fill the
    // device: it is applied to single frames of >= 7/8 of the resident capacity only.
    const double block_bytes = double(kBlock) * resident_pxt * double(elem_size(dtype));
    const bool ramp = ctx->resident_stagger_pct >= 0 ? true : (gr.n_frames == 1 && dtype == D2PC_DTYPE_F32 && gr.total_tiles * 8u >= resident_cap * 7u);
    const int pct = ctx->resident_stagger_pct >= 0 ? ctx->resident_stagger_pct : 50;
    gr.stagger = ramp ? uint32_t(block_bytes / 6.0e12 * 2.4e9 / 64.0 * 1024.0 * pct / 100.0) : 0u;
    a.geom = gr;
    a.pxt = resident_pxt;
  }
#if D2PC_EXPERIMENTS
  if (a.compact_algo == 4) {
    // chunked two-pass (k_compact_chunk): the geometry in its own 512-pixel tiles; chunks of whole frames whose input
    // stays in the Infinity Cache between the launch that counts it and the launch that scatters it
    if (g.roi_n == 0) return fail(ctx, D2PC_ERR_INTERNAL, "empty ROI reached the compaction launch");
    Geom g4 = g;
    retile(&g4, 2);
    if (uint64_t(g4.tiles_per_frame) * g4.n_frames > 0x7fffffffull) return fail(ctx, D2PC_ERR_BAD_SIZE, "batch too large");
    uint32_t gw = 0;
    g4.frame_state_stride = chunk_frame_state_stride(g4.tiles_per_frame, &gw);
    a.geom = g4;
    a.pxt = 2;
    const uint64_t frame_bytes = uint64_t(g.roi_n) * elem_size(dtype);
    uint64_t per = (uint64_t(ctx->chunk_mb) << 20) / (frame_bytes ? frame_bytes : 1);
    if (per < 1) per = 1;
    if (per > g.n_frames) per = g.n_frames;
    a.chunk_frames = uint32_t(per);
    // the first chunk is counted with nothing to run beside it: an eighth of a chunk (a 4K stream: one frame)
    a.chunk_first = ctx->chunk_first_frames > 0 ? uint32_t(ctx->chunk_first_frames) : uint32_t((per + 7) / 8);
    if (a.chunk_first > a.chunk_frames) a.chunk_first = a.chunk_frames;
  }
#endif
  if (a.compact_algo == 2) {
    // the single-pass kernel is software-pipelined over a block's tiles: it
    // wants few, long-lived blocks (about what is resident), not many short ones
    // interleaved sweeps on two devices (profiles/r02_ab_onepass_v2_vs_r1.txt): 4K frames run 1-3 % faster with 3
    // blocks per CU (fewer failed polls), 1080p-class frames 1-4 % faster with 4
    // which single-pass kernel (tuning "onepass_form"; same bytes out): 2 = the count phase packs the survivors, the scatter
    // phase runs dense (round 5: the product's); experiment build: 1 = raw tiles in LDS, every pixel decided in both phases
    // (rounds 2-4), 3 = form 2 with 8 worker waves on tiles of 4,096 pixels, 4 = form 2 with the control wave as the loader
    // 5 = form 2 with 4 runs per worker wave (4,096-pixel tiles), 6 = form 2 with deferred landing (a tile's loads fly for a
    // whole iteration), 7 = 5 + 6: profiles/r05_ab_forms567.txt
    a.onepass_form = ctx->onepass_form ? ctx->onepass_form : kDefaultOnepassForm;
    const bool big_tiles = a.onepass_form == 3 || a.onepass_form == 5 || a.onepass_form == 7;
    const int form_pxt = big_tiles ? 16 : a.onepass_form >= 2 ? 8 : int(g.pxt);
    if (form_pxt != int(g.pxt)) {
      Geom gf = g;
      retile(&gf, form_pxt);
      a.geom = gf;
      a.pxt = form_pxt;
    }
    const int dflt_per_cu = big_tiles ? 2 : (a.geom.tiles_per_frame >= 2048 ? 3 : 4);
    const int per_cu = ctx->onepass_blocks_per_cu ? ctx->onepass_blocks_per_cu : dflt_per_cu;
    const uint32_t persistent = uint32_t(ctx->cu_count) * uint32_t(per_cu);
    a.grid = a.geom.total_tiles < persistent ? a.geom.total_tiles : persistent;
    if (a.grid < g.n_frames) {  // more frames than blocks: every block serves one frame only
      a.compact_algo = 1;
      a.geom = g;
      a.pxt = int(g.pxt);
    }
  }
  if (a.compact_algo == 3) {
    a.grid = a.geom.total_tiles;
    a.epoch = ctx->resident_epoch++;
    if (ctx->resident_epoch >= kEpochEnd) {  // (once in 2^30 launches: start over from clean state)
      std::vector<StateBuf *> all(ctx->states.bufs);
      for (PipeSlot &sl : ctx->slots) all.push_back(&sl.st);  // the pipeline slots' own state buffers carry epochs too
      for (StateBuf *b : all)
        if (b->p) {
          if (!settle(*b)) return fail(ctx, D2PC_ERR_DEVICE, "epoch wrap-around while a stream that uses the context is being captured");
          if (b->pending) D2PC_HIP(ctx, hipEventSynchronize(b->done));
          b->pending = false;
          D2PC_HIP(ctx, hipMemsetAsync(b->p, 0, b->cap, nullptr));
          D2PC_HIP(ctx, hipStreamSynchronize(nullptr));
        }
      ctx->resident_epoch = kEpochBase;  // (only now: a failure above leaves the counter past the end and the next launch tries again)
    }
  }
  if (a.compact_algo == 1) {  // (the two-pass grid is the default one computed above)
    uint32_t want2 = uint32_t(ctx->cu_count) * uint32_t(ctx->blocks_per_cu);
    if (want2 < quarter) want2 = quarter;
    a.grid = g.total_tiles < want2 ? g.total_tiles : want2;
    if (a.grid == 0) a.grid = 1;
  }
  a.state_bytes = compact_state_bytes(a.geom);
  a.stats = ctx->d_stats;
  // the dense single pass cleans up for its successor: two states per buffer (see StateBuf::pp_*)
  const bool self_clean = a.compact_algo == 2 && a.onepass_form >= 2 && a.onepass_form != 4;  // (every dense form)
  const size_t half = (a.state_bytes + 255) & ~size_t(255);
  StateBuf *sb = nullptr;
  int st = acquire_buf(ctx, ctx->states, stream, self_clean ? 2 * half : a.state_bytes, 0, fixed_state, &sb);
  if (st != D2PC_OK) return st;
  a.state = sb->p;
  if (self_clean) {
    const bool clean = sb->pp_clean && sb->pp_half == half && !sb->captured;
    const int h = clean ? sb->pp_next : 0;
    a.state = static_cast<uint8_t *>(sb->p) + size_t(h) * half;
    a.state_is_clean = clean;
    // a captured launch replays on the half baked into it: it keeps the clear kernel in front and cleans nothing
    a.state_other = sb->captured ? nullptr : static_cast<uint8_t *>(sb->p) + size_t(h ^ 1) * half;
    sb->pp_half = half;
    sb->pp_next = h ^ 1;
    sb->pp_clean = !sb->captured;
    sb->hdr_off = size_t(h) * half;
  } else {
    sb->pp_clean = false;
    sb->hdr_off = 0;
  }
#if D2PC_EXPERIMENTS
  if (a.compact_algo == 4) {
    // (tiles per frame, frames): they fix the groups, the padded words and the stride -- two shapes may share a 256-byte-rounded
    // stride and still keep their group totals in different words, and a stale word that is not "empty" would be taken for a total
    const uint64_t sig = (uint64_t(a.geom.tiles_per_frame) << 32) | a.geom.n_frames;
    a.chunk_clear = sb->algo != 4 || sb->chunk_sig != sig;
    sb->chunk_sig = sig;
  }
#endif
  sb->algo = a.compact_algo;
  sb->epoch = a.epoch;
  const hipError_t launched = launch_compact(a);
  if (launched != hipSuccess) sb->pp_clean = false;  // (nothing ran: the half this launch was to zero for its successor is still dirty)
  D2PC_HIP(ctx, launched);
  if (!sb->captured) sb->dirty = true;  // (a captured buffer is never shared; for the others `done` is recorded when somebody asks)
  return D2PC_OK;
}

// Does Q have the structure cv::stereoRectify produces (hpp:104)?
//   [1 0 0 cx; 0 1 0 cy; 0 0 0 f; 0 0 a b], zeros being +0.0 bit patterns.
// Then the nine products with +0.0 / 1.0 are exact and the specialised kernel
// returns bit-identical results (see reproject(QK_STEREO) in d2pc_pixel.hpp).
void classify_q(d2pc_ctx *ctx) {
  const double *q = ctx->q;
  auto pz = [](double x) { uint64_t b; memcpy(&b, &x, 8); return b == 0; };
  const bool stereo = q[0] == 1.0 && q[5] == 1.0 && pz(q[1]) && pz(q[2]) && pz(q[4]) && pz(q[6]) && pz(q[8]) &&
                      pz(q[9]) && pz(q[10]) && pz(q[12]) && pz(q[13]);
  ctx->q_kind = stereo ? QK_STEREO : QK_GENERAL;
  // the row constants as the general evaluation forms them: q_3 + (+0.0)
  volatile double z = 0.0;
  ctx->qs.cx = q[3] + z;
  ctx->qs.cy = q[7] + z;
  ctx->qs.f = q[11] + z;
  ctx->qs.a = q[14];
  ctx->qs.b = q[15] + z;
}

// Device-visible address of `p` when it lies in pinned host memory whose mapping covers `bytes` (memory from
// d2pc_host_alloc / hipHostMalloc / hipHostRegister), else nullptr.
void *pinned_device_view(const void *p, size_t bytes) {
  if (!p) return nullptr;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: not an error here
    return nullptr;
  }
  if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
  hipPointerAttribute_t end;
  if (bytes > 1 && (hipPointerGetAttributes(&end, static_cast<const char *>(p) + bytes - 1) != hipSuccess ||
                    end.type != hipMemoryTypeHost)) {
    (void)hipGetLastError();
    return nullptr;
  }
  return at.devicePointer;
}

// cpp:55-57 + cpp:69-72: the reference filters the whole image and then reads only the inset ROI ("Removing
// borders" -- the inset exists to hide the filter's border artefacts).  The fused entry points therefore
// compute the median of the ROI pixels only (25.5 % fewer at the native 752x480, border 40); the windows
// still read the unfiltered image up to its true edges, so every ROI pixel equals the whole-image result.
void median_roi_only(MedianArgs &m, const Geom &g, int height) {
  const uint32_t roi_h = uint32_t(height) > 2u * g.border ? uint32_t(height) - 2u * g.border : 0u;
  if (g.roi_w == 0 || roi_h == 0) return;
  m.out_x0 = m.out_y0 = g.border;
  m.out_w = g.roi_w;
  m.out_h = roi_h;
}

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

}  // namespace

extern "C" {

int d2pc_abi_version(void) { return D2PC_ABI_VERSION; }

const char *d2pc_status_string(int s) {
  switch (s) {
    case D2PC_OK: return "ok";
    case D2PC_ERR_INVALID_ARG: return "invalid argument";
    case D2PC_ERR_BAD_DTYPE: return "unsupported disparity dtype";
    case D2PC_ERR_BAD_SIZE: return "bad image size or stride";
    case D2PC_ERR_CAPACITY: return "output capacity too small";
    case D2PC_ERR_NO_DEVICE: return "no usable HIP device";
    case D2PC_ERR_DEVICE: return "HIP runtime error";
    case D2PC_ERR_NOT_CALIBRATED: return "Q matrix not set";
    case D2PC_ERR_OUT_OF_MEMORY: return "out of device memory";
    case D2PC_ERR_INTERNAL: return "internal error (compaction hand-off timed out)";
  }
  return "unknown status";
}

int d2pc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// hpp:84-104: cv::stereoRectify closed form for the reference rig (see
// SURVEY.md section 8 row a9): f' = fy; c' = (n-1)/2 - f'((n-1)/2 - c)/f;
// Q = [1 0 0 -cx'; 0 1 0 -cy'; 0 0 0 f'; 0 0 -1/Tx (cx1'-cx2')/Tx], Tx = -b.
int d2pc_make_q_flavour(double fx, double fy, double cx, double cy, double baseline, int nx, int ny, int flavour,
                        double q[16]) {
  if (!q || !(fx > 0) || !(fy > 0) || !(baseline != 0) || nx <= 0 || ny <= 0) return D2PC_ERR_INVALID_ARG;
  const double f = fy;
  double hx, hy, ox, oy;  // centre of the undistorted corners, and the centre the result is re-centred on
  switch (flavour) {
    case D2PC_STEREORECTIFY_CONTINUOUS: ox = hx = double(nx - 1) / 2.0; oy = hy = double(ny - 1) / 2.0; break;
    case D2PC_STEREORECTIFY_CV24: hx = double(nx) / 2.0; hy = double(ny) / 2.0; ox = double(nx / 2); oy = double(ny / 2); break;
    case D2PC_STEREORECTIFY_CV3:
      hx = double(nx - 1) / 2.0; hy = double(ny - 1) / 2.0; ox = double((nx - 1) / 2); oy = double((ny - 1) / 2); break;
    default: return D2PC_ERR_INVALID_ARG;
  }
  const double cxn = ox - f * (hx - cx) / fx;
  const double cyn = oy - f * (hy - cy) / fy;
  const double tx = -baseline;
  for (int i = 0; i < 16; ++i) q[i] = 0.0;
  q[0] = 1.0;  q[3] = -cxn;
  q[5] = 1.0;  q[7] = -cyn;
  q[11] = f;
  q[14] = -1.0 / tx;
  q[15] = (cxn - cxn) / tx;  // 0/Tx: keeps the sign OpenCV produces (-0.0 for Tx < 0)
  return D2PC_OK;
}

int d2pc_make_q(double fx, double fy, double cx, double cy, double baseline, int nx, int ny, double q[16]) {
  return d2pc_make_q_flavour(fx, fy, cx, cy, baseline, nx, ny, D2PC_STEREORECTIFY_CONTINUOUS, q);
}

int d2pc_make_q_disparity_image(double f, double T, double cx, double cy, double q[16]) {
  if (!q || !(f > 0) || !(T > 0) || !std::isfinite(cx) || !std::isfinite(cy)) return D2PC_ERR_INVALID_ARG;
  for (int i = 0; i < 16; ++i) q[i] = 0.0;
  q[0] = 1.0;  q[3] = -cx;
  q[5] = 1.0;  q[7] = -cy;
  q[11] = f;
  q[14] = 1.0 / T;
  return D2PC_OK;
}

void *d2pc_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void d2pc_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int d2pc_config_init(d2pc_config *cfg) {
  if (!cfg) return D2PC_ERR_INVALID_ARG;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof *cfg;
  cfg->device_id = 0;
  cfg->border = 40;  // cpp:70,72
  cfg->mode = D2PC_MODE_PARITY;
  cfg->min_disparity = -std::numeric_limits<float>::infinity();
  cfg->compact_algo = 0;
  return D2PC_OK;
}

int d2pc_create(const d2pc_config *cfg, d2pc_ctx **out) {
  if (!cfg || !out) return D2PC_ERR_INVALID_ARG;
  *out = nullptr;
  if (cfg->struct_size != sizeof(d2pc_config)) return D2PC_ERR_INVALID_ARG;
  if (cfg->border < 0 || cfg->border > 16384) return D2PC_ERR_INVALID_ARG;
  if (cfg->mode != D2PC_MODE_PARITY && cfg->mode != D2PC_MODE_COMPACT) return D2PC_ERR_INVALID_ARG;
  if (cfg->compact_algo < 0 || cfg->compact_algo > (D2PC_EXPERIMENTS ? 4 : 3)) return D2PC_ERR_INVALID_ARG;
  if (std::isnan(cfg->min_disparity)) return D2PC_ERR_INVALID_ARG;
  int n = d2pc_device_count();
  if (n <= 0 || cfg->device_id < 0 || cfg->device_id >= n) return D2PC_ERR_NO_DEVICE;
  d2pc_ctx *ctx = new (std::nothrow) d2pc_ctx();
  if (!ctx) return D2PC_ERR_OUT_OF_MEMORY;
  ctx->cfg = *cfg;
  ctx->states.what = "compaction state";
  ctx->cb_scratch.what = "callback scratch";
  ctx->device = cfg->device_id;
  DeviceGuard guard(ctx->device);
  hipDeviceProp_t prop;
  if (!guard.ok || hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) {
    delete ctx;
    return D2PC_ERR_NO_DEVICE;
  }
  ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&ctx->d_counts), 65536 * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&ctx->d_stats), sizeof(CompactStats)) != hipSuccess ||
      hipMemset(ctx->d_stats, 0, sizeof(CompactStats)) != hipSuccess ||
      hipStreamSynchronize(nullptr) != hipSuccess ||  // (the fill may still be in flight when hipMemset returns)
      hipHostMalloc(reinterpret_cast<void **>(&ctx->h_counts), 65536 * sizeof(uint32_t), hipHostMallocDefault) !=
          hipSuccess) {
    d2pc_destroy(ctx);
    return D2PC_ERR_DEVICE;
  }
  *out = ctx;
  return D2PC_OK;
}

int d2pc_destroy(d2pc_ctx *ctx) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_pool(ctx->states);
  free_pool(ctx->cb_scratch);
  if (ctx->d_in) (void)hipFree(ctx->d_in);
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->d_idx) (void)hipFree(ctx->d_idx);
  if (ctx->d_med) (void)hipFree(ctx->d_med);
  if (ctx->d_cvt) (void)hipFree(ctx->d_cvt);
  if (ctx->d_counts) (void)hipFree(ctx->d_counts);
  if (ctx->d_stats) (void)hipFree(ctx->d_stats);
  if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
  for (PipeSlot &sl : ctx->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.h_in) (void)hipHostFree(sl.h_in);
    if (sl.h_out) (void)hipHostFree(sl.h_out);
    if (sl.h_count) (void)hipHostFree(sl.h_count);
    if (sl.d_in) (void)hipFree(sl.d_in);
    if (sl.d_med) (void)hipFree(sl.d_med);
    if (sl.d_cvt) (void)hipFree(sl.d_cvt);
    if (sl.d_out) (void)hipFree(sl.d_out);
    if (sl.d_idx) (void)hipFree(sl.d_idx);
    if (sl.st.p) (void)hipFree(sl.st.p);
    if (sl.st.done) (void)hipEventDestroy(sl.st.done);
    if (sl.d_count) (void)hipFree(sl.d_count);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  for (hipEvent_t e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->cb_events) (void)hipEventDestroy(e);
  if (ctx->cb_stream_m) (void)hipStreamDestroy(ctx->cb_stream_m);
  if (ctx->cb_stream_r) (void)hipStreamDestroy(ctx->cb_stream_r);
  if (ctx->cb_overlap_done) (void)hipEventDestroy(ctx->cb_overlap_done);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return D2PC_OK;
}

const char *d2pc_last_error(const d2pc_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int d2pc_set_q(d2pc_ctx *ctx, const double q[16]) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!q) return fail(ctx, D2PC_ERR_INVALID_ARG, "q is null");
  memcpy(ctx->q, q, sizeof ctx->q);  // bit copy: keeps -0.0 in Q[3][3]
  ctx->have_q = true;
  ctx->qx_width = 0;
  classify_q(ctx);
  return D2PC_OK;
}

int d2pc_get_q(const d2pc_ctx *ctx, double q[16]) {
  if (!ctx || !q) return D2PC_ERR_INVALID_ARG;
  if (!ctx->have_q) return D2PC_ERR_NOT_CALIBRATED;
  memcpy(q, ctx->q, sizeof ctx->q);
  return D2PC_OK;
}

int d2pc_set_border(d2pc_ctx *ctx, int border) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (border < 0 || border > 16384) return fail(ctx, D2PC_ERR_INVALID_ARG, "bad border %d", border);
  ctx->cfg.border = border;
  return D2PC_OK;
}

int d2pc_set_mode(d2pc_ctx *ctx, int mode) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (mode != D2PC_MODE_PARITY && mode != D2PC_MODE_COMPACT) return fail(ctx, D2PC_ERR_INVALID_ARG, "bad mode %d", mode);
  ctx->cfg.mode = mode;
  return D2PC_OK;
}

int d2pc_set_min_disparity(d2pc_ctx *ctx, float min_disparity) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (std::isnan(min_disparity)) return fail(ctx, D2PC_ERR_INVALID_ARG, "min_disparity is NaN");
  ctx->cfg.min_disparity = min_disparity;
  return D2PC_OK;
}

int d2pc_get_config(const d2pc_ctx *ctx, d2pc_config *cfg) {
  if (!ctx || !cfg) return D2PC_ERR_INVALID_ARG;
  *cfg = ctx->cfg;
  return D2PC_OK;
}

// blob = 16 x f64 Q (bit copy: -0.0 survives) | int32 border | int32 mode, little-endian
int d2pc_calib_pack(const double q[16], int border, int mode, void *blob) {
  if (!q || !blob) return D2PC_ERR_INVALID_ARG;
  if (border < 0 || border > 16384 || (mode != D2PC_MODE_PARITY && mode != D2PC_MODE_COMPACT)) return D2PC_ERR_INVALID_ARG;
  unsigned char *b = static_cast<unsigned char *>(blob);
  memcpy(b, q, 128);
  int32_t tail[2] = {border, mode};
  memcpy(b + 128, tail, 8);
  return D2PC_OK;
}

int d2pc_calib_unpack(const void *blob, size_t bytes, double q[16], int *border, int *mode) {
  if (!blob || !q || !border || !mode || bytes != D2PC_CALIB_BLOB_BYTES) return D2PC_ERR_INVALID_ARG;
  const unsigned char *b = static_cast<const unsigned char *>(blob);
  int32_t tail[2];
  memcpy(tail, b + 128, 8);
  if (tail[0] < 0 || tail[0] > 16384 || (tail[1] != D2PC_MODE_PARITY && tail[1] != D2PC_MODE_COMPACT)) return D2PC_ERR_INVALID_ARG;
  memcpy(q, b, 128);
  *border = tail[0];
  *mode = tail[1];
  return D2PC_OK;
}

int d2pc_export_calibration(const d2pc_ctx *ctx, void *blob) {
  if (!ctx || !blob) return D2PC_ERR_INVALID_ARG;
  if (!ctx->have_q) return D2PC_ERR_NOT_CALIBRATED;
  return d2pc_calib_pack(ctx->q, ctx->cfg.border, ctx->cfg.mode, blob);
}

int d2pc_import_calibration(d2pc_ctx *ctx, const void *blob, size_t bytes) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  double q[16];
  int border = 0, mode = 0;
  if (d2pc_calib_unpack(blob, bytes, q, &border, &mode) != D2PC_OK)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "bad calibration blob (must be %d bytes with a valid border/mode)", D2PC_CALIB_BLOB_BYTES);
  memcpy(ctx->q, q, 128);
  ctx->have_q = true;
  ctx->qx_width = 0;
  classify_q(ctx);
  ctx->cfg.border = border;
  ctx->cfg.mode = mode;
  return D2PC_OK;
}

size_t d2pc_roi_points(int width, int height, int border) {
  if (border < 0) return 0;
  const long long w = (long long)width - 2LL * border, h = (long long)height - 2LL * border;
  return (w > 0 && h > 0) ? size_t(w) * size_t(h) : 0;
}

// cpp:79-85: width = N, height = 1, is_dense = false; toROSMsg's field table.
int d2pc_cloud_meta_fill(const d2pc_ctx *ctx, size_t n, d2pc_cloud_meta *m) {
  if (!ctx || !m) return D2PC_ERR_INVALID_ARG;
  if (n > 0xffffffffull / 16) return D2PC_ERR_BAD_SIZE;
  memset(m, 0, sizeof *m);
  m->height = 1;
  m->width = uint32_t(n);
  m->point_step = 16;
  m->row_step = uint32_t(16 * n);
  m->is_bigendian = 0;
  m->is_dense = ctx->cfg.mode == D2PC_MODE_COMPACT ? 1 : 0;
  m->n_fields = 3;
  const char *names[3] = {"x", "y", "z"};
  for (int i = 0; i < 3; ++i) {
    strncpy(m->fields[i].name, names[i], sizeof m->fields[i].name - 1);
    m->fields[i].offset = uint32_t(4 * i);
    m->fields[i].datatype = 7;  // sensor_msgs::PointField::FLOAT32
    m->fields[i].count = 1;
  }
  return D2PC_OK;
}

int d2pc_set_reproject_form(d2pc_ctx *ctx, int form) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (form != D2PC_FORM_DEFAULT && form != D2PC_FORM_CV24 && form != D2PC_FORM_CV4)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "reproject form %d: not one of D2PC_FORM_*", form);
  ctx->reproject_form = form;
  return D2PC_OK;
}

int d2pc_set_tuning(d2pc_ctx *ctx, const char *key, int value) {
  if (!ctx || !key) return D2PC_ERR_INVALID_ARG;
  if (!strcmp(key, "pxt_parity") && (value == 0 || value == 1 || value == 2 || (D2PC_EXPERIMENTS && tile_shape_supported(value)))) ctx->pxt_parity = value;
  else if (!strcmp(key, "pxt_compact") && tile_shape_supported(value)) ctx->pxt_compact = value;
  else if (!strcmp(key, "blocks_per_cu") && value >= 1 && value <= 4096) ctx->blocks_per_cu = value;
  else if (!strcmp(key, "onepass_blocks_per_cu") && value >= 0 && value <= 64) ctx->onepass_blocks_per_cu = value;
  else if (!strcmp(key, "onepass_form") && (value == 0 || value == 2 || (D2PC_EXPERIMENTS && value >= 1 && value <= 7))) ctx->onepass_form = value;
  else if (!strcmp(key, "no_vec_rows") && (value == 0 || value == 1)) ctx->no_vec_rows = value;
  else if (!strcmp(key, "stage_timing") && (value == 0 || value == 1)) ctx->stage_timing = value;
  else if (!strcmp(key, "spin_timeout_ms") && value >= 1 && value <= 40000) ctx->spin_timeout_ms = value;
  else if (!strcmp(key, "resident_stagger_pct") && value >= -1 && value <= 1000) ctx->resident_stagger_pct = value;
  else if (!strcmp(key, "resident_pair") && (value == 0 || value == 1)) ctx->resident_pair = value;
  else if (!strcmp(key, "resident_pxt") && (value == 0 || value == 32 || value == 64 || tile_shape_supported(value))) ctx->resident_pxt = value;
#if D2PC_EXPERIMENTS  // the laboratory's keys (libd2pc_exp.so): d2pc_ext.h, "experiment build"
  else if (!strcmp(key, "parity_small") && value >= 0 && value <= 2) ctx->parity_small = value;
  else if (!strcmp(key, "big_batch_algo") && (value == 2 || value == 4)) ctx->big_batch_algo = value;
  else if (!strcmp(key, "resident_unbounded") && (value == 0 || value == 1)) ctx->resident_unbounded = value;
  else if (!strcmp(key, "chunk_mb") && value >= 1 && value <= 4096) ctx->chunk_mb = value;
  else if (!strcmp(key, "chunk_first_frames") && value >= 0 && value <= 65535) ctx->chunk_first_frames = value;
#endif
  else if (!strcmp(key, "callback_chunks") && value >= 0 && value <= 64) ctx->cb_chunks = value;
  else if (!strcmp(key, "callback_fused") && (value == 0 || value == 1)) ctx->cb_fused = value;
  else if (!strcmp(key, "callback_fused_compact") && value >= 0 && value <= 2) ctx->cb_fused_compact = value;
  else if (!strcmp(key, "callback_pipe_blocks_per_cu") && value >= 1 && value <= 8) ctx->cb_pipe_blocks_per_cu = value;
  else if (!strcmp(key, "host_direct_read") && (value == 0 || value == 1)) ctx->host_direct_read = value;
  else if (!strcmp(key, "median_algo") && value >= 0 && value <= (D2PC_EXPERIMENTS ? 3 : 2)) ctx->median_algo = value;
  else if (!strcmp(key, "fuse_rows") && (value == 0 || (value >= 2 && value <= 1024))) ctx->fuse_rows = value;
  else if (!strcmp(key, "membench_blocks_per_cu") && value >= 0 && value <= 256) ctx->membench_blocks_per_cu = value;
  else if (!strcmp(key, "membench_unroll") && (value == 1 || value == 2 || value == 4)) ctx->membench_unroll = value;
  else if (!strcmp(key, "membench_nt") && (value == 0 || value == 1)) ctx->membench_nt = value;
  else return fail(ctx, D2PC_ERR_INVALID_ARG, "unknown tuning %s=%d", key, value);
  return D2PC_OK;
}

int d2pc_ext_revision(void) { return D2PC_EXT_REVISION; }

// The two hooks that change the ARITHMETIC (tests compare the specialised kinds with the general kernel through them):
// apart from d2pc_set_tuning, whose keys never change a byte of the result.
int d2pc_ext_set_test_hook(d2pc_ctx *ctx, const char *key, int value) {
  if (!ctx || !key) return D2PC_ERR_INVALID_ARG;
  if (!strcmp(key, "force_general_q") && (value == 0 || value == 1)) ctx->force_general_q = value;
#if D2PC_EXPERIMENTS
  else if (!strcmp(key, "general_q_form") && (value == 0 || value == 1)) ctx->general_q_form = value;
#endif
  else return fail(ctx, D2PC_ERR_INVALID_ARG, "unknown test hook %s=%d", key, value);
  return D2PC_OK;
}

int d2pc_reserve(d2pc_ctx *ctx, int width, int height, int n_frames) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  Geom g;  // the smallest supported tile gives the largest state
  int st = make_geom(ctx, D2PC_DTYPE_U8, 1.f, width, height, size_t(width), size_t(width) * height, n_frames,
                     d2pc_roi_points(width, height, ctx->cfg.border), D2PC_EXPERIMENTS ? 4 : 8, &g);
  if (st != D2PC_OK) return st;
  // Guarantees ONE free (idle, not owned by a captured graph) buffer of this size, and makes it the
  // minimum size of every buffer allocated later.  Call it before each capture that contains a COMPACT launch.
  // (twice: the dense single pass keeps two states per buffer -- one in use, one it cleans for its successor)
  return reserve_buf(ctx, ctx->states, 2 * ((compact_state_bytes(g) + 255) & ~size_t(255)), 0);
}

int d2pc_reserve_mono(d2pc_ctx *ctx, int dtype, int width, int height, int n_frames) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (dtype != D2PC_DTYPE_U8 && dtype != D2PC_DTYPE_MONO16) return fail(ctx, D2PC_ERR_BAD_DTYPE, "dtype %d is not U8 / MONO16", dtype);
  if (width <= 0 || height <= 0 || n_frames <= 0 || n_frames > 65535)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "bad size %dx%d x%d", width, height, n_frames);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  const size_t bytes = ((size_t(width) + 255) & ~size_t(255)) * size_t(height) * size_t(n_frames);
  int st = reserve_buf(ctx, ctx->cb_scratch, bytes, dtype == D2PC_DTYPE_MONO16 ? bytes : 0);
  if (st != D2PC_OK) return st;
  if (ctx->cfg.mode != D2PC_MODE_COMPACT) return D2PC_OK;
  // COMPACT: the state of the two-launch form's compaction and of the tile-fused kernel, whichever is larger
  const long long b = ctx->cfg.border, rw = (long long)width - 2 * b, rh = (long long)height - 2 * b;
  if (rw > 0 && rh > 0 && (rw + 255) / 256 <= (long long)kCbMaxTilesX) {
    const size_t cb = callback_compact_state_bytes(uint32_t((rw + 255) / 256), uint32_t((rh + 31) / 32), uint32_t(n_frames), nullptr);
    if (cb > ctx->states.reserve) ctx->states.reserve = cb;
  }
  return d2pc_reserve(ctx, width, height, n_frames);
}

int d2pc_release_graph_buffers(d2pc_ctx *ctx) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  for (BufPool *pool : {&ctx->states, &ctx->cb_scratch}) {
    for (StateBuf *b : pool->bufs)
      if (b->captured) {
        b->captured = false;
        b->bound = false;
        b->pending = false;  // the caller has destroyed the graphs: nothing of theirs is in flight
        b->dirty = false;
        b->algo = 0;
      }
    // back under the cap on eager buffers: the surplus (idle by the above) is freed
    for (size_t i = pool->bufs.size(); i-- > 0 && eager_bufs(*pool) > kMaxEagerBufs;) {
      StateBuf *b = pool->bufs[i];
      if (b->captured || !state_idle(*b)) continue;
      if (b->p) (void)hipFree(b->p);
      if (b->p2) (void)hipFree(b->p2);
      if (b->done) (void)hipEventDestroy(b->done);
      delete b;
      pool->bufs.erase(pool->bufs.begin() + long(i));
    }
  }
  return D2PC_OK;
}

int d2pc_process_device(d2pc_ctx *ctx, const void *d_disp, int dtype, float scale, int width, int height,
                        size_t row_stride, size_t in_frame_stride, int n_frames, void *d_out, uint32_t *d_idx,
                        size_t out_frame_stride, uint32_t *d_counts, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_disp || !d_out) return fail(ctx, D2PC_ERR_INVALID_ARG, "null device pointer");
  if (!ctx->have_q) return fail(ctx, D2PC_ERR_NOT_CALIBRATED, "Q matrix not set");
  if (reinterpret_cast<uintptr_t>(d_out) % 16 != 0) return fail(ctx, D2PC_ERR_INVALID_ARG, "d_out_points must be 16-byte aligned");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  const bool compact = ctx->cfg.mode == D2PC_MODE_COMPACT;
  Geom g;
  int st = make_geom(ctx, dtype, scale, width, height, row_stride, in_frame_stride, n_frames, out_frame_stride,
                     compact ? ctx->pxt_compact : parity_pxt(ctx, width, height, n_frames, dtype), &g);
  if (st != D2PC_OK) return st;
  if (reinterpret_cast<uintptr_t>(d_disp) % elem_size(dtype) != 0)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "d_disp is not aligned to its sample type");
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = HIP's default stream
  if (g.roi_n == 0) {
    if (d_counts) D2PC_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * size_t(n_frames), s));
    return D2PC_OK;
  }
  return enqueue(ctx, g, d_disp, dtype, d_out, d_idx, d_counts, s);
}

// Reads the header of one state buffer whose last launch was the single pass.
static int state_timed_out(d2pc_ctx *ctx, const StateBuf &b, bool *timed_out) {
  *timed_out = false;
  if (!b.p || (b.algo != 2 && b.algo != 3)) return D2PC_OK;  // the two-pass form has no in-launch hand-off and never reads the flag
  StateHeader h;
  D2PC_HIP(ctx, hipMemcpy(&h, static_cast<const uint8_t *>(b.p) + b.hdr_off, sizeof h, hipMemcpyDeviceToHost));
  // single pass: its state clear zeroed the flag; resident blocks: nothing zeroes it, a give-up stores the launch's epoch
  *timed_out = b.algo == 2 ? h.timeout != 0 : h.timeout == b.epoch;
  return D2PC_OK;
}

int d2pc_check_async_error(d2pc_ctx *ctx) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  // every buffer remembers the algorithm of ITS last launch: a small two-pass launch after a big single-pass
  // one (another buffer, or the same one re-used) neither hides the big launch's flag nor inherits a stale one
  for (const StateBuf *b : ctx->states.bufs) {
    bool bad = false;
    int st = state_timed_out(ctx, *b, &bad);
    if (st != D2PC_OK) return st;
    if (bad) return fail(ctx, D2PC_ERR_INTERNAL, "compaction hand-off spin expired");
  }
  return D2PC_OK;
}

// Single-pass counters: the sum of the slots the launches' blocks added to.  The caller has synchronised its streams.
int d2pc_compact_stats(d2pc_ctx *ctx, d2pc_compact_stats_t *out) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!out || out->struct_size != sizeof(d2pc_compact_stats_t)) return fail(ctx, D2PC_ERR_INVALID_ARG, "bad d2pc_compact_stats_t");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  std::vector<unsigned char> raw(sizeof(CompactStats));
  D2PC_HIP(ctx, hipMemcpy(raw.data(), ctx->d_stats, sizeof(CompactStats), hipMemcpyDeviceToHost));
  const CompactStats &acc = *reinterpret_cast<const CompactStats *>(raw.data());
  uint64_t tiles = 0, polls = 0, ticks = 0;
  for (const CompactStats::Slot &sl : acc.slot) {
    tiles += sl.tiles;
    polls += sl.failed_polls;
    ticks += sl.wait_ticks;
  }
  out->launches = acc.launches;
  out->tiles = tiles;
  out->failed_polls = polls;
  out->wait_us = ticks / (kSpinTicksPerMs / 1000u);
  out->timeouts = acc.timeouts;
  out->twopass_fallbacks = ctx->n_twopass_fallbacks;
  return D2PC_OK;
}

int d2pc_compact_stats_reset(d2pc_ctx *ctx) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  D2PC_HIP(ctx, hipMemset(ctx->d_stats, 0, sizeof(CompactStats)));
  D2PC_HIP(ctx, hipStreamSynchronize(nullptr));
  ctx->n_twopass_fallbacks = 0;
  return D2PC_OK;
}

int d2pc_membench_fill(d2pc_ctx *ctx, void *d_dst, size_t bytes, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_dst || bytes < 16 || bytes % 16 != 0 || reinterpret_cast<uintptr_t>(d_dst) % 16 != 0)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "fill needs a 16-byte aligned buffer of a multiple of 16 bytes");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  D2PC_HIP(ctx, launch_membench_fill(d_dst, bytes, uint32_t(ctx->cu_count * ctx->membench_blocks_per_cu), ctx->membench_unroll,
                                     ctx->membench_nt != 0, static_cast<hipStream_t>(stream)));
  return D2PC_OK;
}

int d2pc_membench_copy(d2pc_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_src || !d_dst || bytes < 16 || bytes % 16 != 0 || reinterpret_cast<uintptr_t>(d_dst) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(d_src) % 16 != 0)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "copy needs 16-byte aligned buffers of a multiple of 16 bytes");
  const uintptr_t s0 = reinterpret_cast<uintptr_t>(d_src), d0 = reinterpret_cast<uintptr_t>(d_dst);
  if (s0 < d0 + bytes && d0 < s0 + bytes) return fail(ctx, D2PC_ERR_INVALID_ARG, "source and destination overlap");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  D2PC_HIP(ctx, launch_membench_copy(d_src, d_dst, bytes, uint32_t(ctx->cu_count * ctx->membench_blocks_per_cu), ctx->membench_unroll,
                                     ctx->membench_nt != 0, static_cast<hipStream_t>(stream)));
  return D2PC_OK;
}

#ifdef D2PC_DIAG
// diagnostic build only: copy the 128-byte state header (phase timers) out
int d2pc_debug_read_header(d2pc_ctx *ctx, void *out64) {
  const StateBuf *last = nullptr;  // diagnostic runs use one stream: the buffer of the last single-pass launch
  if (ctx)
    for (const StateBuf *b : ctx->states.bufs)
      if (b->p && b->algo == 2) last = b;
  if (!last) return D2PC_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  D2PC_HIP(ctx, hipMemcpy(out64, static_cast<const uint8_t *>(last->p) + last->hdr_off, sizeof(StateHeader), hipMemcpyDeviceToHost));
  return D2PC_OK;
}
#endif

// Shared body of d2pc_process / d2pc_process_mono8 / d2pc_process_mono16: H2D copy (packed to a
// 256-byte pitch), optional cv_bridge mono16 -> mono8 rescale, optional device median, kernel(s),
// D2H copy; synchronous.  bridge16: `disp` holds uint16 samples that cpp:50 turns into mono8.
static int process_host_frame(d2pc_ctx *ctx, const void *disp, int dtype, float scale, int width, int height,
                              size_t row_stride, int median_ksize, bool bridge16, void *out_points,
                              uint32_t *out_index, size_t capacity, size_t *n_points) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (n_points) *n_points = 0;
  if (!disp || !n_points) return fail(ctx, D2PC_ERR_INVALID_ARG, "null argument");
  if (!ctx->have_q) return fail(ctx, D2PC_ERR_NOT_CALIBRATED, "Q matrix not set");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  const bool compact = ctx->cfg.mode == D2PC_MODE_COMPACT;
  if (dtype == D2PC_DTYPE_MONO16) {
    bridge16 = true;
    dtype = D2PC_DTYPE_U16;  // layout of the caller's buffer
  }
  if (dtype != D2PC_DTYPE_F32 && dtype != D2PC_DTYPE_U8 && dtype != D2PC_DTYPE_U16)
    return fail(ctx, D2PC_ERR_BAD_DTYPE, "dtype %d is not F32/U8/U16/MONO16", dtype);
  const int kdtype = bridge16 ? D2PC_DTYPE_U8 : dtype;  // what the kernels see
  const bool median = median_ksize > 1;
  if (median && (kdtype != D2PC_DTYPE_U8 || !median_ksize_supported(median_ksize)))
    return fail(ctx, D2PC_ERR_INVALID_ARG, "median needs 8-bit input and an odd ksize in 3..11 (got %d)", median_ksize);
  const size_t es = elem_size(dtype), kes = elem_size(kdtype);
  // device copies of the frame are packed to a 256-byte pitch
  const size_t pitch = (size_t(width > 0 ? width : 0) * es + 255) & ~size_t(255);
  const size_t kpitch = (size_t(width > 0 ? width : 0) * kes + 255) & ~size_t(255);
  const int pxt = compact ? ctx->pxt_compact : parity_pxt(ctx, width, height, 1);
  Geom g;
  int st = make_geom(ctx, dtype, scale, width, height, row_stride, 0, 1, 0, pxt, &g);  // validates the caller's stride
  if (st != D2PC_OK) return st;
  if (g.roi_n == 0) return D2PC_OK;  // cpp:70,72: empty loops => empty cloud
  if (!out_points) return fail(ctx, D2PC_ERR_INVALID_ARG, "out_points is null");
  if (!compact && capacity < g.roi_n)
    return fail(ctx, D2PC_ERR_CAPACITY, "capacity %zu < %u ROI points", capacity, g.roi_n);
  // the kernel sees the packed (and, for mono16, rescaled) copy
  if ((st = make_geom(ctx, kdtype, scale, width, height, kpitch, 0, 1, 0, pxt, &g)) != D2PC_OK) return st;
  // pinned input: the reprojection reads the frame straight from host memory (no staging copy; PCIe is full
  // duplex, so with a pinned output the inbound reads overlap the outbound stores: one 4K fp32 frame 2.99 ->
  // 2.53 ms, the native frame 128 -> 115 us).  Only when the reprojection is the first kernel and reads the frame
  // once: the median's 32-byte row pieces crawl over the link (native frame 117 -> 161 us), and the two-pass
  // compaction would fetch the frame twice.
  const Geom g_caller = [&] { Geom t; (void)make_geom(ctx, dtype, scale, width, height, row_stride, 0, 1, 0, pxt, &t); return t; }();
  const void *direct_in = nullptr;
  if (ctx->host_direct_read && !median && !bridge16 && !compact && reinterpret_cast<uintptr_t>(disp) % es == 0)
    direct_in = pinned_device_view(disp, row_stride * size_t(height - 1) + size_t(width) * es);
  if (!direct_in && (st = grow(ctx, &ctx->d_in, &ctx->in_cap, pitch * size_t(height))) != D2PC_OK) return st;
  if (bridge16 && (st = grow(ctx, &ctx->d_cvt, &ctx->cvt_cap, kpitch * size_t(height))) != D2PC_OK) return st;
  if (median && (st = grow(ctx, &ctx->d_med, &ctx->med_cap, kpitch * size_t(height))) != D2PC_OK) return st;
  // pinned output that holds the whole ROI: the kernels store the final bytes straight into it
  void *direct_out = capacity >= g.roi_n && reinterpret_cast<uintptr_t>(out_points) % 16 == 0
                         ? pinned_device_view(out_points, size_t(g.roi_n) * 16) : nullptr;
  void *direct_idx = direct_out && out_index ? pinned_device_view(out_index, size_t(g.roi_n) * 4) : nullptr;
  if (out_index && !direct_idx) direct_out = nullptr;  // both or neither
  if (!direct_out) {
    if ((st = grow(ctx, &ctx->d_out, &ctx->out_cap, size_t(g.roi_n) * 16)) != D2PC_OK) return st;
    if (out_index && (st = grow(ctx, &ctx->d_idx, &ctx->idx_cap, size_t(g.roi_n) * 4)) != D2PC_OK) return st;
  }
  hipStream_t s = ctx->stream;
  SyncOnExit drain(s);
  const bool timing = ctx->stage_timing != 0;
  ctx->have_times = false;
  if (timing)
    for (hipEvent_t &e : ctx->ev)
      if (!e) D2PC_HIP(ctx, hipEventCreate(&e));
  auto mark = [&](int i) { return timing ? hipEventRecord(ctx->ev[i], s) : hipSuccess; };
  D2PC_HIP(ctx, mark(0));
  if (!direct_in)
    D2PC_HIP(ctx, hipMemcpy2DAsync(ctx->d_in, pitch, disp, row_stride, size_t(width) * es, size_t(height),
                                   hipMemcpyHostToDevice, s));
  D2PC_HIP(ctx, mark(1));
  const void *kernel_in = direct_in ? direct_in : ctx->d_in;
  size_t kernel_in_pitch = direct_in ? row_stride : pitch;
  MedianArgs m;
  m.algo = ctx->median_algo;
  m.width = uint32_t(width);
  m.height = uint32_t(height);
  if (bridge16) {
    m.src_row_stride = uint32_t(kernel_in_pitch);
    m.dst_row_stride = uint32_t(kpitch);
    D2PC_HIP(ctx, launch_mono16_to_mono8(kernel_in, ctx->d_cvt, m, s));
    kernel_in = ctx->d_cvt;
    kernel_in_pitch = kpitch;
  }
  if (median) {
    m.src_row_stride = uint32_t(kernel_in_pitch);
    m.dst_row_stride = uint32_t(kpitch);
    median_roi_only(m, g, height);
    D2PC_HIP(ctx, launch_median(kernel_in, ctx->d_med, m, median_ksize, s));
    kernel_in = ctx->d_med;
    kernel_in_pitch = kpitch;
  }
  if (kernel_in == direct_in) g = g_caller;  // the reprojection itself reads the caller's rows
  D2PC_HIP(ctx, mark(2));
  void *kout = direct_out ? direct_out : ctx->d_out;
  uint32_t *kidx = !out_index ? nullptr : static_cast<uint32_t *>(direct_out ? direct_idx : ctx->d_idx);
  st = enqueue(ctx, g, kernel_in, kdtype, kout, kidx, ctx->d_counts, s);
  if (st != D2PC_OK) return st;
  D2PC_HIP(ctx, mark(3));
  size_t n = g.roi_n;
  if (compact) {
    D2PC_HIP(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    D2PC_HIP(ctx, hipStreamSynchronize(s));
    if (ctx->h_counts[0] == kCountTimedOut) {
      // the single pass gave up waiting for a predecessor (only ever selected here by cfg.compact_algo = 2):
      // this entry point is synchronous, so run the frame again with the two-pass form, which cannot wait.
      // Counted: d2pc_compact_stats reports these reruns (twopass_fallbacks) and the launch that timed out.
      ++ctx->n_twopass_fallbacks;
      st = enqueue(ctx, g, kernel_in, kdtype, kout, kidx, ctx->d_counts, s, nullptr, 1);
      if (st != D2PC_OK) return st;
      D2PC_HIP(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      D2PC_HIP(ctx, hipStreamSynchronize(s));
    }
    n = ctx->h_counts[0];
    if (n > g.roi_n) return fail(ctx, D2PC_ERR_INTERNAL, "compaction reported %zu points for %u ROI pixels", n, g.roi_n);
    if (n > capacity) return fail(ctx, D2PC_ERR_CAPACITY, "capacity %zu < %zu valid points", capacity, n);
  }
  if (n && !direct_out) {
    D2PC_HIP(ctx, hipMemcpyAsync(out_points, ctx->d_out, n * 16, hipMemcpyDeviceToHost, s));
    if (out_index) D2PC_HIP(ctx, hipMemcpyAsync(out_index, ctx->d_idx, n * 4, hipMemcpyDeviceToHost, s));
  }
  D2PC_HIP(ctx, mark(4));
  D2PC_HIP(ctx, hipStreamSynchronize(s));
  drain.armed = false;
  if (timing) {
    float *t[4] = {&ctx->times.h2d_ms, &ctx->times.prep_ms, &ctx->times.kernel_ms, &ctx->times.d2h_ms};
    for (int i = 0; i < 4; ++i) D2PC_HIP(ctx, hipEventElapsedTime(t[i], ctx->ev[i], ctx->ev[i + 1]));
    D2PC_HIP(ctx, hipEventElapsedTime(&ctx->times.total_ms, ctx->ev[0], ctx->ev[4]));
    ctx->have_times = true;
  }
  *n_points = n;
  return D2PC_OK;
}

int d2pc_process(d2pc_ctx *ctx, const void *disp, int dtype, float scale, int width, int height,
                 size_t row_stride, void *out_points, uint32_t *out_index, size_t capacity, size_t *n_points) {
  return process_host_frame(ctx, disp, dtype, scale, width, height, row_stride, 0, false, out_points, out_index,
                            capacity, n_points);
}

int d2pc_process_mono8(d2pc_ctx *ctx, const uint8_t *image, int width, int height, size_t row_stride,
                       int median_ksize, float scale, void *out_points, uint32_t *out_index, size_t capacity,
                       size_t *n_points) {
  return process_host_frame(ctx, image, D2PC_DTYPE_U8, scale, width, height, row_stride, median_ksize, false,
                            out_points, out_index, capacity, n_points);
}

int d2pc_process_mono16(d2pc_ctx *ctx, const uint16_t *image, int width, int height, size_t row_stride,
                        int median_ksize, float scale, void *out_points, uint32_t *out_index, size_t capacity,
                        size_t *n_points) {
  return process_host_frame(ctx, image, D2PC_DTYPE_MONO16, scale, width, height, row_stride, median_ksize, true,
                            out_points, out_index, capacity, n_points);
}

int d2pc_mono16_to_mono8_device(d2pc_ctx *ctx, const void *d_src, int width, int height, size_t src_row_stride,
                                size_t src_frame_stride, int n_frames, void *d_dst, size_t dst_row_stride,
                                size_t dst_frame_stride, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_src || !d_dst) return fail(ctx, D2PC_ERR_INVALID_ARG, "null device pointer");
  if (width <= 0 || height <= 0 || n_frames <= 0 || n_frames > 65535)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "bad size %dx%d x%d", width, height, n_frames);
  if (src_row_stride < size_t(width) * 2 || src_row_stride % 2 != 0 || dst_row_stride < size_t(width) ||
      src_row_stride > 0xffffffffull || dst_row_stride > 0xffffffffull || reinterpret_cast<uintptr_t>(d_src) % 2 != 0)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "bad row stride or alignment (source rows hold %d uint16 samples)", width);
  const size_t src_extent = size_t(height - 1) * src_row_stride + size_t(width) * 2;
  const size_t dst_extent = size_t(height - 1) * dst_row_stride + size_t(width);
  if (n_frames > 1 && (src_frame_stride < src_extent || src_frame_stride % 2 != 0 || dst_frame_stride < dst_extent))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "frame stride too small");
  const uintptr_t s0 = reinterpret_cast<uintptr_t>(d_src), d0 = reinterpret_cast<uintptr_t>(d_dst);
  const uintptr_t s1 = s0 + size_t(n_frames - 1) * src_frame_stride + src_extent;
  const uintptr_t d1 = d0 + size_t(n_frames - 1) * dst_frame_stride + dst_extent;
  if (s0 < d1 && d0 < s1) return fail(ctx, D2PC_ERR_INVALID_ARG, "source and destination overlap");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  MedianArgs m;
  m.algo = ctx->median_algo;
  m.width = uint32_t(width);
  m.height = uint32_t(height);
  m.n_frames = uint32_t(n_frames);
  m.src_row_stride = uint32_t(src_row_stride);
  m.dst_row_stride = uint32_t(dst_row_stride);
  m.src_frame_stride = n_frames > 1 ? src_frame_stride : 0;
  m.dst_frame_stride = n_frames > 1 ? dst_frame_stride : 0;
  D2PC_HIP(ctx, launch_mono16_to_mono8(d_src, d_dst, m, static_cast<hipStream_t>(stream)));
  return D2PC_OK;
}

static int median_device(d2pc_ctx *ctx, const void *d_src, int width, int height, size_t src_row_stride,
                         size_t src_frame_stride, int n_frames, void *d_dst, size_t dst_row_stride,
                         size_t dst_frame_stride, int ksize, void *stream, bool roi_only) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_src || !d_dst || d_src == d_dst) return fail(ctx, D2PC_ERR_INVALID_ARG, "bad device pointers");
  if (!median_ksize_supported(ksize)) return fail(ctx, D2PC_ERR_INVALID_ARG, "ksize %d not in {3,5,7,9,11}", ksize);
  if (width <= 0 || height <= 0 || n_frames <= 0 || n_frames > 65535)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "bad size %dx%d x%d", width, height, n_frames);
  if (src_row_stride < size_t(width) || dst_row_stride < size_t(width) || src_row_stride > 0xffffffffull ||
      dst_row_stride > 0xffffffffull)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "row stride smaller than the width");
  if (n_frames > 1 && (src_frame_stride < size_t(height) * src_row_stride || dst_frame_stride < size_t(height) * dst_row_stride))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "frame stride too small");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  MedianArgs m;
  m.algo = ctx->median_algo;
  m.width = uint32_t(width);
  m.height = uint32_t(height);
  m.n_frames = uint32_t(n_frames);
  m.src_row_stride = uint32_t(src_row_stride);
  m.dst_row_stride = uint32_t(dst_row_stride);
  m.src_frame_stride = src_frame_stride;
  m.dst_frame_stride = dst_frame_stride;
  if (roi_only) {
    const int b = ctx->cfg.border;
    if (width <= 2 * b || height <= 2 * b) return D2PC_OK;  // empty ROI: nothing is read downstream
    m.out_x0 = m.out_y0 = uint32_t(b);
    m.out_w = uint32_t(width - 2 * b);
    m.out_h = uint32_t(height - 2 * b);
  }
  D2PC_HIP(ctx, launch_median(d_src, d_dst, m, ksize, static_cast<hipStream_t>(stream)));
  return D2PC_OK;
}

int d2pc_median_device(d2pc_ctx *ctx, const void *d_src, int width, int height, size_t src_row_stride,
                       size_t src_frame_stride, int n_frames, void *d_dst, size_t dst_row_stride,
                       size_t dst_frame_stride, int ksize, void *stream) {
  return median_device(ctx, d_src, width, height, src_row_stride, src_frame_stride, n_frames, d_dst, dst_row_stride,
                       dst_frame_stride, ksize, stream, false);
}

int d2pc_median_roi_device(d2pc_ctx *ctx, const void *d_src, int width, int height, size_t src_row_stride,
                           size_t src_frame_stride, int n_frames, void *d_dst, size_t dst_row_stride,
                           size_t dst_frame_stride, int ksize, void *stream) {
  return median_device(ctx, d_src, width, height, src_row_stride, src_frame_stride, n_frames, d_dst, dst_row_stride,
                       dst_frame_stride, ksize, stream, true);
}


// ---------------------------------------------------------------------------
// Device-resident callback body for a batch: (rescale ->) median(ROI) -> reproject, the VALU-bound filter of
// one half of the batch overlapped with the HBM-bound reprojection of the other on two streams.
// ---------------------------------------------------------------------------
namespace {

// The filter stream and the reprojection stream (plain streams: see the header for why not CU-masked ones).
int callback_streams(d2pc_ctx *ctx) {
  if (!ctx->cb_stream_m) D2PC_HIP(ctx, hipStreamCreateWithFlags(&ctx->cb_stream_m, hipStreamNonBlocking));
  if (!ctx->cb_stream_r) D2PC_HIP(ctx, hipStreamCreateWithFlags(&ctx->cb_stream_r, hipStreamNonBlocking));
  return D2PC_OK;
}

int callback_event(d2pc_ctx *ctx, size_t i, hipEvent_t *e) {
  while (ctx->cb_events.size() <= i) {
    hipEvent_t ev;
    D2PC_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ctx->cb_events.push_back(ev);
  }
  *e = ctx->cb_events[i];
  return D2PC_OK;
}

// Can the tile-fused kernel (median + points per tile) serve this call?  COMPACT: a band of tiles must fit the
// blocks resident at once (k_callback_bs_compact's hand-off), i.e. ROIs up to 128 x 256 pixels wide.
bool callback_one_kernel_ok(const d2pc_ctx *ctx, bool compact, const Geom &g) {
  if (!compact) return true;
  return ctx->cb_fused_compact >= 1 && (g.roi_w + 255u) / 256u <= kCbMaxTilesX;
}

}  // namespace

extern "C" int d2pc_process_mono_device(d2pc_ctx *ctx, const void *d_image, int dtype, int width, int height,
                                        size_t row_stride, size_t frame_stride, int n_frames, int median_ksize,
                                        float scale, void *d_out, uint32_t *d_idx, size_t out_frame_stride,
                                        uint32_t *d_counts, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_image || !d_out) return fail(ctx, D2PC_ERR_INVALID_ARG, "null device pointer");
  if (!ctx->have_q) return fail(ctx, D2PC_ERR_NOT_CALIBRATED, "Q matrix not set");
  if (reinterpret_cast<uintptr_t>(d_out) % 16 != 0) return fail(ctx, D2PC_ERR_INVALID_ARG, "d_out_points must be 16-byte aligned");
  const bool bridge16 = dtype == D2PC_DTYPE_MONO16;
  if (dtype != D2PC_DTYPE_U8 && !bridge16) return fail(ctx, D2PC_ERR_BAD_DTYPE, "dtype %d is not U8 / MONO16", dtype);
  const bool median = median_ksize > 1;
  if (median && !median_ksize_supported(median_ksize))
    return fail(ctx, D2PC_ERR_INVALID_ARG, "median ksize %d not in {3,5,7,9,11}", median_ksize);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  const bool compact = ctx->cfg.mode == D2PC_MODE_COMPACT;
  const int pxt = compact ? ctx->pxt_compact : parity_pxt(ctx, width, height, n_frames, D2PC_DTYPE_U8);  // (the reprojection sees 8-bit frames)
  Geom gin;  // validates the caller's layout
  int st = make_geom(ctx, bridge16 ? int(D2PC_DTYPE_U16) : int(D2PC_DTYPE_U8), scale, width, height, row_stride,
                     frame_stride, n_frames, out_frame_stride, pxt, &gin);
  if (st != D2PC_OK) return st;
  if (bridge16 && reinterpret_cast<uintptr_t>(d_image) % 2 != 0) return fail(ctx, D2PC_ERR_INVALID_ARG, "d_image is not 2-byte aligned");
  hipStream_t user = static_cast<hipStream_t>(stream);
  if (gin.roi_n == 0) {
    if (d_counts) D2PC_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * size_t(n_frames), user));
    return D2PC_OK;
  }
  if (compact && !d_counts) return fail(ctx, D2PC_ERR_INVALID_ARG, "COMPACT mode needs a d_counts buffer");
  const bool capturing = capture_info(user, nullptr);
  // scratch: the 8-bit frames on a 256-byte pitch
  const size_t kpitch = (size_t(width) + 255) & ~size_t(255), kframe = kpitch * size_t(height);
  // (the same decisions as below: chunked overlap?  filter + points in one kernel, which needs no filtered frames?)
  const int want_chunks = ctx->cb_chunks > n_frames ? n_frames : ctx->cb_chunks;
  const bool will_overlap = want_chunks > 1 && !capturing && (median || bridge16) &&
                            uint64_t(width) * uint64_t(height) * uint64_t(n_frames) / uint64_t(want_chunks) >= (uint64_t(16) << 20);
  bool one_kernel = false;
  if (median && ctx->cb_fused == 1 && !will_overlap) {
    MedianArgs probe;
    probe.algo = ctx->median_algo;
    probe.n_frames = uint32_t(n_frames);
    median_roi_only(probe, gin, height);
    one_kernel = median_uses_bs(probe, median_ksize) && callback_one_kernel_ok(ctx, compact, gin);
  }
  // The filtered (and rescaled) frames of the two-launch form live in a scratch buffer that belongs to THIS stream's
  // work: like the compaction state, one per stream in flight, so that double-buffered use of a context on two
  // streams never overwrites another call's filtered frames (advisor, round 2).  A capture takes an existing idle
  // one (d2pc_reserve_mono, or a call of this size made earlier) and keeps it.
  const size_t need_med = median && !one_kernel ? kframe * size_t(n_frames) : 0;
  const size_t need_cvt = bridge16 ? kframe * size_t(n_frames) : 0;
  StateBuf *scratch = nullptr;
  if (need_med || need_cvt) {
    if ((st = acquire_buf(ctx, ctx->cb_scratch, user, need_med, need_cvt, nullptr, &scratch)) != D2PC_OK) return st;
  }
  void *const d_cb_med = scratch ? scratch->p : nullptr;
  void *const d_cb_cvt = scratch ? scratch->p2 : nullptr;
  // Few, large chunks: a cross-stream dependency costs ~20 us on this runtime (measured: 16 one-frame chunks of
  // 4K frames are 19 % SLOWER than running in order, 2 chunks 8 % faster), so the batch is only cut when every
  // chunk carries well over that in kernel time
  const uint64_t batch_px = uint64_t(width) * uint64_t(height) * uint64_t(n_frames);
  int n_chunks = ctx->cb_chunks;
  if (n_chunks > n_frames) n_chunks = n_frames;
  const bool overlap = n_chunks > 1 && !capturing && (median || bridge16) && batch_px / uint64_t(n_chunks) >= (uint64_t(16) << 20);
  const int chunk = overlap ? (n_frames + n_chunks - 1) / n_chunks : n_frames;
  hipStream_t sm = user, sr = user;
  if (overlap) {
    if ((st = callback_streams(ctx)) != D2PC_OK) return st;
    // the two internal streams and their events are ONE set per context: a second overlapped call (from any
    // stream) is ordered behind the previous one
    if (!ctx->cb_overlap_done) D2PC_HIP(ctx, hipEventCreateWithFlags(&ctx->cb_overlap_done, hipEventDisableTiming));
    if (ctx->cb_overlap_pending) D2PC_HIP(ctx, hipStreamWaitEvent(user, ctx->cb_overlap_done, 0));
    sm = ctx->cb_stream_m;
    sr = ctx->cb_stream_r;
    hipEvent_t fork;
    if ((st = callback_event(ctx, 0, &fork)) != D2PC_OK) return st;
    D2PC_HIP(ctx, hipEventRecord(fork, user));
    D2PC_HIP(ctx, hipStreamWaitEvent(sm, fork, 0));
    D2PC_HIP(ctx, hipStreamWaitEvent(sr, fork, 0));
  }
  size_t ev = 1;
  for (int f0 = 0; f0 < n_frames; f0 += chunk) {
    const int nf = n_frames - f0 < chunk ? n_frames - f0 : chunk;
    const uint8_t *src = static_cast<const uint8_t *>(d_image) + size_t(f0) * frame_stride;
    const void *kin = src;
    size_t kin_pitch = row_stride, kin_frame = frame_stride;
    MedianArgs m;
    m.algo = ctx->median_algo;
    m.width = uint32_t(width);
    m.height = uint32_t(height);
    m.n_frames = uint32_t(nf);
    if (bridge16) {
      m.src_row_stride = uint32_t(row_stride);
      m.dst_row_stride = uint32_t(kpitch);
      m.src_frame_stride = frame_stride;
      m.dst_frame_stride = kframe;
      uint8_t *dst = static_cast<uint8_t *>(d_cb_cvt) + size_t(f0) * kframe;
      D2PC_HIP(ctx, launch_mono16_to_mono8(src, dst, m, sm));
      kin = dst;
      kin_pitch = kpitch;
      kin_frame = kframe;
    }
    if (median) {
      m.src_row_stride = uint32_t(kin_pitch);
      m.dst_row_stride = uint32_t(kpitch);
      m.src_frame_stride = kin_frame;
      m.dst_frame_stride = kframe;
      median_roi_only(m, gin, height);
      if (one_kernel) {
        // filter and points tile by tile in one kernel; the filtered frames never reach memory
        Geom g;
        if ((st = make_geom(ctx, D2PC_DTYPE_U8, scale, width, height, kin_pitch, kin_frame, nf, out_frame_stride, pxt, &g)) != D2PC_OK)
          return st;
        LaunchArgs a;
        a.out_points = static_cast<uint8_t *>(d_out) + size_t(f0) * out_frame_stride * 16;
        a.out_index = d_idx ? d_idx + size_t(f0) * out_frame_stride : nullptr;
        a.counts = d_counts ? d_counts + f0 : nullptr;
        a.dtype = D2PC_DTYPE_U8;
        a.stream = sr;
        if ((st = fill_q(ctx, a, width)) != D2PC_OK) return st;
        a.qs.w_safe = w_safe_for(ctx, g);
        if (compact) {  // the COMPACT form hands row counts over between the tiles of a band: its own state
          // Residency.  Both forms of the kernel hand counts over between the tiles of a BAND, and a frame's blocks are
          // dispatched round-robin over the launch's frames (frame = blockIdx % n_frames): a frame needs tiles_x of its
          // own blocks resident at once, or no frame ever finishes band 0 (advisor, round 3: from ~385 frames of 752x480
          // or ~55 frames of 4K every wave spun out its 4-s budget).  A call with more frames than that is cut into
          // sub-batches of nfc frames with resident / nfc > tiles_x, launched back to back on the same stream (they share
          // the stream's state buffer: a sub-batch's zeroing kernel runs behind the previous sub-batch's last store).
          const uint32_t tiles_x = (g.roi_w + 255u) / 256u, tiles_y = (g.roi_n / g.roi_w + 31u) / 32u, tpf = tiles_x * tiles_y;
          const uint32_t resident = uint32_t(ctx->cu_count * (ctx->cb_pipe_blocks_per_cu < 3 ? ctx->cb_pipe_blocks_per_cu : 3));  // (LDS and registers admit 3 per CU)
          uint32_t nfc = resident / (tiles_x + 1u);  // (tiles_x <= kCbMaxTilesX = 128 < resident: nfc >= 1 on any device of >= 43 CUs)
          if (nfc == 0) nfc = 1;
          for (int s0 = 0; s0 < nf; s0 += int(nfc)) {
            const int ns = nf - s0 < int(nfc) ? nf - s0 : int(nfc);
            Geom gs;
            if ((st = make_geom(ctx, D2PC_DTYPE_U8, scale, width, height, kin_pitch, kin_frame, ns, out_frame_stride, pxt, &gs)) != D2PC_OK)
              return st;
            LaunchArgs as = a;
            as.keep_timeout = s0 > 0;  // one flag for the whole call: a later sub-batch's state clear must not wipe an earlier one's give-up
            MedianArgs ms = m;
            ms.n_frames = uint32_t(ns);
            as.out_points = static_cast<uint8_t *>(a.out_points) + size_t(s0) * out_frame_stride * 16;
            as.out_index = a.out_index ? a.out_index + size_t(s0) * out_frame_stride : nullptr;
            as.counts = a.counts + s0;
            uint32_t stride = 0;
            as.state_bytes = callback_compact_state_bytes(tiles_x, tiles_y, uint32_t(ns), &stride);
            gs.frame_state_stride = stride;
            as.stats = ctx->d_stats;
            StateBuf *sb = nullptr;
            if ((st = acquire_buf(ctx, ctx->states, sr, as.state_bytes, 0, nullptr, &sb)) != D2PC_OK) return st;
            as.state = sb->p;
            sb->algo = 2;  // its header carries the hand-off's timeout flag, like the single pass's
            sb->pp_clean = false;
            sb->hdr_off = 0;
            as.geom = gs;
            as.compact_algo = 1;
            if (ctx->cb_fused_compact == 2) {
              // persistent blocks, a multiple of the frame count; every frame needs more blocks than a band has tiles
              // (or as many as it has tiles): otherwise the one-tile-per-block form serves the launch
              uint32_t per_frame = resident / uint32_t(ns);
              if (per_frame > tpf) per_frame = tpf;
              if (per_frame > tiles_x || per_frame == tpf) {
                as.grid = per_frame * uint32_t(ns);
                as.compact_algo = 2;
              }
            }
            // (the one-tile-per-block form: ns * tiles_x <= resident - ns by the choice of nfc, so every frame has a whole
            // band of blocks resident from the first dispatch round on)
            D2PC_HIP(ctx, launch_callback_bs_compact(as, ms, static_cast<const uint8_t *>(kin) + size_t(s0) * kin_frame, median_ksize));
            if (!sb->captured) sb->dirty = true;
          }
          continue;
        }
        a.geom = g;
        D2PC_HIP(ctx, launch_callback_bs(a, m, kin, median_ksize));
        continue;
      }
      uint8_t *dst = static_cast<uint8_t *>(d_cb_med) + size_t(f0) * kframe;
      D2PC_HIP(ctx, launch_median(kin, dst, m, median_ksize, sm));
      kin = dst;
      kin_pitch = kpitch;
      kin_frame = kframe;
    }
    if (overlap) {
      hipEvent_t done;
      if ((st = callback_event(ctx, ev++, &done)) != D2PC_OK) return st;
      D2PC_HIP(ctx, hipEventRecord(done, sm));
      D2PC_HIP(ctx, hipStreamWaitEvent(sr, done, 0));
    }
    Geom g;
    if ((st = make_geom(ctx, D2PC_DTYPE_U8, scale, width, height, kin_pitch, kin_frame, nf, out_frame_stride, pxt, &g)) != D2PC_OK)
      return st;
    st = enqueue(ctx, g, kin, D2PC_DTYPE_U8, static_cast<uint8_t *>(d_out) + size_t(f0) * out_frame_stride * 16,
                 d_idx ? d_idx + size_t(f0) * out_frame_stride : nullptr, d_counts ? d_counts + f0 : nullptr, sr);
    if (st != D2PC_OK) return st;
  }
  if (overlap) {
    hipEvent_t join;
    if ((st = callback_event(ctx, ev++, &join)) != D2PC_OK) return st;
    D2PC_HIP(ctx, hipEventRecord(join, sr));  // every filter launch is ordered before a reprojection on sr
    D2PC_HIP(ctx, hipStreamWaitEvent(user, join, 0));
    D2PC_HIP(ctx, hipEventRecord(ctx->cb_overlap_done, user));
    ctx->cb_overlap_pending = true;
  }
  if (scratch && !scratch->captured) {  // (inside a capture the record would become a graph node; the buffer is the graph's)
    scratch->dirty = true;
  }
  return D2PC_OK;
}

// ---------------------------------------------------------------------------
// Pipelined host path: up to `depth` frames in flight, each on its own stream
// with its own pinned staging, so the H2D copy of frame i+1, the kernels of
// frame i and the D2H copy of frame i-1 overlap (PCIe is full duplex).
// ---------------------------------------------------------------------------
static int grow_pinned(d2pc_ctx *ctx, void **p, size_t *cap, size_t need) {
  if (need <= *cap) return D2PC_OK;
  if (*p) {
    D2PC_HIP(ctx, hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
  }
  const size_t want = (need + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  D2PC_HIP(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
  *cap = want;
  return D2PC_OK;
}

int d2pc_pipeline_configure(d2pc_ctx *ctx, int depth, int direct_host_write) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (depth < 1 || depth > 8) return fail(ctx, D2PC_ERR_INVALID_ARG, "pipeline depth %d not in 1..8", depth);
  for (const PipeSlot &sl : ctx->slots)
    if (sl.state != 0) return fail(ctx, D2PC_ERR_INVALID_ARG, "frames are still in flight");
  DeviceGuard guard(ctx->device);
  for (int i = 0; i < depth; ++i) {
    PipeSlot &sl = ctx->slots[i];
    if (!sl.stream) D2PC_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    if (!sl.done) D2PC_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.d_count) D2PC_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&sl.d_count), 64));
    if (!sl.h_count) D2PC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_count), 64, hipHostMallocDefault));
  }
  ctx->pipe_depth = depth;
  ctx->pipe_direct = direct_host_write ? 1 : 0;
  return D2PC_OK;
}

int d2pc_pipeline_acquire(d2pc_ctx *ctx, const d2pc_frame_desc *desc, void **host_in, int *slot) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!desc || !host_in || !slot) return fail(ctx, D2PC_ERR_INVALID_ARG, "null argument");
  if (ctx->pipe_depth == 0) return fail(ctx, D2PC_ERR_INVALID_ARG, "call d2pc_pipeline_configure first");
  if (!ctx->have_q) return fail(ctx, D2PC_ERR_NOT_CALIBRATED, "Q matrix not set");
  const bool median = desc->median_ksize > 1;
  const bool bridge16 = desc->dtype == D2PC_DTYPE_MONO16;  // cpp:50's rescale to 8 bits runs on the device
  if (median && ((desc->dtype != D2PC_DTYPE_U8 && !bridge16) || !median_ksize_supported(desc->median_ksize)))
    return fail(ctx, D2PC_ERR_INVALID_ARG, "median needs 8-bit (or MONO16) input and an odd ksize in 3..11");
  DeviceGuard guard(ctx->device);
  Geom g;  // validates dtype / size / stride of the caller's layout
  int st = make_geom(ctx, bridge16 ? int(D2PC_DTYPE_U16) : desc->dtype, desc->scale, desc->width, desc->height,
                     desc->row_stride_bytes, 0, 1, 0,
                     ctx->cfg.mode == D2PC_MODE_COMPACT ? ctx->pxt_compact : parity_pxt(ctx, desc->width, desc->height, 1), &g);
  if (st != D2PC_OK) return st;
  int found = -1;
  for (int i = 0; i < ctx->pipe_depth; ++i)
    if (ctx->slots[i].state == 0) {
      found = i;
      break;
    }
  if (found < 0)
    return fail(ctx, D2PC_ERR_CAPACITY, "all %d pipeline slots are in use: collect and release one", ctx->pipe_depth);
  PipeSlot &sl = ctx->slots[found];
  const size_t in_bytes = size_t(desc->height) * desc->row_stride_bytes;
  if ((st = grow_pinned(ctx, &sl.h_in, &sl.h_in_cap, in_bytes)) != D2PC_OK) return st;
  if ((st = grow(ctx, &sl.d_in, &sl.d_in_cap, in_bytes)) != D2PC_OK) return st;
  // MONO16: the 8-bit copy (and its median) have their own 256-byte pitch
  const size_t k_bytes = bridge16 ? size_t(desc->height) * ((size_t(desc->width) + 255) & ~size_t(255)) : in_bytes;
  if (bridge16 && (st = grow(ctx, &sl.d_cvt, &sl.d_cvt_cap, k_bytes)) != D2PC_OK) return st;
  if (median && (st = grow(ctx, &sl.d_med, &sl.d_med_cap, k_bytes)) != D2PC_OK) return st;
  sl.roi_n = g.roi_n;
  sl.idx_off = (size_t(g.roi_n) * 16 + 255) & ~size_t(255);
  const size_t out_bytes = sl.idx_off + (desc->want_index ? size_t(g.roi_n) * 4 : 0) + 256;
  if ((st = grow_pinned(ctx, &sl.h_out, &sl.h_out_cap, out_bytes)) != D2PC_OK) return st;
  if (!ctx->pipe_direct) {
    if ((st = grow(ctx, &sl.d_out, &sl.d_out_cap, size_t(g.roi_n) * 16 + 16)) != D2PC_OK) return st;
    if (desc->want_index && (st = grow(ctx, &sl.d_idx, &sl.d_idx_cap, size_t(g.roi_n) * 4 + 16)) != D2PC_OK) return st;
  }
  sl.desc = *desc;
  sl.state = 1;
  *host_in = sl.h_in;
  *slot = found;
  return D2PC_OK;
}

int d2pc_pipeline_submit(d2pc_ctx *ctx, int slot) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (slot < 0 || slot >= ctx->pipe_depth || ctx->slots[slot].state != 1)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "slot %d was not acquired", slot);
  DeviceGuard guard(ctx->device);
  PipeSlot &sl = ctx->slots[slot];
  const d2pc_frame_desc &d = sl.desc;
  const bool compact = ctx->cfg.mode == D2PC_MODE_COMPACT;
  const bool bridge16 = d.dtype == D2PC_DTYPE_MONO16;
  const int kdtype = bridge16 ? int(D2PC_DTYPE_U8) : d.dtype;  // what the kernels see
  const size_t kstride = bridge16 ? (size_t(d.width) + 255) & ~size_t(255) : d.row_stride_bytes;
  Geom g;
  int st = make_geom(ctx, kdtype, d.scale, d.width, d.height, kstride, 0, 1, 0,
                     compact ? ctx->pxt_compact : parity_pxt(ctx, d.width, d.height, 1), &g);
  if (st != D2PC_OK) return st;
  // the slot's buffers were sized at acquire time: a d2pc_set_border in between must not overflow them
  if (g.roi_n != sl.roi_n)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "border changed since slot %d was acquired (%zu -> %u ROI points): release it",
                slot, sl.roi_n, g.roi_n);
  hipStream_t s = sl.stream;
  const size_t in_bytes = size_t(d.height) * d.row_stride_bytes;
  // from the first enqueue on, a failure must not hand the slot back while work that reads h_in or
  // writes h_out is in flight: drain the stream and park the slot as "collected" (release frees it)
  struct SlotDrain {
    PipeSlot &sl;
    bool armed = true;
    ~SlotDrain() {
      if (!armed) return;
      (void)hipStreamSynchronize(sl.stream);
      sl.state = 3;
    }
  } drain{sl};
  D2PC_HIP(ctx, hipMemcpyAsync(sl.d_in, sl.h_in, in_bytes, hipMemcpyHostToDevice, s));
  const void *kin = sl.d_in;
  MedianArgs m;
  m.algo = ctx->median_algo;
  m.width = uint32_t(d.width);
  m.height = uint32_t(d.height);
  if (bridge16) {
    m.src_row_stride = uint32_t(d.row_stride_bytes);
    m.dst_row_stride = uint32_t(kstride);
    D2PC_HIP(ctx, launch_mono16_to_mono8(sl.d_in, sl.d_cvt, m, s));
    kin = sl.d_cvt;
  }
  if (d.median_ksize > 1) {
    m.src_row_stride = m.dst_row_stride = uint32_t(kstride);
    median_roi_only(m, g, d.height);
    D2PC_HIP(ctx, launch_median(kin, sl.d_med, m, d.median_ksize, s));
    kin = sl.d_med;
  }
  sl.h_count[0] = 0;
  if (g.roi_n) {
    // direct mode: the kernels store points (and indices) straight into the
    // pinned host buffer over PCIe; staged mode: into HBM, then one D2H copy
    void *kout = ctx->pipe_direct ? sl.h_out : sl.d_out;
    uint32_t *kidx = !d.want_index ? nullptr
                     : ctx->pipe_direct ? reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(sl.h_out) + sl.idx_off)
                                        : static_cast<uint32_t *>(sl.d_idx);
    st = enqueue(ctx, g, kin, kdtype, kout, kidx, sl.d_count, s, &sl.st);
    if (st != D2PC_OK) return st;
    D2PC_HIP(ctx, hipMemcpyAsync(sl.h_count, sl.d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (!ctx->pipe_direct) {
      // COMPACT: the count is not known on the host yet, so the whole ROI
      // capacity is copied; only the first h_count points are meaningful
      D2PC_HIP(ctx, hipMemcpyAsync(sl.h_out, sl.d_out, size_t(g.roi_n) * 16, hipMemcpyDeviceToHost, s));
      if (d.want_index)
        D2PC_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(sl.h_out) + sl.idx_off, sl.d_idx, size_t(g.roi_n) * 4,
                                     hipMemcpyDeviceToHost, s));
    }
  }
  D2PC_HIP(ctx, hipEventRecord(sl.done, s));
  drain.armed = false;
  sl.seq = ++ctx->pipe_seq;
  sl.state = 2;
  return D2PC_OK;
}

int d2pc_pipeline_collect(d2pc_ctx *ctx, int *slot, const void **points, const uint32_t **index, size_t *n_points,
                          uint64_t *tag) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!slot || !points || !n_points) return fail(ctx, D2PC_ERR_INVALID_ARG, "null argument");
  int oldest = -1;
  for (int i = 0; i < ctx->pipe_depth; ++i)
    if (ctx->slots[i].state == 2 && (oldest < 0 || ctx->slots[i].seq < ctx->slots[oldest].seq)) oldest = i;
  if (oldest < 0) return fail(ctx, D2PC_ERR_INVALID_ARG, "no submitted frame to collect");
  DeviceGuard guard(ctx->device);
  PipeSlot &sl = ctx->slots[oldest];
  D2PC_HIP(ctx, hipEventSynchronize(sl.done));
  if (sl.roi_n && sl.h_count[0] == kCountTimedOut) {  // in-band: the slot's own launch reported it
    *slot = oldest;  // the frame is lost, but the slot can be released
    sl.state = 3;
    return fail(ctx, D2PC_ERR_INTERNAL, "compaction hand-off spin expired");
  }
  *slot = oldest;
  *points = sl.h_out;
  if (index)
    *index = sl.desc.want_index ? reinterpret_cast<const uint32_t *>(static_cast<uint8_t *>(sl.h_out) + sl.idx_off)
                                : nullptr;
  *n_points = sl.roi_n ? sl.h_count[0] : 0;
  if (tag) *tag = sl.desc.tag;
  sl.state = 3;
  return D2PC_OK;
}

int d2pc_pipeline_release(d2pc_ctx *ctx, int slot) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (slot < 0 || slot >= ctx->pipe_depth || (ctx->slots[slot].state != 3 && ctx->slots[slot].state != 1))
    return fail(ctx, D2PC_ERR_INVALID_ARG, "slot %d is not collected (or acquired)", slot);
  ctx->slots[slot].state = 0;
  return D2PC_OK;
}

int d2pc_last_stage_times(d2pc_ctx *ctx, d2pc_stage_times *times) {
  if (!ctx || !times) return D2PC_ERR_INVALID_ARG;
  if (!ctx->have_times)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "no timed call yet: d2pc_set_tuning(ctx, \"stage_timing\", 1), then d2pc_process*");
  *times = ctx->times;
  return D2PC_OK;
}

// ---------------------------------------------------------------------------
// Depth-map fusion inner loop
// ---------------------------------------------------------------------------
void d2pc_fuse_desc_init(d2pc_fuse_desc *desc) {
  if (!desc) return;
  memset(desc, 0, sizeof *desc);
  desc->struct_size = sizeof *desc;
  desc->rule = D2PC_FUSE_GRAD_FILTER;  // src/depth_map_fusion.cpp:159
  desc->n_frames = 1;
  desc->crop_left = 0;                 // src/depth_map_fusion.cpp:130
  desc->crop_right = 40;
  desc->crop_top = 30;
  desc->crop_bottom = 10;
}

int d2pc_crop_to_square(int cols, int rows, int offset_x, int offset_y, int member_offset_y, int *x, int *y, int *n) {
  if (!x || !y || !n || cols <= 0 || rows <= 0) return D2PC_ERR_INVALID_ARG;
  const int ax = offset_x < 0 ? -offset_x : offset_x, ay = offset_y < 0 ? -offset_y : offset_y;
  const int am = member_offset_y < 0 ? -member_offset_y : member_offset_y;
  const int free_cols = cols - ax, free_rows = rows - ay;
  *n = (cols < rows ? cols : rows) - (ax > am ? ax : am);
  const bool portrait = free_cols < free_rows;
  const int sx = portrait ? offset_x : offset_x + (free_cols - free_rows) / 2;
  const int sy = portrait ? offset_y + (free_rows - free_cols) / 2 : offset_y;
  *x = sx > 0 ? sx : 0;
  *y = sy > 0 ? sy : 0;
  // cv::Mat::operator()(Rect) asserts that the rectangle lies inside the image
  if (*n <= 0 || *x + *n > cols || *y + *n > rows) return D2PC_ERR_BAD_SIZE;
  return D2PC_OK;
}

int d2pc_fuse_device(d2pc_ctx *ctx, const d2pc_fuse_desc *desc, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!desc || desc->struct_size != sizeof(d2pc_fuse_desc)) return fail(ctx, D2PC_ERR_INVALID_ARG, "bad d2pc_fuse_desc");
  const d2pc_fuse_desc &d = *desc;
  if (d.rule < 0 || d.rule >= FUSE_RULE_COUNT) return fail(ctx, D2PC_ERR_INVALID_ARG, "unknown fusion rule %d", d.rule);
  if (d.width <= 0 || d.height <= 0 || d.n_frames <= 0 || d.n_frames > 65535)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "bad size %dx%d x%d", d.width, d.height, d.n_frames);
  if (d.crop_left < 0 || d.crop_right < 0 || d.crop_top < 0 || d.crop_bottom < 0 ||
      d.crop_left + d.crop_right > d.width || d.crop_top + d.crop_bottom > d.height)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "crop %d/%d/%d/%d does not fit %dx%d", d.crop_left, d.crop_right, d.crop_top,
                d.crop_bottom, d.width, d.height);
  if (!d.fused) return fail(ctx, D2PC_ERR_INVALID_ARG, "null fused output");
  const int n_in = d.combined ? 6 : 4;
  for (int p = 0; p < n_in; ++p) {
    if (!d.planes[p]) return fail(ctx, D2PC_ERR_INVALID_ARG, "input plane %d is null", p);
    if (d.pitch[p] < size_t(d.width) || d.pitch[p] * size_t(d.height) > 0xffffffffull)  // 32-bit row offsets in the kernel
      return fail(ctx, D2PC_ERR_BAD_SIZE, "pitch of plane %d smaller than the width (or plane >= 4 GiB)", p);
    if (d.n_frames > 1 && d.frame_stride[p] < size_t(d.height - 1) * d.pitch[p] + size_t(d.width))
      return fail(ctx, D2PC_ERR_BAD_SIZE, "frame stride of plane %d too small", p);
  }
  const int ow = d.width - d.crop_left - d.crop_right, oh = d.height - d.crop_top - d.crop_bottom;
  // byte extent of a (w x h) x n_frames plane
  auto extent = [&](size_t pitch, size_t fstride, int w, int h) {
    return (w <= 0 || h <= 0) ? size_t(0) : size_t(d.n_frames - 1) * fstride + size_t(h - 1) * pitch + size_t(w);
  };
  if (ow > 0 && oh > 0) {
    if (d.fused_pitch < size_t(ow) || d.fused_pitch * size_t(oh) > 0xffffffffull ||
        (d.n_frames > 1 && d.fused_frame_stride < size_t(oh - 1) * d.fused_pitch + size_t(ow)))
      return fail(ctx, D2PC_ERR_BAD_SIZE, "fused pitch / frame stride too small");
  }
  if (d.combined && (d.combined_pitch < size_t(d.width) || d.combined_pitch * size_t(d.height) > 0xffffffffull ||
                     (d.n_frames > 1 && d.combined_frame_stride < size_t(d.height - 1) * d.combined_pitch + size_t(d.width))))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "combined pitch / frame stride too small");
  struct Range { uintptr_t lo, hi; };
  auto overlaps = [](Range a, Range b) { return a.lo < b.hi && b.lo < a.hi; };
  const Range rf{reinterpret_cast<uintptr_t>(d.fused),
                 reinterpret_cast<uintptr_t>(d.fused) + extent(d.fused_pitch, d.fused_frame_stride, ow, oh)};
  const Range rc{reinterpret_cast<uintptr_t>(d.combined),
                 reinterpret_cast<uintptr_t>(d.combined) +
                     (d.combined ? extent(d.combined_pitch, d.combined_frame_stride, d.width, d.height) : 0)};
  if (d.combined && overlaps(rf, rc)) return fail(ctx, D2PC_ERR_INVALID_ARG, "fused and combined outputs overlap");
  for (int p = 0; p < n_in; ++p) {
    const Range ri{reinterpret_cast<uintptr_t>(d.planes[p]),
                   reinterpret_cast<uintptr_t>(d.planes[p]) + extent(d.pitch[p], d.frame_stride[p], d.width, d.height)};
    if (overlaps(ri, rf) || (d.combined && overlaps(ri, rc)))
      return fail(ctx, D2PC_ERR_INVALID_ARG, "an output overlaps input plane %d (in-place fusion is not supported)", p);
  }
  if (ow <= 0 || oh <= 0) {
    if (!d.combined) return D2PC_OK;  // nothing to write
  }
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  FuseArgs a;
  for (int p = 0; p < 6; ++p) {
    const int q = p < n_in ? p : 0;  // unused grad planes: any valid pointer
    a.in[p] = static_cast<const uint8_t *>(d.planes[q]);
    a.in_pitch[p] = uint32_t(d.pitch[q]);
    a.in_frame_stride[p] = d.n_frames > 1 ? d.frame_stride[q] : 0;
  }
  a.fused = static_cast<uint8_t *>(d.fused);
  a.fused_pitch = uint32_t(d.fused_pitch);
  a.fused_frame_stride = d.n_frames > 1 ? d.fused_frame_stride : 0;
  a.combined = static_cast<uint8_t *>(d.combined);
  a.combined_pitch = uint32_t(d.combined_pitch);
  a.combined_frame_stride = d.n_frames > 1 ? d.combined_frame_stride : 0;
  a.width = uint32_t(d.width);
  a.height = uint32_t(d.height);
  a.n_frames = uint32_t(d.n_frames);
  a.rule = d.rule;
  a.crop_left = uint32_t(d.crop_left);
  a.crop_top = uint32_t(d.crop_top);
  a.out_width = uint32_t(ow > 0 ? ow : 0);
  a.out_height = uint32_t(oh > 0 ? oh : 0);
  D2PC_HIP(ctx, launch_fuse(a, static_cast<hipStream_t>(stream), ctx->fuse_rows));
  return D2PC_OK;
}

int d2pc_rotate_cw_device(d2pc_ctx *ctx, const void *d_src, int cols, int rows, size_t src_pitch,
                          size_t src_frame_stride, int n_frames, void *d_dst, size_t dst_pitch,
                          size_t dst_frame_stride, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_src || !d_dst) return fail(ctx, D2PC_ERR_INVALID_ARG, "null device pointer");
  if (cols <= 0 || rows <= 0 || n_frames <= 0 || n_frames > 65535)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "bad size %dx%d x%d", cols, rows, n_frames);
  if (src_pitch < size_t(cols) || dst_pitch < size_t(rows) || src_pitch > 0xffffffffull || dst_pitch > 0xffffffffull)
    return fail(ctx, D2PC_ERR_BAD_SIZE, "pitch smaller than the row (src rows are %d, dst rows %d pixels)", cols, rows);
  const size_t src_extent = size_t(rows - 1) * src_pitch + size_t(cols), dst_extent = size_t(cols - 1) * dst_pitch + size_t(rows);
  if (n_frames > 1 && (src_frame_stride < src_extent || dst_frame_stride < dst_extent))
    return fail(ctx, D2PC_ERR_BAD_SIZE, "frame stride too small");
  const uintptr_t s0 = reinterpret_cast<uintptr_t>(d_src), d0 = reinterpret_cast<uintptr_t>(d_dst);
  const uintptr_t s1 = s0 + size_t(n_frames - 1) * src_frame_stride + src_extent;
  const uintptr_t d1 = d0 + size_t(n_frames - 1) * dst_frame_stride + dst_extent;
  if (s0 < d1 && d0 < s1) return fail(ctx, D2PC_ERR_INVALID_ARG, "source and destination overlap");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  RotateArgs a;
  a.src = static_cast<const uint8_t *>(d_src);
  a.dst = static_cast<uint8_t *>(d_dst);
  a.src_pitch = uint32_t(src_pitch);
  a.dst_pitch = uint32_t(dst_pitch);
  a.src_frame_stride = n_frames > 1 ? src_frame_stride : 0;
  a.dst_frame_stride = n_frames > 1 ? dst_frame_stride : 0;
  a.cols = uint32_t(cols);
  a.rows = uint32_t(rows);
  a.n_frames = uint32_t(n_frames);
  D2PC_HIP(ctx, launch_rotate_cw(a, static_cast<hipStream_t>(stream)));
  return D2PC_OK;
}

}  // extern "C"
