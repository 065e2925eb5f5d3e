// d2pc_capi_host.hip -- the entry points that take HOST buffers: d2pc_process / _mono8 / _mono16 (one frame, synchronous:
// what DisparityCb calls, reference cpp:46-92) and the pipelined host path d2pc_pipeline_* (several frames in flight).
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

namespace d2pc {
namespace host {

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

}  // namespace host
}  // namespace d2pc

extern "C" {

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

}  // extern "C"
