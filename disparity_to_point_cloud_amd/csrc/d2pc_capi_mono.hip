// d2pc_capi_mono.hip -- what happens to a mono image before the reprojection, on the device (reference cpp:50-61):
// cv_bridge's mono16 rescale, cv::medianBlur, and the whole callback body for a resident batch (d2pc_process_mono_device).
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

namespace d2pc {
namespace host {

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

}  // namespace host
}  // namespace d2pc

extern "C" {

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

}  // extern "C"

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
