// d2pc_capi_ext.hip -- include/d2pc_ext.h (unstable: bench.py, tools/, tests/): tuning keys, test hooks, the device
// calibration kernels' entry points.
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

extern "C" {

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
  else if (!strcmp(key, "resident_pxt") && (value == 0 || value == 32 || value == 64 || tile_shape_supported(value))) ctx->resident_pxt = value;
#if D2PC_EXPERIMENTS  // the laboratory's keys (libd2pc_exp.so): d2pc_ext.h, "experiment build"
  else if (!strcmp(key, "parity_small") && value >= 0 && value <= 2) ctx->parity_small = value;
  else if (!strcmp(key, "resident_stagger_pct") && value >= -1 && value <= 1000) ctx->resident_stagger_pct = value;
  else if (!strcmp(key, "resident_pair") && (value == 0 || value == 1)) ctx->resident_pair = value;
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
  else if (!strcmp(key, "handoff_spin_ticks_first") && value >= -1) ctx->spin_ticks_first = value;
#if D2PC_EXPERIMENTS
  else if (!strcmp(key, "general_q_form") && (value == 0 || value == 1)) ctx->general_q_form = value;
#endif
  else return fail(ctx, D2PC_ERR_INVALID_ARG, "unknown test hook %s=%d", key, value);
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

int d2pc_clock_probe_device(d2pc_ctx *ctx, void *d_out16, uint32_t min_us, void *stream) {
  if (!ctx) return D2PC_ERR_INVALID_ARG;
  if (!d_out16 || reinterpret_cast<uintptr_t>(d_out16) % 8 != 0 || min_us == 0 || min_us > 2000000u)
    return fail(ctx, D2PC_ERR_INVALID_ARG, "clock probe needs an 8-byte aligned buffer of 16 uint64 and 1..2,000,000 us");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(ctx, D2PC_ERR_NO_DEVICE, "cannot select device %d", ctx->device);
  D2PC_HIP(ctx, launch_clock_probe(d_out16, min_us, static_cast<hipStream_t>(stream)));
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

}  // extern "C"
