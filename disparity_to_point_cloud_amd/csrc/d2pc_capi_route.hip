// d2pc_capi_route.hip -- one device-resident call: launch geometry (make_geom), the choice of kernel and shape for it
// (enqueue: PARITY one-shot blocks; COMPACT two-pass / resident one-launch / single pass), d2pc_process_device.
// The seam it serves: reference src/disparity_to_point_cloud.cpp:60-85.
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

namespace d2pc {
namespace host {

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
int parity_pxt(const d2pc_ctx *ctx, int width, int height, int n_frames, int dtype) {
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
            uint32_t *d_counts, hipStream_t stream, StateBuf *fixed_state, int force_algo, uint32_t call_epoch_first) {
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
    // exactly the resident routing's own test below, for ONE frame at 32 pixels per thread (tpf <= 1024 included: on a device
    // of more CUs a single frame could otherwise fall to two two-pass launches, slower than the one launch it replaced), and
    // the pair fits neither in the ordinary tiles nor at 32 pixels per thread
    const bool single_fits_own_tiles = g.tiles_per_frame <= cap && g.tiles_per_frame <= 1024u;
    const bool single_is_resident32 = single_fits_own_tiles || (tpf32 <= cap && tpf32 <= 1024u);  // (<= 32 pixels per thread)
    if (2u * g.tiles_per_frame > cap && 2u * tpf32 > cap && single_is_resident32) {
      // Both launches use the stream's one state buffer, with consecutive epochs: the buffer remembers the first of them, so
      // d2pc_check_async_error covers the whole CALL (a give-up of frame 0 stores epoch E while the buffer's `epoch` is E + 1).
      // A failure of the second enqueue leaves frame 0 launched and counts[0] valid: the call returns the error, the caller's
      // stream stays usable (nothing of frame 1 was enqueued), and the output of frame 1 is untouched.
      const uint32_t first = ctx->resident_epoch;
      for (uint32_t f = 0; f < 2u; ++f) {
        Geom g1 = g;
        g1.n_frames = 1;
        g1.total_tiles = g1.tiles_per_frame;
        if (f == 0 && ctx->spin_ticks_first >= 0) g1.spin_ticks = uint32_t(ctx->spin_ticks_first);  // (test hook)
        int st1 = enqueue(ctx, g1, static_cast<const uint8_t *>(d_disp) + uint64_t(f) * g.in_frame_stride, dtype,
                          static_cast<uint8_t *>(d_out) + uint64_t(f) * g.out_frame_stride * 16u,
                          d_idx ? d_idx + uint64_t(f) * g.out_frame_stride : nullptr, d_counts + f, stream, fixed_state, 0, first);
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
    // clean iff the previous launch's kernel zeroed at least the bytes this one uses, at the same offset (the kernel zeroes
    // `state_bytes`, not the half rounded up: a smaller predecessor leaves a dirty tail; advisor, round 5)
    const bool clean = sb->pp_clean && sb->pp_half == half && sb->pp_bytes >= a.state_bytes && !sb->captured;
    const int h = clean ? sb->pp_next : 0;
    a.state = static_cast<uint8_t *>(sb->p) + size_t(h) * half;
    a.state_is_clean = clean;
    // a captured launch replays on the half baked into it: it keeps the clear kernel in front and cleans nothing
    a.state_other = sb->captured ? nullptr : static_cast<uint8_t *>(sb->p) + size_t(h ^ 1) * half;
    sb->pp_half = half;
    sb->pp_bytes = a.state_bytes;
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
  // (an epoch wrap-around between the two launches of a call starts over below `first`: the range then begins at this launch)
  sb->epoch_first = call_epoch_first && call_epoch_first <= a.epoch ? call_epoch_first : a.epoch;
  const hipError_t launched = launch_compact(a);
  if (launched != hipSuccess) sb->pp_clean = false;  // (nothing ran: the half this launch was to zero for its successor is still dirty)
  D2PC_HIP(ctx, launched);
  if (!sb->captured) sb->dirty = true;  // (a captured buffer is never shared; for the others `done` is recorded when somebody asks)
  return D2PC_OK;
}

}  // namespace host
}  // namespace d2pc

extern "C" {

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

}  // extern "C"
