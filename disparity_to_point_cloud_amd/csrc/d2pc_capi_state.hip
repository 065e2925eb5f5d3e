// d2pc_capi_state.hip -- the pool of state buffers behind the COMPACT launches and the callback scratch: one buffer per stream
// with work in flight, completion found out lazily, buffers baked into captured graphs set aside, the single pass's two
// self-cleaning halves; d2pc_reserve*, d2pc_release_graph_buffers, d2pc_check_async_error, d2pc_compact_stats.
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

namespace d2pc {
namespace host {

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

int state_alloc(d2pc_ctx *ctx, const BufPool *pool, StateBuf &b, size_t need, size_t need2) {
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
    b.epoch = b.epoch_first = 0;
    b.pp_clean = false;
    b.pp_bytes = 0;
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

}  // namespace host
}  // namespace d2pc

extern "C" {

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

// Reads the header of one state buffer whose last launch was the single pass.
static int state_timed_out(d2pc_ctx *ctx, const StateBuf &b, bool *timed_out) {
  *timed_out = false;
  if (!b.p || (b.algo != 2 && b.algo != 3)) return D2PC_OK;  // the two-pass form has no in-launch hand-off and never reads the flag
  StateHeader h;
  D2PC_HIP(ctx, hipMemcpy(&h, static_cast<const uint8_t *>(b.p) + b.hdr_off, sizeof h, hipMemcpyDeviceToHost));
  // single pass: its state clear zeroed the flag; resident blocks: nothing zeroes it, a give-up stores the launch's epoch
  // (a call of two frames is two launches with consecutive epochs on this buffer: any flag in [epoch_first, epoch] is the call's)
  *timed_out = b.algo == 2 ? h.timeout != 0 : (h.timeout >= b.epoch_first && h.timeout <= b.epoch);
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

}  // extern "C"
