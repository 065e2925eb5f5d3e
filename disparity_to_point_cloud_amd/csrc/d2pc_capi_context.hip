// d2pc_capi_context.hip -- include/d2pc.h, part 1: status strings, the Q calibration surface (reference hpp:84-104: closed
// forms, the 136-byte blob, Q classification and the per-launch Q arguments), context creation and destruction.
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

namespace d2pc {
namespace host {

int fail(d2pc_ctx *ctx, int status, const char *fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return status;
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

}  // namespace host
}  // namespace d2pc

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

}  // extern "C"
