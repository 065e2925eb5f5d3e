// d2pc_capi_fusion.hip -- SURVEY.md section 8(f) #4: the per-pixel loop of publishFusedDepthMap
// (reference src/depth_map_fusion.cpp:113-130), rotateMat and cropToSquare's arithmetic.
#include "d2pc_ctx.hpp"

using namespace d2pc;
using namespace d2pc::host;

extern "C" {

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
