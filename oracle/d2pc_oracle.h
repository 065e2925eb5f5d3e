/*
 * d2pc_oracle.h -- CPU restatement of the disparity -> point-cloud hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this code, and only as the checker / the timed CPU
 * baseline.  The product library (libd2pc.so) never links or dlopens it.
 *
 * PARITY UNPINNED.  The reference (PX4/disparity_to_point_cloud) ships no
 * tests and no golden data, and its arithmetic lives in un-vendored,
 * un-pinned third-party libraries (OpenCV calib3d/imgproc/core, cv_bridge,
 * PCL, pcl_conversions; only hint: ROS Indigo => OpenCV 2.4.x, PCL 1.7).
 * None of them can be built in this image, so this file restates their
 * published algorithms and is anchored on the reference's own call sites:
 *
 *   src/disparity_to_point_cloud.cpp:50     cv_bridge::toCvCopy(*msg,"mono8")
 *   src/disparity_to_point_cloud.cpp:55-57  cv::medianBlur(..., 11)
 *   src/disparity_to_point_cloud.cpp:60-61  convertTo(CV_32FC1, 1.0/8.0)
 *   src/disparity_to_point_cloud.cpp:63-64  cv::reprojectImageTo3D(disp, img3d, Q_)
 *   src/disparity_to_point_cloud.cpp:70-76  ROI inset-40 push_back loop
 *   src/disparity_to_point_cloud.cpp:79-85  width/height/is_dense + pcl::toROSMsg
 *   include/disparity_to_point_cloud/disparity_to_point_cloud.hpp:84-104
 *                                           calibration params + cv::stereoRectify
 *
 * It is checked against exact-rational known-answer vectors (tests/golden/).
 */
#ifndef D2PC_ORACLE_H
#define D2PC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Input sample types (same numbering as include/d2pc.h). */
enum { D2PC_ORACLE_F32 = 0, D2PC_ORACLE_U8 = 1, D2PC_ORACLE_U16 = 2 };

/* Which published form of cv::reprojectImageTo3D to follow.
 *  CV24: OpenCV 2.4.x (ROS Indigo, the distro .clang_complete:2 points at):
 *        per-row incremental numerators, iW = 1./W, X = num*iW, all double,
 *        one cast to float.
 *  CV4 : OpenCV 3.x/4.x: Vec4d h = Q*(x,y,d,1); Vec3f p = h.xyz (cast);
 *        p /= h[3]  (float * double reciprocal, cast).
 * Both are double-precision evaluations; results differ by <= 2 float ulp.
 * Both end with `if (fabs(d - minDisparity) <= FLT_EPSILON) Z = bigZ` where
 * minDisparity = FLT_MAX and bigZ = 10000 for handleMissingValues = false
 * (what cpp:64 passes): a pixel with d == FLT_MAX gets Z = 10000.  Restated
 * here and in the kernel (round 1 omitted it: unreachable from the node,
 * whose d <= 31.875, but reachable through the fp32 entry). */
enum { D2PC_ORACLE_FORM_CV24 = 0, D2PC_ORACLE_FORM_CV4 = 1 };

/* hpp:84-104 -- closed form of cv::stereoRectify for the reference's rig
 * (identical pinhole cameras, zero distortion, R = I, t = (-baseline,0,0),
 * CALIB_ZERO_DISPARITY, alpha = -1).  q is row-major 4x4. */
void d2pc_oracle_make_q(double fx, double fy, double cx, double cy,
                        double baseline, int nx, int ny, double q[16]);

/* cpp:60-61 (and the u16 analogue) -- Mat::convertTo(CV_32FC1, scale):
 * d = (float)raw * (float)scale evaluated in fp32. */
float d2pc_oracle_decode(const void *row, int dtype, int x, float scale);

/* cpp:63-85, no filtering (what the reference publishes).
 * Writes (w-2b)*(h-2b) points of 16 bytes {x,y,z,1.0f} in row-major ROI
 * order; returns the number of points (0 when w<=2b or h<=2b).
 * `threads` > 1 splits ROI rows over OpenMP threads (bit-identical). */
size_t d2pc_oracle_reproject(const void *disp, int dtype, float scale,
                             int width, int height, size_t row_stride_bytes,
                             const double q[16], int border, int form,
                             int threads, float *out_points);

/* Same loop with the compaction predicate
 *   isfinite(x) && isfinite(y) && isfinite(z) && !(d <= min_disparity)
 * applied to the float32 results; survivors keep row-major order.
 * out_index (nullable) receives the source pixel index v*width+u. */
size_t d2pc_oracle_reproject_compact(const void *disp, int dtype, float scale,
                                     int width, int height,
                                     size_t row_stride_bytes,
                                     const double q[16], int border, int form,
                                     float min_disparity, float *out_points,
                                     uint32_t *out_index);

/* cpp:50 -- cv_bridge mono16 -> mono8: convertTo(CV_8U, 255./65535.)
 * = saturate_cast<uchar>(cvRound(v * (255./65535.))), round-half-even. */
void d2pc_oracle_mono16_to_mono8(const uint16_t *src, size_t src_stride_bytes,
                                 uint8_t *dst, size_t dst_stride_bytes,
                                 int width, int height);

/* cpp:55-57 -- cv::medianBlur(src, dst, ksize) on CV_8UC1, BORDER_REPLICATE.
 * ksize odd, >= 1. */
void d2pc_oracle_median_u8(const uint8_t *src, size_t src_stride_bytes,
                           uint8_t *dst, size_t dst_stride_bytes, int width,
                           int height, int ksize);

/* cpp:55-57, the same order statistic by the constant-time sliding-histogram
 * algorithm (Perreault & Hebert 2007: what cv::medianBlur runs for CV_8U,
 * ksize > 5 [upstream]).  Single-threaded.  The CPU column of bench.py's
 * callback-body lines; pinned byte for byte to d2pc_oracle_median_u8, which
 * stays the checker. */
void d2pc_oracle_median_u8_fast(const uint8_t *src, size_t src_stride_bytes,
                                uint8_t *dst, size_t dst_stride_bytes,
                                int width, int height, int ksize);

/* ---- depth-map fusion inner loop (d2pc_oracle_fusion.c) ---- */
/* The nine candidate rules of src/depth_map_fusion.cpp:162-235, numbered in
 * source order; GRAD_FILTER is the one getFusedDistance calls (cpp:159). */
enum {
  D2PC_ORACLE_FUSE_WEIGHTED_AVERAGE = 0,
  D2PC_ORACLE_FUSE_MAX_DIST = 1,
  D2PC_ORACLE_FUSE_MAX_DIST_UNLESS_BLACK = 2,
  D2PC_ORACLE_FUSE_BETTER_SCORE = 3,
  D2PC_ORACLE_FUSE_ONLY_GOOD_1 = 4,
  D2PC_ORACLE_FUSE_ONLY_GOOD_AVG = 5,
  D2PC_ORACLE_FUSE_OVERLAP = 6,
  D2PC_ORACLE_FUSE_BLACK_TO_WHITE = 7,
  D2PC_ORACLE_FUSE_GRAD_FILTER = 8
};
/* One pixel of cpp:150-160 (the int the rule returns, before the store to
 * unsigned char); -1 for an unknown rule. */
int d2pc_oracle_fuse_pixel(int rule, int d1, int d2, int s1, int s2, int g1, int g2);
/* cpp:113-130: planes = {depth1, depth2, score1, score2, grad1, grad2}.
 * Returns 0, or <0 on bad arguments / allocation failure. */
int d2pc_oracle_fuse(const uint8_t *const planes[6], const size_t pitch[6], int w, int h, int rule,
                     int crop_left, int crop_right, int crop_top, int crop_bottom,
                     uint8_t *fused, size_t fused_pitch, uint8_t *combined, size_t combined_pitch);
/* cpp:247-265: rect = {x, y, n}. */
void d2pc_oracle_crop_to_square(int cols, int rows, int offset_x, int offset_y, int member_offset_y, int rect[3]);
/* cpp:268-273: dst (cols rows of `rows` pixels) = src rotated 90 degrees clockwise. */
void d2pc_oracle_rotate_cw(const uint8_t *src, size_t src_pitch, int cols, int rows, uint8_t *dst, size_t dst_pitch);

int d2pc_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* D2PC_ORACLE_H */
