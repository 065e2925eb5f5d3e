/*
 * d2pc.h -- C ABI of the MI355X-native disparity -> point-cloud path.
 *
 * Drop-in boundary for ONE path of PX4/disparity_to_point_cloud: the body of
 * d2pc::Disparity2PCloud::DisparityCb between
 *     real_disparity            (src/disparity_to_point_cloud.cpp:60-61)
 * and
 *     sensor_msgs::PointCloud2 output.data / width / ...   (cpp:84-85)
 * i.e. cv::reprojectImageTo3D (cpp:63-64) + the inset-40 ROI push_back loop
 * (cpp:70-76) + cloud metadata (cpp:79-81) + pcl::toROSMsg (cpp:84-85).
 * The reference has no FFI/plugin interface; a maintainer binds these entry
 * points from DisparityCb as shown in INTEGRATION.md.
 *
 * Plain C: pointers and sizes only.  No HIP, torch, ROS, OpenCV or PCL types.
 * Every function returns a d2pc_status (0 = OK) and never throws or aborts.
 * There is NO CPU fallback inside this library: without a usable gfx950
 * device d2pc_create() fails with D2PC_ERR_NO_DEVICE.
 *
 * Threading (mirrors the reference's single-threaded ros::spin(),
 * src/disparity_to_point_cloud_node.cpp:50): a context is not thread-safe;
 * distinct contexts (one per GPU / per camera stream) are independent.
 */
#ifndef D2PC_H
#define D2PC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version policy.  This header is the STABLE surface: what a maintainer binds from DisparityCb.
 * D2PC_ABI_VERSION changes when, and only when, a function here changes its signature or meaning, a struct here
 * changes its layout, an enumerator changes its value, or a symbol is removed; additions keep it (structs carry
 * struct_size / reserved fields for that).  d2pc_abi_version() returns the library's value: refuse a library whose
 * value differs from the header you compiled against.
 *   1  rounds 1-3: one header that also carried the bench / tuning / counter / graph-reservation entry points
 *   2  those moved to include/d2pc_ext.h (unstable: tools, tests and benchmarks only; no compatibility promise);
 *      the string keys that switched the arithmetic ("reproject_form", "force_general_q", "general_q_form") are gone
 *      from d2pc_set_tuning -- d2pc_set_reproject_form is the one way to choose the arithmetic */
#define D2PC_ABI_VERSION 2

typedef enum d2pc_status {
  D2PC_OK = 0,
  D2PC_ERR_INVALID_ARG = 1,   /* null pointer, bad struct_size, bad enum      */
  D2PC_ERR_BAD_DTYPE = 2,     /* dtype not one of D2PC_DTYPE_*                */
  D2PC_ERR_BAD_SIZE = 3,      /* width/height/stride out of range             */
  D2PC_ERR_CAPACITY = 4,      /* output buffer too small for the points       */
  D2PC_ERR_NO_DEVICE = 5,     /* no HIP device / device_id out of range       */
  D2PC_ERR_DEVICE = 6,        /* a HIP call failed (see d2pc_last_error)      */
  D2PC_ERR_NOT_CALIBRATED = 7,/* d2pc_set_q / d2pc_set_calibration not called */
  D2PC_ERR_OUT_OF_MEMORY = 8,
  D2PC_ERR_INTERNAL = 9       /* compaction hand-off timed out (bounded spin) */
} d2pc_status;

/* Sample type of the disparity image handed over.
 * F32 is the reference's `real_disparity` (CV_32FC1, cpp:60-61).
 * U8/U16 fuse the reference's convertTo(CV_32FC1, scale) (cpp:61) into the
 * kernel: d = (float)raw * scale in fp32 (scale = 1/8 in the reference). */
typedef enum d2pc_dtype {
  D2PC_DTYPE_F32 = 0,
  D2PC_DTYPE_U8 = 1,
  D2PC_DTYPE_U16 = 2,       /* raw 16-bit disparities, decoded as (float)v * scale                     */
  D2PC_DTYPE_MONO16 = 3     /* HOST entry points only (d2pc_process, d2pc_pipeline_*): a 16-bit image that
                               cv_bridge::toCvCopy(msg,"mono8") rescales to 8 bits first (cpp:50:
                               convertTo(CV_8U, 255./65535.)); the rescale runs on the device and the
                               frame continues as U8 (median allowed).  d2pc_process_mono16 = this. */
} d2pc_dtype;

typedef enum d2pc_mode {
  /* What the reference publishes (cpp:70-81): every ROI pixel, row-major,
   * nothing filtered (inf/NaN included), N = (W-2b)(H-2b), is_dense=false. */
  D2PC_MODE_PARITY = 0,
  /* Extension named by BASELINE.json's north_star: order-preserving removal
   * of points with a non-finite coordinate (or d <= min_disparity); the
   * survivors keep row-major order, so the cloud is dense (is_dense=true). */
  D2PC_MODE_COMPACT = 1
} d2pc_mode;

typedef struct d2pc_config {
  uint32_t struct_size;   /* = sizeof(d2pc_config); set by d2pc_config_init   */
  int32_t device_id;      /* HIP device ordinal (rank-local GPU)              */
  int32_t border;         /* ROI inset on all four sides; cpp:70,72 => 40     */
  int32_t mode;           /* d2pc_mode                                        */
  float min_disparity;    /* COMPACT only: also drop d <= this; -inf = off    */
  int32_t compact_algo;   /* 0 = library default: single pass (2) for launches
                             of >= 4 frames and >= ~20k tiles; one launch of
                             resident blocks (3) for launches of up to two 4K
                             frames that are not being captured; two-pass (1)
                             else.  1 = two-pass count/scan/scatter;
                             2 = single-pass counted hand-off; 3 = resident
                             blocks (falls back where impossible).  Same bytes
                             out whatever the value. */
  int32_t reserved[4];
} d2pc_config;

typedef struct d2pc_ctx d2pc_ctx;

/* sensor_msgs/PointCloud2 metadata that pcl::toROSMsg (cpp:84-85) fills for a
 * pcl::PointCloud<pcl::PointXYZ> of n points with cpp:79-81's width/height. */
typedef struct d2pc_field { char name[8]; uint32_t offset; uint8_t datatype; uint32_t count; } d2pc_field;
typedef struct d2pc_cloud_meta {
  uint32_t height;        /* 1  (cpp:80)                                      */
  uint32_t width;         /* n  (cpp:79)                                      */
  uint32_t point_step;    /* 16 = sizeof(pcl::PointXYZ)                       */
  uint32_t row_step;      /* 16*n                                             */
  uint8_t is_bigendian;   /* 0                                                */
  uint8_t is_dense;       /* 0 in PARITY (cpp:81), 1 in COMPACT               */
  uint32_t n_fields;      /* 3                                                */
  d2pc_field fields[3];   /* x,y,z FLOAT32(7) @ 0,4,8 count 1                 */
} d2pc_cloud_meta;

/* Calibration blob broadcast once from rank 0 to every GPU's process
 * (RCCL/xGMI broadcast in the multi-GPU harness): 16 x f64 Q + border + mode. */
#define D2PC_CALIB_BLOB_BYTES 136

/* ---- library-level ---------------------------------------------------- */
int d2pc_abi_version(void);
const char *d2pc_status_string(int status);
int d2pc_device_count(void);            /* number of HIP devices, 0 if none  */

/* ---- calibration surface (hpp:66-71,84-104) --------------------------- */
/* Closed form of the reference's cv::stereoRectify call for its rig
 * (identical pinhole cameras, zero distortion, R = I, t = (-baseline,0,0),
 * image size (nx,ny) = (752,480) at hpp:101-103).  Host-only helper for
 * ROS-free callers; the ROS adaptor passes OpenCV's own Q_ to d2pc_set_q.
 * d2pc_make_q = D2PC_STEREORECTIFY_CONTINUOUS below. */
int d2pc_make_q(double fx, double fy, double cx, double cy, double baseline,
                int nx, int ny, double q_out[16]);

/* Where stereoRectify puts the new principal point depends on the OpenCV
 * release (the image-corner coordinates and the centre it re-centres on
 * changed between 2.4 and 3.x), and the reference pins no version.  For the
 * reference's case (no distortion, R = I) with c = source principal point,
 * n = image extent, f' = fy:
 *   CONTINUOUS  c' = (n-1)/2.0 - f'((n-1)/2.0 - c)/f   SURVEY.md section 8 row a9; no release computes
 *               exactly this (real-valued centre), it is "this closed form"
 *   CV24        c' = n/2 - f'(n/2.0 - c)/f     OpenCV 2.4.x (ROS Indigo, what .clang_complete:2 points at):
 *               corners at 0 and n, integer n/2.  752x480 defaults: cx' = 376 exactly
 *   CV3         c' = (n-1)/2 - f'((n-1)/2.0 - c)/f   OpenCV 3.x/4.x: corners at 0 and n-1, INTEGER (n-1)/2.
 *               752x480 defaults: cx' = 375.4995...
 * OpenCV evaluates parts of this in float, so none of the three is bit-for-bit
 * a release's Q: a node that has OpenCV passes ITS Q to d2pc_set_q
 * (ros/disparity_to_point_cloud_node.cpp does); the host mirror without OpenCV
 * defaults to CV24. */
typedef enum d2pc_stereorectify_flavour {
  D2PC_STEREORECTIFY_CONTINUOUS = 0,
  D2PC_STEREORECTIFY_CV24 = 1,
  D2PC_STEREORECTIFY_CV3 = 2
} d2pc_stereorectify_flavour;
int d2pc_make_q_flavour(double fx, double fy, double cx, double cy, double baseline,
                        int nx, int ny, int flavour, double q_out[16]);

/* Q for a stereo_msgs/DisparityImage-style source (SURVEY.md section 8(f) #3:
 * calibration carried by the message instead of ~fx_.. parameters): focal
 * length f (pixels), baseline T (metres, > 0), principal point (cx, cy):
 *   Q = [1 0 0 -cx; 0 1 0 -cy; 0 0 0 f; 0 0 1/T 0]  =>  Z = f*T/d. */
int d2pc_make_q_disparity_image(double f, double T, double cx, double cy, double q_out[16]);

/* Host-only packing of the calibration blob (what rank 0 broadcasts): usable
 * without a device, e.g. by the process that owns the ROS parameters. */
int d2pc_calib_pack(const double q[16], int border, int mode, void *blob /*136 B*/);
int d2pc_calib_unpack(const void *blob, size_t blob_bytes, double q_out[16], int *border, int *mode);

/* ---- context ---------------------------------------------------------- */
int d2pc_config_init(d2pc_config *cfg); /* reference defaults: border 40, PARITY */
int d2pc_create(const d2pc_config *cfg, d2pc_ctx **out_ctx);
int d2pc_destroy(d2pc_ctx *ctx);
const char *d2pc_last_error(const d2pc_ctx *ctx);

/* Q_ (hpp:72), row-major 4x4 doubles exactly as Q_.ptr<double>() yields. */
int d2pc_set_q(d2pc_ctx *ctx, const double q[16]);
int d2pc_get_q(const d2pc_ctx *ctx, double q_out[16]);
int d2pc_set_border(d2pc_ctx *ctx, int border);
int d2pc_set_mode(d2pc_ctx *ctx, int mode);
/* Which cv::reprojectImageTo3D (cpp:64) the points reproduce.  The two OpenCV
 * generations evaluate the same Q*[x y d 1]^T in different orders and differ
 * from each other by up to 1 float ulp (more where a numerator cancels):
 *   D2PC_FORM_DEFAULT  cv::stereoRectify's Q: a specialised kernel, <= 1 ulp
 *                      from BOTH generations (the exact quotient rounded once);
 *                      any other Q: OpenCV 3/4's form bit for bit
 *   D2PC_FORM_CV24     OpenCV 2.4's loop bit for bit (per row qx = q01*y + q03,
 *                      then qx += q00 per column, one rounding per step).  That
 *                      recurrence is sequential in x; it is reproduced in
 *                      parallel only for a Q whose column steps are exact
 *                      (q00 = 1, q01 = q10 = q20 = q30 = +0 -- what
 *                      stereoRectify and d2pc_make_q* produce); any other Q
 *                      makes the next call return D2PC_ERR_INVALID_ARG.  The
 *                      running sum rounds once per binade it crosses inside a
 *                      row; the library holds 18 such segments (a principal
 *                      point of any size on rows of up to 65,536 columns) and
 *                      returns D2PC_ERR_BAD_SIZE beyond
 *   D2PC_FORM_CV4      OpenCV 3/4's form bit for bit for every Q (Matx product
 *                      left to right, numerators cast to float, times 1./W)
 * For cv::stereoRectify's Q the exact forms have kernels of their own and cost
 * about as much as the default; any other Q runs them in the general kernel
 * (double, no contraction), which is arithmetic-bound:
 * profiles/r03_ab_forms.txt has all of them beside the default. */
typedef enum d2pc_reproject_form {
  D2PC_FORM_DEFAULT = 0,
  D2PC_FORM_CV24 = 24,
  D2PC_FORM_CV4 = 4
} d2pc_reproject_form;
int d2pc_set_reproject_form(d2pc_ctx *ctx, int form);
/* COMPACT predicate threshold (e.g. DisparityImage.min_disparity): points with
 * d <= min_disparity are dropped; -inf disables it.  NaN is rejected. */
int d2pc_set_min_disparity(d2pc_ctx *ctx, float min_disparity);
int d2pc_get_config(const d2pc_ctx *ctx, d2pc_config *cfg_out);
int d2pc_export_calibration(const d2pc_ctx *ctx, void *blob /*136 B*/);
int d2pc_import_calibration(d2pc_ctx *ctx, const void *blob, size_t blob_bytes);

/* Number of ROI pixels (= points in PARITY mode): max(w-2b,0)*max(h-2b,0). */
size_t d2pc_roi_points(int width, int height, int border);
int d2pc_cloud_meta_fill(const d2pc_ctx *ctx, size_t n_points, d2pc_cloud_meta *meta);

/* ---- the hot path ----------------------------------------------------- */
/*
 * Replaces cpp:63-85 for one frame held in HOST memory (synchronous).
 *   disp              H x W samples of `dtype`, rows `row_stride_bytes` apart
 *   scale             U8/U16 only (cpp:61: 1/8); ignored for F32
 *   out_points        capacity_points records of 16 bytes {x,y,z,1.0f}: the
 *                     exact bytes of PointCloud2.data (pass output.data.data())
 *   out_index         nullable; source pixel index v*W+u of every emitted
 *                     point (uint32)
 *   n_points          points written: (W-2b)(H-2b) in PARITY, #valid in COMPACT
 * Nothing is retained after return.
 */
int d2pc_process(d2pc_ctx *ctx, const void *disp, int dtype, float scale,
                 int width, int height, size_t row_stride_bytes,
                 void *out_points, uint32_t *out_index,
                 size_t capacity_points, size_t *n_points);

/*
 * Device-resident, batched, asynchronous form (frames already in HBM):
 * n_frames frames, `in_frame_stride_bytes` apart, are converted by ONE
 * kernel sequence enqueued on `stream` (a hipStream_t passed as void*;
 * NULL = HIP's default stream, as in the HIP API).  Frame f's points go to
 * d_out_points + f*out_frame_stride_points*16 (and d_out_index +
 * f*out_frame_stride_points); d_counts[f] (uint32, nullable in PARITY)
 * receives the number of points of frame f.  All device pointers must belong
 * to the context's device; d_out_points must be 16-byte aligned -- 128-byte
 * alignment (of the base and of out_frame_stride_points*16) keeps every wave
 * store on whole 64-byte memory requests (a base that is only 16-byte aligned is slower).
 * Does not synchronise.
 *
 * COMPACT launches in flight at once: calls on ONE stream are ordered and share
 * the context's compaction state; a call on another stream gets a state buffer
 * of its own (up to 8 per context), so double-buffered use of one context on
 * two streams is safe.  The context itself is still used from one host thread.
 * A stream handed to a COMPACT call (here and in d2pc_process_mono_device) must
 * stay valid until the context is destroyed: the context asks it later whether
 * the work it was given has drained, before another stream may take the buffer.
 *
 * hipGraph capture: nothing can be allocated while capturing, so a COMPACT call
 * needs its state buffer beforehand (D2PC_ERR_OUT_OF_MEMORY otherwise): run the
 * largest batch once, or reserve it (d2pc_reserve in d2pc_ext.h) before EACH
 * capture.  The state buffer a capture used belongs to that graph from then on:
 * later calls never free, grow or share it, so the graph stays replayable
 * whatever else the context is asked to do (d2pc_release_graph_buffers in
 * d2pc_ext.h hands them back).  One replay of a given graph in flight at a time.
 *
 * COMPACT failure reporting in-band: if the single-pass kernel gave up waiting
 * for an earlier tile (see d2pc_check_async_error) d_counts[f] of the affected
 * frames is 0xFFFFFFFF instead of a count.
 */
int d2pc_process_device(d2pc_ctx *ctx, const void *d_disp, int dtype,
                        float scale, int width, int height,
                        size_t row_stride_bytes, size_t in_frame_stride_bytes,
                        int n_frames, void *d_out_points,
                        uint32_t *d_out_index,
                        size_t out_frame_stride_points, uint32_t *d_counts,
                        void *stream);

/*
 * The rest of the callback body on the device:
 * k x k median of 8-bit frames with replicated borders, i.e.
 * cv::medianBlur(image, filtered, 11) at cpp:55-57.  ksize odd, 3..11.
 * Device-resident, batched, asynchronous on `stream` like
 * d2pc_process_device; src and dst must not overlap.
 */
int d2pc_median_device(d2pc_ctx *ctx, const void *d_src, int width, int height,
                       size_t src_row_stride_bytes, size_t src_frame_stride_bytes,
                       int n_frames, void *d_dst, size_t dst_row_stride_bytes,
                       size_t dst_frame_stride_bytes, int ksize, void *stream);

/*
 * The same filter restricted to what the callback reads afterwards: only the
 * pixels of the context's inset ROI (cpp:70,72: rows and columns [border,
 * dim - border)) are computed and written; d_dst outside the ROI is left
 * untouched.  Windows still read the whole source image (replicated only at
 * its true edges), so every ROI pixel equals d2pc_median_device's.  At the
 * reference's 752x480 / border 40 this is 25.5 % less work.  The fused entry
 * points (d2pc_process_mono8/16, d2pc_pipeline_*) filter this way.
 */
int d2pc_median_roi_device(d2pc_ctx *ctx, const void *d_src, int width, int height,
                       size_t src_row_stride_bytes, size_t src_frame_stride_bytes,
                       int n_frames, void *d_dst, size_t dst_row_stride_bytes,
                       size_t dst_frame_stride_bytes, int ksize, void *stream);

/*
 * cv_bridge::toCvCopy(msg, "mono8") applied to a mono16 image (cpp:50) on the
 * device: dst = saturate(round_half_even(src * (float)(255./65535.))), i.e.
 * cv::Mat::convertTo(CV_8U, 255./65535.).  src rows hold uint16 samples (strides
 * in bytes, 2-byte aligned); asynchronous on `stream`; src and dst must not overlap.
 */
int d2pc_mono16_to_mono8_device(d2pc_ctx *ctx, const void *d_src, int width, int height,
                                size_t src_row_stride_bytes, size_t src_frame_stride_bytes,
                                int n_frames, void *d_dst, size_t dst_row_stride_bytes,
                                size_t dst_frame_stride_bytes, void *stream);

/*
 * cpp:55-85 in one call for one mono8 frame in HOST memory (what
 * cv_bridge::toCvCopy(msg,"mono8") returned, cpp:50): median (median_ksize,
 * the reference uses 11; 0 or 1 = none) -> x scale (1/8, cpp:61) ->
 * reprojection + ROI pack.  Only 1 byte per pixel crosses PCIe on the way in.
 * Other arguments as d2pc_process.
 */
int d2pc_process_mono8(d2pc_ctx *ctx, const uint8_t *image, int width, int height,
                       size_t row_stride_bytes, int median_ksize, float scale,
                       void *out_points, uint32_t *out_index,
                       size_t capacity_points, size_t *n_points);

/* The same for a mono16 frame (what the node receives when the disparity is
 * published as 16-bit): cpp:50's cv_bridge rescale to mono8 runs on the device
 * in front of the median.  2 bytes per pixel cross PCIe on the way in. */
int d2pc_process_mono16(d2pc_ctx *ctx, const uint16_t *image, int width, int height,
                        size_t row_stride_bytes, int median_ksize, float scale,
                        void *out_points, uint32_t *out_index,
                        size_t capacity_points, size_t *n_points);

/*
 * The whole callback body (cpp:50-85) for a BATCH of frames resident in HBM:
 * [MONO16: cv_bridge rescale ->] k x k median over the inset ROI -> x scale ->
 * reproject + pack, asynchronous on `stream` like d2pc_process_device (same
 * output arguments).  dtype is D2PC_DTYPE_U8 or D2PC_DTYPE_MONO16;
 * median_ksize 0/1 skips the filter.
 *
 * With the default tuning, PARITY mode and a launch of at least 448 tiles of
 * 256 x 32 ROI pixels (one 4K frame has 975; 192 / 320 tiles for 3x3 / 5x5 windows) the call is ONE
 * kernel that works tile by tile: the bit-sliced median of the tile, then the
 * tile's points straight from the filtered bytes in LDS -- for stereoRectify's Q
 * through a per-block table of 1/W and Z over the 256 byte values.  The filtered
 * frames never reach memory.  Everything else -- small launches -- is the filter
 * launch followed by the reprojection launch.  The results are the same bytes
 * either way (measurements: DESIGN.md section 5).
 * (Launch-shape knobs: d2pc_set_tuning in d2pc_ext.h.)
 * In the two-launch form the filtered frames live in a scratch buffer that belongs
 * to the calling stream's work (one per stream in flight, like the compaction
 * state: double-buffered use on two streams is safe); it is grown on demand, so
 * before a capture run the batch size once (or reserve it: d2pc_reserve_mono in
 * d2pc_ext.h); under stream capture the call runs in order on `stream`.  The
 * one-kernel form needs no scratch.
 */
int d2pc_process_mono_device(d2pc_ctx *ctx, const void *d_image, int dtype, int width, int height,
                             size_t row_stride_bytes, size_t frame_stride_bytes, int n_frames,
                             int median_ksize, float scale, void *d_out_points, uint32_t *d_out_index,
                             size_t out_frame_stride_points, uint32_t *d_counts, void *stream);

/*
 * Pinned (page-locked) host buffers for the synchronous entry points above
 * ("write the output into a pinned buffer that IS
 * output.data").  When `out_points` (and `out_index`, if given) of a
 * d2pc_process* call lie in memory from d2pc_host_alloc and hold the whole ROI
 * (capacity_points >= ROI points), the kernels store the final PointCloud2
 * bytes straight into it over PCIe: no device-side copy of the cloud, no
 * D2H copy, no bounce through the runtime's staging buffers.  A pinned `disp`
 * / `image` is read by the reprojection kernel in place when that kernel is the
 * first to touch it (PARITY, no median, no mono16 rescale: the inbound reads
 * then overlap the outbound stores on the full-duplex link); otherwise it makes
 * the upload one DMA.  Pageable buffers keep working as before.
 * host/pinned_allocator.hpp wraps these two in a caching std::allocator so that
 * sensor_msgs::PointCloud2_<Alloc>::data can be such a buffer.
 */
void *d2pc_host_alloc(size_t bytes);
void d2pc_host_free(void *p);

/*
 * Pipelined host path: up to `depth`
 * frames in flight, each on its own HIP stream with its own PINNED staging:
 *   acquire -> the caller decodes the image straight into pinned memory
 *   submit  -> async H2D, (median), kernels, D2H (or, with direct_host_write,
 *              the kernels store the points straight into pinned host memory)
 *   collect -> waits for the OLDEST submitted frame and hands out a view of
 *              its pinned output: the PointCloud2 payload without another copy
 *   release -> the slot can be acquired again
 * The copy in of frame i+1, the kernels of frame i and the copy out of frame
 * i-1 overlap.  Frames are collected in submission order.
 */
typedef struct d2pc_frame_desc {
  int32_t dtype;             /* d2pc_dtype (MONO16 allowed) */
  float scale;               /* U8/U16 decode scale (cpp:61: 1/8) */
  int32_t width, height;
  size_t row_stride_bytes;   /* layout of the pinned input buffer */
  int32_t median_ksize;      /* 0/1 = none; else odd 3..11, U8 only (cpp:55-57) */
  int32_t want_index;        /* also produce source pixel indices */
  uint64_t tag;              /* returned by collect (e.g. the message stamp) */
} d2pc_frame_desc;

int d2pc_pipeline_configure(d2pc_ctx *ctx, int depth /*1..8*/, int direct_host_write);
int d2pc_pipeline_acquire(d2pc_ctx *ctx, const d2pc_frame_desc *desc, void **host_in, int *slot);
int d2pc_pipeline_submit(d2pc_ctx *ctx, int slot);
int d2pc_pipeline_collect(d2pc_ctx *ctx, int *slot, const void **points, const uint32_t **index,
                          size_t *n_points, uint64_t *tag);
int d2pc_pipeline_release(d2pc_ctx *ctx, int slot);

/* After the streams of the context's COMPACT d2pc_process_device calls have been
 * synchronised: D2PC_ERR_INTERNAL if a hand-off wait of the single-pass kernel
 * ran out of its time budget (4 s) in the
 * LAST launch of any of the context's state buffers -- each buffer remembers
 * the algorithm of its own last launch.  The output of such a launch is
 * incomplete and its d_counts entries read 0xFFFFFFFF; relaunch with
 * compact_algo = 1.  d2pc_process* (synchronous) does that relaunch itself;
 * d2pc_pipeline_collect reports the frame as D2PC_ERR_INTERNAL. */
int d2pc_check_async_error(d2pc_ctx *ctx);

/*
 * SURVEY.md section 8(f) #4 -- the inner loop of the sibling node's
 * DepthMapFusion::publishFusedDepthMap, device-resident and batched:
 *   per pixel  fused = rule(depth1, depth2, score1, score2, grad1, grad2)
 *              (src/depth_map_fusion.cpp:115-117,150-160; rules :162-235),
 *              combined = min(grad1, grad2)                  (:118-121)
 *   then       cv::medianBlur(fused, fused, 3)               (:124)
 *   then       cropMat(fused, left, right, top, bottom)      (:130: 0,40,30,10)
 * in one kernel.  Rules are numbered in the reference's source order;
 * GRAD_FILTER is the one getFusedDistance calls (:159).  WEIGHTED_AVERAGE
 * divides by zero in the reference when both scores are non-zero; that case
 * is defined as 0 here.
 */
typedef enum d2pc_fusion_rule {
  D2PC_FUSE_WEIGHTED_AVERAGE = 0,
  D2PC_FUSE_MAX_DIST = 1,
  D2PC_FUSE_MAX_DIST_UNLESS_BLACK = 2,
  D2PC_FUSE_BETTER_SCORE = 3,
  D2PC_FUSE_ONLY_GOOD_1 = 4,
  D2PC_FUSE_ONLY_GOOD_AVG = 5,
  D2PC_FUSE_OVERLAP = 6,
  D2PC_FUSE_BLACK_TO_WHITE = 7,
  D2PC_FUSE_GRAD_FILTER = 8
} d2pc_fusion_rule;

typedef struct d2pc_fuse_desc {
  uint32_t struct_size;        /* sizeof(d2pc_fuse_desc) */
  int32_t rule;                /* d2pc_fusion_rule */
  int32_t width, height;       /* of every input plane (n x n after cropToSquare in the reference) */
  int32_t n_frames;            /* independent plane sets, frame_stride apart */
  int32_t crop_left, crop_right, crop_top, crop_bottom;
  const void *planes[6];       /* DEVICE: depth1, depth2, score1, score2, grad1, grad2 (8-bit).
                                  grad1/grad2 may be NULL when `combined` is NULL; planes may alias each
                                  other (the reference's score1 and grad1 share a buffer, cpp:77) */
  size_t pitch[6];             /* bytes between rows */
  size_t frame_stride[6];      /* bytes between frames (ignored when n_frames == 1) */
  void *fused;                 /* DEVICE out: (width-left-right) x (height-top-bottom) */
  size_t fused_pitch, fused_frame_stride;
  void *combined;              /* DEVICE out, nullable: width x height */
  size_t combined_pitch, combined_frame_stride;
} d2pc_fuse_desc;

/* struct_size, GRAD_FILTER, n_frames 1, crop 0/40/30/10; everything else zero. */
void d2pc_fuse_desc_init(d2pc_fuse_desc *desc);
/* Asynchronous on `stream` (NULL = the HIP default stream).  Outputs must not
 * overlap the inputs or each other: the reference writes `combined` over
 * score1 in place (cpp:113), which a caller reproduces by swapping buffers. */
int d2pc_fuse_device(d2pc_ctx *ctx, const d2pc_fuse_desc *desc, void *stream);
/* rotateMat (src/depth_map_fusion.cpp:268-273: cv::transpose + cv::flip(.,1) = 90 degrees clockwise) of 8-bit
 * frames on the device: dst has `cols` rows of `rows` pixels, dst(i,j) = src(rows-1-j, i).  What DisparityCb2 and
 * MatchingScoreCb2 apply to camera 2's images before cropToSquare (cpp:56,84).  Asynchronous on `stream`. */
int d2pc_rotate_cw_device(d2pc_ctx *ctx, const void *d_src, int cols, int rows, size_t src_pitch,
                          size_t src_frame_stride, int n_frames, void *d_dst, size_t dst_pitch,
                          size_t dst_frame_stride, void *stream);
/* cropToSquare (src/depth_map_fusion.cpp:247-265): the square view of a
 * cols x rows image, shifted by the offsets.  The side length uses the class
 * member offset_y_ (cpp:253), the origin the argument: pass both. */
int d2pc_crop_to_square(int cols, int rows, int offset_x, int offset_y, int member_offset_y,
                        int *x, int *y, int *n);

#ifdef __cplusplus
}
#endif
#endif /* D2PC_H */
