/*
 * d2pc_oracle.c -- CPU restatement of Disparity2PCloud::DisparityCb's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see d2pc_oracle.h).  PARITY UNPINNED: the
 * reference has no tests or golden data, and OpenCV / cv_bridge / PCL are
 * neither vendored in /root/reference nor buildable here; the third-party
 * algorithms are restated from their published sources and tagged [upstream].
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off -fopenmp; contraction
 * is OFF so every multiply and add rounds separately, as the x86-64/SSE2
 * builds of OpenCV 2.4 the reference ran against did).
 */
#include "d2pc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------
 * hpp:84-104.  cv::stereoRectify [upstream calib3d] specialised to the
 * reference's call: K1 == K2 == [[fx,0,cx],[0,fy,cy],[0,0,1]], zero
 * distortion, R = I, T = (-b,0,0), flags = CALIB_ZERO_DISPARITY, alpha = -1,
 * newImageSize = imageSize = (nx,ny) (hpp:101-104 hard-codes 752x480).
 *
 *   horizontal rig (|Tx| > |Ty|)  => idx = 0, fc_new = K[1][1] = fy
 *   corners (0,0),(nx-1,0),(0,ny-1),(nx-1,ny-1) are normalised with K and
 *   re-projected with focal fc_new, principal point 0; their mean is
 *   fc_new*((nx-1)/2 - cx)/fx  (x)  and  fc_new*((ny-1)/2 - cy)/fy  (y)
 *   cc_new = (n-1)/2 - mean;   both cameras equal => cx1' == cx2'
 *   Q = [1 0 0 -cx'; 0 1 0 -cy'; 0 0 0 fc_new; 0 0 -1/Tx (cx1'-cx2')/Tx]
 *
 * With Tx = -b: Q[3][2] = +1/b and Q[3][3] = 0/(-b) = -0.0 (negative zero is
 * kept: it decides the sign of W for d == 0).
 * The exact last digits OpenCV produces depend on its version (float32 corner
 * storage, nx vs nx-1): the product never relies on this helper for parity --
 * Q crosses the C-ABI as data -- it only feeds the ROS-free harness.
 * ------------------------------------------------------------------------- */
void d2pc_oracle_make_q(double fx, double fy, double cx, double cy,
                        double baseline, int nx, int ny, double q[16]) {
  const double fc_new = fy;
  const double hx = (double)(nx - 1) / 2.0, hy = (double)(ny - 1) / 2.0;
  const double cxn = hx - fc_new * (hx - cx) / fx;
  const double cyn = hy - fc_new * (hy - cy) / fy;
  const double tx = -baseline;
  const double row[16] = {1, 0, 0,         -cxn,
                          0, 1, 0,         -cyn,
                          0, 0, 0,         fc_new,
                          0, 0, -1.0 / tx, (cxn - cxn) / tx};
  memcpy(q, row, sizeof row);
}

/* cpp:60-61.  Mat::convertTo(CV_32FC1, alpha) [upstream core/convert.cpp,
 * cvtScale_<uchar|ushort, float, float>]: dst = src*(float)alpha + 0.f,
 * evaluated in float.  For the reference's alpha = 1/8 this is exact. */
float d2pc_oracle_decode(const void *row, int dtype, int x, float scale) {
  switch (dtype) {
    case D2PC_ORACLE_U8:
      return (float)((const uint8_t *)row)[x] * scale;
    case D2PC_ORACLE_U16:
      return (float)((const uint16_t *)row)[x] * scale;
    default:
      return ((const float *)row)[x];
  }
}

typedef struct {
  float x, y, z;
} xyz_t;

/* cpp:63-64.  One image row of cv::reprojectImageTo3D(disp, out, Q,
 * handleMissingValues=false, ddepth=-1) for columns [u0,u1).
 *
 * FORM_CV24 [upstream calib3d/calibration.cpp, OpenCV 2.4.x]:
 *   qx = q01*y + q03, qy = q11*y + q13, qz = q21*y + q23, qw = q31*y + q33
 *   for x = 0..cols-1:                       (qx += q00, ... after each x)
 *     d  = sptr[x]
 *     iW = 1./(qw + q32*d)
 *     X = (qx + q02*d)*iW; Y = (qy + q12*d)*iW; Z = (qz + q22*d)*iW
 *     dptr[x] = Vec3f((float)X,(float)Y,(float)Z)
 *   The x-recurrence starts at column 0, so it is replayed from 0 even when
 *   only [u0,u1) is emitted.
 *
 * FORM_CV4 [upstream calib3d/calibration.cpp + core/matx.hpp, OpenCV 3/4]:
 *   Vec4d h = Q*Vec4d(x,y,d,1)  (row dot products, left to right)
 *   Vec3f p = Vec3d(h.val)      (cast of the numerators to float)
 *   p /= h[3]                   (ia = 1./h[3]; p[i] = (float)(p[i]*ia))
 *
 * Both forms end with [upstream, same file]
 *   if (fabs(d - minDisparity) <= FLT_EPSILON) Z = bigZ;      bigZ = 10000.
 * where minDisparity stays at its initial FLT_MAX unless handleMissingValues
 * is set (cpp:64 leaves it false): the test is true only for d == FLT_MAX,
 * which then gets Z = 10000 (X and Y are left as computed).  Unreachable from
 * the node (d <= 31.875 after cpp:61), reachable through the fp32 seam.
 */
#define D2PC_ORACLE_MIN_DISPARITY ((double)FLT_MAX)
#define D2PC_ORACLE_BIG_Z 10000.
static void reproject_row(const void *row, int dtype, float scale, int y,
                          int u0, int u1, const double *q, int form,
                          xyz_t *out) {
  if (form == D2PC_ORACLE_FORM_CV24) {
    double qx = q[1] * y + q[3], qy = q[5] * y + q[7];
    double qz = q[9] * y + q[11], qw = q[13] * y + q[15];
    for (int x = 0; x < u1; x++, qx += q[0], qy += q[4], qz += q[8],
             qw += q[12]) {
      if (x < u0) continue;
      const double d = d2pc_oracle_decode(row, dtype, x, scale);
      const double iW = 1. / (qw + q[14] * d);
      const double X = (qx + q[2] * d) * iW;
      const double Y = (qy + q[6] * d) * iW;
      double Z = (qz + q[10] * d) * iW;
      if (fabs(d - D2PC_ORACLE_MIN_DISPARITY) <= FLT_EPSILON) Z = D2PC_ORACLE_BIG_Z;
      out[x - u0].x = (float)X;
      out[x - u0].y = (float)Y;
      out[x - u0].z = (float)Z;
    }
  } else {
    for (int x = u0; x < u1; x++) {
      const double d = d2pc_oracle_decode(row, dtype, x, scale);
      double h[4];
      for (int r = 0; r < 4; r++)
        h[r] = 0.0 + q[4 * r] * x + q[4 * r + 1] * y + q[4 * r + 2] * d +
               q[4 * r + 3] * 1.0; /* Matx product: s = 0; s += a_k*b_k */
      /* Vec3f p = Vec3d(h.val): the numerators exist as floats before the division (p /= h[3] multiplies each
       * by ia = 1./h[3] in double and casts back).  `volatile`: gcc 11.4 -O3 vectorises the x/y pair and drops
       * the float round trip of (float)h * ia there (seen in the disassembly: vmulpd + one vcvtpd2ps), i.e. the
       * binary computed (float)(h * ia) for x and y -- one cast less than the published form.
       * tests/test_oracle.py pins the binary against a numpy restatement of both forms. */
      volatile float p0 = (float)h[0], p1 = (float)h[1], p2 = (float)h[2];
      const double ia = 1. / h[3];
      out[x - u0].x = (float)((double)p0 * ia);
      out[x - u0].y = (float)((double)p1 * ia);
      out[x - u0].z = (float)((double)p2 * ia);
      if (fabs(d - D2PC_ORACLE_MIN_DISPARITY) <= FLT_EPSILON)
        out[x - u0].z = (float)D2PC_ORACLE_BIG_Z;
    }
  }
}

/* cpp:70-85.  ROI gather + pcl::PointXYZ + pcl::toROSMsg:
 *   for v in [b, h-b): for u in [b, w-b): push_back(PointXYZ(x,y,z))
 * pcl::PointXYZ is 16 bytes {x,y,z,data[3]=1.0f} [upstream PCL point_types];
 * toROSMsg memcpy's the point array, so the PointCloud2 payload is exactly
 * these 16-byte records in push order. */
size_t d2pc_oracle_reproject(const void *disp, int dtype, float scale,
                             int width, int height, size_t row_stride_bytes,
                             const double q[16], int border, int form,
                             int threads, float *out_points) {
  const int rw = width - 2 * border, rh = height - 2 * border;
  if (rw <= 0 || rh <= 0) return 0;
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) if (threads > 1)
  {
    xyz_t *tmp = (xyz_t *)malloc((size_t)rw * sizeof(xyz_t));
#pragma omp for schedule(static)
    for (int v = border; v < height - border; v++) {
      const char *row = (const char *)disp + (size_t)v * row_stride_bytes;
      reproject_row(row, dtype, scale, v, border, width - border, q, form,
                    tmp);
      float *dst = out_points + (size_t)(v - border) * rw * 4;
      for (int i = 0; i < rw; i++) {
        dst[4 * i + 0] = tmp[i].x;
        dst[4 * i + 1] = tmp[i].y;
        dst[4 * i + 2] = tmp[i].z;
        dst[4 * i + 3] = 1.0f;
      }
    }
    free(tmp);
  }
  return (size_t)rw * rh;
}

size_t d2pc_oracle_reproject_compact(const void *disp, int dtype, float scale,
                                     int width, int height,
                                     size_t row_stride_bytes,
                                     const double q[16], int border, int form,
                                     float min_disparity, float *out_points,
                                     uint32_t *out_index) {
  const int rw = width - 2 * border, rh = height - 2 * border;
  if (rw <= 0 || rh <= 0) return 0;
  xyz_t *tmp = (xyz_t *)malloc((size_t)rw * sizeof(xyz_t));
  size_t n = 0;
  for (int v = border; v < height - border; v++) {
    const char *row = (const char *)disp + (size_t)v * row_stride_bytes;
    reproject_row(row, dtype, scale, v, border, width - border, q, form, tmp);
    for (int i = 0; i < rw; i++) {
      const float d = d2pc_oracle_decode(row, dtype, border + i, scale);
      if (!(isfinite(tmp[i].x) && isfinite(tmp[i].y) && isfinite(tmp[i].z)))
        continue;
      if (d <= min_disparity) continue;
      out_points[4 * n + 0] = tmp[i].x;
      out_points[4 * n + 1] = tmp[i].y;
      out_points[4 * n + 2] = tmp[i].z;
      out_points[4 * n + 3] = 1.0f;
      if (out_index) out_index[n] = (uint32_t)v * (uint32_t)width + (uint32_t)(border + i);
      n++;
    }
  }
  free(tmp);
  return n;
}

/* cpp:50.  cv_bridge::toCvCopy(msg,"mono8") on a mono16 source [upstream
 * cv_bridge.cpp, SAME_FORMAT branch]: image.convertTo(out, CV_8U, 255./65535.)
 * => cvtScale_<ushort,uchar,float>: saturate_cast<uchar>(src*(float)alpha),
 * product in float, cvRound = round-half-to-even. */
void d2pc_oracle_mono16_to_mono8(const uint16_t *src, size_t src_stride_bytes,
                                 uint8_t *dst, size_t dst_stride_bytes,
                                 int width, int height) {
  const float a = (float)(255. / 65535.);
  for (int y = 0; y < height; y++) {
    const uint16_t *s = (const uint16_t *)((const char *)src + y * src_stride_bytes);
    uint8_t *d = dst + y * dst_stride_bytes;
    for (int x = 0; x < width; x++) {
      long r = lrintf((float)s[x] * a); /* default rounding mode: nearest-even */
      d[x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
  }
}

/* cpp:55-57.  cv::medianBlur(src,dst,ksize) for CV_8UC1 [upstream
 * imgproc/smooth.cpp]: every output pixel is the (k*k/2)-th order statistic
 * of its k x k neighbourhood, taps clamped to the image (BORDER_REPLICATE).
 * OpenCV's O(1)/O(m) histogram variants return exactly this value. */
void d2pc_oracle_median_u8(const uint8_t *src, size_t src_stride_bytes,
                           uint8_t *dst, size_t dst_stride_bytes, int width,
                           int height, int ksize) {
  const int r = ksize / 2, half = (ksize * ksize) / 2;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; y++) {
    int hist[256];
    for (int x = 0; x < width; x++) {
      memset(hist, 0, sizeof hist);
      for (int dy = -r; dy <= r; dy++) {
        int yy = y + dy;
        yy = yy < 0 ? 0 : yy >= height ? height - 1 : yy;
        const uint8_t *s = src + (size_t)yy * src_stride_bytes;
        for (int dx = -r; dx <= r; dx++) {
          int xx = x + dx;
          xx = xx < 0 ? 0 : xx >= width ? width - 1 : xx;
          hist[s[xx]]++;
        }
      }
      int acc = 0, m = 0;
      for (; m < 256; m++) {
        acc += hist[m];
        if (acc > half) break;
      }
      dst[(size_t)y * dst_stride_bytes + x] = (uint8_t)m;
    }
  }
}

/* cpp:55-57 again, FAST form: the constant-time median of Perreault & Hebert
 * ("Median Filtering in Constant Time", IEEE TIP 2007), the algorithm
 * cv::medianBlur uses for CV_8U and ksize > 5 [upstream imgproc/smooth.cpp,
 * medianBlur_8u_O1]: one 256-bin histogram per image column (16 coarse + 256
 * fine counters) slides down one row at a time (one pixel out, one in); the
 * kernel histogram slides along the row by adding the entering and removing
 * the leaving column's COARSE histogram, and brings only the fine segment the
 * median falls into up to date (lazily, `luc` = first virtual column not yet
 * in it).  Taps are clamped to the image (BORDER_REPLICATE): virtual column j
 * stands for column clamp(j).  Same order statistic as d2pc_oracle_median_u8,
 * which stays the CHECKER (tests/test_oracle.py pins this one to it byte for
 * byte); this one is the CPU COLUMN of bench.py's callback-body lines.
 * Single-threaded, like the reference's spinner. */
void d2pc_oracle_median_u8_fast(const uint8_t *src, size_t src_stride_bytes,
                                uint8_t *dst, size_t dst_stride_bytes,
                                int width, int height, int ksize) {
  const int r = ksize / 2, half = (ksize * ksize) / 2;
  if (width <= 0 || height <= 0) return;
  /* column stripes (as upstream: the histograms of one stripe stay in cache): STRIPE output columns + r halo
   * columns on either side; local column c stands for image column clamp(x0 - r + c) */
  enum { STRIPE = 384 };
  const int max_cols = (width < STRIPE ? width : STRIPE) + 2 * r;
  uint16_t *coarse = (uint16_t *)malloc((size_t)max_cols * 16 * sizeof(uint16_t));
  uint16_t *fine = (uint16_t *)malloc((size_t)max_cols * 256 * sizeof(uint16_t));
  int *colx = (int *)malloc((size_t)max_cols * sizeof(int));
  if (!coarse || !fine || !colx) {  /* out of memory: the checker's algorithm needs none and gives the same bytes */
    free(coarse); free(fine); free(colx);
    d2pc_oracle_median_u8(src, src_stride_bytes, dst, dst_stride_bytes, width, height, ksize);
    return;
  }
#define D2PC_ROW(yy) (src + (size_t)((yy) < 0 ? 0 : (yy) >= height ? height - 1 : (yy)) * src_stride_bytes)
  for (int x0 = 0; x0 < width; x0 += STRIPE) {
    const int sw = width - x0 < STRIPE ? width - x0 : STRIPE, nc = sw + 2 * r;
    for (int c = 0; c < nc; c++) {
      const int j = x0 - r + c;
      colx[c] = j < 0 ? 0 : j >= width ? width - 1 : j;
    }
    memset(coarse, 0, (size_t)nc * 16 * sizeof(uint16_t));
    memset(fine, 0, (size_t)nc * 256 * sizeof(uint16_t));
    /* column histograms of the window rows of y = -1 (rows -1-r .. -1+r, clamped): the loop's first step moves them to y = 0 */
    for (int dy = -1 - r; dy <= -1 + r; dy++) {
      const uint8_t *s = D2PC_ROW(dy);
      for (int c = 0; c < nc; c++) {
        const uint8_t v = s[colx[c]];
        coarse[(size_t)c * 16 + (v >> 4)]++;
        fine[(size_t)c * 256 + v]++;
      }
    }
    for (int y = 0; y < height; y++) {
      const uint8_t *out_row = D2PC_ROW(y - r - 1), *in_row = D2PC_ROW(y + r);
      for (int c = 0; c < nc; c++) {
        const uint8_t vo = out_row[colx[c]], vi = in_row[colx[c]];
        coarse[(size_t)c * 16 + (vo >> 4)]--;
        fine[(size_t)c * 256 + vo]--;
        coarse[(size_t)c * 16 + (vi >> 4)]++;
        fine[(size_t)c * 256 + vi]++;
      }
      uint16_t hc[16], hf[16][16];
      int luc[16];  /* first local column NOT yet in fine segment k; the segment holds columns luc-ksize .. luc-1 */
      memset(hc, 0, sizeof hc);
      /* the kernel's coarse histogram for output 0 of the stripe minus its last column: local columns 0 .. 2r-1 */
      for (int c = 0; c < 2 * r; c++)
        for (int k = 0; k < 16; k++) hc[k] = (uint16_t)(hc[k] + coarse[(size_t)c * 16 + k]);
      for (int k = 0; k < 16; k++) luc[k] = -ksize;  /* "holds nothing" */
      uint8_t *d = dst + (size_t)y * dst_stride_bytes + x0;
      for (int x = 0; x < sw; x++) {  /* output x of the stripe: local columns x .. x + 2r */
        const uint16_t *cin = coarse + (size_t)(x + 2 * r) * 16;
        for (int k = 0; k < 16; k++) hc[k] = (uint16_t)(hc[k] + cin[k]);
        int acc = 0, k = 0;
        for (; k < 16; k++) {
          if (acc + hc[k] > half) break;
          acc += hc[k];
        }
        uint16_t *f = hf[k];
        if (luc[k] <= x) {  /* nothing of what it holds is still inside: rebuild */
          memset(f, 0, 16 * sizeof(uint16_t));
          for (int c = x; c <= x + 2 * r; c++) {
            const uint16_t *p = fine + (size_t)c * 256 + 16 * k;
            for (int b = 0; b < 16; b++) f[b] = (uint16_t)(f[b] + p[b]);
          }
        } else {
          for (int c = luc[k]; c <= x + 2 * r; c++) {
            const uint16_t *pi = fine + (size_t)c * 256 + 16 * k, *po = fine + (size_t)(c - ksize) * 256 + 16 * k;
            for (int b = 0; b < 16; b++) f[b] = (uint16_t)(f[b] + pi[b] - po[b]);
          }
        }
        luc[k] = x + 2 * r + 1;
        int b = 0;
        for (; b < 16; b++) {
          acc += f[b];
          if (acc > half) break;
        }
        d[x] = (uint8_t)(16 * k + b);
        const uint16_t *cout = coarse + (size_t)x * 16;  /* local column x leaves */
        for (int k2 = 0; k2 < 16; k2++) hc[k2] = (uint16_t)(hc[k2] - cout[k2]);
      }
    }
  }
#undef D2PC_ROW
  free(coarse);
  free(fine);
  free(colx);
}

int d2pc_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
