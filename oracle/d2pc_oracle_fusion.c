/*
 * d2pc_oracle_fusion.c -- CPU restatement of the depth-map fusion inner loop
 * (SURVEY.md section 8(f) #4).  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED --
 * see d2pc_oracle.h.  Anchors in the reference:
 *
 *   src/depth_map_fusion.cpp:113-123  per-pixel fused distance + combined score
 *   src/depth_map_fusion.cpp:124      cv::medianBlur(image, image, 3)
 *   src/depth_map_fusion.cpp:130      cropMat(image, 0, 40, 30, 10)
 *   src/depth_map_fusion.cpp:150-160  getFusedDistance (gradFilter is the one wired in)
 *   src/depth_map_fusion.cpp:162-235  the nine candidate fusion rules
 *   src/depth_map_fusion.cpp:237-273  cropMat / cropToSquare / rotateMat
 *
 * Every rule is evaluated with the reference's own types (int, float, double)
 * so that the implicit conversions land where they do there.
 */
#include "d2pc_oracle.h"

#include <stdlib.h>
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }

/* cpp:162-167.  The weights are ints initialised from doubles, so they are
 * 1 for score 0 and 0 otherwise.  Both zero divides by zero in the reference
 * (undefined; SIGFPE on x86): that case is DEFINED here as 0. */
static int rule_weighted_average(int d1, int d2, int s1, int s2) {
  const int w1 = (int)dmax(0.01, 1.0 - 0.5 * s1), w2 = (int)dmax(0.01, 1.0 - 0.5 * s2);
  if (w1 + w2 == 0) return 0;
  return (w1 * d1 + w2 * d2) / (w1 + w2);
}
/* cpp:219-235.  grad1/grad2 are accepted and ignored by the reference. */
static int rule_grad_filter(int d1, int d2, int s1, int s2) {
  const int thres = 100, too_close = 230;
  const float ratio = (float)d1 / (float)d2; /* inf or NaN when d2 == 0 */
  if (s1 < s2 && s1 < thres && d1 < too_close) return d1;
  if (s2 < s1 && s2 < thres && d2 < too_close) return d2;
  if (0.8 < ratio && ratio < 1.25 && s1 < 1.25 * thres && s2 < 1.25 * thres)
    return (int)((float)(d1 + d2) / 2.0);
  return 0;
}

int d2pc_oracle_fuse_pixel(int rule, int d1, int d2, int s1, int s2, int g1, int g2) {
  (void)g1;
  (void)g2;
  switch (rule) {
    case D2PC_ORACLE_FUSE_WEIGHTED_AVERAGE: return rule_weighted_average(d1, d2, s1, s2);
    case D2PC_ORACLE_FUSE_MAX_DIST: return imin(d1, d2);                                        /* cpp:169-172 */
    case D2PC_ORACLE_FUSE_MAX_DIST_UNLESS_BLACK: return (d1 == 0 || d2 == 0) ? imax(d1, d2) : imin(d1, d2); /* :174-180 */
    case D2PC_ORACLE_FUSE_BETTER_SCORE: return s1 < s2 ? d1 : d2;                               /* cpp:182-188 */
    case D2PC_ORACLE_FUSE_ONLY_GOOD_1: return s2 < 50 ? d2 : 0;                                 /* cpp:190-196 */
    case D2PC_ORACLE_FUSE_ONLY_GOOD_AVG: return (s1 < 100 && s2 < 100) ? (d1 + d2) / 2 : 0;     /* cpp:198-203 */
    case D2PC_ORACLE_FUSE_OVERLAP: return (s1 < s2 && s1 < 20) ? 150 : (s2 < s1 && s2 < 20) ? 255 : 0; /* :205-213 */
    case D2PC_ORACLE_FUSE_BLACK_TO_WHITE: return 255 - s1;                                      /* cpp:215-217 */
    case D2PC_ORACLE_FUSE_GRAD_FILTER: return rule_grad_filter(d1, d2, s1, s2);
    default: return -1;
  }
}

/* cpp:113-130 for one set of n-by-n (here: w-by-h) planes.  `combined`
 * (nullable) is w x h; `fused` is (w-l-r) x (h-t-b).  In the reference
 * cropped_score_combined_ shares its buffer with cropped_score_1_ (cpp:113),
 * which is read at (i,j) before it is written at (i,j): identical to writing
 * a separate plane, which is what this function does. */
int d2pc_oracle_fuse(const uint8_t *const planes[6], const size_t pitch[6], int w, int h, int rule, int crop_left,
                     int crop_right, int crop_top, int crop_bottom, uint8_t *fused, size_t fused_pitch,
                     uint8_t *combined, size_t combined_pitch) {
  const int ow = w - crop_left - crop_right, oh = h - crop_top - crop_bottom;
  if (w <= 0 || h <= 0 || ow < 0 || oh < 0 || crop_left < 0 || crop_right < 0 || crop_top < 0 || crop_bottom < 0)
    return -1;
  uint8_t *sel = (uint8_t *)malloc((size_t)w * h), *med = (uint8_t *)malloc((size_t)w * h);
  if (!sel || !med) { free(sel); free(med); return -2; }
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      int v[6];
      for (int p = 0; p < 6; p++) v[p] = planes[p][(size_t)i * pitch[p] + j];
      sel[(size_t)i * w + j] = (uint8_t)d2pc_oracle_fuse_pixel(rule, v[0], v[1], v[2], v[3], v[4], v[5]); /* :117 */
      if (combined) combined[(size_t)i * combined_pitch + j] = (uint8_t)imin(v[4], v[5]);                  /* :118-121 */
    }
  d2pc_oracle_median_u8(sel, (size_t)w, med, (size_t)w, w, h, 3);                                          /* :124 */
  for (int i = 0; i < oh; i++)                                                                             /* :130 */
    memcpy(fused + (size_t)i * fused_pitch, med + (size_t)(i + crop_top) * w + crop_left, (size_t)ow);
  free(sel);
  free(med);
  return 0;
}

/* cpp:247-265.  rect = {x, y, n}.  The side length uses the MEMBER offset_y_
 * where the argument offset_y was presumably meant (cpp:253) -- both are
 * parameters here so that the quirk can be reproduced (DisparityCb2 passes
 * -offset_y_ while the member stays +offset_y_; abs() hides the difference). */
void d2pc_oracle_crop_to_square(int cols, int rows, int offset_x, int offset_y, int member_offset_y, int rect[3]) {
  const int num_cols = cols - abs(offset_x), num_rows = rows - abs(offset_y);
  rect[2] = imin(cols, rows) - imax(abs(offset_x), abs(member_offset_y));
  if (num_cols < num_rows) {
    rect[0] = imax(0, offset_x);
    rect[1] = imax(0, offset_y + (num_rows - num_cols) / 2);
  } else {
    rect[0] = imax(0, offset_x + (num_cols - num_rows) / 2);
    rect[1] = imax(0, offset_y);
  }
}

/* cpp:268-273.  transpose, then flip around the vertical axis = 90 degrees
 * clockwise: dst has `cols` rows of `rows` pixels, dst(i,j) = src(rows-1-j, i). */
void d2pc_oracle_rotate_cw(const uint8_t *src, size_t src_pitch, int cols, int rows, uint8_t *dst, size_t dst_pitch) {
  for (int i = 0; i < cols; i++)
    for (int j = 0; j < rows; j++) dst[(size_t)i * dst_pitch + j] = src[(size_t)(rows - 1 - j) * src_pitch + i];
}
