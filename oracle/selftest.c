/*
 * selftest.c -- drives every oracle entry point on small inputs; built with
 * -fsanitize=address,undefined by `make -C oracle sanitize` (SURVEY.md section 5:
 * sanitizers on the CPU code; GPU sanitizers are not available on this pool).
 * TEST INFRASTRUCTURE ONLY.  Exit code 0 = every call stayed in bounds and the
 * cheap invariants hold.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "d2pc_oracle.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
  double q[16];
  d2pc_oracle_make_q(714.24, 713.5, 376.0, 240.0, 0.09, 752, 480, q);
  CHECK(q[0] == 1.0 && q[11] == 713.5 && signbit(q[15]));
  const int sizes[][3] = {{97, 83, 40}, {81, 81, 40}, {80, 200, 40}, {1, 1, 0}, {33, 17, 3}, {64, 64, 0}};
  for (unsigned s = 0; s < sizeof sizes / sizeof sizes[0]; s++) {
    const int w = sizes[s][0], h = sizes[s][1], b = sizes[s][2];
    const size_t n = (size_t)w * h;
    float *f = malloc(n * sizeof *f);
    uint8_t *u8 = malloc(n), *m8 = malloc(n), *m8b = malloc(n);
    uint16_t *u16 = malloc(n * sizeof *u16);
    for (size_t i = 0; i < n; i++) {
      u8[i] = (uint8_t)(i * 37u + s);
      u16[i] = (uint16_t)(i * 2654435761u >> 7);
      f[i] = (i % 7 == 0) ? 0.f : (float)u8[i] * 0.125f + 0.5f;
    }
    const int rw = w - 2 * b > 0 ? w - 2 * b : 0, rh = h - 2 * b > 0 ? h - 2 * b : 0;
    const size_t cap = (size_t)rw * rh;
    float *p = malloc((cap + 1) * 16), *pc = malloc((cap + 1) * 16);
    uint32_t *idx = malloc((cap + 1) * 4);
    for (int form = 0; form < 2; form++) {
      CHECK(d2pc_oracle_reproject(f, D2PC_ORACLE_F32, 1.f, w, h, (size_t)w * 4, q, b, form, 1 + form, p) == cap);
      CHECK(d2pc_oracle_reproject(u8, D2PC_ORACLE_U8, 0.125f, w, h, (size_t)w, q, b, form, 1, p) == cap);
      CHECK(d2pc_oracle_reproject(u16, D2PC_ORACLE_U16, 1.f / 64, w, h, (size_t)w * 2, q, b, form, 1, p) == cap);
      const size_t k = d2pc_oracle_reproject_compact(f, D2PC_ORACLE_F32, 1.f, w, h, (size_t)w * 4, q, b, form,
                                                     -INFINITY, pc, idx);
      CHECK(k <= cap);
      for (size_t i = 1; i < k; i++) CHECK(idx[i] > idx[i - 1]);
    }
    d2pc_oracle_mono16_to_mono8(u16, (size_t)w * 2, m8, (size_t)w, w, h);
    for (int ks = 1; ks <= 11; ks += 2) {
      d2pc_oracle_median_u8(m8, (size_t)w, m8b, (size_t)w, w, h, ks);
      uint8_t *m8c = malloc(n);
      d2pc_oracle_median_u8_fast(m8, (size_t)w, m8c, (size_t)w, w, h, ks);
      CHECK(memcmp(m8c, m8b, n) == 0);
      free(m8c);
    }
    /* fusion: rules, median, crop, rotate, crop-to-square */
    const uint8_t *planes[6] = {u8, m8, m8b, u8, m8, m8b};
    const size_t pitch[6] = {(size_t)w, (size_t)w, (size_t)w, (size_t)w, (size_t)w, (size_t)w};
    uint8_t *fused = malloc(n + 1), *comb = malloc(n + 1), *rot = malloc(n + 1);
    for (int rule = 0; rule < 9; rule++) {
      const int l = w > 4 ? 1 : 0, r = w > 4 ? 2 : 0, t = h > 4 ? 1 : 0, bt = h > 4 ? 1 : 0;
      CHECK(d2pc_oracle_fuse(planes, pitch, w, h, rule, l, r, t, bt, fused, (size_t)(w - l - r), comb, (size_t)w) == 0);
    }
    CHECK(d2pc_oracle_fuse(planes, pitch, w, h, 8, w, 1, 0, 0, fused, 1, comb, (size_t)w) < 0); /* crop too large */
    d2pc_oracle_rotate_cw(u8, (size_t)w, w, h, rot, (size_t)h);
    CHECK(rot[0] == u8[(size_t)(h - 1) * w]);
    int rect[3];
    d2pc_oracle_crop_to_square(w, h, -1, 1, 1, rect);
    CHECK(rect[2] <= (w < h ? w : h));
    free(f), free(u8), free(m8), free(m8b), free(u16), free(p), free(pc), free(idx), free(fused), free(comb), free(rot);
  }
  for (int d1 = 0; d1 < 256; d1 += 5)
    for (int d2 = 0; d2 < 256; d2 += 3)
      for (int rule = 0; rule < 9; rule++) {
        const int v = d2pc_oracle_fuse_pixel(rule, d1, d2, (d1 * 7) & 255, (d2 * 11) & 255, 0, 0);
        CHECK(v >= 0 && v <= 255);
      }
  CHECK(d2pc_oracle_fuse_pixel(99, 1, 2, 3, 4, 5, 6) == -1);
  puts("oracle selftest ok");
  return 0;
}
