#!/usr/bin/env python3
"""Known answers for the depth-map fusion inner loop.

NOT outputs of the reference (it cannot be built here and ships no fixtures): an independent
restatement of src/depth_map_fusion.cpp:113-130,219-235 in exact arithmetic --

  * gradFilter's `float(dist1) / float(dist2)` is the exact rational d1/d2 rounded ONCE to float32
    (ties-to-even, `round_fraction_to_f32`), then compared AS A RATIONAL with the exact binary values
    of the double literals 0.8 and 1.25 -- no floating-point compare of the platform is involved;
  * the 3x3 median with replicated borders is numpy's sort of the nine samples;
  * the crop is slicing.

Run:  python tests/golden/make_fusion_golden.py   (pure python + numpy, ~10 s) -> fusion_rules.npz
"""
import os
from fractions import Fraction

import numpy as np

from make_golden import round_fraction_to_f32

HERE = os.path.dirname(os.path.abspath(__file__))
LO, HI = Fraction(0.8), Fraction(1.25)  # the exact values of the double literals


def ratio_ok(d1: int, d2: int) -> bool:
    if d2 == 0:
        return False  # +inf fails `< 1.25`, NaN (0/0) fails both
    q = Fraction(float(round_fraction_to_f32(Fraction(d1, d2))))
    return LO < q < HI


def grad_filter(d1, d2, s1, s2, ok):
    if s1 < s2 and s1 < 100 and d1 < 230:
        return d1
    if s2 < s1 and s2 < 100 and d2 < 230:
        return d2
    if ok and 4 * s1 < 5 * 100 and 4 * s2 < 5 * 100:  # score < 1.25 * thres, in integers
        return (d1 + d2) // 2                          # float(d1 + d2) / 2.0 truncated (values are exact)
    return 0


def median3_replicate(img):
    p = np.pad(img, 1, mode="edge")
    h, w = img.shape
    win = np.stack([p[i:i + h, j:j + w] for i in range(3) for j in range(3)], axis=-1)
    return np.sort(win, axis=-1)[..., 4]


def main():
    ok = np.array([[ratio_ok(a, b) for b in range(256)] for a in range(256)])
    arrays = {"ratio_ok": ok}
    score_pairs = [(110, 110), (99, 99), (124, 124), (125, 110), (110, 125), (10, 20), (20, 10), (100, 99), (99, 100),
                   (0, 0), (255, 255)]
    arrays["score_pairs"] = np.array(score_pairs, dtype=np.int32)
    for s1, s2 in score_pairs:
        arrays[f"grad_filter__{s1}_{s2}"] = np.array(
            [[grad_filter(a, b, s1, s2, bool(ok[a, b])) for b in range(256)] for a in range(256)], dtype=np.uint8)
    # one small image through rule + combined + median + crop (the reference's crop 0/40/30/10)
    rng = np.random.default_rng(20161103)
    h, w = 57, 71
    planes = [rng.integers(0, 256, size=(h, w)).astype(np.uint8) for _ in range(6)]
    planes[2] = rng.choice(np.array([0, 50, 99, 100, 124, 125, 200], dtype=np.uint8), size=(h, w))
    planes[3] = rng.choice(np.array([0, 50, 99, 100, 124, 125, 200], dtype=np.uint8), size=(h, w))
    sel = np.array([[grad_filter(int(planes[0][i, j]), int(planes[1][i, j]), int(planes[2][i, j]), int(planes[3][i, j]),
                                 bool(ok[planes[0][i, j], planes[1][i, j]])) for j in range(w)] for i in range(h)],
                   dtype=np.uint8)
    med = median3_replicate(sel)
    for k, p in enumerate(planes):
        arrays[f"image__plane{k}"] = p
    arrays["image__fused"] = med[30:h - 10, 0:w - 40]
    arrays["image__combined"] = np.minimum(planes[4], planes[5])
    np.savez_compressed(os.path.join(HERE, "fusion_rules.npz"), **arrays)
    print("ratio passes for", int(ok.sum()), "of 65536 pairs; 4/5 exact:", bool(ok[80, 100]), " 5/4 exact:", bool(ok[100, 80]))


if __name__ == "__main__":
    main()
