#!/usr/bin/env python3
"""Generate the known-answer vectors under tests/golden/.

The reference (PX4/disparity_to_point_cloud) has no tests, fixtures or golden
data, and neither it nor its OpenCV/PCL dependencies can be built or imported
here, so these vectors are NOT outputs of the reference.  They are the exact
mathematical definition of the path

    [X Y Z W]^T = Q . [u v d 1]^T ;  point = (X/W, Y/W, Z/W)        (cpp:63-64)
    ROI = border-inset, row-major, 16-byte {x,y,z,1.0f} records        (cpp:70-85)

evaluated in exact rational arithmetic (python `fractions`) on the exact
binary values of Q (float64) and d (float32), then rounded ONCE to the nearest
float32 (ties-to-even).  Any double-precision implementation (OpenCV 2.4's or
4.x's loop, the oracle, the HIP kernel) must agree with them to <= 1-2 float32
ulp; that is what tests/test_oracle.py and the GPU parity tests assert.

Run:  python tests/golden/make_golden.py       (pure python + numpy, a few seconds)
Deterministic: fixed seeds, no dependence on the oracle or the product.
"""
import json
import os
import struct
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLT_MAX = np.float32(3.4028234663852886e38)


def f32_bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def round_fraction_to_f32(fr: Fraction) -> np.float32:
    """Correctly rounded (nearest, ties-to-even) float32 of an exact rational."""
    if fr == 0:
        return np.float32(0.0)
    # float(Fraction) is correctly rounded to float64; casting that to float32
    # can double-round, so repair against the exact value.
    c = np.float32(float(fr))
    if not np.isfinite(c):
        return c
    lo = np.nextafter(c, np.float32(-np.inf))
    hi = np.nextafter(c, np.float32(np.inf))
    best = None
    for cand in (lo, c, hi):
        if not np.isfinite(cand):
            continue
        err = abs(Fraction(float(cand)) - fr)
        key = (err, f32_bits(float(cand)) & 1)  # ties -> even mantissa
        if best is None or key < best[0]:
            best = (key, cand)
    return best[1]


def make_q_default(fx=714.24, fy=713.5, cx=376.0, cy=240.0, b=0.09, nx=752, ny=480):
    """hpp:66-71,84-104 -- closed form of stereoRectify for the reference rig
    (same formula as SURVEY.md section 8 row a9), evaluated in float64."""
    fc = fy
    hx, hy = (nx - 1) / 2.0, (ny - 1) / 2.0
    cxn = hx - fc * (hx - cx) / fx
    cyn = hy - fc * (hy - cy) / fy
    tx = -b
    return np.array(
        [1, 0, 0, -cxn, 0, 1, 0, -cyn, 0, 0, 0, fc, 0, 0, -1.0 / tx, (cxn - cxn) / tx],
        dtype=np.float64,
    )


def exact_points(q, disp, border, rows=None):
    """Exact-rational reprojection of the ROI -> (R,4) uint32 bit patterns.
    rows=(r0,r1) restricts the evaluation to image rows r0..r1-1."""
    h, w = disp.shape
    qf = [Fraction(float(v)) for v in q]
    r0, r1 = rows if rows is not None else (border, h - border)
    out = []
    for v in range(r0, r1):
        for u in range(border, w - border):
            d = Fraction(float(disp[v, u]))
            num = [qf[4 * r] * u + qf[4 * r + 1] * v + qf[4 * r + 2] * d + qf[4 * r + 3] for r in range(4)]
            assert num[3] != 0, "exact W == 0: not a finite known-answer point"
            xyz = [f32_bits(float(round_fraction_to_f32(num[r] / num[3]))) for r in range(3)]
            if disp[v, u] == FLT_MAX:  # reprojectImageTo3D: |d - minDisparity| <= FLT_EPSILON => Z = bigZ = 10000,
                xyz[2] = f32_bits(10000.0)  # minDisparity = FLT_MAX when handleMissingValues is false (cpp:64)
            out.append(xyz + [0x3F800000])
    return np.array(out, dtype=np.uint32).reshape(-1, 4)


def main():
    cases = {}
    rng = np.random.default_rng(0xD2C0)

    # Case A: default rig, reference-like quantised disparity d = k/8, k in 1..255
    # (cpp:60-61), strip that contains the principal-point column u = 376 where
    # a pure-fp32 evaluation loses 1e-3 (SURVEY.md finding 4).
    qa = make_q_default()
    da = (rng.integers(1, 256, size=(88, 752)).astype(np.float32)) / np.float32(8)
    cases["A_default_q_k8"] = dict(q=qa, disp=da, border=40)

    # Case B: dense (all 16 entries non-zero) random Q with a W row that stays
    # well away from 0, continuous disparities.
    qb = rng.uniform(-2.0, 2.0, size=16)
    qb[12:16] = [1e-4, -2e-4, 0.05, 1.0]
    db = rng.uniform(0.5, 128.0, size=(60, 100)).astype(np.float32)
    cases["B_dense_q"] = dict(q=qb.astype(np.float64), disp=db, border=8)

    # Case C: default rig, extremes of the finite float32 disparity range:
    # results overflow to +inf on the float cast (tiny d) or land in the
    # float32 subnormal range (huge d near the principal column).  Border 0.
    ext = np.array([2.0 ** -20, 0.125, 1.0, 31.875, 64.215, 1e-3, 1e3, 3.0e38, 1.17549435e-38, 255.0, 1e30],
                   dtype=np.float32)
    dc = ext[rng.integers(0, len(ext), size=(3, 752))]
    cases["C_extremes"] = dict(q=qa, disp=dc, border=0)

    # Case D: default rig at the native geometry's centre rows (v = 238..242
    # crosses cy' = 240 where Y changes sign and is exactly 0 at v = 240),
    # 8-bit input decoded with 1/8 as at cpp:60-61.
    raw = rng.integers(1, 256, size=(243, 752)).astype(np.uint8)
    dd = raw.astype(np.float32) * np.float32(0.125)
    cases["D_centre_rows_u8"] = dict(q=qa, disp=dd, border=0, rows=(238, 243), raw=raw)

    # Case E: the missing-value sentinel.  d == FLT_MAX is the one disparity reprojectImageTo3D treats
    # specially with handleMissingValues = false: Z = 10000 exactly, X and Y as computed (subnormal or 0
    # here).  Its float neighbour just below FLT_MAX must NOT get the rule.  Drawn after A-D so those
    # cases keep their values.
    below = np.nextafter(FLT_MAX, np.float32(0))
    pool = np.array([FLT_MAX, below, 1.0, 31.875, FLT_MAX, 0.125], dtype=np.float32)
    de = pool[rng.integers(0, len(pool), size=(4, 752))]
    de[0, :4] = [FLT_MAX, below, FLT_MAX, 1.0]
    cases["E_flt_max_sentinel"] = dict(q=qa, disp=de, border=0)
    cases["E_flt_max_sentinel_dense_q"] = dict(q=qb.astype(np.float64), disp=de[:, :100].copy(), border=0)

    arrays = {}
    for name, c in cases.items():
        exp = exact_points(c["q"], c["disp"], c["border"], c.get("rows"))
        arrays[name + "__q"] = c["q"]
        if "raw" not in c:
            arrays[name + "__disp"] = c["disp"]
        arrays[name + "__border"] = np.array(c["border"], dtype=np.int32)
        arrays[name + "__expected_bits"] = exp
        if "rows" in c:
            arrays[name + "__rows"] = np.array(c["rows"], dtype=np.int32)
        if "raw" in c:
            # only the evaluated rows are stored; the test pads rows above
            r0, r1 = c["rows"]
            arrays[name + "__raw_u8_rows"] = c["raw"][r0:r1]
        print(name, c["disp"].shape, "->", exp.shape[0], "points")
    np.savez_compressed(os.path.join(HERE, "reproject_exact.npz"), **arrays)

    # Byte-level PointCloud2 payload for an 82x82 frame, border 40 => R = 4
    # points (cpp:70-85; PCL PointXYZ = 16 B with pad 1.0f = 0x3F800000).
    d82 = np.full((82, 82), 4.0, dtype=np.float32)
    d82[40, 40], d82[40, 41], d82[41, 40], d82[41, 41] = 1.0, 0.125, 31.875, 8.0
    exp82 = exact_points(qa, d82, 40)
    blob = exp82.astype("<u4").tobytes()
    meta = {
        "comment": "PointCloud2 for an 82x82 frame, default Q, border 40 (4 points)",
        "q_hex": [float(v).hex() for v in qa],
        "roi_disparities": [1.0, 0.125, 31.875, 8.0],
        "width": 4,
        "height": 1,
        "point_step": 16,
        "row_step": 64,
        "is_bigendian": False,
        "is_dense": False,
        "fields": [
            {"name": "x", "offset": 0, "datatype": 7, "count": 1},
            {"name": "y", "offset": 4, "datatype": 7, "count": 1},
            {"name": "z", "offset": 8, "datatype": 7, "count": 1},
        ],
        "frame_id": "/camera_optical_frame",
        "data_hex": blob.hex(),
    }
    with open(os.path.join(HERE, "pointcloud2_82x82.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("pointcloud2_82x82.json", len(blob), "bytes")


if __name__ == "__main__":
    main()
