"""CPU tests: the oracle (oracle/d2pc_oracle.c) against the exact-rational
known-answer vectors in tests/golden/ and against closed forms.

The reference has no tests; these are the T0/T1 tiers of SURVEY.md section 4.
"""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import assert_points_close, ulp_distance, synth_disparity


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "reproject_exact.npz"))


def _case(golden, name):
    q = golden[name + "__q"]
    border = int(golden[name + "__border"])
    exp = golden[name + "__expected_bits"].view(np.float32)
    return q, border, exp


def test_make_q_default_matches_survey_constants():
    # SURVEY.md section 8 row a9 / hpp:66-71,84-104
    q = oracle.make_q()
    assert q[0] == 1 and q[5] == 1 and q[1] == 0 and q[2] == 0
    assert abs(q[3] - (-375.999481966846)) < 1e-9
    assert q[7] == -240.0
    assert q[11] == 713.5
    assert abs(q[14] - 1 / 0.09) < 1e-12
    assert q[15] == 0.0 and np.signbit(q[15]), "Q[3][3] must be negative zero"
    assert q[10] == 0 and q[12] == 0 and q[13] == 0


@pytest.mark.parametrize("form", [oracle.FORM_CV24, oracle.FORM_CV4])
@pytest.mark.parametrize("name", ["A_default_q_k8", "B_dense_q", "C_extremes", "E_flt_max_sentinel",
                                  "E_flt_max_sentinel_dense_q"])
def test_oracle_vs_exact_rational(golden, name, form):
    q, border, exp = _case(golden, name)
    disp = golden[name + "__disp"]
    got = oracle.reproject(disp, q, border=border, form=form)
    if name.startswith("E_"):  # d == FLT_MAX => Z = 10000 exactly, in both forms; its float neighbour below: not
        sent = disp[border:disp.shape[0] - border, border:disp.shape[1] - border].ravel() == np.finfo(np.float32).max
        assert sent.any() and np.all(got[sent, 2] == np.float32(10000.0)) and not np.any(got[~sent, 2] == 10000.0)
        if form == oracle.FORM_CV4 and name.endswith("dense_q"):
            # the 3.x/4.x form casts the NUMERATORS to float before dividing: with a dense Q and d ~ 3.4e38
            # they overflow to inf there, which the exact quotient (and the 2.4 form) does not -- a real
            # difference between the two published forms, outside any calibration's range
            return
    # a double evaluation + one cast is within 1 ulp of the correctly rounded
    # exact value; the OpenCV-4 form casts twice => 2 ulp
    assert_points_close(got, exp, max_ulp=1 if form == oracle.FORM_CV24 else 2, what=name)
    exact_frac = (ulp_distance(got[:, :3], exp[:, :3]) == 0).mean()
    assert exact_frac > (0.99 if form == oracle.FORM_CV24 else 0.5), exact_frac


def test_oracle_u8_decode_centre_rows(golden):
    name = "D_centre_rows_u8"
    q, border, exp = _case(golden, name)
    r0, r1 = golden[name + "__rows"]
    rows = golden[name + "__raw_u8_rows"]
    raw = np.ones((r1, rows.shape[1]), dtype=np.uint8)
    raw[r0:r1] = rows
    got = oracle.reproject(raw, q, border=0, scale=0.125)
    w = raw.shape[1]
    sub = got[r0 * w : r1 * w]
    assert_points_close(sub, exp, max_ulp=1, what=name)
    # the same frame pre-decoded on the host, as the reference does (cpp:60-61)
    got_f = oracle.reproject(raw.astype(np.float32) * np.float32(0.125), q, border=0)
    assert np.array_equal(got.view(np.uint32), got_f.view(np.uint32))
    # Y is exactly +0 on row v = 240 (cy' = 240)
    y240 = got[240 * w : 241 * w, 1]
    assert np.all(y240 == 0) and not np.signbit(y240).any()


def test_default_q_closed_form():
    # Z = f'*b/d = 64.215/d, X = (u-cx')*b/d, Y = (v-240)*b/d
    q = oracle.make_q()
    disp = synth_disparity(2, 0, 752, 480, "k8")
    pts = oracle.reproject(disp, q, border=40)
    assert pts.shape == (672 * 400, 4)
    v, u = np.mgrid[40:440, 40:712]
    d = disp[40:440, 40:712].astype(np.float64)
    z = 713.5 * 0.09 / d
    x = (u - 375.999481966846) * 0.09 / d
    y = (v - 240.0) * 0.09 / d
    ref = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    err = np.abs(pts[:, :3] - ref) / np.maximum(np.abs(ref), 1e-30)
    assert err[ref != 0].max() < 5e-7
    assert np.all(pts[:, 3].view(np.uint32) == 0x3F800000)


def test_zero_disparity_inf_nan_pattern():
    # d = 0 with the default Q => W = +0, iW = +inf,
    # X = sign(u-cx')*inf, Y = sign(v-240)*inf (NaN on v = 240), Z = +inf.
    q = oracle.make_q()
    disp = np.zeros((330, 752), dtype=np.float32)
    for form in (oracle.FORM_CV24, oracle.FORM_CV4):
        pts = oracle.reproject(disp, q, border=40, form=form).reshape(250, 672, 4)
        v, u = np.mgrid[40:290, 40:712]
        assert np.all(np.isposinf(pts[..., 2]))
        assert np.array_equal(np.isposinf(pts[..., 0]), u > 375.9995)
        assert np.array_equal(np.isneginf(pts[..., 0]), u < 375.9995)
        assert np.array_equal(np.isnan(pts[..., 1]), v == 240)
        assert np.array_equal(np.isposinf(pts[..., 1]), v > 240)
        assert np.array_equal(np.isneginf(pts[..., 1]), v < 240)


def test_forms_agree_within_2ulp():
    rng = np.random.default_rng(7)
    q = rng.uniform(-1, 1, 16)
    q[12:16] = [2e-4, 1e-4, 0.03, 0.7]
    disp = rng.uniform(0.5, 128, size=(97, 131)).astype(np.float32)
    a = oracle.reproject(disp, q, border=5, form=oracle.FORM_CV24)
    b = oracle.reproject(disp, q, border=5, form=oracle.FORM_CV4)
    assert_points_close(a, b, max_ulp=2)


def test_roi_geometry_and_index_map():
    # cpp:70-76: 96x96 frame, border 40 -> 16x16 points, i = (v-40)*16 + (u-40)
    q = np.eye(4).reshape(16)  # identity: point = (u, v, d)
    disp = np.arange(96 * 96, dtype=np.float32).reshape(96, 96) / 4 + 1
    pts = oracle.reproject(disp, q, border=40)
    assert pts.shape == (256, 4)
    v, u = np.mgrid[40:56, 40:56]
    assert np.array_equal(pts[:, 0], u.reshape(-1).astype(np.float32))
    assert np.array_equal(pts[:, 1], v.reshape(-1).astype(np.float32))
    assert np.array_equal(pts[:, 2], disp[40:56, 40:56].reshape(-1))
    cpts, idx = oracle.reproject_compact(disp, q, border=40)
    assert np.array_equal(cpts, pts)
    assert np.array_equal(idx, (v * 96 + u).reshape(-1).astype(np.uint32))


@pytest.mark.parametrize("w,h", [(80, 80), (80, 200), (200, 80), (79, 300), (1, 1), (81, 81)])
def test_degenerate_sizes(w, h):
    # cpp:70,72: loops are empty when a dimension is <= 2*border
    q = oracle.make_q()
    disp = np.ones((h, w), dtype=np.float32)
    pts = oracle.reproject(disp, q, border=40)
    assert pts.shape[0] == max(w - 80, 0) * max(h - 80, 0)


def test_row_stride_is_honoured():
    q = oracle.make_q()
    big = synth_disparity(2, 1, 200, 100, "k8")
    view = big[:, :150]  # 150 px rows with a 200 px stride
    a = oracle.reproject(view, q, border=10)
    b = oracle.reproject(np.ascontiguousarray(view), q, border=10)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_threads_are_bit_identical():
    q = oracle.make_q()
    disp = synth_disparity(4, 0, 640, 480, "holes")
    a = oracle.reproject(disp, q, border=40, threads=1)
    b = oracle.reproject(disp, q, border=40, threads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_compact_is_filter_of_parity():
    q = oracle.make_q()
    disp = synth_disparity(3, 0, 320, 240, "holes")
    full = oracle.reproject(disp, q, border=40)
    cpts, idx = oracle.reproject_compact(disp, q, border=40)
    keep = np.isfinite(full[:, :3]).all(axis=1)
    assert 0.6 < keep.mean() < 0.8
    assert np.array_equal(cpts.view(np.uint32), full[keep].view(np.uint32))
    v, u = np.mgrid[40:200, 40:280]
    assert np.array_equal(idx, (v * 320 + u).reshape(-1)[keep].astype(np.uint32))
    assert np.all(np.diff(idx.astype(np.int64)) > 0)
    # min_disparity predicate: !(d <= t)
    cpts2, idx2 = oracle.reproject_compact(disp, q, border=40, min_disparity=64.0)
    dsel = disp.reshape(-1)[idx2]
    assert np.all(dsel > 64.0)
    assert idx2.size == np.count_nonzero(disp[40:200, 40:280] > 64.0)


def test_pointcloud2_blob_82x82(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "pointcloud2_82x82.json")))
    q = np.array([float.fromhex(h) for h in meta["q_hex"]])
    disp = np.full((82, 82), 4.0, dtype=np.float32)
    disp[40:42, 40:42] = np.array(meta["roi_disparities"], dtype=np.float32).reshape(2, 2)
    pts = oracle.reproject(disp, q, border=40)
    assert pts.nbytes == meta["row_step"] == meta["point_step"] * meta["width"]
    want = np.frombuffer(bytes.fromhex(meta["data_hex"]), dtype=np.float32).reshape(-1, 4)
    assert_points_close(pts, want, max_ulp=1)
    assert pts.tobytes()[12:16] == bytes.fromhex("0000803f")


def test_mono16_to_mono8_exact_on_k257():
    # values k*257 map back to k exactly
    img = synth_disparity(1, 0, 640, 480, "mono16")
    out = oracle.mono16_to_mono8(img)
    assert np.array_equal(out, (img // 257).astype(np.uint8))
    # general values: round-half-even of v*255/65535 evaluated in float
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, size=(50, 70)).astype(np.uint16)
    ref = np.rint(img.astype(np.float32) * np.float32(255.0 / 65535.0)).clip(0, 255).astype(np.uint8)
    assert np.array_equal(oracle.mono16_to_mono8(img), ref)


def test_median_u8_against_numpy():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(40, 57)).astype(np.uint8)
    for k in (3, 11):
        r = k // 2
        pad = np.pad(img, r, mode="edge")
        win = np.lib.stride_tricks.sliding_window_view(pad, (k, k)).reshape(40, 57, k * k)
        ref = np.sort(win, axis=-1)[..., (k * k) // 2]
        assert np.array_equal(oracle.median_u8(img, k), ref)


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
def test_median_u8_against_scipy(k):
    """An implementation nobody here wrote: scipy.ndimage.median_filter with replicated borders ('nearest' =
    cv::BORDER_REPLICATE, what cv::medianBlur uses) on images smaller than, equal to and larger than the window,
    with ties and constant regions."""
    from scipy import ndimage
    rng = np.random.default_rng(k)
    for h, w in ((1, 1), (2, 9), (k, k), (k - 1, 3 * k), (64, 97), (131, 45)):
        img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
        if (h * w) % 2:
            img = (img // 64 * 85).astype(np.uint8)
        assert np.array_equal(oracle.median_u8(img, k), ndimage.median_filter(img, size=k, mode="nearest")), (h, w, k)


@pytest.mark.parametrize("k", [1, 3, 5, 7, 9, 11])
def test_fast_median_is_pinned_to_the_checker(k):
    """d2pc_oracle_median_u8_fast (Perreault & Hebert's constant-time sliding histogram, bench.py's CPU column of the
    callback body, cpp:55-57) against d2pc_oracle_median_u8 (the per-pixel histogram walk, the CHECKER) byte for byte: images
    smaller than, equal to and larger than the window, strided rows, ties, constant regions, the extremes 0 and 255."""
    rng = np.random.default_rng(100 + k)
    for h, w in ((1, 1), (1, 40), (33, 1), (2, 9), (k, k), (max(k - 1, 1), 3 * k), (64, 97), (131, 45), (40, 300)):
        for kind in ("uniform", "few_levels", "extremes", "ramp"):
            if kind == "uniform":
                img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
            elif kind == "few_levels":
                img = (rng.integers(0, 4, size=(h, w)) * 85).astype(np.uint8)
            elif kind == "extremes":
                img = np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8)
            else:
                img = ((np.arange(h)[:, None] * 7 + np.arange(w)[None, :] * 3) % 256).astype(np.uint8)
            assert np.array_equal(oracle.median_u8_fast(img, k), oracle.median_u8(img, k)), (h, w, k, kind)
    # a view with a row stride larger than its width
    big = rng.integers(0, 256, size=(50, 128)).astype(np.uint8)
    view = big[:, 5:90]
    assert np.array_equal(oracle.median_u8_fast(view, k), oracle.median_u8(np.ascontiguousarray(view), k))


def test_fast_median_full_frame_752x480():
    """the reference's native frame (hpp:102-103), 11 x 11 (cpp:57): the two oracle medians agree on every byte"""
    img = np.random.default_rng(7).integers(0, 256, size=(480, 752)).astype(np.uint8)
    img[100:200, 300:500] = 0          # a no-match region
    assert np.array_equal(oracle.median_u8_fast(img, 11), oracle.median_u8(img, 11))


# ---- depth-map fusion inner loop ---------------------------
def test_fusion_rules_hand_derived():
    """Answers worked out by hand from the text of src/depth_map_fusion.cpp:162-235."""
    G = oracle.FUSE_GRAD_FILTER
    assert oracle.fuse_pixel(G, 50, 60, 10, 20) == 50          # score1 better, < 100, dist1 < 230
    assert oracle.fuse_pixel(G, 50, 60, 20, 10) == 60          # score2 better
    assert oracle.fuse_pixel(G, 230, 60, 10, 20) == 0          # dist1 too close, ratio 3.8 -> 0
    assert oracle.fuse_pixel(G, 229, 200, 10, 20) == 229
    assert oracle.fuse_pixel(G, 100, 100, 110, 110) == 100     # equal scores -> third branch, averaged
    assert oracle.fuse_pixel(G, 100, 101, 124, 124) == 100     # float(201)/2.0 = 100.5 -> int 100
    assert oracle.fuse_pixel(G, 100, 101, 125, 110) == 0       # score1 !< 125
    assert oracle.fuse_pixel(G, 80, 100, 110, 110) == 90       # float(0.8f) > 0.8 (double): the exact ratio 4/5 passes
    assert oracle.fuse_pixel(G, 100, 80, 110, 110) == 0        # 1.25 is exact in float: 5/4 fails
    assert oracle.fuse_pixel(G, 0, 0, 110, 110) == 0           # NaN compares false
    assert oracle.fuse_pixel(G, 7, 0, 110, 110) == 0           # +inf fails the upper bound
    assert oracle.fuse_pixel(G, 100, 90, 99, 99) == 95         # equal scores below thres still average
    W = oracle.FUSE_WEIGHTED_AVERAGE                           # int weights: 1 iff score == 0
    assert oracle.fuse_pixel(W, 10, 21, 0, 0) == 15
    assert oracle.fuse_pixel(W, 10, 21, 0, 1) == 10
    assert oracle.fuse_pixel(W, 10, 21, 5, 0) == 21
    assert oracle.fuse_pixel(W, 10, 21, 1, 1) == 0             # 0/0 in the reference: defined as 0
    assert oracle.fuse_pixel(oracle.FUSE_MAX_DIST, 10, 21, 0, 0) == 10
    assert oracle.fuse_pixel(oracle.FUSE_MAX_DIST_UNLESS_BLACK, 0, 21, 0, 0) == 21
    assert oracle.fuse_pixel(oracle.FUSE_MAX_DIST_UNLESS_BLACK, 30, 21, 0, 0) == 21
    assert oracle.fuse_pixel(oracle.FUSE_BETTER_SCORE, 1, 2, 5, 5) == 2
    assert oracle.fuse_pixel(oracle.FUSE_BETTER_SCORE, 1, 2, 4, 5) == 1
    assert oracle.fuse_pixel(oracle.FUSE_ONLY_GOOD_1, 1, 2, 0, 49) == 2
    assert oracle.fuse_pixel(oracle.FUSE_ONLY_GOOD_1, 1, 2, 0, 50) == 0
    assert oracle.fuse_pixel(oracle.FUSE_ONLY_GOOD_AVG, 255, 254, 99, 99) == 254
    assert oracle.fuse_pixel(oracle.FUSE_ONLY_GOOD_AVG, 255, 254, 100, 99) == 0
    assert oracle.fuse_pixel(oracle.FUSE_OVERLAP, 1, 2, 19, 20) == 150
    assert oracle.fuse_pixel(oracle.FUSE_OVERLAP, 1, 2, 20, 19) == 255
    assert oracle.fuse_pixel(oracle.FUSE_OVERLAP, 1, 2, 19, 19) == 0
    assert oracle.fuse_pixel(oracle.FUSE_BLACK_TO_WHITE, 1, 2, 55, 0) == 200
    assert oracle.fuse_pixel(99, 1, 2, 3, 4) == -1


def test_fusion_grad_filter_integer_form_is_exhaustively_equal():
    """The device evaluates cpp:224-232 in integers; check that form against the literal one for all inputs
    the ratio test can see (every distance pair, scores either side of each threshold)."""
    d1, d2 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    for s1, s2 in ((110, 110), (99, 99), (124, 125), (10, 10)):
        lit = np.array([[oracle.fuse_pixel(oracle.FUSE_GRAD_FILTER, a, b, s1, s2) for b in range(256)]
                        for a in range(256)])
        ok = (5 * d1 >= 4 * d2) & (4 * d1 < 5 * d2) & (s1 < 125) & (s2 < 125)
        assert np.array_equal(lit, np.where(ok, (d1 + d2) >> 1, 0))


def test_fusion_image_path_median_and_crop():
    # a constant pair of planes: every rule output is constant, the median keeps it, the crop only sizes it
    planes = [np.full((50, 60), v, dtype=np.uint8) for v in (100, 101, 124, 124, 7, 9)]
    fused, comb = oracle.fuse(planes)
    assert fused.shape == (50 - 30 - 10, 60 - 40) and (fused == 100).all()
    assert comb.shape == (50, 60) and (comb == 7).all()
    # a single outlier is removed by the 3x3 median, a 2x2 block at the corner survives by replication
    d = np.full((40, 45), 50, dtype=np.uint8)
    d[20, 20] = 200
    d[0:2, 0:2] = 90
    z = np.zeros_like(d)
    planes = [d, d, z, z + 1, z, z]           # score1 < score2 -> dist1
    fused, _ = oracle.fuse(planes, crop=(0, 0, 0, 0), want_combined=False)
    assert fused[20, 20] == 50 and fused[0, 0] == 90 and fused[1, 1] == 50 and fused[0, 1] == 90
    want = oracle.median_u8(d, 3)
    assert np.array_equal(fused, want)
    fused_c, _ = oracle.fuse(planes, crop=(3, 4, 5, 6), want_combined=False)
    assert np.array_equal(fused_c, want[5:40 - 6, 3:45 - 4])
    with pytest.raises(ValueError):
        oracle.fuse(planes, crop=(30, 30, 0, 0))


def test_crop_to_square_and_rotate():
    # launch/depth_map_fusion.launch: offset_x -7, offset_y 15 on the 752x480 camera image
    assert oracle.crop_to_square(752, 480, -7, 15) == (133, 15, 465)        # landscape: x = -7 + (745-465)/2
    # DisparityCb2 rotates first (480 wide, 752 high) and passes the NEGATED offsets (cpp:57)
    assert oracle.crop_to_square(480, 752, 7, -15, 15) == (7, 117, 465)     # portrait: y = -15 + (737-473)/2
    assert oracle.crop_to_square(640, 480) == (80, 0, 480)
    assert oracle.crop_to_square(480, 640) == (0, 80, 480)
    a = np.arange(6, dtype=np.uint8).reshape(2, 3)
    assert np.array_equal(oracle.rotate_cw(a), np.array([[3, 0], [4, 1], [5, 2]], dtype=np.uint8))
    img = np.random.default_rng(1).integers(0, 256, size=(37, 91)).astype(np.uint8)
    assert np.array_equal(oracle.rotate_cw(img), np.rot90(img, -1))


def test_fusion_oracle_vs_exact_golden(golden_dir):
    """tests/golden/fusion_rules.npz: gradFilter evaluated with exact rationals (make_fusion_golden.py)."""
    g = np.load(os.path.join(golden_dir, "fusion_rules.npz"))
    assert bool(g["ratio_ok"][80, 100]) and not bool(g["ratio_ok"][100, 80])
    for s1, s2 in g["score_pairs"]:
        want = g[f"grad_filter__{s1}_{s2}"]
        got = np.array([[oracle.fuse_pixel(oracle.FUSE_GRAD_FILTER, a, b, int(s1), int(s2)) for b in range(256)]
                        for a in range(256)], dtype=np.uint8)
        assert np.array_equal(got, want), (s1, s2)
    planes = [np.ascontiguousarray(g[f"image__plane{k}"]) for k in range(6)]
    fused, comb = oracle.fuse(planes)
    assert np.array_equal(fused, g["image__fused"])
    assert np.array_equal(comb, g["image__combined"])


def test_oracle_under_asan_ubsan():
    """sanitizers on the CPU code.  oracle/selftest.c calls every oracle entry point on small
    and degenerate inputs in an -fsanitize=address,undefined build."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.run(["make", "-C", here, "sanitize"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               OMP_NUM_THREADS="2")
    p = subprocess.run([os.path.join(here, "_selftest_asan")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "oracle selftest ok" in p.stdout


def _numpy_cv4(disp, q, border):
    """OpenCV 3/4's reprojectImageTo3D association in numpy float64 / float32 operations, one rounding per
    operation: Vec4d h = Q * (x, y, d, 1) left to right, Vec3f p = h (cast), p /= h[3] (ia = 1./h[3], product in
    double, one cast)."""
    h, w = disp.shape
    v, u = np.mgrid[border:h - border, border:w - border]
    x, y, d = u.astype(np.float64), v.astype(np.float64), disp[border:h - border, border:w - border].astype(np.float64)
    hh = []
    with np.errstate(all="ignore"):
        for r in range(4):
            s = 0.0 + q[4 * r] * x
            s = s + q[4 * r + 1] * y
            s = s + q[4 * r + 2] * d
            hh.append(s + q[4 * r + 3])
        ia = 1.0 / hh[3]
        return np.stack([(hh[c].astype(np.float32).astype(np.float64) * ia).astype(np.float32) for c in range(3)], axis=-1)


def _numpy_cv24(disp, q, border):
    """OpenCV 2.4's loop: per row qx = q01*y + q03 ..., then qx += q00 per column (replayed from column 0),
    iW = 1./(qw + q32*d), X = (qx + q02*d)*iW, one cast."""
    h, w = disp.shape
    out = np.empty((h - 2 * border, w - 2 * border, 3), dtype=np.float32)
    with np.errstate(all="ignore"):
        for yi, y in enumerate(range(border, h - border)):
            acc = [np.float64(q[4 * r + 1]) * y + q[4 * r + 3] for r in range(4)]
            cols = []
            for r in range(4):  # the x-recurrence: a running sum of q_r0, one rounding per step
                steps = np.full(w, q[4 * r], dtype=np.float64)
                steps[0] = acc[r]
                run = np.empty(w, dtype=np.float64)
                a = steps[0]
                run[0] = a
                for xx in range(1, w):
                    a = a + q[4 * r]
                    run[xx] = a
                cols.append(run[border:w - border])
            d = disp[y, border:w - border].astype(np.float64)
            iw = 1.0 / (cols[3] + q[14] * d)
            for c, k in enumerate((2, 6, 10)):
                out[yi, :, c] = ((cols[c] + q[k] * d) * iw).astype(np.float32)
    return out


@pytest.mark.parametrize("form", [oracle.FORM_CV24, oracle.FORM_CV4])
def test_oracle_binary_equals_a_numpy_restatement_bit_for_bit(form):
    """Pins the COMPILED oracle against compiler liberties (round 3: gcc 11.4 -O3 had vectorised FORM_CV4's x/y pair
    and dropped the float cast of the numerators there -- the binary did not compute what its source says)."""
    rng = np.random.default_rng(5)
    for dense in (True, False):
        q = rng.uniform(-2, 2, 16)
        q[12:14] = rng.uniform(0, 1e-3, 2)
        q[14], q[15] = rng.uniform(0.01, 1), rng.uniform(0.1, 2)
        if not dense:
            q = oracle.make_q()
        disp = rng.uniform(0.5, 128, size=(61, 83)).astype(np.float32)
        disp[rng.random(disp.shape) < 0.2] = 0
        got = oracle.reproject(disp, q, border=3, form=form).reshape(55, 77, 4)[..., :3]
        want = (_numpy_cv4 if form == oracle.FORM_CV4 else _numpy_cv24)(disp, q, 3)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), f"form {form}, dense {dense}: {int((~same).sum())} of {same.size} values differ"
