"""GPU parity tests: the HIP path, called through the C ABI (include/d2pc.h),
against the CPU oracle and the exact-rational golden vectors.

Bar (BASELINE.json north_star): pixel indices and point counts bit-exact;
XYZ within 1e-5 relative of the CPU loop -- asserted here far tighter, at
<= 1 float32 ulp, with identical NaN/inf classes and the pad word 0x3F800000.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import disparity_to_point_cloud_amd as d2pc
import oracle
from helpers import assert_points_close, synth_disparity, ulp_distance, variant_for

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star tolerance
MAX_ULP = 1     # what we actually hold


@pytest.fixture(scope="module")
def q_default():
    return d2pc.make_q()


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "reproject_exact.npz"))


def ctx_for(q, **kw):
    return d2pc.Context(q=q, **kw)


# ---------------------------------------------------------------- parity mode
@pytest.mark.parametrize("name", ["A_default_q_k8", "B_dense_q", "C_extremes", "E_flt_max_sentinel",
                                  "E_flt_max_sentinel_dense_q"])
def test_golden_exact_rational(golden, name):
    q = golden[name + "__q"]
    border = int(golden[name + "__border"])
    exp = golden[name + "__expected_bits"].view(np.float32)
    disp = golden[name + "__disp"]
    with ctx_for(q, border=border) as ctx:
        got = ctx.process(disp)
    with ctx_for(q, border=border, variant="exp") as ctx:
        ctx.set_test_hook("general_q_form", 1)   # round 2's fused multiply-add evaluation of a general Q (experiment build)
        got_fma = ctx.process(disp)
    if name.endswith("dense_q"):
        # a GENERAL Q is evaluated in OpenCV 3/4's association, bit for bit (oracle FORM_CV4): two casts => within
        # 2 ulp of the exact value, and its float-cast numerators overflow where d ~ FLT_MAX meets a dense Q (a real
        # property of that published form: tests/test_oracle.py)
        want4 = oracle.reproject(disp, q, border=border, form=oracle.FORM_CV4)
        assert np.array_equal(got.view(np.uint32), want4.view(np.uint32)), "general Q is not OpenCV 4's form bit for bit"
        if not name.startswith("E_"):
            assert_points_close(got, exp, max_ulp=2, rel=REL_TOL, what=name)
        assert_points_close(got_fma, exp, max_ulp=MAX_ULP, rel=REL_TOL, what=name + " (fma form)")
        return
    assert np.array_equal(got.view(np.uint32), got_fma.view(np.uint32))   # the key only touches the general path
    assert_points_close(got, exp, max_ulp=MAX_ULP, rel=REL_TOL, what=name)
    assert (ulp_distance(got[:, :3], exp[:, :3]) == 0).mean() > 0.99


def test_golden_u8_centre_rows(golden):
    name = "D_centre_rows_u8"
    q = golden[name + "__q"]
    exp = golden[name + "__expected_bits"].view(np.float32)
    r0, r1 = golden[name + "__rows"]
    rows = golden[name + "__raw_u8_rows"]
    raw = np.ones((r1, rows.shape[1]), dtype=np.uint8)
    raw[r0:r1] = rows
    w = raw.shape[1]
    with ctx_for(q, border=0) as ctx:
        got = ctx.process(raw, scale=0.125)  # fused cpp:61 decode
        got_f = ctx.process(raw.astype(np.float32) * np.float32(0.125))
    assert_points_close(got[r0 * w : r1 * w], exp, max_ulp=MAX_ULP, rel=REL_TOL, what=name)
    assert np.array_equal(got.view(np.uint32), got_f.view(np.uint32))
    y240 = got[240 * w : 241 * w, 1]
    assert np.all(y240 == 0)


def test_pointcloud2_blob_and_meta(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "pointcloud2_82x82.json")))
    q = np.array([float.fromhex(h) for h in meta["q_hex"]])
    disp = np.full((82, 82), 4.0, dtype=np.float32)
    disp[40:42, 40:42] = np.array(meta["roi_disparities"], dtype=np.float32).reshape(2, 2)
    with ctx_for(q) as ctx:
        pts = ctx.process(disp)
        m = ctx.cloud_meta(len(pts))
    want = np.frombuffer(bytes.fromhex(meta["data_hex"]), dtype=np.float32).reshape(-1, 4)
    assert_points_close(pts, want, max_ulp=MAX_ULP)
    assert (m.height, m.width, m.point_step, m.row_step) == (1, 4, 16, 64)
    assert (m.is_bigendian, m.is_dense, m.n_fields) == (0, 0, 3)
    for f, ref in zip(m.fields, meta["fields"]):
        assert (f.name.decode(), f.offset, f.datatype, f.count) == (ref["name"], ref["offset"], ref["datatype"],
                                                                  ref["count"])


def test_c2_640x480_fp32_vs_cpu(q_default):
    """BASELINE.json configs[1]: 640x480 fp32, d = k/8, validate XYZ vs CPU."""
    disp = synth_disparity(2, 0, 640, 480, "k8")
    want = oracle.reproject(disp, q_default, border=40)
    with ctx_for(q_default) as ctx:
        got, idx = ctx.process(disp, want_index=True)
    assert got.shape == (224000, 4)
    assert_points_close(got, want, max_ulp=MAX_ULP, rel=REL_TOL, what="C2")
    v, u = np.mgrid[40:440, 40:600]
    assert np.array_equal(idx, (v * 640 + u).reshape(-1).astype(np.uint32))
    # also within tolerance of the OpenCV-4 form of the same loop
    want4 = oracle.reproject(disp, q_default, border=40, form=oracle.FORM_CV4)
    assert_points_close(got, want4, max_ulp=2, rel=REL_TOL, what="C2/cv4")


def test_native_752x480_u8_as_the_reference_feeds_it(q_default):
    """The reference's own geometry and input: 8-bit disparity, x1/8."""
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, size=(480, 752)).astype(np.uint8)  # zeros included
    want = oracle.reproject(raw, q_default, border=40, scale=0.125)
    with ctx_for(q_default) as ctx:
        got = ctx.process(raw, scale=0.125)
    assert got.shape == (268800, 4)
    assert_points_close(got, want, max_ulp=MAX_ULP, rel=REL_TOL, what="752x480 u8")
    assert np.isinf(got[:, 2]).sum() == np.count_nonzero(raw[40:440, 40:712] == 0)


def test_c1_mono16_plumbing(q_default):
    """configs[0]: uint16 k*257 published as mono16; cv_bridge maps it to k
    (cpp:50), then the reference's /8 decode.  Here: U16 input, scale such
    that d = (v/257)/8."""
    img = synth_disparity(1, 0, 640, 480, "mono16")
    mono8 = oracle.mono16_to_mono8(img)
    want = oracle.reproject(mono8, q_default, border=40, scale=0.125)
    with ctx_for(q_default) as ctx:
        got = ctx.process(mono8, scale=0.125)
        got16 = ctx.process(img, scale=float(np.float32(0.125) / np.float32(257)))
    assert got.shape == (224000, 4)
    assert_points_close(got, want, max_ulp=MAX_ULP, rel=REL_TOL, what="C1")
    m8 = oracle.reproject(img, q_default, border=40, scale=float(np.float32(0.125) / np.float32(257)))
    assert_points_close(got16, m8, max_ulp=MAX_ULP, rel=REL_TOL, what="C1/u16")


def test_zero_disparity_inf_nan_pattern(q_default):
    disp = np.zeros((330, 752), dtype=np.float32)
    want = oracle.reproject(disp, q_default, border=40)
    with ctx_for(q_default) as ctx:
        got = ctx.process(disp)
    assert_points_close(got, want, what="d=0")
    g = got.reshape(250, 672, 4)
    v, u = np.mgrid[40:290, 40:712]
    assert np.all(np.isposinf(g[..., 2]))
    assert np.array_equal(np.isnan(g[..., 1]), v == 240)
    assert np.array_equal(np.isneginf(g[..., 0]), u < 375.9995)


def test_nan_and_inf_disparities(q_default):
    disp = synth_disparity(2, 5, 200, 120, "uniform")
    disp[50, 60] = np.nan
    disp[51, 61] = np.inf
    disp[52, 62] = -np.inf
    disp[53, 63] = -3.0
    want = oracle.reproject(disp, q_default, border=40)
    with ctx_for(q_default) as ctx:
        got = ctx.process(disp)
    assert_points_close(got, want, max_ulp=MAX_ULP, what="nan/inf d")


@pytest.mark.parametrize("w,h,border", [(96, 96, 40), (81, 81, 40), (97, 83, 40), (131, 97, 5), (1, 1, 0),
                                        (3, 1000, 0), (1000, 3, 0), (1025, 9, 0), (257, 263, 7), (4099, 5, 1)])
def test_ragged_sizes(w, h, border):
    rng = np.random.default_rng(w * 10007 + h)
    q = rng.uniform(-1, 1, 16)
    q[12:16] = [2e-4, 1e-4, 0.03, 0.7]
    disp = rng.uniform(0.5, 128, size=(h, w)).astype(np.float32)
    want = oracle.reproject(disp, q, border=border, form=oracle.FORM_CV4)   # a dense Q: OpenCV 3/4's association, bit for bit
    with ctx_for(q, border=border) as ctx:
        got, idx = ctx.process(disp, want_index=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{w}x{h}"
    assert_points_close(got, oracle.reproject(disp, q, border=border), max_ulp=8, rel=REL_TOL, what=f"{w}x{h} vs the 2.4 form")
    v, u = np.mgrid[border : h - border, border : w - border]
    assert np.array_equal(idx, (v * w + u).reshape(-1).astype(np.uint32))


@pytest.mark.parametrize("w,h", [(80, 80), (80, 200), (200, 80), (79, 300), (40, 40)])
def test_empty_roi(w, h, q_default):
    disp = np.ones((h, w), dtype=np.float32)
    with ctx_for(q_default) as ctx:
        got = ctx.process(disp)
        assert got.shape == (0, 4)
        ctx.set_mode(d2pc.MODE_COMPACT)
        got = ctx.process(disp)
        assert got.shape == (0, 4)


def test_row_stride(q_default):
    big = synth_disparity(2, 1, 300, 150, "k8")
    view = big[:, 17:250]
    want = oracle.reproject(np.ascontiguousarray(view), q_default, border=10)
    with ctx_for(q_default, border=10) as ctx:
        got = ctx.process(view)
    assert_points_close(got, want, max_ulp=MAX_ULP)


@pytest.mark.parametrize("pxt", [4, 8, 16])
def test_tile_shapes_agree_bitwise(q_default, pxt):
    disp = synth_disparity(3, 2, 500, 300, "holes")
    with ctx_for(q_default) as ctx:
        ref = ctx.process(disp)
    with ctx_for(q_default, variant="exp") as ctx:   # the tile-walking kernel of rounds 1-2: experiment build
        ctx.set_tuning("pxt_parity", pxt)
        ctx.set_tuning("blocks_per_cu", 1)
        got = ctx.process(disp)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))


# --------------------------------------------------------------- compact mode
@pytest.mark.parametrize("algo", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["holes", "blocky", "uniform"])
def test_compact_vs_oracle_small(q_default, kind, algo):
    disp = synth_disparity(3, 0, 640, 360, kind)
    wp, wi = oracle.reproject_compact(disp, q_default, border=40)
    with ctx_for(q_default, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        gp, gi = ctx.process(disp, want_index=True)
        m = ctx.cloud_meta(len(gp))
    assert len(gp) == len(wp), "point count must be bit-exact"
    assert np.array_equal(gi, wi), "pixel indices must be bit-exact and in row-major order"
    assert_points_close(gp, wp, max_ulp=MAX_ULP, rel=REL_TOL, what=f"compact {kind}")
    assert m.is_dense == 1 and m.width == len(gp)


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
def test_c3_1080p_30pct_invalid(q_default, algo):
    """BASELINE.json configs[2]: 1920x1080 fp32, ~30 % invalid, compaction on."""
    for kind in ("holes", "blocky"):
        disp = synth_disparity(3, 1, 1920, 1080, kind)
        wp, wi = oracle.reproject_compact(disp, q_default, border=40)
        with ctx_for(q_default, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
            gp, gi = ctx.process(disp, want_index=True)
        assert len(gp) == len(wp)
        assert np.array_equal(gi, wi)
        assert_points_close(gp, wp, max_ulp=MAX_ULP, rel=REL_TOL, what=f"C3 {kind}")
        assert 0.6 < len(gp) / 1840000 < 0.8


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
@pytest.mark.parametrize("pxt", [4, 8, 16])
def test_compact_edge_patterns(q_default, algo, pxt):
    rng = np.random.default_rng(99)
    base = rng.uniform(0.5, 128, size=(200, 333)).astype(np.float32)
    patterns = {
        "all_invalid": np.zeros_like(base),
        "all_valid": base,
        "checker": np.where((np.indices(base.shape).sum(0) & 1) == 0, base, 0).astype(np.float32),
        "one_valid": np.where(np.arange(base.size).reshape(base.shape) == 12345, base, 0).astype(np.float32),
        "last_only": np.where(np.arange(base.size).reshape(base.shape) == base.size - 1, base, 0).astype(np.float32),
    }
    with ctx_for(q_default, border=0, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo, pxt)) as ctx:
        ctx.set_tuning("pxt_compact", pxt)
        for name, disp in patterns.items():
            wp, wi = oracle.reproject_compact(disp, q_default, border=0)
            gp, gi = ctx.process(disp, want_index=True)
            assert len(gp) == len(wp), name
            assert np.array_equal(gi, wi), name
            assert_points_close(gp, wp, max_ulp=MAX_ULP, what=name)


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
def test_compact_tiny_w_takes_exact_slow_path(algo):
    """W so small that coordinates overflow float32 for part of the frame:
    the count pass's cheap predicate must fall back to the real arithmetic
    and agree with the scatter pass and the oracle point for point."""
    q = d2pc.make_q()
    q[14] = 1e-36  # W = 1e-36*d: between 'certainly finite' and zero
    rng = np.random.default_rng(5)
    disp = rng.uniform(1.0, 40.0, size=(300, 900)).astype(np.float32)
    disp[rng.random(disp.shape) < 0.1] = 0.0
    wp, wi = oracle.reproject_compact(disp, q, border=0)
    assert 0.05 < len(wp) / disp.size < 0.95
    with ctx_for(q, border=0, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        gp, gi = ctx.process(disp, want_index=True)
    assert np.array_equal(gi, wi)
    assert_points_close(gp, wp, max_ulp=MAX_ULP)


def test_general_and_stereo_kernels_agree_bitwise(q_default):
    """The stereoRectify-structured specialisation drops only exact products of the fused multiply-add evaluation
    (tuning general_q_form = 1): it must reproduce that general kernel bit for bit, NaN/inf inputs included.  The
    DEFAULT general evaluation is OpenCV 3/4's association: checked against the oracle's FORM_CV4 bit for bit."""
    disp = synth_disparity(3, 7, 700, 500, "holes")
    disp[100, 100:110] = [np.nan, np.inf, -np.inf, -1.0, 1e-30, 3e38, 0.0, -0.0, 1e-45, 5.0]
    with ctx_for(q_default) as ctx:
        a = ctx.process(disp)
        ctx.set_test_hook("force_general_q", 1)
        b4 = ctx.process(disp)
        w4 = oracle.reproject(disp, q_default, border=40, form=oracle.FORM_CV4)
        nan4 = np.isnan(w4)
        assert np.array_equal(nan4, np.isnan(b4)) and np.array_equal(b4.view(np.uint32)[~nan4], w4.view(np.uint32)[~nan4])
        ctx.set_mode(d2pc.MODE_COMPACT)
        c4, i4 = ctx.process(disp, want_index=True)
        wc4, wi4 = oracle.reproject_compact(disp, q_default, border=40, form=oracle.FORM_CV4)
        assert np.array_equal(i4, wi4) and np.array_equal(c4.view(np.uint32), wc4.view(np.uint32))
    with ctx_for(q_default, variant="exp") as ctx:   # round 2's fused form exists in the experiment build only
        a_exp = ctx.process(disp)
        assert np.array_equal(a_exp.view(np.uint32), a.view(np.uint32)), "product and experiment build differ on the product's path"
        ctx.set_test_hook("force_general_q", 1)
        ctx.set_test_hook("general_q_form", 1)
        b = ctx.process(disp)
        ctx.set_mode(d2pc.MODE_COMPACT)
        cg, ig = ctx.process(disp, want_index=True)
        ctx.set_test_hook("force_general_q", 0)
        cs, i_s = ctx.process(disp, want_index=True)
    nan = np.isnan(a)
    assert np.array_equal(nan, np.isnan(b)) and nan.sum() >= 9  # NaN payloads may differ, NaN-ness may not
    assert np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan])
    assert np.array_equal(ig, i_s) and np.array_equal(cg.view(np.uint32), cs.view(np.uint32))
    want = oracle.reproject(disp, q_default, border=40)
    assert_points_close(a, want, max_ulp=MAX_ULP)


def _same_bits(got, want, what=""):
    nan = np.isnan(want)
    assert np.array_equal(nan, np.isnan(got)), what + ": NaN pattern"
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan]), what


@pytest.mark.parametrize("q33", [None, 0.37, -1.0 / 3.0])
@pytest.mark.parametrize("w,h,border,cx", [(752, 480, 40, 376.0), (3840, 2160, 40, 1919.5), (1025, 67, 0, 511.37),
                                           (4099, 5, 1, 2050.123456789), (333, 200, 7, 0.1), (640, 360, 0, -3.75),
                                           (65, 33, 0, 1e-9)])
def test_reproject_form_selects_one_opencv_generation_bit_for_bit(w, h, border, cx, q33):
    """Tuning reproject_form: 24 = OpenCV 2.4's loop (running column sum qx += q00, replayed by the host into a table of
    its roundings), 4 = OpenCV 3/4's Matx product with float numerators.  Each against the oracle's form of the same
    name at 0 ulp on cv::stereoRectify's Q -- the calibrated path --, PARITY and COMPACT, with holes, NaN and inf in
    the input.  (The default for that Q is the specialised kernel: <= 1 ulp from both, ~25 % less arithmetic.)
    q33 != 0 is stereoRectify WITHOUT CALIB_ZERO_DISPARITY (different principal points): W = q33 + RN(q32 * d) then
    rounds twice in both generations, where the default kind's fused multiply-add rounds once."""
    q = d2pc.make_q(cx=cx, cy=h / 2 - 0.3, nx=w, ny=h)
    if q33 is not None:
        q[15] = q33
    disp = synth_disparity(3, w + h, w, h, "holes")
    disp[h // 2, w // 2 : w // 2 + 6] = [np.nan, np.inf, -np.inf, -1.0, 3.4028235e38, 1e-45][: min(6, w - w // 2)]
    for form, oform in ((24, oracle.FORM_CV24), (4, oracle.FORM_CV4)):
        want = oracle.reproject(disp, q, border=border, form=oform)
        wp, wi = oracle.reproject_compact(disp, q, border=border, form=oform)
        # two independent routes to the same bytes: the specialised kinds (QK_STEREO_CV24 / _CV4: the generation's
        # roundings on stereoRectify's structure) and the general kernel in that generation's form
        for general in (0, 1):
            with ctx_for(q, border=border) as ctx:
                ctx.set_reproject_form(form)
                ctx.set_test_hook("force_general_q", general)
                _same_bits(ctx.process(disp), want, f"form {form} general={general} {w}x{h}")
            with ctx_for(q, border=border, variant="exp") as ctx:   # the tile-walking kernel too (experiment build)
                ctx.set_reproject_form(form)
                ctx.set_test_hook("force_general_q", general)
                ctx.set_tuning("pxt_parity", 8)
                _same_bits(ctx.process(disp), want, f"form {form} general={general} {w}x{h} (tiles)")
            for algo in (1, 2, 3, 4):
                with ctx_for(q, border=border, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
                    ctx.set_reproject_form(form)
                    ctx.set_test_hook("force_general_q", general)
                    gp, gi = ctx.process(disp, want_index=True)
                assert np.array_equal(gi, wi)
                _same_bits(gp, wp, f"compact (algo {algo}) form {form} general={general} {w}x{h}")


@pytest.mark.parametrize("q03", [-0.1, -0.7, -1e-9, 0.3, -2047.9])
def test_reproject_form_24_with_a_principal_point_that_crosses_many_binades(q03):
    """Advisor, round 3: OpenCV 2.4's running column sum qx += q00 rounds once per binade it crosses; a SMALL non-dyadic
    principal point crosses one per doubling of the column (q03 = -0.1: 13 over 4100 columns), more than the 7 segments the
    host's table used to hold, and the table was replayed over 4096 columns whatever the width -- D2PC_ERR_BAD_SIZE for a
    valid calibration.  Now 18 segments, replayed over the frame's own width: bit for bit on both routes, wide and narrow."""
    for w, h in ((4100, 6), (70, 9)):
        q = d2pc.make_q(nx=w, ny=h)
        q[3] = q03
        disp = synth_disparity(3, w, w, h, "holes")
        want = oracle.reproject(disp, q, border=0, form=oracle.FORM_CV24)
        wp, wi = oracle.reproject_compact(disp, q, border=0, form=oracle.FORM_CV24)
        for general in (0, 1):
            with ctx_for(q, border=0) as ctx:
                ctx.set_reproject_form(d2pc.FORM_CV24)
                ctx.set_test_hook("force_general_q", general)
                _same_bits(ctx.process(disp), want, f"q03={q03} {w}x{h} general={general}")
                ctx.set_mode(d2pc.MODE_COMPACT)
                gp, gi = ctx.process(disp, want_index=True)
            assert np.array_equal(gi, wi)
            _same_bits(gp, wp, f"compact q03={q03} {w}x{h} general={general}")


def test_reproject_form_24_u8_and_scale(q_default):
    """The node's own input (uint8 disparities times 1/8, cpp:61) through OpenCV 2.4's form, bit for bit."""
    rng = np.random.default_rng(24)
    img = rng.integers(0, 256, size=(480, 752)).astype(np.uint8)
    want = oracle.reproject(img, q_default, border=40, scale=0.125, form=oracle.FORM_CV24)
    with ctx_for(q_default) as ctx:
        ctx.set_reproject_form(24)
        _same_bits(ctx.process(img, scale=0.125), want, "u8 x 1/8, 2.4 form")
    with ctx_for(q_default) as ctx:
        base = ctx.process(img, scale=0.125)
    assert_points_close(base, want, max_ulp=MAX_ULP, what="default kernel against the same frame")


def test_reproject_form_24_refuses_a_q_without_exact_column_steps():
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, 16)
    q[12:16] = [2e-4, 1e-4, 0.03, 0.7]
    disp = np.ones((50, 60), dtype=np.float32)
    with ctx_for(q, border=0) as ctx:
        ctx.set_reproject_form(24)
        with pytest.raises(d2pc.D2pcError) as e:
            ctx.process(disp)
        assert e.value.status == 1 and "column" in str(e.value)
        with pytest.raises(d2pc.D2pcError):
            ctx.set_reproject_form(3)
        ctx.set_reproject_form(d2pc.FORM_CV4)
        got = ctx.process(disp)
    assert np.array_equal(got.view(np.uint32), oracle.reproject(disp, q, border=0, form=oracle.FORM_CV4).view(np.uint32))
    # a new Q on the same context drops the cached table of the old one
    qa, qb = d2pc.make_q(cx=100.3), d2pc.make_q(cx=517.77)
    disp = synth_disparity(3, 5, 700, 90, "k8")
    with ctx_for(qa, border=0) as ctx:
        ctx.set_reproject_form(24)
        _same_bits(ctx.process(disp), oracle.reproject(disp, qa, border=0), "first Q")
        ctx.set_q(qb)
        _same_bits(ctx.process(disp), oracle.reproject(disp, qb, border=0), "second Q")


def test_compact_min_disparity(q_default):
    disp = synth_disparity(3, 3, 400, 300, "holes")
    wp, wi = oracle.reproject_compact(disp, q_default, border=40, min_disparity=64.0)
    with ctx_for(q_default, mode=d2pc.MODE_COMPACT, min_disparity=64.0) as ctx:
        gp, gi = ctx.process(disp, want_index=True)
    assert np.array_equal(gi, wi)
    assert_points_close(gp, wp, max_ulp=MAX_ULP)


def test_compact_capacity_error(q_default):
    disp = synth_disparity(3, 4, 300, 200, "uniform")
    with ctx_for(q_default, mode=d2pc.MODE_COMPACT) as ctx:
        with pytest.raises(d2pc.D2pcError) as e:
            ctx.process(disp, capacity=100)
        assert e.value.status == 4


# ------------------------------------------------------------- ABI misuse (T4)
def test_abi_misuse_returns_codes(q_default):
    lib = d2pc.load_library()
    disp = np.ones((100, 100), dtype=np.float32)
    out = np.empty((400, 4), dtype=np.float32)
    n = ctypes.c_size_t()
    with d2pc.Context(border=40) as ctx:
        h = ctx.handle
        args = lambda **k: [  # noqa: E731
            h, k.get("disp", disp.ctypes.data), k.get("dtype", 0), 1.0, k.get("w", 100), k.get("h", 100),
            k.get("stride", 400), k.get("out", out.ctypes.data), None, k.get("cap", 400), ctypes.byref(n)]
        assert lib.d2pc_process(*args()) == 7  # Q not set
        ctx.set_q(q_default)
        assert lib.d2pc_process(*args()) == 0 and n.value == 400
        assert lib.d2pc_process(*args(disp=None)) == 1
        assert lib.d2pc_process(*args(out=None)) == 1
        assert lib.d2pc_process(*args(dtype=5)) == 2
        assert lib.d2pc_process(*args(w=0)) == 3
        assert lib.d2pc_process(*args(h=-4)) == 3
        assert lib.d2pc_process(*args(stride=396)) == 3
        assert lib.d2pc_process(*args(stride=402)) == 3
        assert lib.d2pc_process(*args(cap=399)) == 4
        assert b"capacity" in lib.d2pc_last_error(h)
        assert n.value == 0
        # context still usable after errors
        assert lib.d2pc_process(*args()) == 0 and n.value == 400


def test_calibration_blob_roundtrip(q_default):
    with ctx_for(q_default, border=13, mode=d2pc.MODE_COMPACT) as a, d2pc.Context() as b:
        blob = a.export_calibration()
        assert len(blob) == d2pc.CALIB_BLOB_BYTES
        b.import_calibration(blob)
        assert b.get_q().tobytes() == q_default.tobytes()
        cfg = b.config()
        assert (cfg.border, cfg.mode) == (13, d2pc.MODE_COMPACT)
        with pytest.raises(d2pc.D2pcError):
            b.import_calibration(blob[:-1])


def test_disparity_image_style_calibration_and_threshold():
    """f, T, min_disparity carried by the message."""
    q = d2pc.make_q_disparity_image(520.0, 0.11, 319.5, 239.5)
    disp = synth_disparity(3, 9, 640, 480, "holes")
    with ctx_for(q, mode=d2pc.MODE_COMPACT) as ctx:
        ctx.set_min_disparity(2.0)
        gp, gi = ctx.process(disp, want_index=True)
    wp, wi = oracle.reproject_compact(disp, q, border=40, min_disparity=2.0)
    assert np.array_equal(gi, wi)
    assert_points_close(gp, wp, max_ulp=MAX_ULP)
    z = 520.0 * 0.11 / disp.reshape(-1)[gi].astype(np.float64)
    assert np.allclose(gp[:, 2], z, rtol=2e-7)


def test_stage_timing_of_the_host_entry_points(q_default):
    """per-stage HIP-event times of the synchronous host path."""
    img = np.random.default_rng(1).integers(0, 256, size=(480, 752)).astype(np.uint8)
    with d2pc.Context(q=q_default) as ctx:
        ctx.process_mono8(img)
        with pytest.raises(d2pc.D2pcError):
            ctx.last_stage_times()                      # timing was off
        ctx.set_tuning("stage_timing", 1)
        ctx.process_mono8(img, median_ksize=11)
        t = ctx.last_stage_times()
        assert all(t[k] > 0 for k in ("h2d_ms", "prep_ms", "kernel_ms", "d2h_ms", "total_ms")), t
        assert t["total_ms"] >= 0.99 * (t["h2d_ms"] + t["prep_ms"] + t["kernel_ms"] + t["d2h_ms"]), t
        assert t["total_ms"] < 100.0
        ctx.process(img.astype(np.float32))             # no median: prep stage is (nearly) empty
        assert ctx.last_stage_times()["prep_ms"] < t["prep_ms"]


# ------------------------------------------------- pinned caller buffers (d2pc_host_alloc)
@pytest.mark.parametrize("mode", [d2pc.MODE_PARITY, d2pc.MODE_COMPACT])
def test_pinned_output_is_written_directly_and_matches_the_staged_path(q_default, mode):
    """out_points / out_index in memory from d2pc_host_alloc: the kernels store the final bytes there (no
    device-side copy of the cloud, no D2H copy).  Bit-identical to the pageable path, nothing written past n."""
    disp = synth_disparity(2, 9, 752, 480, "holes")
    cap = d2pc.roi_points(752, 480, 40)
    pts_buf, idx_buf = d2pc.PinnedBuffer((cap + 8, 4), np.float32), d2pc.PinnedBuffer(cap + 8, np.uint32)
    pts_buf.array[:] = np.float32(-7.0)
    idx_buf.array[:] = 0xDEADBEEF
    with ctx_for(q_default, mode=mode) as ctx:
        want_p, want_i = ctx.process(disp, want_index=True)
        got_p, got_i = ctx.process(disp, out=pts_buf.array, out_index=idx_buf.array)
        n = len(got_p)
        assert n == len(want_p)
        assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)) and np.array_equal(got_i, want_i)
        # the same memory the caller handed in, and untouched behind the points that were produced
        assert got_p.ctypes.data == pts_buf.array.ctypes.data
        if mode == d2pc.MODE_PARITY:
            assert np.all(pts_buf.array[n:] == np.float32(-7.0)) and np.all(idx_buf.array[n:] == 0xDEADBEEF)
        # a pinned buffer too small for the whole ROI (legal in COMPACT): the staged path takes over
        if mode == d2pc.MODE_COMPACT:
            small = d2pc.PinnedBuffer((n + 1, 4), np.float32)
            again = ctx.process(disp, capacity=n + 1, out=small.array)
            assert np.array_equal(again.view(np.uint32), want_p.view(np.uint32))
            small.close()
    pts_buf.close()
    idx_buf.close()


def test_pinned_array_from_a_temporary_owner_stays_valid(q_default):
    """The docstring's pattern `out=PinnedBuffer(...).array`: the owner object dies at once, the array keeps the
    pinned allocation (advisor, round 2: it used to be freed, and the kernels then stored into freed memory)."""
    import gc

    disp = synth_disparity(2, 9, 752, 480, "holes")
    cap = d2pc.roi_points(752, 480, 40)
    out = d2pc.PinnedBuffer((cap, 4), np.float32).array
    gc.collect()
    filler = [d2pc.PinnedBuffer((cap, 4), np.float32) for _ in range(3)]  # would reuse a freed block
    for f in filler:
        f.array[:] = np.float32(3.0)
    with ctx_for(q_default) as ctx:
        want = ctx.process(disp)
        got = ctx.process(disp, out=out)
        assert got.ctypes.data == out.ctypes.data
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert all(np.all(f.array == np.float32(3.0)) for f in filler)


@pytest.mark.parametrize("mode", [d2pc.MODE_PARITY, d2pc.MODE_COMPACT])
def test_pinned_input_is_read_in_place(mode, q_default):
    """A frame that lies in pinned host memory is read by the first kernel straight from there (no staging copy):
    fp32 through d2pc_process (rows with a pitch), mono8 and mono16 through the median entry points.  Same bytes
    as from pageable memory and as with the tuning key host_direct_read = 0."""
    rng = np.random.default_rng(31 + mode)
    h, w, pitch = 300, 412, 420
    f32 = d2pc.PinnedBuffer((h, pitch), np.float32)
    f32.array[:] = synth_disparity(2, 3, pitch, h, "holes")
    u8 = d2pc.PinnedBuffer((h, pitch), np.uint8)
    u8.array[:] = rng.integers(0, 256, size=(h, pitch)).astype(np.uint8)
    u16 = d2pc.PinnedBuffer((h, pitch), np.uint16)
    u16.array[:] = rng.integers(0, 65536, size=(h, pitch)).astype(np.uint16)
    with ctx_for(q_default, mode=mode) as ctx:
        for direct in (1, 0):
            ctx.set_tuning("host_direct_read", direct)
            for pinned, run in ((f32.array[:, :w], lambda a: ctx.process(a, want_index=True)),
                                (u8.array[:, :w], lambda a: ctx.process_mono8(a, 11, 0.125, want_index=True)),
                                (u8.array[:, :w], lambda a: ctx.process_mono8(a, 0, 0.125, want_index=True)),
                                (u16.array[:, :w], lambda a: ctx.process_mono16(a, 11, 0.125, want_index=True))):
                got_p, got_i = run(pinned)
                want_p, want_i = run(np.ascontiguousarray(pinned))   # a pageable copy
                assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)) and np.array_equal(got_i, want_i)
    for b in (f32, u8, u16):
        b.close()
