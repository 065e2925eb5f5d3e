"""Property tests (tier T3) with hypothesis: random
non-singular Q, random sizes/borders/strides (odd widths, ROI <= 0), random
validity patterns.  CPU part: oracle invariants.  GPU part: HIP path == oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from helpers import assert_points_close

finite = st.floats(min_value=-4.0, max_value=4.0, allow_nan=False, allow_infinity=False, width=64)


@st.composite
def q_matrices(draw):
    q = np.array([draw(finite) for _ in range(16)], dtype=np.float64)
    # keep W = q30*u + q31*v + q32*d + q33 away from 0 for u,v < 512, d in [0.5,128]
    q[12] = draw(st.floats(min_value=0.0, max_value=1e-3))
    q[13] = draw(st.floats(min_value=0.0, max_value=1e-3))
    q[14] = draw(st.floats(min_value=0.01, max_value=1.0))
    q[15] = draw(st.floats(min_value=0.1, max_value=2.0))
    return q


@st.composite
def frames(draw, max_side=96):
    w = draw(st.integers(1, max_side))
    h = draw(st.integers(1, max_side))
    border = draw(st.integers(0, 12))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    p_invalid = draw(st.sampled_from([0.0, 0.1, 0.5, 1.0]))
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.5, 128.0, size=(h, w)).astype(np.float32)
    d[rng.random((h, w)) < p_invalid] = 0.0
    pad = draw(st.integers(0, 5))
    if pad:  # non-trivial row stride
        big = np.zeros((h, w + pad), dtype=np.float32)
        big[:, :w] = d
        d = big[:, :w]
    return d, border


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(q=q_matrices(), fb=frames())
def test_oracle_invariants(q, fb):
    disp, border = fb
    h, w = disp.shape
    full = oracle.reproject(disp, q, border=border)
    rw, rh = max(w - 2 * border, 0), max(h - 2 * border, 0)
    assert full.shape == (rw * rh, 4)
    assert np.all(full[:, 3].view(np.uint32) == 0x3F800000)
    # both published forms of reprojectImageTo3D agree to 2 ulp
    assert_points_close(full, oracle.reproject(disp, q, border=border, form=oracle.FORM_CV4), max_ulp=2, rel=1e-5)
    # compact == parity filtered by isfinite, order preserved, indices increasing and inside the ROI
    cp, ci = oracle.reproject_compact(disp, q, border=border)
    keep = np.isfinite(full[:, :3]).all(axis=1)
    assert np.array_equal(cp.view(np.uint32), full[keep].view(np.uint32))
    if len(ci):
        assert np.all(np.diff(ci.astype(np.int64)) > 0)
        v, u = np.divmod(ci.astype(np.int64), w)
        assert v.min() >= border and v.max() < h - border and u.min() >= border and u.max() < w - border
    # threads do not change a bit
    assert np.array_equal(full.view(np.uint32), oracle.reproject(disp, q, border=border, threads=3).view(np.uint32))


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(q=q_matrices(), fb=frames(max_side=160), stereo=st.booleans())
def test_gpu_matches_oracle_on_random_inputs(q, fb, stereo):
    import disparity_to_point_cloud_amd as d2pc

    disp, border = fb
    if stereo:  # the structure cv::stereoRectify produces (specialised kernel)
        q = np.array([1, 0, 0, q[3], 0, 1, 0, q[7], 0, 0, 0, q[11], 0, 0, q[14], q[15]], dtype=np.float64)
    # a general Q is evaluated in OpenCV 3/4's association (oracle FORM_CV4): BIT FOR BIT, points, classes and which
    # pixels survive; the structure cv::stereoRectify produces takes the specialised kernel: 1 ulp from the 2.4 form.
    # derandomize: the same examples in every run.
    form = oracle.FORM_CV24 if stereo else oracle.FORM_CV4
    want = oracle.reproject(disp, q, border=border, form=form)
    wp, wi = oracle.reproject_compact(disp, q, border=border, form=form)
    with d2pc.Context(q=q, border=border) as ctx:
        got = ctx.process(disp)
        ctx.set_mode(d2pc.MODE_COMPACT)
        gp, gi = ctx.process(disp, want_index=True)
    ulp = 1 if stereo else 0
    assert_points_close(got, want, max_ulp=ulp, rel=1e-5)
    assert np.array_equal(gi, wi)
    assert_points_close(gp, wp, max_ulp=ulp, rel=1e-5)
