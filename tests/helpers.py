"""Shared helpers for the parity tests (numpy only)."""
import numpy as np

DEFAULT_CALIB = dict(fx=714.24, fy=713.5, cx=376.0, cy=240.0, baseline=0.09, nx=752, ny=480)


def ulp_distance(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance in float32 ulps between finite values (0 when both are the
    same inf or both NaN; a huge number when the classes differ)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    # map the sign-magnitude float ordering onto a monotone integer line
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    d = np.abs(ai - bi)
    both_nan = np.isnan(a) & np.isnan(b)
    one_nan = np.isnan(a) ^ np.isnan(b)
    d = np.where(both_nan, 0, d)
    d = np.where(one_nan, 1 << 40, d)
    return d


def assert_points_close(got: np.ndarray, want: np.ndarray, max_ulp=2, rel=1e-5, what=""):
    """The parity bar for XYZ: identical NaN/inf classes, pad word bit-exact,
    finite values within `rel` relative (BASELINE.json north_star: 1e-5) and,
    tighter, within `max_ulp` float32 ulps."""
    got = np.asarray(got, dtype=np.float32).reshape(-1, 4)
    want = np.asarray(want, dtype=np.float32).reshape(-1, 4)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.array_equal(got[:, 3].view(np.uint32), want[:, 3].view(np.uint32)), f"{what}: pad word"
    g, w = got[:, :3], want[:, :3]
    assert np.array_equal(np.isnan(g), np.isnan(w)), f"{what}: NaN positions differ"
    assert np.array_equal(np.isposinf(g), np.isposinf(w)), f"{what}: +inf positions differ"
    assert np.array_equal(np.isneginf(g), np.isneginf(w)), f"{what}: -inf positions differ"
    fin = np.isfinite(w)
    if fin.any():
        gw, ww = g[fin].astype(np.float64), w[fin].astype(np.float64)
        denom = np.maximum(np.abs(ww), np.finfo(np.float32).tiny)
        relerr = np.abs(gw - ww) / denom
        assert relerr.max() <= rel, f"{what}: max rel err {relerr.max():.3e} > {rel}"
        d = ulp_distance(g[fin], w[fin])
        assert d.max() <= max_ulp, f"{what}: max ulp distance {d.max()} > {max_ulp}"


def frame_seed(config_id: int, frame_id: int) -> int:
    """seed = 0xD2C00000 + config_id*1000 + frame_id."""
    return 0xD2C00000 + config_id * 1000 + frame_id


def synth_disparity(config_id: int, frame_id: int, width: int, height: int, kind: str) -> np.ndarray:
    """Seeded synthetic disparity frames for BASELINE.json's configs.
      'k8'      d = k/8, k in U{1..255}   (C2: all valid, reference quantisation)
      'uniform' d ~ U(0.5,128)            (C4: all valid)
      'holes'   'uniform' with iid 30 % zeros         (C3)
      'blocky'  'uniform' with 64x64-block holes ~30 % (C3 variant)
      'mono16'  uint16 k*257, k in U{0..255}          (C1)
    """
    rng = np.random.default_rng(frame_seed(config_id, frame_id))
    if kind == "k8":
        return rng.integers(1, 256, size=(height, width)).astype(np.float32) * np.float32(0.125)
    if kind == "mono16":
        return (rng.integers(0, 256, size=(height, width)) * 257).astype(np.uint16)
    d = rng.uniform(0.5, 128.0, size=(height, width)).astype(np.float32)
    if kind == "uniform":
        return d
    if kind == "holes":
        d[rng.random(size=(height, width)) < 0.3] = 0.0
        return d
    if kind == "blocky":
        by, bx = (height + 63) // 64, (width + 63) // 64
        m = rng.random(size=(by, bx)) < 0.3
        m = np.repeat(np.repeat(m, 64, axis=0), 64, axis=1)[:height, :width]
        d[m] = 0.0
        return d
    raise ValueError(kind)
