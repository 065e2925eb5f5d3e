"""ctypes binding of the CPU oracle (oracle/d2pc_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see d2pc_oracle.h).  Importable
from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package `disparity_to_point_cloud_amd` never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libd2pc_oracle.so")

F32, U8, U16 = 0, 1, 2
FORM_CV24, FORM_CV4 = 0, 1
_NP_DTYPE = {F32: np.float32, U8: np.uint8, U16: np.uint16}

_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "d2pc_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("d2pc_oracle.c", "d2pc_oracle_fusion.c", "d2pc_oracle.h", "Makefile")
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    assert os.path.exists(_LIB_PATH), src
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c_dp = ctypes.POINTER(ctypes.c_double)
        L.d2pc_oracle_make_q.argtypes = [ctypes.c_double] * 5 + [ctypes.c_int, ctypes.c_int, c_dp]
        L.d2pc_oracle_make_q.restype = None
        L.d2pc_oracle_reproject.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
            c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.d2pc_oracle_reproject.restype = ctypes.c_size_t
        L.d2pc_oracle_reproject_compact.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
            c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        L.d2pc_oracle_reproject_compact.restype = ctypes.c_size_t
        L.d2pc_oracle_mono16_to_mono8.argtypes = [
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.d2pc_oracle_mono16_to_mono8.restype = None
        L.d2pc_oracle_median_u8.argtypes = [
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
            ctypes.c_int]
        L.d2pc_oracle_median_u8.restype = None
        L.d2pc_oracle_median_u8_fast.argtypes = L.d2pc_oracle_median_u8.argtypes
        L.d2pc_oracle_median_u8_fast.restype = None
        L.d2pc_oracle_max_threads.restype = ctypes.c_int
        L.d2pc_oracle_fuse_pixel.argtypes = [ctypes.c_int] * 7
        L.d2pc_oracle_fuse_pixel.restype = ctypes.c_int
        L.d2pc_oracle_fuse.argtypes = [
            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t]
        L.d2pc_oracle_fuse.restype = ctypes.c_int
        L.d2pc_oracle_crop_to_square.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)]
        L.d2pc_oracle_crop_to_square.restype = None
        L.d2pc_oracle_rotate_cw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_size_t]
        L.d2pc_oracle_rotate_cw.restype = None
        _lib = L
    return _lib


def _q_ptr(q):
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(16))
    return q, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def dtype_code(arr: np.ndarray) -> int:
    for code, dt in _NP_DTYPE.items():
        if arr.dtype == dt:
            return code
    raise TypeError(f"unsupported disparity dtype {arr.dtype}")


def make_q(fx=714.24, fy=713.5, cx=376.0, cy=240.0, baseline=0.09, nx=752, ny=480) -> np.ndarray:
    q = np.zeros(16, dtype=np.float64)
    lib().d2pc_oracle_make_q(fx, fy, cx, cy, baseline, nx, ny, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return q


def reproject(disp: np.ndarray, q, border=40, scale=1.0, form=FORM_CV24, threads=1, out=None) -> np.ndarray:
    """(H,W) disparity -> (R,4) float32 points, reference (unfiltered) semantics.
    `out` lets a timing loop reuse one output buffer."""
    assert disp.ndim == 2 and disp.strides[1] == disp.itemsize
    h, w = disp.shape
    rw, rh = max(w - 2 * border, 0), max(h - 2 * border, 0)
    if out is None:
        out = np.empty((rw * rh, 4), dtype=np.float32)
    assert out.shape == (rw * rh, 4) and out.dtype == np.float32 and out.flags.c_contiguous
    qk, qp = _q_ptr(q)
    n = lib().d2pc_oracle_reproject(disp.ctypes.data, dtype_code(disp), scale, w, h, disp.strides[0], qp,
                                    border, form, threads, out.ctypes.data)
    assert n == rw * rh
    return out


def reproject_compact(disp: np.ndarray, q, border=40, scale=1.0, form=FORM_CV24,
                      min_disparity=-np.inf):
    """-> ((P,4) float32 points, (P,) uint32 source pixel indices)."""
    assert disp.ndim == 2 and disp.strides[1] == disp.itemsize
    h, w = disp.shape
    rw, rh = max(w - 2 * border, 0), max(h - 2 * border, 0)
    out = np.empty((rw * rh, 4), dtype=np.float32)
    idx = np.empty(rw * rh, dtype=np.uint32)
    qk, qp = _q_ptr(q)
    n = lib().d2pc_oracle_reproject_compact(disp.ctypes.data, dtype_code(disp), scale, w, h, disp.strides[0],
                                            qp, border, form, min_disparity, out.ctypes.data, idx.ctypes.data)
    return out[:n].copy(), idx[:n].copy()


def mono16_to_mono8(img: np.ndarray) -> np.ndarray:
    assert img.dtype == np.uint16 and img.ndim == 2 and img.strides[1] == 2
    out = np.empty(img.shape, dtype=np.uint8)
    lib().d2pc_oracle_mono16_to_mono8(img.ctypes.data, img.strides[0], out.ctypes.data, out.strides[0],
                                      img.shape[1], img.shape[0])
    return out


def median_u8(img: np.ndarray, ksize=11) -> np.ndarray:
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    out = np.empty(img.shape, dtype=np.uint8)
    lib().d2pc_oracle_median_u8(img.ctypes.data, img.strides[0], out.ctypes.data, out.strides[0],
                                img.shape[1], img.shape[0], ksize)
    return out


def median_u8_fast(img: np.ndarray, ksize=11) -> np.ndarray:
    """The same order statistic by the constant-time sliding-histogram algorithm (Perreault & Hebert 2007, what
    cv::medianBlur runs for 8-bit images and ksize > 5 [upstream]); single-threaded.  bench.py's CPU column."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    out = np.empty(img.shape, dtype=np.uint8)
    lib().d2pc_oracle_median_u8_fast(img.ctypes.data, img.strides[0], out.ctypes.data, out.strides[0],
                                     img.shape[1], img.shape[0], ksize)
    return out


def max_threads() -> int:
    return lib().d2pc_oracle_max_threads()


# ---- depth-map fusion inner loop (src/depth_map_fusion.cpp:113-130, 150-273) ----
(FUSE_WEIGHTED_AVERAGE, FUSE_MAX_DIST, FUSE_MAX_DIST_UNLESS_BLACK, FUSE_BETTER_SCORE, FUSE_ONLY_GOOD_1,
 FUSE_ONLY_GOOD_AVG, FUSE_OVERLAP, FUSE_BLACK_TO_WHITE, FUSE_GRAD_FILTER) = range(9)
REFERENCE_CROP = (0, 40, 30, 10)  # left, right, top, bottom at cpp:130


def fuse_pixel(rule, d1, d2, s1, s2, g1=0, g2=0) -> int:
    return lib().d2pc_oracle_fuse_pixel(rule, d1, d2, s1, s2, g1, g2)


def fuse(planes, rule=FUSE_GRAD_FILTER, crop=REFERENCE_CROP, want_combined=True):
    """planes = (depth1, depth2, score1, score2, grad1, grad2), equal-shape uint8 images
    -> (fused (h-t-b, w-l-r), combined (h, w) or None)."""
    assert len(planes) == 6
    h, w = planes[0].shape
    for p in planes:
        assert p.dtype == np.uint8 and p.shape == (h, w) and p.strides[1] == 1
    l, r, t, b = crop
    fused = np.empty((max(h - t - b, 0), max(w - l - r, 0)), dtype=np.uint8)
    combined = np.empty((h, w), dtype=np.uint8) if want_combined else None
    ptrs = (ctypes.c_void_p * 6)(*[p.ctypes.data for p in planes])
    pitch = (ctypes.c_size_t * 6)(*[p.strides[0] for p in planes])
    st = lib().d2pc_oracle_fuse(ptrs, pitch, w, h, rule, l, r, t, b, fused.ctypes.data, max(fused.strides[0], 1),
                                combined.ctypes.data if want_combined else None, w)
    if st != 0:
        raise ValueError("d2pc_oracle_fuse: bad arguments (%d)" % st)
    return fused, combined


def crop_to_square(cols, rows, offset_x=0, offset_y=0, member_offset_y=None):
    """-> (x, y, n) of the square region cropToSquare returns (cpp:247-265)."""
    rect = (ctypes.c_int * 3)()
    lib().d2pc_oracle_crop_to_square(cols, rows, offset_x, offset_y,
                                     offset_y if member_offset_y is None else member_offset_y, rect)
    return tuple(rect)


def rotate_cw(img: np.ndarray) -> np.ndarray:
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    rows, cols = img.shape
    out = np.empty((cols, rows), dtype=np.uint8)
    lib().d2pc_oracle_rotate_cw(img.ctypes.data, img.strides[0], cols, rows, out.ctypes.data, out.strides[0])
    return out
