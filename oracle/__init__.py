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
        for f in ("d2pc_oracle.c", "d2pc_oracle.h", "Makefile")
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
        L.d2pc_oracle_max_threads.restype = ctypes.c_int
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


def max_threads() -> int:
    return lib().d2pc_oracle_max_threads()
