"""ctypes binding of include/d2pc.h (the C-ABI drop-in boundary) and of include/d2pc_ext.h (unstable: bench / tools /
tests).

One-to-one with the two headers: every function they declare is bound here
and nothing else.  The binding never computes points itself.
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# D2PC_LIBRARY_VARIANT=x loads libd2pc_x.so (tuning builds of `make variant NAME=x`; tools and soak runs only)
_LIB_NAME = ("libd2pc_%s.so" % os.environ["D2PC_LIBRARY_VARIANT"]) if os.environ.get("D2PC_LIBRARY_VARIANT") else "libd2pc.so"

DTYPE_F32, DTYPE_U8, DTYPE_U16, DTYPE_MONO16 = 0, 1, 2, 3
MODE_PARITY, MODE_COMPACT = 0, 1
CALIB_BLOB_BYTES = 136

# every symbol include/d2pc.h declares: the STABLE surface (tests check the library exports all, and nothing else is listed)
ABI_SYMBOLS = [
    "d2pc_abi_version", "d2pc_status_string", "d2pc_device_count", "d2pc_make_q", "d2pc_config_init",
    "d2pc_calib_pack", "d2pc_calib_unpack", "d2pc_make_q_disparity_image", "d2pc_set_min_disparity",
    "d2pc_create", "d2pc_destroy", "d2pc_last_error", "d2pc_set_q", "d2pc_get_q", "d2pc_set_border",
    "d2pc_set_mode", "d2pc_get_config", "d2pc_export_calibration", "d2pc_import_calibration",
    "d2pc_roi_points", "d2pc_cloud_meta_fill", "d2pc_process", "d2pc_process_device",
    "d2pc_check_async_error", "d2pc_median_device", "d2pc_process_mono8",
    "d2pc_pipeline_configure", "d2pc_pipeline_acquire", "d2pc_pipeline_submit", "d2pc_pipeline_collect",
    "d2pc_pipeline_release", "d2pc_fuse_desc_init", "d2pc_fuse_device", "d2pc_crop_to_square",
    "d2pc_rotate_cw_device", "d2pc_mono16_to_mono8_device", "d2pc_process_mono16",
    "d2pc_median_roi_device", "d2pc_host_alloc", "d2pc_host_free", "d2pc_make_q_flavour",
    "d2pc_process_mono_device", "d2pc_set_reproject_form",
]
# include/d2pc_ext.h: unstable, for bench.py / tools / tests only
EXT_SYMBOLS = [
    "d2pc_ext_revision", "d2pc_reserve", "d2pc_reserve_mono", "d2pc_release_graph_buffers", "d2pc_compact_stats",
    "d2pc_compact_stats_reset", "d2pc_membench_fill", "d2pc_membench_copy", "d2pc_last_stage_times", "d2pc_set_tuning",
    "d2pc_ext_set_test_hook", "d2pc_clock_probe_device",
]
ABI_VERSION = 2
FORM_DEFAULT, FORM_CV24, FORM_CV4 = 0, 24, 4   # d2pc_reproject_form
# d2pc_fusion_rule (source order of the reference's src/depth_map_fusion.cpp:162-235)
(FUSE_WEIGHTED_AVERAGE, FUSE_MAX_DIST, FUSE_MAX_DIST_UNLESS_BLACK, FUSE_BETTER_SCORE, FUSE_ONLY_GOOD_1,
 FUSE_ONLY_GOOD_AVG, FUSE_OVERLAP, FUSE_BLACK_TO_WHITE, FUSE_GRAD_FILTER) = range(9)


class Config(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("device_id", ctypes.c_int32),
        ("border", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("min_disparity", ctypes.c_float),
        ("compact_algo", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 4),
    ]


class Field(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 8), ("offset", ctypes.c_uint32), ("datatype", ctypes.c_uint8),
                ("count", ctypes.c_uint32)]


class CloudMeta(ctypes.Structure):
    _fields_ = [
        ("height", ctypes.c_uint32), ("width", ctypes.c_uint32), ("point_step", ctypes.c_uint32),
        ("row_step", ctypes.c_uint32), ("is_bigendian", ctypes.c_uint8), ("is_dense", ctypes.c_uint8),
        ("n_fields", ctypes.c_uint32), ("fields", Field * 3),
    ]


class FrameDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", ctypes.c_int32), ("scale", ctypes.c_float), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("row_stride_bytes", ctypes.c_size_t), ("median_ksize", ctypes.c_int32), ("want_index", ctypes.c_int32),
        ("tag", ctypes.c_uint64),
    ]


class FuseDesc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("rule", ctypes.c_int32), ("width", ctypes.c_int32),
        ("height", ctypes.c_int32), ("n_frames", ctypes.c_int32), ("crop_left", ctypes.c_int32),
        ("crop_right", ctypes.c_int32), ("crop_top", ctypes.c_int32), ("crop_bottom", ctypes.c_int32),
        ("planes", ctypes.c_void_p * 6), ("pitch", ctypes.c_size_t * 6), ("frame_stride", ctypes.c_size_t * 6),
        ("fused", ctypes.c_void_p), ("fused_pitch", ctypes.c_size_t), ("fused_frame_stride", ctypes.c_size_t),
        ("combined", ctypes.c_void_p), ("combined_pitch", ctypes.c_size_t), ("combined_frame_stride", ctypes.c_size_t),
    ]


class StageTimes(ctypes.Structure):
    _fields_ = [("h2d_ms", ctypes.c_float), ("prep_ms", ctypes.c_float), ("kernel_ms", ctypes.c_float),
                ("d2h_ms", ctypes.c_float), ("total_ms", ctypes.c_float)]


class CompactStats(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("launches", ctypes.c_uint64),
                ("tiles", ctypes.c_uint64), ("failed_polls", ctypes.c_uint64), ("wait_us", ctypes.c_uint64),
                ("timeouts", ctypes.c_uint64), ("twopass_fallbacks", ctypes.c_uint64)]


class PinnedBuffer:
    """Page-locked host memory from d2pc_host_alloc with a numpy view (d2pc_process* stores the cloud
    straight into such a buffer).

    The ALLOCATION belongs to the array, not to this object: `.array` (and every view or slice taken from
    it) keeps the memory alive, and d2pc_host_free runs when the last of them is gone.  So
    `out=PinnedBuffer(...).array` is safe, and close() / garbage collection of the PinnedBuffer only drop
    this object's own reference."""

    def __init__(self, shape, dtype):
        lib = load_library()
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib.d2pc_host_alloc(max(nbytes, 1))
        if not self.ptr:
            raise MemoryError(f"d2pc_host_alloc({nbytes}) failed")
        buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(self.ptr)
        # np.frombuffer keeps `buf` alive through the array's base chain; the finalizer hangs on `buf`
        self._finalizer = weakref.finalize(buf, lib.d2pc_host_free, self.ptr)
        self._finalizer.atexit = False  # at interpreter exit the HIP runtime may be gone already: leave it to the OS
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    @property
    def alive(self) -> bool:
        """False once the memory has been handed back to d2pc_host_free."""
        return self._finalizer.alive

    def close(self):
        """Drop this object's reference.  The memory is freed now if no other view of `.array` exists,
        otherwise when the last one is collected."""
        self.array = None
        self.ptr = None


class D2pcError(RuntimeError):
    def __init__(self, status, message=""):
        self.status = status
        super().__init__(f"d2pc status {status} ({status_string(status)}): {message}")


_lib = None   # (kept for tools that reset it; the cache below is keyed by file name)
_libs = {}


def library_path(variant=None) -> str:
    return os.path.join(_HERE, _LIB_NAME if variant is None else "libd2pc_%s.so" % variant)


def load_library(variant=None):
    """Load libd2pc.so -- or, for `variant`, libd2pc_<variant>.so (tests and tools: "exp" is the experiment build of
    `make exp`).  Raises (never falls back) when it is not built."""
    global _lib
    path = library_path(variant)
    fname = os.path.basename(path)
    if fname in _libs:
        return _libs[fname]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7; import it first so both share ONE
    # HIP runtime (same SONAME => the loader reuses the already-mapped copy).
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only; the ABI works without it
        pass
    L = ctypes.CDLL(path)
    if fname != "libd2pc.so":
        # tuning / A-B builds (tools only) may predate an entry point: calling a missing one raises
        class _Missing:
            argtypes = restype = None

            def __call__(self, *a):
                raise AttributeError(f"{fname} does not export this entry point")
        for name in ABI_SYMBOLS + EXT_SYMBOLS:
            if not hasattr(L, name):
                setattr(L, name, _Missing())
    vp, cp = ctypes.c_void_p, ctypes.c_char_p
    dp = ctypes.POINTER(ctypes.c_double)
    L.d2pc_abi_version.restype = ctypes.c_int
    L.d2pc_status_string.argtypes = [ctypes.c_int]
    L.d2pc_status_string.restype = cp
    L.d2pc_device_count.restype = ctypes.c_int
    L.d2pc_make_q.argtypes = [ctypes.c_double] * 5 + [ctypes.c_int, ctypes.c_int, dp]
    L.d2pc_make_q_flavour.argtypes = [ctypes.c_double] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
    L.d2pc_make_q_disparity_image.argtypes = [ctypes.c_double] * 4 + [dp]
    L.d2pc_set_min_disparity.argtypes = [vp, ctypes.c_float]
    L.d2pc_calib_pack.argtypes = [dp, ctypes.c_int, ctypes.c_int, vp]
    L.d2pc_calib_unpack.argtypes = [vp, ctypes.c_size_t, dp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.d2pc_config_init.argtypes = [ctypes.POINTER(Config)]
    L.d2pc_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.d2pc_destroy.argtypes = [vp]
    L.d2pc_last_error.argtypes = [vp]
    L.d2pc_last_error.restype = cp
    L.d2pc_set_q.argtypes = [vp, dp]
    L.d2pc_get_q.argtypes = [vp, dp]
    L.d2pc_set_border.argtypes = [vp, ctypes.c_int]
    L.d2pc_set_mode.argtypes = [vp, ctypes.c_int]
    L.d2pc_get_config.argtypes = [vp, ctypes.POINTER(Config)]
    L.d2pc_export_calibration.argtypes = [vp, vp]
    L.d2pc_import_calibration.argtypes = [vp, vp, ctypes.c_size_t]
    L.d2pc_roi_points.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.d2pc_roi_points.restype = ctypes.c_size_t
    L.d2pc_cloud_meta_fill.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(CloudMeta)]
    L.d2pc_process.argtypes = [vp, vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                               vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.d2pc_process_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp, vp, ctypes.c_size_t, vp,
                                      vp]
    L.d2pc_median_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                     vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp]
    L.d2pc_median_roi_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                     vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp]
    L.d2pc_process_mono8.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_float,
                                     vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.d2pc_process_mono16.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_float,
                                      vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.d2pc_mono16_to_mono8_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
                                              ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.d2pc_last_stage_times.argtypes = [vp, ctypes.POINTER(StageTimes)]
    L.d2pc_process_mono_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                           ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp,
                                           ctypes.c_size_t, vp, vp]
    L.d2pc_host_alloc.argtypes = [ctypes.c_size_t]
    L.d2pc_host_alloc.restype = ctypes.c_void_p
    L.d2pc_host_free.argtypes = [vp]
    L.d2pc_host_free.restype = None
    L.d2pc_pipeline_configure.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.d2pc_pipeline_acquire.argtypes = [vp, ctypes.POINTER(FrameDesc), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)]
    L.d2pc_pipeline_submit.argtypes = [vp, ctypes.c_int]
    L.d2pc_pipeline_collect.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64)]
    L.d2pc_pipeline_release.argtypes = [vp, ctypes.c_int]
    L.d2pc_reserve.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.d2pc_fuse_desc_init.argtypes = [ctypes.POINTER(FuseDesc)]
    L.d2pc_fuse_desc_init.restype = None
    L.d2pc_fuse_device.argtypes = [vp, ctypes.POINTER(FuseDesc), vp]
    L.d2pc_rotate_cw_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                        vp, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.d2pc_crop_to_square.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)] * 3
    L.d2pc_check_async_error.argtypes = [vp]
    L.d2pc_reserve_mono.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.d2pc_release_graph_buffers.argtypes = [vp]
    L.d2pc_compact_stats.argtypes = [vp, ctypes.POINTER(CompactStats)]
    L.d2pc_compact_stats_reset.argtypes = [vp]
    L.d2pc_membench_fill.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.d2pc_membench_copy.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
    L.d2pc_clock_probe_device.argtypes = [vp, vp, ctypes.c_uint32, vp]
    L.d2pc_set_tuning.argtypes = [vp, cp, ctypes.c_int]
    L.d2pc_ext_set_test_hook.argtypes = [vp, cp, ctypes.c_int]
    L.d2pc_set_reproject_form.argtypes = [vp, ctypes.c_int]
    for name in ABI_SYMBOLS + EXT_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int or fn.restype is None:
            fn.restype = ctypes.c_int
    _libs[fname] = L
    if variant is None:
        _lib = L
    return L


def abi_version() -> int:
    return load_library().d2pc_abi_version()


def status_string(status: int) -> str:
    try:
        return load_library().d2pc_status_string(int(status)).decode()
    except ImportError:
        return "?"


def device_count() -> int:
    return load_library().d2pc_device_count()


def roi_points(width: int, height: int, border: int) -> int:
    return int(load_library().d2pc_roi_points(width, height, border))


def make_q(fx=714.24, fy=713.5, cx=376.0, cy=240.0, baseline=0.09, nx=752, ny=480) -> np.ndarray:
    """hpp:66-71,84-104 defaults -> row-major 4x4 Q (host-only helper)."""
    q = np.zeros(16, dtype=np.float64)
    st = load_library().d2pc_make_q(fx, fy, cx, cy, baseline, nx, ny,
                                    q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if st:
        raise D2pcError(st, "d2pc_make_q")
    return q


STEREORECTIFY_CONTINUOUS, STEREORECTIFY_CV24, STEREORECTIFY_CV3 = 0, 1, 2


def make_q_flavour(fx=714.24, fy=713.5, cx=376.0, cy=240.0, baseline=0.09, nx=752, ny=480,
                   flavour=STEREORECTIFY_CV24) -> np.ndarray:
    """d2pc_make_q_flavour: the closed form of hpp:104 with a release's corner / centre convention."""
    q = np.zeros(16, dtype=np.float64)
    st = load_library().d2pc_make_q_flavour(ctypes.c_double(fx), ctypes.c_double(fy), ctypes.c_double(cx),
                                            ctypes.c_double(cy), ctypes.c_double(baseline), nx, ny, flavour,
                                            q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if st:
        raise D2pcError(st, "d2pc_make_q_flavour")
    return q


def make_q_disparity_image(f, T, cx, cy) -> np.ndarray:
    """Q for a stereo_msgs/DisparityImage-style source: Z = f*T/d."""
    q = np.zeros(16, dtype=np.float64)
    st = load_library().d2pc_make_q_disparity_image(f, T, cx, cy, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if st:
        raise D2pcError(st, "d2pc_make_q_disparity_image")
    return q


def calib_pack(q, border=40, mode=MODE_PARITY) -> bytes:
    """Host-only: 16 x f64 Q + border + mode -> the 136-byte broadcast blob."""
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(16))
    buf = ctypes.create_string_buffer(CALIB_BLOB_BYTES)
    st = load_library().d2pc_calib_pack(q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), border, mode, buf)
    if st:
        raise D2pcError(st, "d2pc_calib_pack")
    return buf.raw


def calib_unpack(blob: bytes):
    """Host-only: blob -> (q[16] float64, border, mode)."""
    q = np.zeros(16, dtype=np.float64)
    b, m = ctypes.c_int(), ctypes.c_int()
    buf = ctypes.create_string_buffer(bytes(blob), len(blob))
    st = load_library().d2pc_calib_unpack(buf, len(blob), q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                         ctypes.byref(b), ctypes.byref(m))
    if st:
        raise D2pcError(st, "d2pc_calib_unpack")
    return q, b.value, m.value


_NP2DT = {np.dtype(np.float32): DTYPE_F32, np.dtype(np.uint8): DTYPE_U8, np.dtype(np.uint16): DTYPE_U16}


class Context:
    """RAII wrapper of d2pc_ctx."""

    def __init__(self, device_id=0, border=40, mode=MODE_PARITY, min_disparity=-np.inf, compact_algo=0, q=None,
                 variant=None):
        self._L = load_library(variant)
        cfg = Config()
        self._check(self._L.d2pc_config_init(ctypes.byref(cfg)), None)
        cfg.device_id, cfg.border, cfg.mode = device_id, border, mode
        cfg.min_disparity, cfg.compact_algo = min_disparity, compact_algo
        h = ctypes.c_void_p()
        st = self._L.d2pc_create(ctypes.byref(cfg), ctypes.byref(h))
        if st:
            raise D2pcError(st, "d2pc_create")
        self._h = h
        if q is not None:
            self.set_q(q)

    # -- plumbing ---------------------------------------------------------
    def _check(self, st, h="self"):
        if st:
            msg = ""
            if h == "self" and getattr(self, "_h", None):
                msg = self._L.d2pc_last_error(self._h).decode()
            raise D2pcError(st, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._L.d2pc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    # -- calibration ------------------------------------------------------
    def set_q(self, q):
        q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(16))
        self._check(self._L.d2pc_set_q(self._h, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))

    def get_q(self) -> np.ndarray:
        q = np.zeros(16, dtype=np.float64)
        self._check(self._L.d2pc_get_q(self._h, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return q

    def set_border(self, border: int):
        self._check(self._L.d2pc_set_border(self._h, border))

    def set_mode(self, mode: int):
        self._check(self._L.d2pc_set_mode(self._h, mode))

    def set_min_disparity(self, min_disparity: float):
        self._check(self._L.d2pc_set_min_disparity(self._h, min_disparity))

    def config(self) -> Config:
        cfg = Config()
        self._check(self._L.d2pc_get_config(self._h, ctypes.byref(cfg)))
        return cfg

    def export_calibration(self) -> bytes:
        buf = ctypes.create_string_buffer(CALIB_BLOB_BYTES)
        self._check(self._L.d2pc_export_calibration(self._h, buf))
        return buf.raw

    def import_calibration(self, blob: bytes):
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        self._check(self._L.d2pc_import_calibration(self._h, buf, len(blob)))

    def cloud_meta(self, n_points: int) -> CloudMeta:
        m = CloudMeta()
        self._check(self._L.d2pc_cloud_meta_fill(self._h, n_points, ctypes.byref(m)))
        return m

    def set_reproject_form(self, form: int):
        """FORM_DEFAULT / FORM_CV24 / FORM_CV4: which OpenCV generation's reprojectImageTo3D arithmetic (d2pc.h)."""
        self._check(self._L.d2pc_set_reproject_form(self._h, form))

    def set_tuning(self, key: str, value: int):
        """d2pc_ext.h: launch-shape knobs; the bytes of the result never depend on them."""
        self._check(self._L.d2pc_set_tuning(self._h, key.encode(), value))

    def set_test_hook(self, key: str, value: int):
        """d2pc_ext.h: "force_general_q" / "general_q_form" -- the two hooks that DO change the arithmetic (tests only)."""
        self._check(self._L.d2pc_ext_set_test_hook(self._h, key.encode(), value))

    def reserve(self, width, height, n_frames=1):
        self._check(self._L.d2pc_reserve(self._h, width, height, n_frames))

    def check_async_error(self):
        self._check(self._L.d2pc_check_async_error(self._h))

    def reserve_mono(self, dtype, width, height, n_frames=1):
        self._check(self._L.d2pc_reserve_mono(self._h, dtype, width, height, n_frames))

    def release_graph_buffers(self):
        self._check(self._L.d2pc_release_graph_buffers(self._h))

    def compact_stats(self) -> dict:
        """d2pc_compact_stats: counters of the single-pass compaction since creation / the last reset."""
        s = CompactStats()
        s.struct_size = ctypes.sizeof(CompactStats)
        self._check(self._L.d2pc_compact_stats(self._h, ctypes.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in CompactStats._fields_ if k not in ("struct_size", "reserved")}

    def compact_stats_reset(self):
        self._check(self._L.d2pc_compact_stats_reset(self._h))

    def membench_fill(self, d_dst_ptr, nbytes, stream_ptr=None):
        self._check(self._L.d2pc_membench_fill(self._h, d_dst_ptr, nbytes, stream_ptr))

    def clock_probe(self, d_out16_ptr, min_us, stream_ptr=None):
        """d2pc_clock_probe_device: 8 x {shader cycles, 100-MHz ticks} into 16 uint64 of device memory, asynchronous."""
        self._check(self._L.d2pc_clock_probe_device(self._h, d_out16_ptr, int(min_us), stream_ptr))

    def membench_copy(self, d_src_ptr, d_dst_ptr, nbytes, stream_ptr=None):
        self._check(self._L.d2pc_membench_copy(self._h, d_src_ptr, d_dst_ptr, nbytes, stream_ptr))

    # -- hot path: host buffers (d2pc_process) ------------------------------
    def process(self, disp: np.ndarray, scale=1.0, want_index=False, capacity=None, out=None, out_index=None):
        """(H,W) numpy disparity -> ((n,4) float32 points[, (n,) uint32 index]).  `out` / `out_index`: caller
        buffers to fill (e.g. PinnedBuffer(...).array, which the kernels then store into directly; the array
        owns the pinned allocation, so no separate reference to the PinnedBuffer is needed)."""
        if disp.ndim != 2 or disp.strides[1] != disp.itemsize:
            raise ValueError("disp must be a 2-D array with contiguous rows")
        dt = _NP2DT.get(disp.dtype)
        if dt is None:
            raise D2pcError(2, f"numpy dtype {disp.dtype}")
        h, w = disp.shape
        cfg = self.config()
        cap = roi_points(w, h, cfg.border) if capacity is None else capacity
        if out is None:
            out = np.empty((max(cap, 1), 4), dtype=np.float32)
        else:
            assert out.dtype == np.float32 and out.flags.c_contiguous and out.size >= 4 * cap
            want_index = want_index or out_index is not None
        if want_index and out_index is not None:
            assert out_index.dtype == np.uint32 and out_index.flags.c_contiguous and out_index.size >= cap
            idx = out_index
        else:
            idx = np.empty(max(cap, 1), dtype=np.uint32) if want_index else None
        n = ctypes.c_size_t(0)
        st = self._L.d2pc_process(self._h, disp.ctypes.data, dt, scale, w, h, disp.strides[0], out.ctypes.data,
                                  idx.ctypes.data if want_index else None, cap, ctypes.byref(n))
        self._check(st)
        if want_index:
            return out[: n.value], idx[: n.value]
        return out[: n.value]

    def process_mono8(self, image: np.ndarray, median_ksize=11, scale=0.125, want_index=False, capacity=None):
        """cpp:55-85 for one mono8 frame: device median -> x scale -> points."""
        if image.ndim != 2 or image.dtype != np.uint8 or image.strides[1] != 1:
            raise ValueError("image must be a 2-D uint8 array with contiguous rows")
        h, w = image.shape
        cfg = self.config()
        cap = roi_points(w, h, cfg.border) if capacity is None else capacity
        out = np.empty((max(cap, 1), 4), dtype=np.float32)
        idx = np.empty(max(cap, 1), dtype=np.uint32) if want_index else None
        n = ctypes.c_size_t(0)
        st = self._L.d2pc_process_mono8(self._h, image.ctypes.data, w, h, image.strides[0], median_ksize, scale,
                                        out.ctypes.data, idx.ctypes.data if want_index else None, cap, ctypes.byref(n))
        self._check(st)
        return (out[: n.value], idx[: n.value]) if want_index else out[: n.value]

    def process_mono16(self, image: np.ndarray, median_ksize=11, scale=0.125, want_index=False, capacity=None):
        """cpp:50-85 for one mono16 frame: device cv_bridge rescale -> median -> x scale -> points."""
        assert image.dtype == np.uint16 and image.ndim == 2 and image.strides[1] == 2
        h, w = image.shape
        cfg = self.config()
        cap = roi_points(w, h, cfg.border) if capacity is None else capacity
        out = np.empty((max(cap, 1), 4), dtype=np.float32)
        idx = np.empty(max(cap, 1), dtype=np.uint32) if want_index else None
        n = ctypes.c_size_t(0)
        st = self._L.d2pc_process_mono16(self._h, image.ctypes.data, w, h, image.strides[0], median_ksize, scale,
                                         out.ctypes.data, idx.ctypes.data if want_index else None, cap, ctypes.byref(n))
        self._check(st)
        return (out[: n.value], idx[: n.value]) if want_index else out[: n.value]

    def mono16_to_mono8_device(self, d_src_ptr, width, height, src_row_stride, src_frame_stride, n_frames, d_dst_ptr,
                               dst_row_stride, dst_frame_stride, stream_ptr=None):
        self._check(self._L.d2pc_mono16_to_mono8_device(self._h, d_src_ptr, width, height, src_row_stride,
                                                        src_frame_stride, n_frames, d_dst_ptr, dst_row_stride,
                                                        dst_frame_stride, stream_ptr))

    def last_stage_times(self) -> dict:
        """Times of the last synchronous host call made with set_tuning("stage_timing", 1)."""
        t = StageTimes()
        self._check(self._L.d2pc_last_stage_times(self._h, ctypes.byref(t)))
        return {k: getattr(t, k) for k, _ in StageTimes._fields_}

    def median_device(self, d_src_ptr, width, height, src_row_stride, src_frame_stride, n_frames, d_dst_ptr,
                      dst_row_stride, dst_frame_stride, ksize=11, stream_ptr=None):
        self._check(self._L.d2pc_median_device(self._h, d_src_ptr, width, height, src_row_stride, src_frame_stride,
                                               n_frames, d_dst_ptr, dst_row_stride, dst_frame_stride, ksize,
                                               stream_ptr))

    def median_roi_device(self, d_src_ptr, width, height, src_row_stride, src_frame_stride, n_frames, d_dst_ptr,
                          dst_row_stride, dst_frame_stride, ksize=11, stream_ptr=None):
        """d2pc_median_roi_device: the same filter, computed for the context's inset ROI only."""
        self._check(self._L.d2pc_median_roi_device(self._h, d_src_ptr, width, height, src_row_stride,
                                                   src_frame_stride, n_frames, d_dst_ptr, dst_row_stride,
                                                   dst_frame_stride, ksize, stream_ptr))

    def process_mono_device(self, d_image_ptr, dtype, width, height, row_stride, frame_stride, n_frames, median_ksize,
                            scale, d_out_ptr, d_index_ptr, out_frame_stride_points, d_counts_ptr, stream_ptr=None):
        """d2pc_process_mono_device: (rescale ->) median(ROI) -> reproject for a device-resident batch.  PARITY
        mode and a launch of >= 448 tiles (192 / 320 for 3x3 / 5x5): one kernel, tile by tile (bit-sliced median + the tile's
        points); otherwise the filter launch followed by the reprojection launch (tuning "callback_fused")."""
        self._check(self._L.d2pc_process_mono_device(self._h, d_image_ptr, dtype, width, height, row_stride,
                                                     frame_stride, n_frames, median_ksize, scale, d_out_ptr,
                                                     d_index_ptr, out_frame_stride_points, d_counts_ptr, stream_ptr))

    def fuse_device(self, desc: "FuseDesc", stream_ptr=None):
        """d2pc_fuse_device: fusion rule + combined confidence + 3x3 median + crop on device planes."""
        self._check(self._L.d2pc_fuse_device(self._h, ctypes.byref(desc), stream_ptr))

    def rotate_cw_device(self, d_src_ptr, cols, rows, src_pitch, src_frame_stride, n_frames, d_dst_ptr, dst_pitch,
                         dst_frame_stride, stream_ptr=None):
        """d2pc_rotate_cw_device: dst(i, j) = src(rows-1-j, i) for 8-bit device frames."""
        self._check(self._L.d2pc_rotate_cw_device(self._h, d_src_ptr, cols, rows, src_pitch, src_frame_stride, n_frames,
                                                  d_dst_ptr, dst_pitch, dst_frame_stride, stream_ptr))

    # -- pipelined host path (d2pc_pipeline_*) ------------------------------
    def pipeline_configure(self, depth=3, direct_host_write=False):
        self._check(self._L.d2pc_pipeline_configure(self._h, depth, int(direct_host_write)))

    def pipeline_submit(self, image: np.ndarray, scale=1.0, median_ksize=0, want_index=False, tag=0,
                        mono16=False) -> int:
        """Acquire a slot, copy `image` into its pinned input buffer (a real
        producer would decode straight into it) and submit.  Returns the slot.
        mono16: a uint16 image that cv_bridge would rescale to mono8 first (DTYPE_MONO16)."""
        dt = DTYPE_MONO16 if mono16 else _NP2DT[image.dtype]
        h, w = image.shape
        desc = FrameDesc(dt, scale, w, h, w * image.itemsize, median_ksize, int(want_index), tag)
        host_in, slot = ctypes.c_void_p(), ctypes.c_int()
        self._check(self._L.d2pc_pipeline_acquire(self._h, ctypes.byref(desc), ctypes.byref(host_in), ctypes.byref(slot)))
        dst = np.ctypeslib.as_array(ctypes.cast(host_in, ctypes.POINTER(ctypes.c_uint8)), shape=(h * w * image.itemsize,))
        dst[:] = np.ascontiguousarray(image).view(np.uint8).reshape(-1)
        self._check(self._L.d2pc_pipeline_submit(self._h, slot.value))
        return slot.value

    def pipeline_collect(self, copy=True):
        """Oldest submitted frame -> (points, index or None, tag, slot).  With
        copy=False the arrays are VIEWS of the pinned buffers, valid until
        pipeline_release(slot)."""
        slot, pts, idx = ctypes.c_int(-1), ctypes.c_void_p(), ctypes.c_void_p()
        n, tag = ctypes.c_size_t(), ctypes.c_uint64()
        st = self._L.d2pc_pipeline_collect(self._h, ctypes.byref(slot), ctypes.byref(pts), ctypes.byref(idx),
                                           ctypes.byref(n), ctypes.byref(tag))
        if st:
            msg = self._L.d2pc_last_error(self._h).decode()
            if slot.value >= 0:  # the frame is lost but its slot was handed back: free it before raising
                self._L.d2pc_pipeline_release(self._h, slot.value)
            raise D2pcError(st, msg)
        if n.value:
            p = np.ctypeslib.as_array(ctypes.cast(pts, ctypes.POINTER(ctypes.c_float)), shape=(n.value, 4))
            i = (np.ctypeslib.as_array(ctypes.cast(idx, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value,))
                 if idx.value else None)
        else:
            p, i = np.empty((0, 4), np.float32), (np.empty(0, np.uint32) if idx.value else None)
        if copy:
            p, i = p.copy(), (i.copy() if i is not None else None)
            self.pipeline_release(slot.value)
        return p, i, tag.value, slot.value

    def pipeline_release(self, slot: int):
        self._check(self._L.d2pc_pipeline_release(self._h, slot))

    # -- hot path: device-resident batch (d2pc_process_device) --------------
    def process_device(self, d_disp_ptr, dtype, scale, width, height, row_stride, in_frame_stride, n_frames,
                       d_out_ptr, d_index_ptr, out_frame_stride_points, d_counts_ptr, stream_ptr=None):
        st = self._L.d2pc_process_device(self._h, d_disp_ptr, dtype, scale, width, height, row_stride,
                                         in_frame_stride, n_frames, d_out_ptr, d_index_ptr,
                                         out_frame_stride_points, d_counts_ptr, stream_ptr)
        self._check(st)


def fuse_desc_init() -> FuseDesc:
    d = FuseDesc()
    load_library().d2pc_fuse_desc_init(ctypes.byref(d))
    return d


def crop_to_square(cols, rows, offset_x=0, offset_y=0, member_offset_y=None):
    """d2pc_crop_to_square -> (x, y, n); raises D2pcError when the square leaves the image."""
    x, y, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    st = load_library().d2pc_crop_to_square(cols, rows, offset_x, offset_y,
                                   offset_y if member_offset_y is None else member_offset_y,
                                   ctypes.byref(x), ctypes.byref(y), ctypes.byref(n))
    if st != 0:
        raise D2pcError(st, "crop_to_square(%d,%d,%d,%d)" % (cols, rows, offset_x, offset_y))
    return x.value, y.value, n.value
