"""CPU tests of the C-ABI surface: the library loads, exports every symbol
include/d2pc.h declares, and its host-only entry points behave.  No compute
calls (there is no GPU here and the library has no CPU path)."""
import ctypes
import os
import re

import numpy as np
import pytest

import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(name="d2pc.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d2pc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    """Both headers against the library and the binding: include/d2pc.h is the stable drop-in surface, include/d2pc_ext.h
    the unstable one for bench / tools / tests."""
    lib = d2pc.load_library()
    stable, ext = _header_functions("d2pc.h"), _header_functions("d2pc_ext.h")
    assert len(stable) >= 20 and len(ext) >= 8 and not set(stable) & set(ext)
    for name in stable + ext:
        assert hasattr(lib, name), f"libd2pc.so does not export {name}"
    assert sorted(capi.ABI_SYMBOLS) == stable, "python binding and d2pc.h disagree"
    assert sorted(capi.EXT_SYMBOLS) == ext, "python binding and d2pc_ext.h disagree"


def test_library_exports_nothing_undeclared():
    """Every d2pc_* symbol the shared object exports is declared in one of the two headers (no hidden surface)."""
    import subprocess
    so = os.path.join(ROOT, "disparity_to_point_cloud_amd", "libd2pc.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("d2pc_")
                       and ln.split()[-2] in ("T", "W")})
    assert exported == sorted(_header_functions("d2pc.h") + _header_functions("d2pc_ext.h"))


def test_stable_header_holds_no_tool_surface():
    """What round 3's verdict asked to move out: no tuning strings, bench kernels, counters or graph plumbing in d2pc.h;
    the adaptor and the host mirror include d2pc_ext.h only where they print stage times."""
    stable = _header_functions("d2pc.h")
    for name in ("d2pc_set_tuning", "d2pc_membench_fill", "d2pc_membench_copy", "d2pc_compact_stats", "d2pc_reserve",
                 "d2pc_reserve_mono", "d2pc_release_graph_buffers", "d2pc_last_stage_times", "d2pc_ext_set_test_hook"):
        assert name not in stable
    assert "d2pc_set_reproject_form" in stable
    # ... and nothing of the laboratory (round 4's verdict): the stable header names neither the experiment build's
    # algorithm nor any of its keys
    text = open(os.path.join(ROOT, "include", "d2pc.h")).read().lower()
    for word in ("chunk", "big_batch", "unbounded", "experiment"):
        assert word not in text, word
    lib = d2pc.load_library()
    assert lib.d2pc_ext_revision() >= 5
    for rel in ("ros/disparity_to_point_cloud_node.cpp", "host/multi_gpu.hpp", "host/image_prep.hpp", "host/pinned_allocator.hpp",
                "host/replay_main.cpp"):
        assert "d2pc_ext.h" not in open(os.path.join(ROOT, rel)).read(), rel
    ext_calls = set(re.findall(r"\b(d2pc_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "host", "disparity_to_point_cloud_amd.hpp")).read()))
    assert ext_calls & set(capi.EXT_SYMBOLS) <= {"d2pc_set_tuning", "d2pc_last_stage_times"}


def test_product_library_carries_no_laboratory():
    """The shipped libd2pc.so holds the product's kernels only (< 6 MB); the chunked two-pass, the tile-walking PARITY
    kernel and the 1,024- / 4,096-pixel tile shapes are instantiated in the experiment build (`make exp`) alone."""
    import subprocess

    def kernels(path):
        out = subprocess.run(["strings", "-n", "12", path], capture_output=True, text=True, check=True).stdout
        return set(re.findall(r"_ZN4d2pc\w+", out))

    prod_path = capi.library_path()
    exp_path = capi.library_path("exp")
    assert os.path.getsize(prod_path) < 6 * 1024 * 1024, os.path.getsize(prod_path)
    prod = kernels(prod_path)
    assert prod, "no kernel names found"
    assert not [k for k in prod if "k_compact_chunk" in k or "k_chunk_clear" in k]
    assert not [k for k in prod if "16k_reproject_packI" in k]          # the tile-walking kernel
    assert [k for k in prod if "k_reproject_pack_small" in k] and [k for k in prod if "k_compact_onepass" in k]
    if os.path.exists(exp_path):
        exp = kernels(exp_path)
        assert [k for k in exp if "k_compact_chunk" in k] and [k for k in exp if "16k_reproject_packI" in k]
        assert prod <= exp, sorted(prod - exp)[:5]


def test_abi_version_and_status_strings():
    assert d2pc.abi_version() == capi.ABI_VERSION == 2
    assert d2pc.status_string(0) == "ok"
    for s in range(1, 10):
        assert d2pc.status_string(s) not in ("", "ok", "unknown status")
    assert d2pc.status_string(99) == "unknown status"


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Config) == 40
    cfg = capi.Config()
    assert d2pc.load_library().d2pc_config_init(ctypes.byref(cfg)) == 0
    # reference defaults: border 40 (cpp:70,72), unfiltered output (cpp:81)
    assert (cfg.struct_size, cfg.border, cfg.mode, cfg.device_id) == (40, 40, d2pc.MODE_PARITY, 0)
    assert cfg.min_disparity == -np.inf
    assert ctypes.sizeof(capi.Field) == 20 and ctypes.sizeof(capi.CloudMeta) == 84


def test_make_q_matches_oracle_bitwise():
    import oracle

    for kw in (dict(), dict(fx=500.0, fy=510.0, cx=300.5, cy=200.25, baseline=0.043, nx=640, ny=480)):
        a = d2pc.make_q(**kw)
        b = oracle.make_q(**kw)
        assert a.tobytes() == b.tobytes()
    q = d2pc.make_q()
    assert np.signbit(q[15]) and q[15] == 0.0  # Q[3][3] = -0.0
    lib = d2pc.load_library()
    assert lib.d2pc_make_q(0.0, 1.0, 0.0, 0.0, 0.1, 10, 10, q.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 1
    assert lib.d2pc_make_q(1.0, 1.0, 0.0, 0.0, 0.1, 10, 10, None) == 1


def test_roi_points():
    assert d2pc.roi_points(752, 480, 40) == 672 * 400  # native geometry, BASELINE.md
    assert d2pc.roi_points(640, 480, 40) == 224000
    assert d2pc.roi_points(1920, 1080, 40) == 1840000
    assert d2pc.roi_points(3840, 2160, 40) == 7820800
    assert d2pc.roi_points(80, 480, 40) == 0 and d2pc.roi_points(81, 81, 40) == 1
    assert d2pc.roi_points(10, 10, -1) == 0


def test_create_fails_loudly_without_gpu_or_with_bad_config():
    lib = d2pc.load_library()
    h = ctypes.c_void_p()
    cfg = capi.Config()
    lib.d2pc_config_init(ctypes.byref(cfg))
    assert lib.d2pc_create(None, ctypes.byref(h)) == 1
    assert lib.d2pc_create(ctypes.byref(cfg), None) == 1
    bad = capi.Config.from_buffer_copy(cfg)
    bad.struct_size = 12
    assert lib.d2pc_create(ctypes.byref(bad), ctypes.byref(h)) == 1
    bad = capi.Config.from_buffer_copy(cfg)
    bad.mode = 7
    assert lib.d2pc_create(ctypes.byref(bad), ctypes.byref(h)) == 1
    bad = capi.Config.from_buffer_copy(cfg)
    bad.border = -1
    assert lib.d2pc_create(ctypes.byref(bad), ctypes.byref(h)) == 1
    if d2pc.device_count() == 0:
        # no CPU fallback: a valid request without a GPU is an error, not a slow path
        assert lib.d2pc_create(ctypes.byref(cfg), ctypes.byref(h)) == 5
        with pytest.raises(d2pc.D2pcError) as e:
            d2pc.Context()
        assert e.value.status == 5
    bad = capi.Config.from_buffer_copy(cfg)
    bad.device_id = 1 << 20
    assert lib.d2pc_create(ctypes.byref(bad), ctypes.byref(h)) == 5


def test_null_context_calls_return_invalid_arg():
    lib = d2pc.load_library()
    q = np.zeros(16)
    qp = q.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    n = ctypes.c_size_t()
    assert lib.d2pc_set_q(None, qp) == 1
    assert lib.d2pc_get_q(None, qp) == 1
    assert lib.d2pc_destroy(None) == 1
    assert lib.d2pc_set_border(None, 3) == 1 and lib.d2pc_set_mode(None, 0) == 1
    assert lib.d2pc_process(None, None, 0, 1.0, 1, 1, 4, None, None, 0, ctypes.byref(n)) == 1
    assert lib.d2pc_process_device(None, None, 0, 1.0, 1, 1, 4, 4, 1, None, None, 0, None, None) == 1
    assert lib.d2pc_reserve(None, 1, 1, 1) == 1 and lib.d2pc_check_async_error(None) == 1
    assert lib.d2pc_last_error(None) == b"null context"


def test_fastdiv_reference_model():
    """The kernels' exact division (csrc/d2pc_device.hpp make_fastdiv/fdiv),
    modelled in python: q == n // d for all probed 32-bit n."""
    rng = np.random.default_rng(1)

    def mk(d):
        l = 0
        while (1 << l) < d:
            l += 1
        m = ((1 << 32) * ((1 << l) - d)) // d + 1
        return m & 0xFFFFFFFF, min(l, 1), max(l - 1, 0)

    ds = [1, 2, 3, 5, 7, 16, 560, 672, 1840, 3760, 3840, 65535, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]
    ds += list(rng.integers(1, 2 ** 32, size=200))
    for d in ds:
        d = int(d)
        m, s1, s2 = mk(d)
        ns = np.concatenate([rng.integers(0, 2 ** 32, size=2000, dtype=np.uint64),
                             np.array([0, 1, d - 1, d, d + 1, 2 ** 32 - 1, (2 ** 32 - 1) // d * d], dtype=np.uint64)])
        ns = ns[ns < 2 ** 32]
        t = (ns * np.uint64(m)) >> np.uint64(32)
        q = ((t + ((ns - t) >> np.uint64(s1))) & np.uint64(0xFFFFFFFF)) >> np.uint64(s2)
        assert np.array_equal(q, ns // np.uint64(d)), d


def test_make_q_disparity_image():
    q = d2pc.make_q_disparity_image(700.0, 0.12, 320.5, 240.25)
    want = np.zeros(16)
    want[[0, 5]] = 1
    want[3], want[7], want[11], want[14] = -320.5, -240.25, 700.0, 1 / 0.12
    assert np.array_equal(q, want)
    lib = d2pc.load_library()
    qp = q.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.d2pc_make_q_disparity_image(0.0, 0.1, 1.0, 1.0, qp) == 1
    assert lib.d2pc_make_q_disparity_image(1.0, -0.1, 1.0, 1.0, qp) == 1
    assert lib.d2pc_set_min_disparity(None, 1.0) == 1


def test_crop_to_square_matches_oracle_without_a_gpu():
    """d2pc_crop_to_square is host arithmetic (cropToSquare, src/depth_map_fusion.cpp:247-265)."""
    import oracle
    for cols, rows, ox, oy, my in ((752, 480, -7, 15, 15), (480, 752, 7, -15, 15), (640, 480, 0, 0, 0),
                                   (480, 640, 0, 0, 0), (100, 100, 3, -4, 4), (100, 100, -3, 4, 4), (65, 33, 1, 1, 1)):
        assert d2pc.crop_to_square(cols, rows, ox, oy, my) == oracle.crop_to_square(cols, rows, ox, oy, my)
    with pytest.raises(d2pc.D2pcError):       # the square leaves the image: cv::Mat(Rect) would assert
        d2pc.crop_to_square(100, 100, 0, 20, 0)   # member offset_y_ stale: n = 100 but x = 10
    d = d2pc.fuse_desc_init()
    assert (d.struct_size, d.rule, d.n_frames) == (ctypes.sizeof(d2pc.FuseDesc), d2pc.FUSE_GRAD_FILTER, 1)
    assert (d.crop_left, d.crop_right, d.crop_top, d.crop_bottom) == (0, 40, 30, 10)   # cpp:130


def test_public_header_is_plain_c99_and_cxx11(tmp_path):
    """The boundary is a C ABI: include/d2pc.h must compile as strict C99 (and C++11) with no warnings."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "d2pc.h"\n#include "d2pc_ext.h"\nint main(void) { d2pc_config c; d2pc_fuse_desc f; d2pc_frame_desc d; '
                   'd2pc_stage_times t; d2pc_cloud_meta m; d2pc_compact_stats_t s; (void)c; (void)f; (void)d; (void)t; (void)m; (void)s; '
                   'return D2PC_ABI_VERSION == 2 && D2PC_EXT_REVISION >= 4 ? 0 : 1; }\n')
    inc = os.path.join(root, "include")
    for cmd in (["gcc", "-std=c99"], ["g++", "-std=c++11", "-x", "c++"]):
        p = subprocess.run(cmd + ["-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(src), "-o",
                                  str(tmp_path / "hdr.o")], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr


def test_ros_adaptor_parses():
    """SYNTAX CHECK ONLY: ros/disparity_to_point_cloud_node.cpp cannot be built here (no ROS, OpenCV, cv_bridge in
    the image); it is parsed and type-checked against the declaration-only stubs under tests/stubs/ so that typos,
    missing members and template errors in the adaptor are caught.  Nothing is linked or run, and the stubs pin no
    behaviour of ROS, OpenCV or the reference."""
    import subprocess

    src = os.path.join(ROOT, "ros", "disparity_to_point_cloud_node.cpp")
    p = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I",
                        os.path.join(ROOT, "tests", "stubs"), src], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-3000:]


def test_stereorectify_flavours():
    """d2pc_make_q_flavour: where the new principal point lands depends on the OpenCV release; none of the
    three conventions is claimed to be a release's Q bit for bit (OpenCV computes part of it in float)."""
    d = dict(fx=714.24, fy=713.5, cx=376.0, cy=240.0, baseline=0.09, nx=752, ny=480)
    cont = d2pc.make_q_flavour(flavour=d2pc.STEREORECTIFY_CONTINUOUS, **d)
    cv24 = d2pc.make_q_flavour(flavour=d2pc.STEREORECTIFY_CV24, **d)
    cv3 = d2pc.make_q_flavour(flavour=d2pc.STEREORECTIFY_CV3, **d)
    assert cont.tobytes() == d2pc.make_q(**d).tobytes()             # d2pc_make_q is "this closed form"
    assert abs(-cont[3] - 375.999481966846) < 1e-9                  # SURVEY.md row a9's constant: the CONTINUOUS form
    assert -cv24[3] == 376.0 and -cv24[7] == 240.0                  # 2.4: corners 0..n, integer n/2
    assert abs(-cv3[3] - (375.0 - 713.5 * (375.5 - 376.0) / 714.24)) < 1e-12 and abs(-cv3[3] - 375.4995) < 1e-4
    for q in (cont, cv24, cv3):                                     # everything else is common
        assert q[11] == 713.5 and abs(q[14] - 1 / 0.09) < 1e-12 and q[15] == 0.0 and np.signbit(q[15])
    with pytest.raises(d2pc.D2pcError):
        d2pc.make_q_flavour(flavour=7, **d)


def test_pinned_buffer_memory_lives_as_long_as_any_view(monkeypatch):
    """capi.PinnedBuffer: the numpy views own the allocation (advisor, round 2: a temporary
    PinnedBuffer(...).array used to be freed at once and handed d2pc_process a dangling pointer).
    Counting stand-ins replace the two ABI calls -- there is no GPU here."""
    import gc

    class _Lib:
        def __init__(self):
            self.blocks, self.freed = {}, []

        def d2pc_host_alloc(self, n):
            b = ctypes.create_string_buffer(n)
            self.blocks[ctypes.addressof(b)] = b
            return ctypes.addressof(b)

        def d2pc_host_free(self, p):
            self.freed.append(p)

    lib = _Lib()
    monkeypatch.setattr(capi, "load_library", lambda: lib)
    view = capi.PinnedBuffer((4, 4), np.float32).array  # the owner object is a temporary
    addr = view.ctypes.data
    gc.collect()
    assert lib.freed == [], "memory freed while a view of it is alive"
    view[:] = 7.0  # still writable memory of the stand-in allocation
    assert np.frombuffer(lib.blocks[addr], dtype=np.float32, count=16).tolist() == [7.0] * 16
    tail = view[2:]
    del view
    gc.collect()
    assert lib.freed == [], "a slice keeps the allocation too"
    del tail
    gc.collect()
    assert lib.freed == [addr]
    # close() only drops the owner's reference
    pb = capi.PinnedBuffer(8, np.uint32)
    keep = pb.array
    pb.close()
    gc.collect()
    assert pb.ptr is None and pb.array is None and len(lib.freed) == 1
    del keep
    gc.collect()
    assert len(lib.freed) == 2 and not pb.alive
