#!/usr/bin/env python3
"""Per-frame latency of the synchronous host entry points at the reference's
native geometry (what DisparityCb would call once per camera frame)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
q = d2pc.make_q()
L = d2pc.load_library()
rng = np.random.default_rng(0)
pin_in = len(sys.argv) > 1 and sys.argv[1] == "pinned_io"   # input frame in pinned memory as well: read in place
if pin_in: sys.argv[1] = "pinned"
for (w, h) in ((752, 480), (640, 480), (1920, 1080), (3840, 2160)):
    img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    f32 = img.astype(np.float32) * np.float32(0.125)
    if pin_in:
        keep_in = (d2pc.PinnedBuffer((h, w), np.uint8), d2pc.PinnedBuffer((h, w), np.float32))
        keep_in[0].array[:] = img; keep_in[1].array[:] = f32
        img, f32 = keep_in[0].array, keep_in[1].array
    for mode in (d2pc.MODE_PARITY, d2pc.MODE_COMPACT):
        with d2pc.Context(q=q, mode=mode) as ctx:
            cap = d2pc.roi_points(w, h, 40)
            pinned = len(sys.argv) > 1 and sys.argv[1] == "pinned"  # output buffer from d2pc_host_alloc
            keep = d2pc.PinnedBuffer((cap, 4), np.float32) if pinned else None
            out = keep.array if pinned else np.empty((cap, 4), dtype=np.float32)
            n = ctypes.c_size_t()
            def mono8(k):
                assert L.d2pc_process_mono8(ctx.handle, img.ctypes.data, w, h, w, k, 0.125, out.ctypes.data, None, cap, ctypes.byref(n)) == 0
            def fp32():
                assert L.d2pc_process(ctx.handle, f32.ctypes.data, 0, 1.0, w, h, w * 4, out.ctypes.data, None, cap, ctypes.byref(n)) == 0
            for name, fn in (("process_mono8 median11", lambda: mono8(11)), ("process_mono8 no median", lambda: mono8(0)), ("process fp32", fp32)):
                for _ in range(20): fn()
                ts = []
                for _ in range(200 if w < 3000 else 40):
                    t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
                ts = np.array(ts) * 1e6
                print(f"{w}x{h} {'pinned io' if pin_in else 'pinned  ' if pinned else 'pageable'} {'parity ' if mode == 0 else 'compact'} {name:24s}: median {np.median(ts):7.1f} us  p10 {np.percentile(ts,10):7.1f}  p90 {np.percentile(ts,90):7.1f}  ({n.value} points)", flush=True)
