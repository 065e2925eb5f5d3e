import ctypes, sys, numpy as np
sys.path.insert(0, '/root/repo')
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.capi import StageTimes
L = d2pc.load_library()
rng = np.random.default_rng(0)
w, h = 752, 480
img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
cap = d2pc.roi_points(w, h, 40)
for pinned in (False, True):
    keep = d2pc.PinnedBuffer((cap, 4), np.float32) if pinned else None
    out = keep.array if pinned else np.empty((cap, 4), dtype=np.float32)
    pin_in = d2pc.PinnedBuffer((h, w), np.uint8); pin_in.array[:] = img
    for src, name in ((img, "pageable in"), (pin_in.array, "pinned in")):
        with d2pc.Context(q=d2pc.make_q()) as ctx:
            ctx.set_tuning("stage_timing", 1)
            n = ctypes.c_size_t()
            acc = np.zeros(5)
            for i in range(120):
                assert L.d2pc_process_mono8(ctx.handle, src.ctypes.data, w, h, w, 11, 0.125, out.ctypes.data, None, cap, ctypes.byref(n)) == 0
                t = StageTimes(); assert L.d2pc_last_stage_times(ctx.handle, ctypes.byref(t)) == 0
                if i >= 20: acc += [t.h2d_ms, t.prep_ms, t.kernel_ms, t.d2h_ms, t.total_ms]
            acc /= 100
            print(f"{'pinned out' if pinned else 'pageable out'}, {name}: h2d {acc[0]*1e3:.1f} us, median {acc[1]*1e3:.1f}, reproject(+direct write) {acc[2]*1e3:.1f}, d2h {acc[3]*1e3:.1f}, total(device) {acc[4]*1e3:.1f}")
