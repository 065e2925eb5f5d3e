#!/usr/bin/env python3
"""Diagnostic: where a block of k_callback_bs<11> (the callback body tile by tile: bit-sliced median of a 256 x 32 tile, then its
points) and of k_callback_bs_compact_pipe<11> (the same with the ordered compaction) spends its cycles.  Needs `make -C disparity_to_point_cloud_amd/csrc diag` (s_memtime stamps at the stage barriers,
summed by lane 0 of the block's first and last wave; in the product no stamp executes).  GPU box only.

  python tools/diag_callback.py > profiles/rNN_callback_phases.txt"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

VARIANT = sys.argv[1] if len(sys.argv) > 1 else "diag"
lib = d2pc.load_library(VARIANT)
lib.d2pc_debug_read_bs_diag.argtypes = [ctypes.c_void_p]
W, H, F = 3840, 2160, 16
gen = torch.Generator(device="cuda").manual_seed(3)
raw = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device="cuda", generator=gen)
blocky = raw.clone()
m = torch.rand((F, (H + 63) // 64, (W + 63) // 64), device="cuda", generator=gen) < 0.3
blocky[m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H, :W]] = 0
s = torch.cuda.current_stream().cuda_stream
N = 10
FILTER = ["rows requested -> staged in LDS", "plane words", "the select", "plane words -> bytes"]


def measure(mode, src, idx):
    ctx = d2pc.Context(q=d2pc.make_q(), mode=mode, variant=VARIANT)
    b = DeviceBatch(ctx, F, H, W, dtype=torch.uint8, want_index=idx)
    run = lambda: ctx.process_mono_device(src.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, b.points.data_ptr(),
                                          b.index.data_ptr() if idx else None, b.stride, b.counts.data_ptr(), s)
    for _ in range(40):
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert lib.d2pc_debug_read_bs_diag(buf) == 0   # (reads and resets)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        run()
    e1.record(); torch.cuda.synchronize()
    assert lib.d2pc_debug_read_bs_diag(buf) == 0
    ctx.close()
    return [buf[i] for i in range(32)], e0.elapsed_time(e1) / N * 1e3


def table(v, names, slots):
    tiles = max(v[0], 1)
    for wave, off in (("first wave", 0), ("last wave ", 16)):
        t = [v[off + i] / tiles for i in slots]
        print(f"{wave}: " + "  ".join(f"{n} {x:7.0f}" for n, x in zip(names, t)) + f"   sum {sum(t):7.0f}" +
              (f"   block lifetime per tile {v[6] / tiles:7.0f} cycles" if off == 0 else ""))
    t = [v[i] / tiles for i in slots]
    print("shares (first wave): " + "  ".join(f"{100 * x / sum(t):4.1f} %" for x in t))


v, us = measure(d2pc.MODE_PARITY, raw, False)
print(f"# {torch.cuda.get_device_name(0)}; k_callback_bs<11, stereo>, 16 x 4K u8, {N} launches, {max(v[0], 1) // N} tiles per launch; "
      f"kernel {us:.1f} us per launch under the stamps; block clock {v[6] / max(v[7], 1) * 0.1:.2f} GHz")
table(v, FILTER + ["epilogue"], [1, 2, 3, 4, 5])
v, us = measure(d2pc.MODE_COMPACT, blocky, True)
print(f"# k_callback_bs_compact_pipe<11, stereo>, 30 % zero pixels in 64 x 64 blocks, + indices; kernel {us:.1f} us per launch under the "
      f"stamps; block clock {v[6] / max(v[7], 1) * 0.1:.2f} GHz (persistent blocks: cycles per TILE of the block)")
table(v, FILTER + ["table + barrier", "count + barrier", "publish + place (+ barrier)", "scatter", "tile kept + barriers"], [1, 2, 3, 4, 5, 8, 9, 10, 11])
v, us = measure(d2pc.MODE_COMPACT, blocky, False)
print(f"# the same without indices; kernel {us:.1f} us per launch under the stamps; block clock {v[6] / max(v[7], 1) * 0.1:.2f} GHz")
table(v, FILTER + ["table + barrier", "count + barrier", "publish + place (+ barrier)", "scatter", "tile kept + barriers"], [1, 2, 3, 4, 5, 8, 9, 10, 11])
