#!/usr/bin/env python3
"""Diagnostic: where a block of k_callback_bs<11> (the callback body tile by tile: bit-sliced median of a 256 x 32 tile, then its
points) spends its cycles.  Needs `make -C disparity_to_point_cloud_amd/csrc diag` (s_memtime stamps at the stage barriers,
summed by lane 0 of the block's first and last wave; in the product no stamp executes).  GPU box only.

  python tools/diag_callback.py > profiles/rNN_callback_phases.txt"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

lib = d2pc.load_library("diag")
lib.d2pc_debug_read_bs_diag.argtypes = [ctypes.c_void_p]
W, H, F = 3840, 2160, 16
ctx = d2pc.Context(q=d2pc.make_q(), variant="diag")
raw = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
b = DeviceBatch(ctx, F, H, W, dtype=torch.uint8)
s = torch.cuda.current_stream().cuda_stream
run = lambda: ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, b.points.data_ptr(), None,
                                      b.stride, b.counts.data_ptr(), s)
for _ in range(40):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
assert lib.d2pc_debug_read_bs_diag(buf) == 0   # (reads and resets)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 10
e0.record()
for _ in range(N):
    run()
e1.record(); torch.cuda.synchronize()
assert lib.d2pc_debug_read_bs_diag(buf) == 0
v = [buf[i] for i in range(16)]
tiles = max(v[0], 1)
clock = v[6] / max(v[14], 1) * 0.1
print(f"# {torch.cuda.get_device_name(0)}; k_callback_bs<11, stereo>, 16 x 4K u8, {N} launches, {tiles // N} tiles per launch; "
      f"kernel {e0.elapsed_time(e1) / N * 1e3:.1f} us per launch under the stamps; block clock {clock:.2f} GHz")
names = ["rows requested -> staged in LDS (the global loads' round trip)", "plane words (gather + bit transposes)",
         "the select (8 planes x count + update)", "plane words -> bytes", "epilogue: table of 1/W, points, stores"]
for wave, off in (("first wave", 0), ("last wave ", 7)):
    t = [v[off + 1 + i] / tiles for i in range(5)]
    life = v[6] / tiles if off == 0 else sum(t)
    print(f"{wave}: " + "  ".join(f"{n.split(':')[0].split('(')[0].strip()} {x:7.0f}" for n, x in zip(names, t)) +
          f"   sum {sum(t):7.0f}" + (f"   block lifetime {life:7.0f} cycles" if off == 0 else ""))
t = [v[1 + i] / tiles for i in range(5)]
print("shares (first wave): " + "  ".join(f"{100 * x / sum(t):4.1f} %" for x in t))
