#!/usr/bin/env python3
"""Time d2pc_fuse_device (fusion rule + combined confidence + 3x3 median + crop)
on batches of 8-bit plane sets.  Algorithmic bytes per pixel: 6 read (5 when
score1 aliases grad1, as in the reference) + 1 combined + the cropped fused
image.  GPU only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd import capi
if len(sys.argv) > 1:  # tuning build: make -C disparity_to_point_cloud_amd/csrc variant NAME=x DEFS=...
    capi._LIB_NAME = f"libd2pc_{sys.argv[1]}.so"
    print("library:", capi._LIB_NAME)
from disparity_to_point_cloud_amd.torch_api import fuse_planes

def t(fn, iters=10, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return float(np.median(ts))

ctx = d2pc.Context(q=d2pc.make_q())
s = torch.cuda.current_stream().cuda_stream
for (n, f) in ((465, 1), (465, 64), (1080, 32), (2160, 16), (2160, 64)):
    planes = [torch.randint(0, 256, (f, n, n), dtype=torch.uint8, device="cuda") for _ in range(6)]
    crop = (0, 40, 30, 10)
    desc = d2pc.fuse_desc_init()
    desc.width = desc.height = n
    desc.n_frames = f
    for i, p in enumerate(planes):
        desc.planes[i], desc.pitch[i], desc.frame_stride[i] = p.data_ptr(), n, n * n
    ow, oh = n - 40, n - 40
    fused = torch.empty((f, oh, ow), dtype=torch.uint8, device="cuda")
    comb = torch.empty((f, n, n), dtype=torch.uint8, device="cuda")
    desc.fused, desc.fused_pitch, desc.fused_frame_stride = fused.data_ptr(), ow, ow * oh
    for with_comb in (True, False):
        if with_comb:
            desc.combined, desc.combined_pitch, desc.combined_frame_stride = comb.data_ptr(), n, n * n
        else:
            desc.combined = None
        px = f * n * n
        byts = px * ((6 + 1) if with_comb else 4) + f * ow * oh
        line = f"{n}x{n} x{f:3d} combined={int(with_comb)}:"
        for rows in (0, 4, 8, 16, 32):
            ctx.set_tuning("fuse_rows", rows)
            us = t(lambda: ctx.fuse_device(desc, s))
            line += f"  rows={rows:2d} {us:7.1f} us {px/us/1e3:6.1f} Gpix/s {byts/us/1e3:5.0f} GB/s |"
        print(line, flush=True)
ctx.close()

# ---- rotateMat (d2pc_rotate_cw_device): 1 B read + 1 B written per pixel ----
ctx = d2pc.Context(q=d2pc.make_q())
for (rows, cols, f) in ((480, 752, 64), (2160, 3840, 16)):
    src = torch.randint(0, 256, (f, rows, cols), dtype=torch.uint8, device="cuda")
    dst = torch.empty((f, cols, rows), dtype=torch.uint8, device="cuda")
    us = t(lambda: ctx.rotate_cw_device(src.data_ptr(), cols, rows, cols, cols * rows, f, dst.data_ptr(), rows, rows * cols, s))
    px = f * rows * cols
    print(f"rotate_cw {cols}x{rows} x{f:3d}: {us:8.1f} us  {px/us/1e3:8.1f} Gpix/s  {2*px/us/1e3:8.1f} GB/s algorithmic", flush=True)
ctx.close()
