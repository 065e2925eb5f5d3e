#!/usr/bin/env python3
"""Per-pixel select (median_algo 1) against the bit-sliced kernel (2) over launch sizes: the rule
kBsMinTiles in d2pc_median.hip.  GPU only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc


def t(fn, iters=20, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return float(np.median(ts))


torch.cuda.set_stream(torch.cuda.Stream())
s = torch.cuda.current_stream().cuda_stream
ctx = d2pc.Context(q=d2pc.make_q())
K = int(sys.argv[1]) if len(sys.argv) > 1 else 11
print(f"k = {K}; size x frames: 256x32 tiles (ROI, border 40) | per-pixel us | bit-sliced us | ratio")
for (w, h) in ((752, 480), (1920, 1080), (3840, 2160)):
    for n in (1, 2, 4, 8, 16, 32):
        if w * h * n > 3840 * 2160 * 16: continue
        raw = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda")
        out = torch.empty_like(raw)
        tiles = -(-(w - 80) // 256) * -(-(h - 80) // 32) * n
        us = []
        for algo in (1, 2):
            ctx.set_tuning("median_algo", algo)
            us.append(t(lambda: ctx.median_roi_device(raw.data_ptr(), w, h, w, w * h, n, out.data_ptr(), w, w * h, K, s)))
        print(f"{w}x{h} x{n:2d}: {tiles:6d} tiles  {us[0]:8.1f}  {us[1]:8.1f}  {us[0]/us[1]:.2f}", flush=True)
