#!/usr/bin/env python3
"""(round 5) The bit-sliced median's select, one lane per (t, row) (median_algo 2: 121 candidate words per thread, 3 waves per
SIMD) against the lane-PAIR form (median_algo 3, experiment build: 66 words per lane, 4 waves per SIMD, ~+20 % instructions),
interleaved in one process on one set of buffers, every window size.  d2pc_median_device over the inset ROI of 16 x 4K, 16 x
1080p and 64 x 752x480 frames.  GPU box only:  python tools/ab_median_select.py > profiles/r05_ab_median_select.txt"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc

s = torch.cuda.current_stream().cuda_stream
for (w, h, n) in ((3840, 2160, 16), (1920, 1080, 16), (752, 480, 64)):
    raw = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(raw)
    for k in (11, 9, 7, 5, 3):
        cands = []
        for algo in (2, 3):
            ctx = d2pc.Context(q=d2pc.make_q(), variant="exp")
            ctx.set_tuning("median_algo", algo)
            cands.append((algo, ctx, []))
        for _, ctx, _ in cands:
            for _ in range(10):
                ctx.median_roi_device(raw.data_ptr(), w, h, w, w * h, n, dst.data_ptr(), w, w * h, k, s)
        torch.cuda.synchronize()
        for r in range(7):
            for algo, ctx, ts in cands:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ctx.median_roi_device(raw.data_ptr(), w, h, w, w * h, n, dst.data_ptr(), w, w * h, k, s)
                e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        a, b = (float(np.median(c[2])) for c in cands)
        print(f"{n:3d} x {w}x{h}  median {k:2d} x {k:<2d}  one lane per (t,row) {a:8.1f} us   lane pair {b:8.1f} us   pair/one = {b/a:.3f}", flush=True)
        for _, ctx, _ in cands:
            ctx.close()
