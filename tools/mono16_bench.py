#!/usr/bin/env python3
"""d2pc_mono16_to_mono8_device (cv_bridge's mono16 -> mono8 rescale, cpp:50) over batch sizes: 2 B read + 1 B written per pixel.
GPU box:  python tools/mono16_bench.py [variant]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disparity_to_point_cloud_amd import capi
if len(sys.argv) > 1:   # a variant build: libd2pc_<name>.so
    capi._LIB_NAME = f"libd2pc_{sys.argv[1]}.so"
import disparity_to_point_cloud_amd as d2pc

with d2pc.Context(q=d2pc.make_q()) as ctx:
    s = torch.cuda.current_stream().cuda_stream
    for n, h, w in ((16, 2160, 3840), (16, 1080, 1920), (64, 480, 752), (1, 480, 752)):
        src = torch.randint(0, 32767, (n, h, w), dtype=torch.int16, device="cuda")
        dst = torch.empty((n, h, w), dtype=torch.uint8, device="cuda")
        def launch():
            ctx.mono16_to_mono8_device(src.data_ptr(), w, h, 2 * w, 2 * w * h, n, dst.data_ptr(), w, w * h, s)
        for _ in range(20): launch()
        torch.cuda.synchronize()
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): launch()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        us = float(np.median(ts))
        print(f"{n:3d} x {w}x{h}: {us:8.1f} us  {3 * n * h * w / us / 1e6:7.2f} TB/s  {n * h * w / us / 1e3:8.1f} Gpixel/s", flush=True)
