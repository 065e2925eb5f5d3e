#!/usr/bin/env python3
"""Time the device median and the whole device-resident callback body
(median 11 -> x1/8 -> reproject+pack) on batches of 8-bit frames. GPU only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd import capi
if len(sys.argv) > 1:  # tuning build: make -C disparity_to_point_cloud_amd/csrc variant NAME=x
    capi._LIB_NAME = f"libd2pc_{sys.argv[1]}.so"
    print("library:", capi._LIB_NAME)
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

def t(fn, iters=10, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return float(np.median(ts))

q = d2pc.make_q()
if not os.environ.get("D2PC_BENCH_DEFAULT_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream())  # not the legacy default stream (it synchronises with every blocking stream)
s = torch.cuda.current_stream().cuda_stream
for (w, h, n) in ((752, 480, 64), (1920, 1080, 16), (3840, 2160, 16)):
    ctx = d2pc.Context(q=q)
    raw = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda")
    b = DeviceBatch(ctx, n, h, w, dtype=torch.uint8)
    for algo in (1, 2):  # per-pixel select, bit-sliced (k = 9, 11)
        try:
            ctx.set_tuning("median_algo", algo)
        except Exception:  # a library built before the bit-sliced kernel
            break
        us = t(lambda: ctx.median_device(raw.data_ptr(), w, h, w, w * h, n, b.disp.data_ptr(), w, w * h, 11, s))
        print(f"{w}x{h} x{n}: median11 algo {algo} {us:8.1f} us  = {us/n:7.2f} us/frame  {n*w*h/us:9.1f} Mpix/s", flush=True)
    try:
        ctx.set_tuning("median_algo", 0)
    except Exception:
        pass
    for k in (3, 11):
        us = t(lambda: ctx.median_device(raw.data_ptr(), w, h, w, w * h, n, b.disp.data_ptr(), w, w * h, k, s))
        print(f"{w}x{h} x{n}: median{k:2d} {us:8.1f} us  = {us/n:7.2f} us/frame  {n*w*h/us:9.1f} Mpix/s", flush=True)
        ur = t(lambda: ctx.median_roi_device(raw.data_ptr(), w, h, w, w * h, n, b.disp.data_ptr(), w, w * h, k, s))
        print(f"{w}x{h} x{n}: median{k:2d} ROI only (border 40) {ur:8.1f} us  = {ur/n:7.2f} us/frame  ({100*(1-ur/us):.1f} % less; "
              f"ROI is {100*(1-(w-80)*(h-80)/(w*h)):.1f} % fewer pixels)", flush=True)
    def body():
        ctx.median_roi_device(raw.data_ptr(), w, h, w, w * h, n, b.disp.data_ptr(), w, w * h, 11, s)
        b.launch(scale=0.125)
    us = t(body)
    us_r = t(lambda: b.launch(scale=0.125))
    for fused in (1, 0):  # one kernel tile by tile (k_callback_bs), two launches
        ctx.set_tuning("callback_fused", fused); ctx.set_tuning("callback_chunks", 1)
        uf = t(lambda: ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, w, h, w, w * h, n, 11, 0.125,
                                               b.points.data_ptr(), None, b.stride, b.counts.data_ptr(), s))
        print(f"{w}x{h} x{n}: d2pc_process_mono_device callback_fused={fused}: {uf:8.1f} us = "
              f"{n*w*h/uf:9.1f} Mpix/s  ({us/uf:.3f}x the two launches in order)", flush=True)
    ctx.set_tuning("callback_fused", 0)
    for chunks in (2, 4):
        ctx.set_tuning("callback_chunks", chunks)
        uf = t(lambda: ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, w, h, w, w * h, n, 11, 0.125,
                                               b.points.data_ptr(), None, b.stride, b.counts.data_ptr(), s))
        print(f"{w}x{h} x{n}: d2pc_process_mono_device callback_chunks={chunks}: {uf:8.1f} us = "
              f"{n*w*h/uf:9.1f} Mpix/s  ({us/uf:.3f}x the two launches in order)", flush=True)
    print(f"{w}x{h} x{n}: callback body (ROI median11 + reproject u8) {us:8.1f} us = {us/n:7.2f} us/frame {n*w*h/us:9.1f} Mpix/s; reproject alone {us_r:8.1f} us", flush=True)
    ctx.close()
