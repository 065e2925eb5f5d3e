#!/usr/bin/env python3
"""Does a CU-masked stream confine our kernels, and what do the median (VALU-bound) and the u8 reprojection
(HBM-bound) keep of their rate on a subset of the CUs?  And do two masked streams run concurrently?
(Evidence for the overlap attempt of d2pc_process_mono_device.)  GPU box only."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]

def masked_stream(pred, cus=256):
    words = (ctypes.c_uint32 * ((cus + 31) // 32))()
    n = 0
    for i in range(cus):
        if pred(i):
            words[i // 32] |= 1 << (i % 32); n += 1
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), words) == 0
    return torch.cuda.ExternalStream(s.value), n

W, H, N = 3840, 2160, 16
ctx = d2pc.Context(q=d2pc.make_q())
raw = torch.randint(0, 256, (N, H, W), dtype=torch.uint8, device="cuda")
b = DeviceBatch(ctx, N, H, W, dtype=torch.uint8)
torch.cuda.synchronize()

def med(stream):
    ctx.median_roi_device(raw.data_ptr(), W, H, W, W * H, N, b.disp.data_ptr(), W, W * H, 11, stream.cuda_stream)
def rep(stream):
    b.launch(scale=0.125, stream=stream)

def timed(fn, stream, iters=5):
    fn(stream); stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters): fn(stream)
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

pats = [("all", lambda i: True), ("7 of 8", lambda i: i % 8 < 7), ("6 of 8", lambda i: i % 8 < 6), ("4 of 8", lambda i: i % 8 < 4),
        ("2 of 8 (hi)", lambda i: i % 8 >= 6), ("1 of 8 (hi)", lambda i: i % 8 >= 7), ("first 192", lambda i: i < 192),
        ("last 64", lambda i: i >= 192), ("first 128", lambda i: i < 128)]
streams = {}
for name, pred in pats:
    s, n = masked_stream(pred)
    streams[name] = s
    print(f"{name:12s} {n:3d} CUs: median11 {timed(med, s):8.1f} us   reproject u8 {timed(rep, s):8.1f} us", flush=True)

# the fp32 headline kernel (16 x 4K, PARITY, border 40) on CU subsets: is it bound by the CUs or by the memory system?
bf = DeviceBatch(ctx, N, H, W)
bf.disp.copy_(torch.rand((N, H, W), device="cuda") * 127.5 + 0.5)
for name in ("all", "first 192", "first 128", "last 64"):
    us = timed(lambda s: bf.launch(stream=s), streams[name])
    print(f"{name:12s}: k_reproject_pack<F32> {us:8.1f} us = {16 * 3760 * 2080 * 20 / us / 1e3:7.0f} GB/s", flush=True)
del bf

# concurrency: median on one masked stream, reprojection on the complementary one, started together
for a, c in (("6 of 8", "2 of 8 (hi)"), ("7 of 8", "1 of 8 (hi)"), ("first 192", "last 64"), ("all", "all")):
    sa, sc = streams[a], (streams[c] if c != a else masked_stream(lambda i: True)[0])
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    cur = torch.cuda.current_stream()
    e0.record(cur)
    sa.wait_event(e0); sc.wait_event(e0)
    for _ in range(3):
        med(sa); rep(sc)
    e1.record(sa); e2.record(sc)
    e1.synchronize(); e2.synchronize()
    print(f"concurrent median on [{a}] + reproject on [{c}] x3: median stream {e0.elapsed_time(e1)/3*1e3:8.1f} us/iter, "
          f"reproject stream {e0.elapsed_time(e2)/3*1e3:8.1f} us/iter", flush=True)

# ---- the real pipeline: reproject(chunk c) depends on median(chunk c); median(chunk c+1) overlaps it ----
stride = b.stride
def pipeline(sm, sr, chunk):
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(cur)
    sm.wait_event(fork); sr.wait_event(fork)
    for f0 in range(0, N, chunk):
        nf = min(chunk, N - f0)
        ctx.median_roi_device(raw.data_ptr() + f0 * W * H, W, H, W, W * H, nf, b.disp.data_ptr() + f0 * W * H, W, W * H, 11,
                              sm.cuda_stream)
        ev = torch.cuda.Event(); ev.record(sm)
        sr.wait_event(ev)
        ctx.process_device(b.disp.data_ptr() + f0 * W * H, d2pc.DTYPE_U8, 0.125, W, H, W, W * H, nf,
                           b.points.data_ptr() + f0 * stride * 16, None, stride, b.counts.data_ptr() + 4 * f0, sr.cuda_stream)
    join = torch.cuda.Event(); join.record(sr)
    cur.wait_event(join)

def time_pipeline(sm, sr, chunk, iters=5):
    cur = torch.cuda.current_stream()
    pipeline(sm, sr, chunk); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(iters): pipeline(sm, sr, chunk)
    e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

side = torch.cuda.Stream()
with torch.cuda.stream(side):   # a non-default caller stream (the legacy default stream synchronises with blocking streams)
    seq = timed(lambda s: (med(s), rep(s)), side)
    print(f"in order on one stream: {seq:8.1f} us", flush=True)
    plain_m, plain_r = torch.cuda.Stream(), torch.cuda.Stream()
    for label, sm, sr in (("two plain streams", plain_m, plain_r), ("first 192 | last 64", streams["first 192"], streams["last 64"]),
                          ("first 128 | last 128 -> use first 128 + all", streams["first 128"], plain_r)):
        for chunk in (1, 2, 4, 8):
            t = time_pipeline(sm, sr, chunk)
            print(f"pipeline [{label}] chunk={chunk}: {t:8.1f} us  ({seq / t:.3f}x in-order)", flush=True)
