#!/usr/bin/env python3
"""Does the kernel time depend on WHERE the buffers sit in HBM?  Same library,
same launch shape, output/input buffers shifted by various offsets. GPU only."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd import capi

F, H, W = 16, 2160, 3840
q = d2pc.make_q()
ctx = d2pc.Context(q=q, border=int(sys.argv[1]) if len(sys.argv) > 1 else 40)
roi_n = d2pc.roi_points(W, H, ctx.config().border)
stride = (roi_n + 15) // 16 * 16
g = torch.Generator(device="cuda").manual_seed(1)
pad = 64 << 20
inbuf = torch.empty(F * H * W * 4 + pad, dtype=torch.uint8, device="cuda")
outbuf = torch.empty(F * stride * 16 + pad, dtype=torch.uint8, device="cuda")
counts = torch.zeros(F, dtype=torch.int32, device="cuda")
src = torch.rand((F, H, W), generator=g, device="cuda") * 127.5 + 0.5
s = torch.cuda.current_stream().cuda_stream
print("inbuf %x outbuf %x" % (inbuf.data_ptr(), outbuf.data_ptr()))
def run(io, oo, iters=10, rounds=5):
    d = inbuf[io:io + F * H * W * 4].view(torch.float32).view(F, H, W)
    d.copy_(src)
    o = outbuf[oo:oo + F * stride * 16]
    def launch():
        ctx.process_device(d.data_ptr(), 0, 1.0, W, H, W * 4, H * W * 4, F, o.data_ptr(), None, stride, counts.data_ptr(), s)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): launch()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return np.median(ts)
for io in (0, 4096, 1 << 20):
    for oo in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 17 << 20, 33 << 20):
        print(f"in+{io:>8d} out+{oo:>9d}: {run(io, oo):7.1f} us", flush=True)
