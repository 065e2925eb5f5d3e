#!/usr/bin/env python3
"""Sweep of the calibration kernels' grid (tuning membench_blocks_per_cu) against hipMemsetAsync / a torch copy on the
bench's 2-GB buffer.  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
import bench

ctx = d2pc.Context(q=d2pc.make_q())
n = 16 * 7820800 * 16
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
base, half = buf.data_ptr(), n // 2 // 32 * 32


class L:
    def __init__(self, f):
        self.launch = f


def rate(f, nbytes):
    sp = bench.spread(bench.timed_rounds(L(f), 5))
    return nbytes / (sp["median"] * 1e-3) / 1e9


print(f"torch zero_ (hipMemsetAsync): {rate(lambda: buf.zero_(), n):7.1f} GB/s")
print(f"torch copy_ (half -> half)  : {rate(lambda: buf[half:2 * half].copy_(buf[:half]), 2 * half):7.1f} GB/s")
for bpc in (1, 2, 4, 8, 16, 32, 64, 128):
    ctx.set_tuning("membench_blocks_per_cu", bpc)
    f = rate(lambda: ctx.membench_fill(base, n, s), n)
    c = rate(lambda: ctx.membench_copy(base, base + half, half, s), 2 * half)
    print(f"blocks per CU {bpc:3d}: fill {f:7.1f} GB/s   copy {c:7.1f} GB/s", flush=True)
