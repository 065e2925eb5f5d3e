#!/usr/bin/env python3
"""Time the library's kernels over launch-shape variants (GPU box only).
Usage: python tools/tune.py [--frames 16] [--w 3840 --h 2160]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch


def time_launch(b, iters=10, reps=5):
    for _ in range(3):
        b.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            b.launch()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--h", type=int, default=2160)
    ap.add_argument("--holes", type=float, default=0.0)
    ap.add_argument("--modes", default="parity,compact")
    ap.add_argument("--borders", default="40,0")
    ap.add_argument("--algos", default="1,2")
    ap.add_argument("--novec", type=int, default=0)
    a = ap.parse_args()
    q = d2pc.make_q()
    g = torch.Generator(device="cuda").manual_seed(1)
    for mode, name in ((d2pc.MODE_PARITY, "parity"), (d2pc.MODE_COMPACT, "compact")):
        if name not in a.modes.split(","):
            continue
        for border in [int(x) for x in a.borders.split(",")]:
            for algo in ((0,) if mode == d2pc.MODE_PARITY else [int(x) for x in a.algos.split(",")]):
                for idx in (False, True) if mode == d2pc.MODE_COMPACT else (False,):
                    ctx = d2pc.Context(q=q, border=border, mode=mode, compact_algo=algo)
                    b = DeviceBatch(ctx, a.frames, a.h, a.w, want_index=idx)
                    b.disp.copy_(torch.rand(b.disp.shape, generator=g, device="cuda") * 127.5 + 0.5)
                    if a.holes > 0:
                        b.disp.mul_((torch.rand(b.disp.shape, generator=g, device="cuda") >= a.holes).float())
                    b.launch()
                    torch.cuda.synchronize()
                    npts = int(b.counts.sum().item())
                    rin = a.frames * b.roi_n
                    alg_bytes = 4 * rin + (20 if idx else 16) * npts
                    for pxt, bpc, gen, novec in [(p, b, 0, nv) for p in (8, 16) for b in (8, 16) for nv in (0, 1)] + [(8, 16, 1, 0)]:
                        if True:
                            ctx.set_tuning("force_general_q", gen)
                            ctx.set_tuning("no_vec_rows", novec)
                            ctx.set_tuning("pxt_parity" if mode == d2pc.MODE_PARITY else "pxt_compact", pxt)
                            ctx.set_tuning("blocks_per_cu", bpc)
                            med, mn = time_launch(b)
                            print(f"{name:7s} border={border:2d} algo={algo} idx={int(idx)} pxt={pxt:2d} bpc={bpc:2d} gen={gen} novec={novec} "
                                  f"med={med*1e3:8.1f}us min={mn*1e3:8.1f}us  {alg_bytes/med/1e6:7.1f} GB/s "
                                  f"{a.frames*a.w*a.h/med/1e3:8.1f} Mpix/s pts={npts}", flush=True)
                    if mode == d2pc.MODE_COMPACT:
                        ctx.check_async_error()
                    del b
                    ctx.close()


if __name__ == "__main__":
    main()
