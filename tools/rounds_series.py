#!/usr/bin/env python3
"""Per-round time series of the compaction variants right after their buffers were allocated (is a slow round a
warm-up effect or sporadic?).  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.synth import synth_disparity
from disparity_to_point_cloud_amd.torch_api import DeviceBatch
import bench

q = d2pc.make_q()
for name, kind, idx in (("all_valid", "uniform", False), ("holes", "holes", False), ("holes_index", "holes", True),
                        ("blocky_index", "blocky", True)):
    ctx = d2pc.Context(border=40, mode=d2pc.MODE_COMPACT, q=q)
    b = DeviceBatch(ctx, 16, 2160, 3840, want_index=idx)
    for f in range(16):
        b.disp[f].copy_(torch.from_numpy(synth_disparity(4, f, 3840, 2160, kind)))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for r in range(60):      # rounds of 5 launches, NO warm-up: round 0 is the first use of the buffers
        for _ in range(5):
            b.launch()
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = [ev[r].elapsed_time(ev[r + 1]) / 5 for r in range(60)]
    ctx.compact_stats_reset()
    print(name, " ".join(f"{x*1e3:.0f}" for x in ms), flush=True)
    del b
    ctx.close()
