#!/usr/bin/env python3
"""(round 5) Library builds of the tile-fused callback kernels (d2pc_process_mono_device: median 11 + points per tile) against
each other, interleaved in ONE process on ONE set of buffers -- 16 x 4K u8: PARITY (k_callback_bs), COMPACT + indices with
30 % zero pixels in 64 x 64 blocks and iid (k_callback_bs_compact_pipe).  GPU box only.

  make -C disparity_to_point_cloud_amd/csrc variant NAME=nohoist DEFS=-DD2PC_CB_NO_HOIST=1
  python tools/ab_callback.py base,nohoist [key=value ...]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

libs = (sys.argv[1] if len(sys.argv) > 1 else "base").split(",")
W, H, F = 3840, 2160, 16
q = d2pc.make_q()
gen = torch.Generator(device="cuda").manual_seed(7)
raw = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device="cuda", generator=gen)
blocky = raw.clone()
m = torch.rand((F, (H + 63) // 64, (W + 63) // 64), device="cuda", generator=gen) < 0.3
blocky[m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H, :W]] = 0
iid = raw.clone()
iid[torch.rand(raw.shape, device="cuda", generator=gen) < 0.3] = 0
s = torch.cuda.current_stream().cuda_stream
points = torch.empty((F, (W * H + 15) // 16 * 16, 4), dtype=torch.float32, device="cuda")   # ONE set of output buffers for all
index = torch.empty((F, points.shape[1]), dtype=torch.int32, device="cuda")
counts = torch.zeros((F,), dtype=torch.int32, device="cuda")
cands = []
for lib in libs:
    variant = None if lib == "base" else lib
    for name, mode, idx, src in (("parity              ", d2pc.MODE_PARITY, False, raw),
                                 ("compact blocky + idx", d2pc.MODE_COMPACT, True, blocky),
                                 ("compact iid + idx   ", d2pc.MODE_COMPACT, True, iid)):
        ctx = d2pc.Context(q=q, mode=mode, variant=variant)
        for kv in sys.argv[2:]:
            k, v = kv.split("=")
            ctx.set_tuning(k, int(v))
        ctx.reserve_mono(d2pc.DTYPE_U8, W, H, F)

        def launch(ctx=ctx, src=src, idx=idx):
            ctx.process_mono_device(src.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, points.data_ptr(),
                                    index.data_ptr() if idx else None, points.shape[1], counts.data_ptr(), s)
        cands.append((f"{lib:8s} {name}", launch, ctx, []))
for _, launch, _, _ in cands:
    for _ in range(20):
        launch()
torch.cuda.synchronize()
for r in range(7):
    for name, launch, ctx, ts in cands:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(10):
            launch()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
for name, launch, ctx, ts in cands:
    ctx.check_async_error()
    print(f"{name}: median {np.median(ts)*1e3:7.1f} us  min {min(ts)*1e3:7.1f}  max {max(ts)*1e3:7.1f}", flush=True)
