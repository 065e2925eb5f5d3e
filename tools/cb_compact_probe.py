#!/usr/bin/env python3
"""Where the COMPACT tile-fused callback kernel's time goes: PARITY / COMPACT, with and without indices, the three
COMPACT forms, all valid vs blocky holes.  Interleaved rounds in one process.  GPU box only."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

W, H, F = 3840, 2160, 16
q = d2pc.make_q()
gen = torch.Generator(device="cuda").manual_seed(7)
raw = torch.randint(1, 256, (F, H, W), dtype=torch.uint8, device="cuda", generator=gen)   # no zero: all valid after the filter
holes = raw.clone()
m = torch.rand((F, (H + 63) // 64, (W + 63) // 64), device="cuda", generator=gen) < 0.3
holes[m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H, :W]] = 0
s = torch.cuda.current_stream().cuda_stream
cands = []
for name, mode, form, idx, src in (
        ("parity fused            ", d2pc.MODE_PARITY, None, False, raw), ("parity fused +idx       ", d2pc.MODE_PARITY, None, True, raw),
        ("compact form1 all valid ", d2pc.MODE_COMPACT, 1, False, raw), ("compact form1 valid +idx", d2pc.MODE_COMPACT, 1, True, raw),
        ("compact form2 all valid ", d2pc.MODE_COMPACT, 2, False, raw), ("compact form2 valid +idx", d2pc.MODE_COMPACT, 2, True, raw),
        ("compact form1 holes     ", d2pc.MODE_COMPACT, 1, False, holes), ("compact form1 holes +idx", d2pc.MODE_COMPACT, 1, True, holes),
        ("compact form2 holes     ", d2pc.MODE_COMPACT, 2, False, holes), ("compact form2 holes +idx", d2pc.MODE_COMPACT, 2, True, holes),
        ("compact 2 launches holes+idx", d2pc.MODE_COMPACT, 0, True, holes), ("compact 2 launches valid+idx", d2pc.MODE_COMPACT, 0, True, raw)):
    ctx = d2pc.Context(q=q, mode=mode)
    if form is not None:
        ctx.set_tuning("callback_fused_compact", form)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        ctx.set_tuning(k, int(v))
    b = DeviceBatch(ctx, F, H, W, dtype=torch.uint8, want_index=idx)

    def launch(ctx=ctx, b=b, src=src, idx=idx):
        ctx.process_mono_device(src.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, b.points.data_ptr(),
                                b.index.data_ptr() if idx else None, b.stride, b.counts.data_ptr(), s)
    cands.append((name, launch, ctx, b, []))
for _, launch, _, _, _ in cands:
    for _ in range(30):
        launch()
torch.cuda.synchronize()
for r in range(6):
    for name, launch, ctx, b, ts in cands:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(10):
            launch()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
for name, launch, ctx, b, ts in cands:
    print(f"{name}: median {np.median(ts)*1e3:7.1f} us  min {min(ts)*1e3:7.1f}  points {int(b.counts.sum().item())}", flush=True)
