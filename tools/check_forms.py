#!/usr/bin/env python3
"""The single pass's kernel forms must write the same bytes: runs each `onepass_form` given on 16 x 4K (or --frames/--w/--h)
with holes and compares points, indices and counts with form 2's.  GPU box only; experiment build.
  python tools/check_forms.py 2,6,8,9 [frames w h]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

forms = [int(x) for x in sys.argv[1].split(",")]
frames, w, h = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (16, 3840, 2160)
g = torch.Generator(device="cuda").manual_seed(11)
ref = None
bad = 0
for holes in (0.0, 0.3, 0.9):
    disp = torch.rand((frames, h, w), generator=g, device="cuda") * 127.5 + 0.5
    if holes:
        disp.mul_((torch.rand(disp.shape, generator=g, device="cuda") >= holes).float())
    ref = None
    for form in forms:
        ctx = d2pc.Context(q=d2pc.make_q(), mode=d2pc.MODE_COMPACT, compact_algo=2, variant="exp")
        ctx.set_tuning("onepass_form", form)
        b = DeviceBatch(ctx, frames, h, w, want_index=True)
        b.disp.copy_(disp)
        for _ in range(3):
            b.launch()
        torch.cuda.synchronize()
        counts = b.counts.clone()
        n = [int(c) for c in counts.tolist()]
        got = (counts, [b.points[f, :n[f]].clone() for f in range(frames)], [b.index[f, :n[f]].clone() for f in range(frames)])
        st = ctx.compact_stats() if hasattr(ctx, "compact_stats") else None
        if ref is None:
            ref = got
            print(f"holes {holes}: form {form} counts[0..3] = {n[:4]}  stats {st}")
        else:
            same = torch.equal(got[0], ref[0]) and all(torch.equal(a.view(torch.int32), r.view(torch.int32)) for a, r in zip(got[1], ref[1])) \
                and all(torch.equal(a, r) for a, r in zip(got[2], ref[2]))
            print(f"holes {holes}: form {form} {'same bytes' if same else 'DIFFERS'}  counts[0..3] = {n[:4]}  stats {st}")
            bad += 0 if same else 1
        ctx.close()
sys.exit(1 if bad else 0)
