#!/usr/bin/env python3
"""Stress of the single-pass compaction's inter-block hand-off: random launch shapes (frames, sizes, tile size,
persistent blocks per CU, hole patterns), every launch compared BIT FOR BIT on the device with the two-pass
result (which has no in-launch synchronisation) and checked for the sticky timeout flag.
GPU box:  python tools/stress_onepass.py [launches] [seed]"""
import os, sys, time
os.environ.setdefault("D2PC_LIBRARY_VARIANT", "exp")   # draws from the laboratory too (compact_algo 4, tile shapes 4 / 16): the experiment build
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
contend = len(sys.argv) > 3 and sys.argv[3] == "contend"   # keep the GPU busy with 11x11 medians on another stream
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
q = d2pc.make_q()
t0 = time.time()
done = 0
if contend:
    side = torch.cuda.Stream()
    cctx = d2pc.Context(q=q)
    craw = torch.randint(0, 256, (8, 2160, 3840), dtype=torch.uint8, device="cuda")
    cout = torch.empty_like(craw)
while done < launches:
    n = int(rng.choice([1, 2, 4, 5, 8, 16, 24, 40]))
    w, h = (int(rng.integers(64, 2000)), int(rng.integers(64, 1200)))
    if rng.random() < 0.2: w, h = 3840, 2160; n = min(n, 16)
    border = int(rng.choice([0, 8, 40]))
    if w <= 2 * border or h <= 2 * border: continue
    form = int(rng.choice([0, 0, 0, 2, 1, 3, 4, 5, 6, 7]))   # the single pass's kernel form (0 / 2: the product's dense pass)
    pxt = int(rng.choice([4, 8, 16])); opbpc = int(rng.choice([1, 2, 3, 4, 6, 8])); idx = bool(rng.integers(0, 2))
    holes = float(rng.choice([0.0, 0.02, 0.3, 0.7, 0.98]))
    disp = torch.rand((n, h, w), generator=g, device="cuda") * 127.5 + 0.5
    kind = rng.integers(0, 3)
    if kind == 0: disp.mul_((torch.rand(disp.shape, generator=g, device="cuda") >= holes).float())
    elif kind == 1:  # blocky holes
        m = (torch.rand((n, (h + 63) // 64, (w + 63) // 64), generator=g, device="cuda") >= holes).float()
        disp.mul_(m.repeat_interleave(64, 1).repeat_interleave(64, 2)[:, :h, :w])
    else:            # whole frames empty / full
        disp.mul_((torch.rand((n, 1, 1), generator=g, device="cuda") >= holes).float())
    outs = []
    for algo in (1, 2):
        with d2pc.Context(q=q, border=border, mode=d2pc.MODE_COMPACT, compact_algo=algo) as ctx:
            ctx.set_tuning("pxt_compact", pxt); ctx.set_tuning("onepass_blocks_per_cu", opbpc)
            if algo == 2: ctx.set_tuning("onepass_form", form)
            b = DeviceBatch(ctx, n, h, w, want_index=idx)
            b.points.fill_(-7.0)
            if idx: b.index.fill_(-7)
            b.disp.copy_(disp)
            reps = 1 if algo == 1 else int(rng.integers(1, 4))
            if contend and algo == 2:
                for _ in range(3):
                    cctx.median_device(craw.data_ptr(), 3840, 2160, 3840, 3840 * 2160, 8, cout.data_ptr(), 3840, 3840 * 2160,
                                       11, side.cuda_stream)
            for _ in range(reps): b.launch()
            torch.cuda.synchronize()
            ctx.check_async_error()
            outs.append((b.counts.clone(), b.points.clone(), b.index.clone() if idx else None))
    what = f"launch {done}: form={form} n={n} {w}x{h} b={border} pxt={pxt} opbpc={opbpc} idx={idx} holes={holes} kind={kind}"
    assert torch.equal(outs[0][0], outs[1][0]), what + " counts differ"
    # compare the first count[f] points of every frame (the rest of each frame's slab is untouched: -7)
    assert torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32)), what + " points differ"
    if idx: assert torch.equal(outs[0][2], outs[1][2]), what + " indices differ"
    done += 1
    if done % 50 == 0: print(f"{done} launches ok ({time.time() - t0:.0f} s)", flush=True)
print("onepass stress ok:", launches, "launches")
