#!/usr/bin/env python3
"""Randomised soak of the 8-bit image kernels against the oracle: d2pc_median_device (all k, random
sizes / pitches / batches), d2pc_fuse_device (random rules, crops, strip heights, aliasing) and
d2pc_rotate_cw_device.  GPU box:  python tools/soak_filters.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import fuse_planes
import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
ctx = d2pc.Context(q=d2pc.make_q())
s = torch.cuda.current_stream().cuda_stream
t0 = time.time()

def image(h, w):
    kind = rng.integers(0, 4)
    if kind == 0: return rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    if kind == 1: return (rng.integers(0, 3, size=(h, w)) * 127).astype(np.uint8)        # many ties
    if kind == 2: return np.clip(np.add.outer(np.arange(h) * 2, np.arange(w)) % 300, 0, 255).astype(np.uint8)
    a = np.full((h, w), rng.integers(0, 256), dtype=np.uint8); a[rng.random((h, w)) < 0.02] = rng.integers(0, 256); return a

for c in range(cases):
    # ---- median ----
    k = int(rng.choice([3, 5, 7, 9, 11])); n = int(rng.integers(1, 4))
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 400))
    if rng.random() < 0.2: w = int(rng.integers(400, 1100))   # several 256-wide tiles of the bit-sliced kernel
    sp, dp = w + int(rng.integers(0, 9)), w + int(rng.integers(0, 9))
    ctx.set_tuning("median_algo", int(rng.integers(0, 3)))   # 0 the library's choice, 1 per pixel, 2 bit-sliced (k = 9, 11)
    imgs = [image(h, w) for _ in range(n)]
    src = torch.zeros((n, h, sp), dtype=torch.uint8, device="cuda"); src[:, :, :w] = torch.from_numpy(np.stack(imgs)).cuda()
    dst = torch.full((n, h, dp), 9, dtype=torch.uint8, device="cuda")
    ctx.median_device(src.data_ptr(), w, h, sp, sp * h, n, dst.data_ptr(), dp, dp * h, k, s)
    got = dst.cpu().numpy()
    for f in range(n):
        assert np.array_equal(got[f, :, :w], oracle.median_u8(imgs[f], k)), f"median case {c}: k={k} {w}x{h} n={n}"
        assert (got[f, :, w:] == 9).all(), "median wrote outside the row"
    # ---- fusion ----
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 600))
    rule = int(rng.integers(0, 9)); rows = int(rng.choice([0, 2, 4, 8, 16, 34]))
    l, r = sorted(rng.integers(0, w + 1, 2)); r = w - r; t, b = sorted(rng.integers(0, h + 1, 2)); b = h - b
    planes = [image(h, w) for _ in range(6)]
    if rng.random() < 0.3: planes[4] = planes[2]          # grad1 aliases score1 as in the reference
    comb = bool(rng.random() < 0.7)
    ctx.set_tuning("fuse_rows", rows)
    dev = [torch.from_numpy(p).cuda() for p in planes]
    if planes[4] is planes[2]: dev[4] = dev[2]
    fused, cmb = fuse_planes(ctx, dev, rule=rule, crop=(int(l), int(r), int(t), int(b)), want_combined=comb)
    wf, wc = oracle.fuse(planes, rule=rule, crop=(int(l), int(r), int(t), int(b)), want_combined=comb)
    what = f"fusion case {c}: rule={rule} {w}x{h} crop={l},{r},{t},{b} rows={rows} comb={comb}"
    assert np.array_equal(fused.cpu().numpy(), wf), what
    if comb: assert np.array_equal(cmb.cpu().numpy(), wc), what
    # ---- rotate ----
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    img = image(h, w)
    o = torch.zeros((w, h), dtype=torch.uint8, device="cuda")
    ctx.rotate_cw_device(torch.from_numpy(img).cuda().data_ptr(), w, h, w, 0, 1, o.data_ptr(), h, 0, s)
    assert np.array_equal(o.cpu().numpy(), oracle.rotate_cw(img)), f"rotate case {c}: {w}x{h}"
    if c % 50 == 49: print(f"{c + 1} cases ok ({time.time() - t0:.0f} s)", flush=True)
print("filter soak ok:", cases, "cases")
