#!/usr/bin/env python3
"""Randomised soak of the one-launch COMPACT forms on CAMERA-TO-4K-SIZE frames through the DEFAULT routing (k_compact_resident for
<= 1024 ordinary tiles, k_compact_resident_lean<32 / 64> beyond, two-pass where nothing is resident): one or two frames, random sizes
around 2-9 Mpixel, borders, dtypes, hole patterns, stereoRectify and general Q, both OpenCV forms, against the oracle.  GPU box:
    python tools/soak_lean.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch
import oracle
from helpers import assert_points_close

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
t0 = time.time()
launched = {"resident": 0, "other": 0}
for c in range(cases):
    n = int(rng.choice([1, 1, 2, 3]))
    w, h = int(rng.integers(1500, 4300)), int(rng.integers(900, 2400))
    border = int(rng.choice([0, 3, 40, 57]))
    dt = rng.choice(["f32", "f32", "u8", "u16"])
    holes = float(rng.choice([0.0, 0.3, 0.3, 0.9]))
    stereo = bool(rng.random() < 0.8)
    q = d2pc.make_q(cx=w / 2 - 0.37, cy=h / 2 + 0.21, nx=w, ny=h)
    if not stereo:
        q = rng.uniform(-2, 2, 16)
        q[12:14] = rng.uniform(0, 1e-3, 2); q[14] = rng.uniform(0.01, 1); q[15] = rng.uniform(0.1, 2)
    elif rng.random() < 0.3:
        q[15] = rng.uniform(-1, 1)
    if dt == "f32":
        frames = rng.uniform(0.5, 128, size=(n, h, w)).astype(np.float32); scale = 1.0; tdt = torch.float32
    elif dt == "u8":
        frames = rng.integers(1, 256, size=(n, h, w)).astype(np.uint8); scale = 0.125; tdt = torch.uint8
    else:
        frames = rng.integers(1, 65536, size=(n, h, w)).astype(np.uint16); scale = 1.0 / 64; tdt = torch.uint16
    if rng.random() < 0.5:
        frames[rng.random((n, h, w)) < holes] = 0
    else:   # blocky holes
        m = rng.random((n, (h + 63) // 64, (w + 63) // 64)) < holes
        frames[np.repeat(np.repeat(m, 64, axis=1), 64, axis=2)[:, :h, :w]] = 0
    rform = int(rng.choice([0, 0, 24, 4])) if stereo else int(rng.choice([0, 4]))
    with d2pc.Context(q=q, border=border, mode=d2pc.MODE_COMPACT) as ctx:
        ctx.set_reproject_form(rform)
        b = DeviceBatch(ctx, n, h, w, dtype=tdt, want_index=True)
        b.disp.copy_(torch.from_numpy(frames.view(np.int16) if dt == "u16" else frames).view(tdt))
        ctx.compact_stats_reset()
        for _ in range(2):
            b.points.fill_(0)
            b.launch(scale=scale)
        res = b.results()
        ctx.check_async_error()
        st = ctx.compact_stats()
        assert st["timeouts"] == 0
        launched["resident" if st["launches"] == 2 else "other"] += 1
    ulp, form = (1, oracle.FORM_CV24) if stereo else (0, oracle.FORM_CV4)
    if rform:
        ulp, form = 0, (oracle.FORM_CV24 if rform == 24 else oracle.FORM_CV4)
    what = f"case {c}: n={n} {w}x{h} b={border} {dt} holes={holes} stereo={stereo} form={rform}"
    for f in range(n):
        wp, wi = oracle.reproject_compact(frames[f], q, border=border, scale=scale, form=form)
        assert len(res[f][0]) == len(wp), what + f" frame {f}: {len(res[f][0])} vs {len(wp)} points"
        assert np.array_equal(res[f][1], wi), what
        assert_points_close(res[f][0], wp, max_ulp=ulp, rel=1e-5, what=what)
    if c % 10 == 9:
        print(f"{c + 1} cases ok ({time.time() - t0:.0f} s); one-launch resident forms served {launched['resident']}, other forms {launched['other']}", flush=True)
print("lean soak ok:", cases, "cases;", launched)
