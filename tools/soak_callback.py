#!/usr/bin/env python3
"""Randomised soak of d2pc_process_mono_device against the oracle: the tile-fused kernels (k_callback_bs in PARITY mode,
k_callback_bs_compact and k_callback_bs_compact_pipe in COMPACT mode: bit-sliced median of a tile + the tile's points)
forced onto random sizes, borders, pitches, scales, hole patterns, both forms of Q, with and without indices, U8 and
MONO16 input, all window sizes; every case also runs as two launches and must give the same bytes.
GPU box:  python tools/soak_callback.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch
import oracle
from helpers import assert_points_close

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time()
for c in range(cases):
    k = int(rng.choice([3, 5, 7, 9, 11])); n = int(rng.integers(1, 4))
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 900))
    border = int(rng.choice([0, 1, 7, 40]))
    scale = float(rng.choice([0.125, 1.0, 0.37, 1e-3]))
    mono16 = bool(rng.random() < 0.3)
    want_idx = bool(rng.random() < 0.5)
    general = int(rng.random() < 0.4)
    compact = bool(rng.random() < 0.6)
    holes = float(rng.choice([0.0, 0.3, 0.6, 1.0]))
    q = d2pc.make_q(fx=float(rng.uniform(300, 900)), fy=float(rng.uniform(300, 900)), cx=float(rng.uniform(100, 500)),
                    cy=float(rng.uniform(100, 300)), baseline=float(rng.uniform(0.05, 0.3)))
    pitch = w + int(rng.integers(0, 17))
    if mono16:
        imgs = rng.integers(0, 65536, size=(n, h, pitch)).astype(np.uint16)
        src = torch.from_numpy(imgs.view(np.int16)).cuda()
        m8 = np.stack([oracle.mono16_to_mono8(np.ascontiguousarray(i[:, :w])) for i in imgs])
        dt, rs = d2pc.DTYPE_MONO16, 2 * pitch
    else:
        imgs = rng.integers(0, 256, size=(n, h, pitch)).astype(np.uint8)
        if rng.random() < 0.3: imgs = (imgs // 64 * 85).astype(np.uint8)   # many ties
        if holes:   # regions without a match: zero in blocks of 16 x 16
            m = rng.random((n, (h + 15) // 16, (pitch + 15) // 16)) < holes
            imgs[np.repeat(np.repeat(m, 16, axis=1), 16, axis=2)[:, :h, :pitch]] = 0
        src = torch.from_numpy(imgs).cuda()
        m8 = np.ascontiguousarray(imgs[:, :, :w])
        dt, rs = d2pc.DTYPE_U8, pitch
    res = {}
    forms = (2, 1, 0) if compact else (1, 0)
    rform = int(rng.choice([0, 0, 24, 4]))   # d2pc_set_reproject_form: one OpenCV generation bit for bit
    with d2pc.Context(q=q, border=border, mode=d2pc.MODE_COMPACT if compact else d2pc.MODE_PARITY) as ctx:
        ctx.set_test_hook("force_general_q", general)
        ctx.set_reproject_form(rform)
        ctx.set_tuning("median_algo", 2)
        b = DeviceBatch(ctx, n, h, w, dtype=torch.uint8, want_index=want_idx)
        s = torch.cuda.current_stream().cuda_stream
        for fused in forms:
            ctx.set_tuning("callback_fused_compact" if compact else "callback_fused", fused)
            b.points.fill_(0); b.counts.fill_(0)
            if want_idx: b.index.fill_(-1)
            ctx.process_mono_device(src.data_ptr(), dt, w, h, rs, rs * h, n, k, scale, b.points.data_ptr(),
                                    b.index.data_ptr() if want_idx else None, b.stride, b.counts.data_ptr(), s)
            torch.cuda.synchronize()
            res[fused] = [b.points.cpu().numpy().copy(), b.counts.cpu().numpy().copy()] + ([b.index.cpu().numpy().copy()] if want_idx else [])
        ctx.check_async_error()
    what = f"case {c}: k={k} {w}x{h} n={n} border={border} scale={scale} mono16={mono16} idx={want_idx} general={general} compact={compact} holes={holes} form={rform}"
    for fused in forms[:-1]:
        for x, y in zip(res[fused], res[0]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), what + f": fused form {fused} != two launches"
    form, ulp = (oracle.FORM_CV4, 0) if general else (oracle.FORM_CV24, 1)   # a general Q is OpenCV 4's association bit for bit
    if rform:
        form, ulp = (oracle.FORM_CV24 if rform == 24 else oracle.FORM_CV4), 0
    for f in range(n):
        filt = oracle.median_u8(m8[f], k)
        if compact:
            want, wi = oracle.reproject_compact(filt, q, border=border, scale=scale, form=form)
            if want_idx:
                assert np.array_equal(res[forms[0]][2].view(np.uint32)[f][:len(wi)], wi), what
        else:
            want = oracle.reproject(filt, q, border=border, scale=scale, form=form)
        assert res[forms[0]][1].view(np.uint32)[f] == len(want), what
        if len(want):
            assert_points_close(res[forms[0]][0][f][:len(want)], want, max_ulp=ulp, rel=1e-5, what=what + f" frame {f}")
    if c % 20 == 19: print(f"{c + 1} cases ok ({time.time() - t0:.0f} s)", flush=True)
print("all", cases, "cases ok")
