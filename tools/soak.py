#!/usr/bin/env python3
"""Randomised soak of the device-resident entry point against the oracle: random frame counts,
sizes, borders, dtypes, hole patterns, both compaction algorithms, index on/off, general and
stereoRectify-structured Q.  Not part of the test suites (minutes); run on the GPU box:
    python tools/soak.py [cases] [seed]
"""
import os, sys, time
os.environ.setdefault("D2PC_LIBRARY_VARIANT", "exp")   # draws from the laboratory too (compact_algo 4, tile shapes 4 / 16): the experiment build
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch
import oracle
from helpers import assert_points_close


def check(got, want, ulp, what):
    return assert_points_close(got, want, max_ulp=ulp, rel=1e-5, what=what)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t0 = time.time()
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 20]))
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    if rng.random() < 0.15:
        w, h = int(rng.choice([752, 1920, 640])), int(rng.choice([480, 1080]))
        n = min(n, 5)
    border = int(rng.choice([0, 0, 1, 3, 8, 40, 40, 57]))
    dt = rng.choice(["f32", "u8", "u16"])
    mode = rng.choice(["parity", "compact"])
    algo = int(rng.choice([0, 1, 2, 3, 4]))
    idx = bool(rng.random() < 0.5)
    holes = float(rng.choice([0.0, 0.05, 0.3, 0.9, 1.0]))
    stereo = bool(rng.random() < 0.7)
    q = d2pc.make_q()
    if not stereo:
        q = rng.uniform(-2, 2, 16)
        q[12:14] = rng.uniform(0, 1e-3, 2); q[14] = rng.uniform(0.01, 1); q[15] = rng.uniform(0.1, 2)
    elif rng.random() < 0.3:
        q = d2pc.make_q(fx=500 + 300 * rng.random(), fy=600 + 200 * rng.random(), cx=w / 2, cy=h / 2, baseline=0.05 + rng.random(), nx=w + 1, ny=h + 1)
    if stereo and rng.random() < 0.3:
        q[15] = rng.uniform(-1, 1)   # stereoRectify without CALIB_ZERO_DISPARITY: W = q33 + RN(q32 * d) rounds twice in both OpenCV forms
    if dt == "f32":
        frames = rng.uniform(0.5, 128, size=(n, h, w)).astype(np.float32); scale = 1.0; tdt = torch.float32
    elif dt == "u8":
        frames = rng.integers(1, 256, size=(n, h, w)).astype(np.uint8); scale = 0.125; tdt = torch.uint8
    else:
        frames = rng.integers(1, 65536, size=(n, h, w)).astype(np.uint16); scale = 1.0 / 64; tdt = torch.uint16
    frames[rng.random((n, h, w)) < holes] = 0
    m = d2pc.MODE_PARITY if mode == "parity" else d2pc.MODE_COMPACT
    rform = int(rng.choice([0, 0, 24, 4])) if stereo else int(rng.choice([0, 4]))   # d2pc_set_reproject_form
    fgen = bool(stereo and rng.random() < 0.3)   # stereoRectify's Q through the general kernel (the other route to the same bytes)
    with d2pc.Context(q=q, border=border, mode=m, compact_algo=algo) as ctx:
        ctx.set_reproject_form(rform)
        ctx.set_test_hook("force_general_q", int(fgen))
        if algo == 4:
            ctx.set_tuning("chunk_mb", int(rng.choice([1, 2, 96])))
        if algo == 2:   # the single pass in its kernel forms (0 = the product's default, form 2; 1 / 3 / 4: experiment build)
            ctx.set_tuning("onepass_form", int(rng.choice([0, 0, 1, 2, 3, 4, 5, 6, 7])))
        if algo == 3:   # the ordinary resident tile, or the register-resident blocks of 32 / 64 pixels per thread
            ctx.set_tuning("resident_pxt", int(rng.choice([0, 32, 64])))
            ctx.set_tuning("resident_stagger_pct", int(rng.choice([-1, 0, 100])))
        b = DeviceBatch(ctx, n, h, w, dtype=tdt, want_index=idx)
        b.disp.copy_(torch.from_numpy(frames.view(np.int16) if dt == "u16" else frames).view(tdt))
        b.launch(scale=scale)
        res = b.results()
        if m == d2pc.MODE_COMPACT:
            ctx.check_async_error()
    # A general (dense) Q is evaluated in OpenCV 3/4's association: 0 ulp against the oracle's FORM_CV4 (round 2's fused
    # multiply-adds were up to 71 float ulp from the 2.4 form where a numerator cancels: that loosened bar is gone).
    # stereoRectify-structured Q: the specialised kernel, 1 ulp from the 2.4 form.
    # An explicit form (24: OpenCV 2.4's loop, 4: OpenCV 3/4's) is that generation bit for bit, for stereoRectify's Q too.
    ulp, form = (1, oracle.FORM_CV24) if stereo and not fgen else (0, oracle.FORM_CV4)
    if rform:
        ulp, form = 0, (oracle.FORM_CV24 if rform == 24 else oracle.FORM_CV4)
    what = f"case {c}: n={n} {w}x{h} b={border} {dt} {mode} algo={algo} idx={idx} holes={holes} stereo={stereo} form={rform} general_route={fgen}"
    for f in range(n):
        if m == d2pc.MODE_PARITY:
            want = oracle.reproject(frames[f], q, border=border, scale=scale, form=form)
            check(res[f][0], want, ulp, what)
            if idx:
                rw, rh = max(w - 2 * border, 0), max(h - 2 * border, 0)
                vv, uu = np.divmod(np.arange(rw * rh), max(rw, 1))
                assert np.array_equal(res[f][1], ((vv + border) * w + uu + border).astype(np.uint32)), what
        else:
            wp, wi = oracle.reproject_compact(frames[f], q, border=border, scale=scale, form=form)
            assert len(res[f][0]) == len(wp), what + f" frame {f}: {len(res[f][0])} vs {len(wp)} points"
            check(res[f][0], wp, ulp, what)
            if idx:
                assert np.array_equal(res[f][1], wi), what
    if c % 20 == 19:
        print(f"{c + 1} cases ok ({time.time() - t0:.0f} s)", flush=True)
print("soak ok:", cases, "cases")
