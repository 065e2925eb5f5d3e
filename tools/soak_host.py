#!/usr/bin/env python3
"""Randomised soak of the HOST entry points against the oracle: d2pc_process (f32/u8/u16, strided rows),
d2pc_process_mono8 / _mono16 (device median), and the pipelined path (random depth, direct/staged,
MONO16 frames).  GPU box:  python tools/soak_host.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import disparity_to_point_cloud_amd as d2pc
import oracle
from helpers import assert_points_close

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
q = d2pc.make_q()
t0 = time.time()

def want_for(img8_or_f, scale, border, mode, k):
    filt = oracle.median_u8(img8_or_f, k) if k else img8_or_f
    if mode == d2pc.MODE_PARITY:
        return oracle.reproject(filt, q, border=border, scale=scale), None
    return oracle.reproject_compact(filt, q, border=border, scale=scale)

for c in range(cases):
    w, h = int(rng.integers(1, 500)), int(rng.integers(1, 400))
    border = int(rng.choice([0, 2, 40, 40]))
    mode = int(rng.choice([d2pc.MODE_PARITY, d2pc.MODE_COMPACT]))
    k = int(rng.choice([0, 3, 5, 7, 9, 11]))
    holes = float(rng.choice([0.0, 0.3, 1.0]))
    u8 = rng.integers(1, 256, size=(h, w)).astype(np.uint8); u8[rng.random((h, w)) < holes] = 0
    u16 = rng.integers(0, 65536, size=(h, w)).astype(np.uint16)
    f32 = np.zeros((h, w + 3), dtype=np.float32)[:, :w]  # strided rows
    f32[:] = rng.uniform(0.5, 128, size=(h, w)); f32[rng.random((h, w)) < holes] = 0
    what = f"case {c}: {w}x{h} b={border} mode={mode} k={k} holes={holes}"
    with d2pc.Context(q=q, border=border, mode=mode) as ctx:
        ctx.set_tuning("stage_timing", int(rng.integers(0, 2)))
        # synchronous entry points
        got = ctx.process(f32, want_index=(mode == d2pc.MODE_COMPACT))
        wp, wi = want_for(f32, 1.0, border, mode, 0)
        if mode == d2pc.MODE_COMPACT:
            assert np.array_equal(got[1], wi), what
            got = got[0]
        assert_points_close(got, wp, max_ulp=1, what=what + " process f32")
        got = ctx.process_mono8(u8, median_ksize=k)
        assert_points_close(got, want_for(u8, 0.125, border, mode, k)[0], max_ulp=1, what=what + " mono8")
        got = ctx.process_mono16(u16, median_ksize=k)
        assert_points_close(got, want_for(oracle.mono16_to_mono8(u16), 0.125, border, mode, k)[0], max_ulp=1, what=what + " mono16")
        # pipelined path
        depth = int(rng.integers(1, 4)); direct = bool(rng.integers(0, 2))
        ctx.pipeline_configure(depth=depth, direct_host_write=direct)
        frames = [("u8", u8), ("m16", u16), ("f32", np.ascontiguousarray(f32)), ("u8", u8[::-1].copy())][: int(rng.integers(1, 5))]
        inflight, results = 0, []
        for i, (kind, img) in enumerate(frames):
            if inflight == depth:
                results.append(ctx.pipeline_collect()); inflight -= 1
            kk = k if kind in ("u8", "m16") else 0
            ctx.pipeline_submit(img, scale=1.0 if kind == "f32" else 0.125, median_ksize=kk, want_index=True, tag=i, mono16=(kind == "m16"))
            inflight += 1
        while inflight:
            results.append(ctx.pipeline_collect()); inflight -= 1
        for i, (p, idx, tag, _) in enumerate(results):
            assert tag == i, what
            kind, img = frames[i]
            base = oracle.mono16_to_mono8(img) if kind == "m16" else img
            wp, wi = want_for(base, 1.0 if kind == "f32" else 0.125, border, mode, k if kind != "f32" else 0)
            assert_points_close(p, wp, max_ulp=1, what=what + f" pipeline frame {i} {kind}")
            if mode == d2pc.MODE_COMPACT:
                assert np.array_equal(idx, wi), what
    if c % 20 == 19: print(f"{c + 1} cases ok ({time.time() - t0:.0f} s)", flush=True)
print("host soak ok:", cases, "cases")
