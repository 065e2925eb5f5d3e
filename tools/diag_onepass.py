#!/usr/bin/env python3
"""Diagnostic: phase timers of the single-pass compaction kernel k_compact_onepass (needs
`make -C disparity_to_point_cloud_amd/csrc diag` -> libd2pc_diag.so; in the product no stamp executes).  GPU box only.

  python tools/diag_onepass.py [variant [forms]] > profiles/rNN_onepass_phases.txt      (forms: 1,2 = tuning "onepass_form")

Per workload (16 x 4K and 32 x 1080p, border 40; 0 / 30 / 90 % iid holes; with and without indices) one line per
blocks-per-CU setting: the launch's time by HIP events, and -- summed over the blocks by lane 0 of worker wave 0 and of
the control wave, divided by the iterations -- the shader cycles of every phase of one pipeline iteration:

  worker : count | wait at barrier 1 | wait for scan + publish (barrier 2) | next tile's loads land | reproject + scatter
  control: ticket + prefix polls | wait at barrier 1 | scan + publish + barrier 2 | rest

and the floor these imply: iterations per block x cycles per iteration / clock (the blocks' own clock: shader cycles
of their lifetime / 100 MHz ticks of their lifetime)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "diag"
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

lib = d2pc.load_library(VARIANT)
lib.d2pc_debug_read_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
q = d2pc.make_q()
g = torch.Generator(device="cuda").manual_seed(7)


def run(frames, h, w, holes, idx, bpc, form=1):
    ctx = d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=2, variant=VARIANT)
    ctx.set_tuning("onepass_blocks_per_cu", bpc)
    ctx.set_tuning("onepass_form", form)
    b = DeviceBatch(ctx, frames, h, w, want_index=bool(idx))
    b.disp.copy_(torch.rand(b.disp.shape, generator=g, device="cuda") * 127.5 + 0.5)
    if holes > 0:
        b.disp.mul_((torch.rand(b.disp.shape, generator=g, device="cuda") >= holes).float())
    for _ in range(60):  # (the clock ramp after idleness: profiles/r03_warmup_ramp.txt)
        b.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.launch(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    buf = (ctypes.c_ulonglong * 16)()  # the 128-byte StateHeader of the LAST launch: flag, stats pointer, diag[7], pad2[7]
    assert lib.d2pc_debug_read_header(ctx.handle, buf) == 0
    its, spins, wA, wB, wC, wD, cA = [buf[i] for i in range(2, 9)]
    wE, life_cyc, life_ticks, blocks, cB, cC, cD = [buf[i] for i in range(9, 16)]
    its = max(its, 1)
    blocks = max(blocks, 1)
    clock_ghz = life_cyc / max(life_ticks, 1) * 0.1
    per_it = (wA + wB + wC + wD) / its
    npts = int(b.counts.sum().item())
    roi = d2pc.roi_points(w, h, 40)
    alg = 4 * frames * roi + (20 if idx else 16) * npts
    med = float(np.median(ts))
    print(f"form {form} {frames}x{w}x{h} holes={holes:.1f} idx={idx} bpc={bpc}: kernel {med:7.1f} us ({alg/med/1e3/8000:.3f} of 8 TB/s) "
          f"blocks={blocks} it/block={its/blocks:5.1f} failed polls/it={spins/its:5.2f} clock={clock_ghz:4.2f} GHz\n"
          f"    worker  cycles/it: count={wA/its:6.0f} barrier1={wB/its:6.0f} scan-wait={wC/its:6.0f} loads-land={wE/its:6.0f} "
          f"reproject+scatter={(wD-wE)/its:6.0f} sum={per_it:6.0f}\n"
          f"    control cycles/it: ticket+prefix={cA/its:6.0f} barrier1={cB/its:6.0f} scan+publish+barrier2={cC/its:6.0f} rest={cD/its:6.0f}\n"
          f"    floor = it/block x cycles/it / clock = {its/blocks*per_it/clock_ghz/1e3:7.1f} us; block lifetime {life_ticks/blocks/100:7.1f} us",
          flush=True)
    ctx.close()


print(f"# {torch.cuda.get_device_name(0)}; library variant '{VARIANT}' (in-kernel s_memtime stamps: ~+10 % wave cycles, "
      "so the times below are NOT the product's)")
FORMS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
for form in FORMS:
    for holes in (0.0, 0.3, 0.9):
        for idx in (0, 1):
            for bpc in (3, 4):
                run(16, 2160, 3840, holes, idx, bpc, form)
    for holes in (0.0, 0.3, 0.9):
        run(32, 1080, 1920, holes, 1, 4, form)
