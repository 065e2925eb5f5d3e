#!/usr/bin/env python3
"""Diagnostic: phase timers of the single-pass compaction kernel (needs
`make -C disparity_to_point_cloud_amd/csrc diag`).  GPU box only."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disparity_to_point_cloud_amd import capi
capi._LIB_NAME = 'libd2pc_%s.so' % (sys.argv[1] if len(sys.argv) > 1 else 'diag')
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

lib = d2pc.load_library()
hip = ctypes.CDLL("libamdhip64.so.7")
q = d2pc.make_q()
for pxt in (8,):
    for bpc in (3, 4, 5):
        ctx = d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=2)
        ctx.set_tuning("pxt_compact", pxt); ctx.set_tuning("onepass_blocks_per_cu", bpc)
        b = DeviceBatch(ctx, 16, 2160, 3840)
        b.disp.copy_(torch.rand(b.disp.shape, device="cuda") * 127.5 + 0.5)
        for _ in range(3):
            b.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.launch(); e1.record(); torch.cuda.synchronize()
        # read the header back: ctx keeps d_state private, so find it via a tiny helper: check_async_error copies the
        # header; we re-read it here through hipMemcpy using the pointer stored at a known offset is not exposed =>
        # use the debug export below.
        buf = (ctypes.c_ulonglong * 16)()  # the 128-byte StateHeader: flag, stats pointer, diag[7]
        lib.d2pc_debug_read_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.d2pc_debug_read_header(ctx.handle, buf)
        its, spins, wA, wB, wC, wD, cA = [buf[i] for i in range(2, 9)]
        its = max(its, 1)
        print(f"pxt={pxt:2d} bpc={bpc:2d} kernel={e0.elapsed_time(e1)*1e3:8.1f}us iterations={its} spins/it={spins/its:6.2f} "
              f"cycles/it worker: count={wA/its:6.0f} barrier1={wB/its:6.0f} scan-wait={wC/its:6.0f} loads+scatter={wD/its:6.0f} "
              f"sum={(wA+wB+wC+wD)/its:6.0f} | control ticket+prefix={cA/its:6.0f}", flush=True)
        ctx.close()
