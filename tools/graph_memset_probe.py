#!/usr/bin/env python3
"""Does the LIBRARY's captured single-pass launch replay correctly when the state is zeroed by hipMemsetAsync (round 1)
instead of k_state_clear (round 2)?  Needs the experiment build:
    make -C disparity_to_point_cloud_amd/csrc variant NAME=memset DEFS=-DD2PC_CLEAR_WITH_MEMSET=1
    D2PC_LIBRARY_VARIANT=memset D2PC_TRACE_MEMSET=1 python tools/graph_memset_probe.py
Prints, per replay, the frames' counts against the expected ones and the timeout flag.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc  # noqa: E402
from disparity_to_point_cloud_amd.synth import synth_disparity  # noqa: E402
from disparity_to_point_cloud_amd.torch_api import DeviceBatch  # noqa: E402

print("library:", d2pc.capi.library_path())
q = d2pc.make_q()
for (w, h, n) in ((640, 480, 6), (1920, 1080, 8)):
    frames = [synth_disparity(2, f, w, h, "holes") for f in range(n)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=2) as ctx:
        ctx.set_tuning("spin_timeout_ms", 200)
        b = DeviceBatch(ctx, n, h, w, want_index=True)
        b.disp.copy_(torch.from_numpy(np.stack(frames)))
        b.launch()
        torch.cuda.synchronize()
        want = b.counts.cpu().numpy().copy()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            b.launch()
        for r in range(4):
            b.points.fill_(0)
            b.index.fill_(0)
            b.counts.fill_(0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            got = b.counts.cpu().numpy()
            try:
                ctx.check_async_error()
                flag = "clear"
            except d2pc.D2pcError as e:
                flag = "TIMEOUT FLAG SET"
            print(f"{w}x{h} x{n} replay {r}: counts {'ok' if np.array_equal(got, want) else 'WRONG ' + str(got.tolist())}, {flag}", flush=True)
