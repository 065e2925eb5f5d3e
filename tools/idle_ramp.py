#!/usr/bin/env python3
"""How long may the device idle before the next launches run slow?  Heat with 100 headline launches, idle for T (a host
synchronisation plus a sleep), then time 24 launches one by one.  GPU box only."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

ctx = d2pc.Context(q=d2pc.make_q())
b = DeviceBatch(ctx, 16, 2160, 3840)
b.disp.copy_(torch.rand(b.disp.shape, device="cuda") * 127.5 + 0.5)
for idle_ms in (0.0, 0.05, 0.5, 5.0, 50.0, 500.0):
    for _ in range(100):
        b.launch()
    torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms / 1e3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
    ev[0].record()
    for i in range(24):
        b.launch()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(24)]
    print(f"idle {idle_ms:6.2f} ms after the synchronisation: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
