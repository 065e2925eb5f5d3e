#!/usr/bin/env python3
"""(round 6) Energy per launch of the callback-body kernels, by library build: time x socket power over ~1 s of back-to-back launches, the
sysfs telemetry of the computing card sampled every 4 ms (disparity_to_point_cloud_amd/telemetry.py).  The PARITY body runs at the socket power
cap, so what a variant changes in ENERGY is what it can change in time.  Variants: `make variant NAME=nolds DEFS=-DD2PC_BS_NO_LDS=1` (the
select's instructions without its LDS reads: wrong results, timing only).
    python tools/energy_probe.py base,nolds"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd import telemetry
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

libs = (sys.argv[1] if len(sys.argv) > 1 else "base").split(",")
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
dev, (W, H, F) = "cuda:0", (3840, 2160, 16)
card = telemetry.find_card(pci_address=telemetry.torch_pci_address(0))
raw = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(0xD2C))
s = torch.cuda.current_stream().cuda_stream
idle = None
with telemetry.Sampler(card, 0.004) as smp:
    time.sleep(0.4)
idle_w = smp.summary().get("metrics_socket_power_W", {}).get("median")
print(f"# card {card}; idle {idle_w} W; energy = median power x time per launch; dynamic = (power - idle) x time")
print(f"{'build':8s} {'kernel':44s} {'us/launch':>9s} {'sclk MHz':>9s} {'power W':>8s} {'mJ/launch':>10s} {'dynamic mJ':>10s}")
rows = {}
for rep in range(2):            # twice, interleaved: the two passes must agree
    for lib in libs:
        ctx = d2pc.Context(q=d2pc.make_q(), border=40, mode=d2pc.MODE_PARITY, variant=None if lib == "base" else lib)
        b3 = DeviceBatch(ctx, F, H, W, dtype=torch.uint8, device=dev)

        def body():
            ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, b3.points.data_ptr(), None, b3.stride,
                                    b3.counts.data_ptr(), s)

        def med():
            ctx.median_roi_device(raw.data_ptr(), W, H, W, W * H, F, b3.disp.data_ptr(), W, W * H, 11, s)

        for name, fn in (("callback body PARITY (k_callback_bs<11>)", body), ("median 11 x 11 over the ROI (k_median_bs_u8)", med)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n, ms = 0, 0.0
            with telemetry.Sampler(card, 0.004) as smp:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < seconds:
                    e0.record()
                    for _ in range(32):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ms += e0.elapsed_time(e1)
                    n += 32
            t = smp.summary()
            per = ms / n
            pw = t.get("metrics_socket_power_W", {}).get("median", float("nan"))
            clk = t.get("sclk_MHz", {}).get("median", float("nan"))
            print(f"{lib:8s} {name:44s} {per * 1e3:9.1f} {clk:9.0f} {pw:8.0f} {pw * per:10.1f} {(pw - (idle_w or 0)) * per:10.1f}", flush=True)
        del b3
        ctx.close()
