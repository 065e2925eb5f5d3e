#!/usr/bin/env python3
"""What distinguishes a "slow-class" from a "fast-class" device?

Runs the bench's kernels one after the other for ~1.5 s each while a sampler thread reads the amdgpu sysfs telemetry
(disparity_to_point_cloud_amd/telemetry.py): shader / memory / fabric clock, socket power, temperatures, partition modes.
Prints one line per phase: the kernel's time per launch and rate, and the telemetry's median (min..max).

    python tools/devclass_probe.py [--dump] [--seconds 1.5]      # --dump: list the sysfs files and a gpu_metrics hexdump first
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from disparity_to_point_cloud_amd import telemetry  # noqa: E402  (no GPU call in there)


def dump(card):
    print("# card:", card)
    if not card:
        return
    for root in (card, telemetry._hwmon(card) or ""):
        if not root:
            continue
        for name in sorted(os.listdir(root)):
            p = os.path.join(root, name)
            if os.path.isfile(p):
                v = telemetry._read(p)
                if v is None:
                    v = "<unreadable>"
                elif len(v) > 200 or "\x00" in v:
                    v = f"<{len(v)} bytes>"
                print(f"#   {os.path.relpath(p, card)} = {v.strip()!r}")
    blob = telemetry._read(os.path.join(card, "gpu_metrics"), binary=True)
    if blob:
        print("# gpu_metrics", len(blob), "bytes:", blob[:160].hex())
        print("#", telemetry.parse_gpu_metrics(blob))


def fmt(summary, keys):
    parts = []
    for k in keys:
        if k in summary:
            s = summary[k]
            parts.append(f"{k} {s['median']:g} ({s['min']:g}..{s['max']:g})")
    return "  ".join(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump", action="store_true")
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--json", default=None, help="also write the phases as JSON to this path")
    a = ap.parse_args()
    import numpy as np
    import torch
    addr = telemetry.torch_pci_address(0)
    print("# cards visible in sysfs:", [(d, a_, telemetry._read(os.path.join(d, "gpu_busy_percent"))) for d, a_ in telemetry.list_cards()])
    print("# torch cuda:0 is at PCI", addr, "| HIP_VISIBLE_DEVICES", os.environ.get("HIP_VISIBLE_DEVICES"), "| ROCR_VISIBLE_DEVICES",
          os.environ.get("ROCR_VISIBLE_DEVICES"))
    card = telemetry.find_card(pci_address=addr) if addr else telemetry.find_card(0)
    if card is None:
        print("# the computing device's sysfs node is not visible in this container: no telemetry")
    if a.dump:
        dump(card)
    print("# static:", json.dumps(telemetry.static_state(card)))
    import disparity_to_point_cloud_amd as d2pc
    from disparity_to_point_cloud_amd.synth import synth_disparity
    from disparity_to_point_cloud_amd.torch_api import DeviceBatch

    dev = "cuda:0"
    W, H, F = 3840, 2160, 16
    q = d2pc.make_q()
    keys = ("sclk_MHz", "dpm_sclk_MHz", "mclk_MHz", "dpm_mclk_MHz", "dpm_fclk_MHz", "dpm_socclk_MHz", "power_W", "power_input_W",
            "metrics_socket_power_W", "metrics_gfx_activity", "metrics_umc_activity", "temp_junction_C", "metrics_temp_hotspot_C",
            "temp_mem_C", "metrics_temp_mem_C", "gpu_busy_percent")
    phases = []

    def run(name, launch, unit_bytes=None):
        s = torch.cuda.current_stream()
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # batches of launches between synchronisations, so the sampler sees a busy device and the queue stays short
        e0.record()
        for _ in range(5):
            launch()
        e1.record()
        torch.cuda.synchronize()
        per = max(e0.elapsed_time(e1) / 5, 1e-3)
        batch = max(int(20.0 / per), 1)   # ~20 ms per batch
        n = 0
        ms = 0.0
        import time
        with telemetry.Sampler(card, 0.004) as smp:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.seconds:
                e0.record()
                for _ in range(batch):
                    launch()
                e1.record()
                torch.cuda.synchronize()
                ms += e0.elapsed_time(e1)
                n += batch
        summ = smp.summary()
        per = ms / n
        rate = f"  {unit_bytes / (per * 1e-3) / 1e9:8.1f} GB/s" if unit_bytes else ""
        print(f"{name:44s} {per * 1e3:9.1f} us/launch{rate}  [{summ.get('samples', 0)} samples]  {fmt(summ, keys)}", flush=True)
        phases.append({"phase": name, "us_per_launch": round(per * 1e3, 2), "GBs": round(unit_bytes / (per * 1e-3) / 1e9, 1) if unit_bytes else None,
                       "telemetry": summ})

    import time
    with telemetry.Sampler(card, 0.004) as smp:
        time.sleep(0.5)
    idle = smp.summary()
    print(f"{'idle':44s} {'':9s}            [{idle.get('samples', 0)} samples]  {fmt(idle, keys)}")
    phases.append({"phase": "idle", "telemetry": idle})

    ctx = d2pc.Context(q=q, border=40, mode=d2pc.MODE_PARITY)
    batch = DeviceBatch(ctx, F, H, W, want_index=False, device=dev)
    for f in range(F):
        batch.disp[f].copy_(torch.from_numpy(synth_disparity(4, f, W, H, "uniform")))
    s = torch.cuda.current_stream().cuda_stream
    nbytes = batch.points.numel() * 4 // 32 * 32
    base = batch.points.data_ptr()
    run("fill, persistent blocks (8/CU, plain)", lambda: ctx.membench_fill(base, nbytes, s), nbytes)
    ctx.set_tuning("membench_blocks_per_cu", 0)
    ctx.set_tuning("membench_unroll", 1)
    run("fill, one-shot blocks (plain)", lambda: ctx.membench_fill(base, nbytes, s), nbytes)
    ctx.set_tuning("membench_blocks_per_cu", 8)
    ctx.set_tuning("membench_unroll", 4)
    batch.launch()
    torch.cuda.synchronize()
    alg = 4 * F * batch.roi_n + 16 * int(batch.counts.sum().item())
    run("PARITY 16 x 4K (k_reproject_pack_small)", batch.launch, alg)

    c2 = d2pc.Context(q=q, border=40, mode=d2pc.MODE_COMPACT)
    b2 = DeviceBatch(c2, F, H, W, want_index=False, device=dev)
    for f in range(F):
        b2.disp[f].copy_(torch.from_numpy(synth_disparity(4, f, W, H, "holes")))
    b2.launch()
    torch.cuda.synchronize()
    alg2 = 4 * F * b2.roi_n + 16 * int(b2.counts.sum().item())
    run("COMPACT 30 % holes 16 x 4K (single pass)", b2.launch, alg2)
    c2.check_async_error()

    raw = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(0xD2C))
    b3 = DeviceBatch(ctx, F, H, W, dtype=torch.uint8, device=dev)

    def body():
        ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, b3.points.data_ptr(), None, b3.stride,
                                b3.counts.data_ptr(), s)
    run("callback body PARITY (k_callback_bs<11>)", body, 17 * F * b3.roi_n)

    # the same body on a DISPARITY-LIKE batch (what a block matcher delivers: piecewise-smooth surfaces, a little noise, 64 x 64
    # no-match blocks): a power-capped kernel's clock depends on how many bits toggle, and iid uniform bytes toggle the most
    gen = torch.Generator(device=dev).manual_seed(0xD2E)
    yy = torch.arange(H, device=dev, dtype=torch.float32)[None, :, None]
    xx = torch.arange(W, device=dev, dtype=torch.float32)[None, None, :]
    ff = torch.arange(F, device=dev, dtype=torch.float32)[:, None, None]
    surf = 40.0 + 30.0 * torch.sin(xx / 517.0 + ff) * torch.cos(yy / 389.0) + 0.02 * xx + 0.03 * yy
    steps = (torch.rand((F, (H + 127) // 128, (W + 127) // 128), device=dev, generator=gen) * 4).floor() * 24.0
    surf = surf + steps.repeat_interleave(128, dim=1).repeat_interleave(128, dim=2)[:, :H, :W]
    surf = surf + torch.randn((F, H, W), device=dev, generator=gen) * 1.5
    smooth = surf.clamp(1, 255).to(torch.uint8)
    holes = torch.rand((F, (H + 63) // 64, (W + 63) // 64), device=dev, generator=gen) < 0.1
    smooth[holes.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H, :W]] = 0
    del surf, steps, holes

    def body_smooth():
        ctx.process_mono_device(smooth.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125, b3.points.data_ptr(), None, b3.stride,
                                b3.counts.data_ptr(), s)
    run("callback body PARITY, disparity-like frames", body_smooth, 17 * F * b3.roi_n)

    def med():
        ctx.median_roi_device(raw.data_ptr(), W, H, W, W * H, F, b3.disp.data_ptr(), W, W * H, 11, s)
    run("median 11 x 11 over the ROI (k_median_bs_u8)", med, 2 * F * b3.roi_n)
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"static": telemetry.static_state(card), "phases": phases}, f, indent=1)


if __name__ == "__main__":
    main()
