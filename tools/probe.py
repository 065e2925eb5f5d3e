#!/usr/bin/env python3
"""One workload, a few launches -- the program rocprofv3 --pmc / --stats passes are run on when a
kernel's counters must not be mixed with other workloads' (bench.py runs several variants of the same
kernel).  GPU box only.  Prints the HIP-event time per launch and the algorithmic bytes.

  rocprofv3 --pmc SQ_INSTS_VMEM_WR ... --kernel-trace --output-format csv -d gpurun_out/x -- \
      python3 tools/probe.py --workload compact_holes_idx --launches 6
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc
from disparity_to_point_cloud_amd.torch_api import DeviceBatch

WORKLOADS = {  # name: (mode, frames, W, H, border, hole fraction, blocky, indices, dtype)
    "parity":             ("parity", 16, 3840, 2160, 40, 0.0, 0, False, "f32"),
    "parity_u8":          ("parity", 16, 3840, 2160, 40, 0.0, 0, False, "u8"),
    "compact_allvalid":   ("compact", 16, 3840, 2160, 40, 0.0, 0, False, "f32"),
    "compact_holes_idx":  ("compact", 16, 3840, 2160, 40, 0.3, 0, True, "f32"),
    "compact_blocky_idx": ("compact", 16, 3840, 2160, 40, 0.3, 1, True, "f32"),
    "compact_1080p_x32":  ("compact", 32, 1920, 1080, 40, 0.3, 0, True, "f32"),
    "compact_1080p_x1":   ("compact", 1, 1920, 1080, 40, 0.3, 0, True, "f32"),
    "compact_4k_x1":      ("compact", 1, 3840, 2160, 40, 0.3, 0, True, "f32"),
    "compact_4k_x1_allvalid": ("compact", 1, 3840, 2160, 40, 0.0, 0, False, "f32"),
    "parity_4k_x1":       ("parity", 1, 3840, 2160, 40, 0.0, 0, False, "f32"),
    "median11_roi":       ("median", 16, 3840, 2160, 40, 0.0, 0, False, "u8"),
    "callback_u8":        ("callback", 16, 3840, 2160, 40, 0.0, 0, False, "u8"),
    "callback_u8_compact_blocky": ("callback", 16, 3840, 2160, 40, 0.3, 1, True, "u8"),   # COMPACT + indices, 30 % zero pixels in 64 x 64 blocks
    "callback_u8_compact_iid":    ("callback", 16, 3840, 2160, 40, 0.3, 0, True, "u8"),
}


def make_batch(name, algo=0, device="cuda:0", tuning=()):
    mode, F, W, H, border, holes, blocky, idx, dt = WORKLOADS[name]
    ctx = d2pc.Context(q=d2pc.make_q(), border=border,
                       mode=d2pc.MODE_PARITY if mode == "parity" else d2pc.MODE_COMPACT, compact_algo=algo)
    for k, v in tuning:
        ctx.set_tuning(k, v)
    b = DeviceBatch(ctx, F, H, W, dtype=torch.float32 if dt == "f32" else torch.uint8, want_index=idx, device=device)
    g = torch.Generator(device=device).manual_seed(0xD2C)
    if dt == "u8":
        b.disp.copy_(torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device=device, generator=g))
    else:
        b.disp.copy_(torch.rand((F, H, W), generator=g, device=device) * 127.5 + 0.5)
        if holes > 0 and blocky:
            m = (torch.rand((F, (H + 63) // 64, (W + 63) // 64), generator=g, device=device) >= holes).float()
            b.disp.mul_(m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H, :W])
        elif holes > 0:
            b.disp.mul_((torch.rand((F, H, W), generator=g, device=device) >= holes).float())
    return ctx, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="compact_holes_idx", choices=sorted(WORKLOADS))
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--scale", type=float, default=None)
    ap.add_argument("--tune", action="append", default=[], help="key=value for d2pc_set_tuning (repeatable)")
    a = ap.parse_args()
    if WORKLOADS[a.workload][0] in ("median", "callback"):   # the 11 x 11 median over the inset ROI, 16 x 4K (cpp:55-57); --tune median_algo=1|2 picks the kernel
        _, F, W, H, border, holes, blocky, want_idx, _ = WORKLOADS[a.workload]
        ctx = d2pc.Context(q=d2pc.make_q(), border=border, mode=d2pc.MODE_COMPACT if want_idx else d2pc.MODE_PARITY)
        for kv in a.tune:
            ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
        gen = torch.Generator(device="cuda").manual_seed(0xD2C)
        raw = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device="cuda", generator=gen)
        if holes > 0 and blocky:
            m = torch.rand((F, (H + 63) // 64, (W + 63) // 64), device="cuda", generator=gen) < holes
            raw[m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H, :W]] = 0
        elif holes > 0:
            raw[torch.rand(raw.shape, device="cuda", generator=gen) < holes] = 0
        dst = torch.empty_like(raw)
        s = torch.cuda.current_stream().cuda_stream
        if WORKLOADS[a.workload][0] == "callback":   # d2pc_process_mono_device: k_callback_bs<11> at this size
            from disparity_to_point_cloud_amd.torch_api import DeviceBatch as _DB
            bb = _DB(ctx, F, H, W, dtype=torch.uint8, want_index=want_idx)
            run = lambda: ctx.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, W, H, W, W * H, F, 11, 0.125,
                                                  bb.points.data_ptr(), bb.index.data_ptr() if want_idx else None, bb.stride,
                                                  bb.counts.data_ptr(), s)
        else:
            run = lambda: ctx.median_roi_device(raw.data_ptr(), W, H, W, W * H, F, dst.data_ptr(), W, W * H, 11, s)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.launches
        px = F * (W - 2 * border) * (H - 2 * border)
        print(json.dumps({"workload": a.workload, "launches": a.launches, "ms_per_launch": round(ms, 4), "points": px,
                          "roi_pixels": px, "algorithmic_bytes": (17 if WORKLOADS[a.workload][0] == "callback" else 2) * px, "algorithmic_GBs": round((17 if WORKLOADS[a.workload][0] == "callback" else 2) * px / ms / 1e6, 1)}))
        return
    ctx, b = make_batch(a.workload, a.algo, tuning=[(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.tune])
    scale = a.scale if a.scale is not None else (0.125 if b.disp.dtype == torch.uint8 else 1.0)
    b.launch(scale=scale)
    torch.cuda.synchronize()
    npts = int(b.counts.sum().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.launches):
        b.launch(scale=scale)
    e1.record()
    torch.cuda.synchronize()
    ctx.check_async_error()
    ms = e0.elapsed_time(e1) / a.launches
    es = b.disp.element_size()
    alg = es * b.n_frames * b.roi_n + (20 if b.index is not None else 16) * npts
    print(json.dumps({"workload": a.workload, "launches": a.launches, "ms_per_launch": round(ms, 4), "points": npts,
                      "roi_pixels": b.n_frames * b.roi_n, "algorithmic_bytes": alg,
                      "algorithmic_GBs": round(alg / ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
