#!/usr/bin/env python3
"""Interleaved A/B timing of library builds and launch shapes in ONE process
on ONE device (devices and runs differ by ~10 %, so only this kind of
comparison ranks variants).  GPU box only.

  python tools/ab.py --libs base,ntload,ntstore --modes parity --borders 40,0
"""
import argparse, ctypes, itertools, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disparity_to_point_cloud_amd import capi

def load_variant(name):
    capi._lib = None
    capi._LIB_NAME = "libd2pc.so" if name == "base" else f"libd2pc_{name}.so"
    return capi.load_library()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="base")
    ap.add_argument("--modes", default="parity")
    ap.add_argument("--borders", default="40")
    ap.add_argument("--pxts", default="16")
    ap.add_argument("--bpcs", default="128", help="tuning blocks_per_cu (library default 128: two-pass and tile-walking grids = min(T, max(CUs x this, T/4)))")
    ap.add_argument("--novecs", default="0")
    ap.add_argument("--algos", default="1")
    ap.add_argument("--idx", type=int, default=0)
    ap.add_argument("--holes", type=float, default=0.0)
    ap.add_argument("--blocky", type=int, default=0, help="holes come in 64x64 blocks instead of iid pixels")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--opbpc", type=int, default=4, help="single-pass kernel: persistent blocks per CU")
    ap.add_argument("--oaligns", default="16", help="output frame stride rounded up to this many points (several: A/B)")
    ap.add_argument("--ooffs", default="0", help="output base offset in points (several: A/B)")
    ap.add_argument("--ioffs", default="0", help="index base offset in 4-byte elements (several: A/B) -- does the relation of the index stream's addresses to the point stream's matter?")
    ap.add_argument("--small", type=int, default=None, help="tuning parity_small (1: one-shot blocks also for pxt 4)")
    ap.add_argument("--forms", default="0", help="tuning reproject_form (0 per Q kind, 24 / 4: one OpenCV generation bit for bit)")
    ap.add_argument("--tunes", default="", help="alternatives separated by ';', each a comma-separated list of d2pc_set_tuning key=value (e.g. 'chunk_mb=96;chunk_mb=48,chunk_first_frames=1')")
    ap.add_argument("--dtype", default="f32", choices=["f32", "u8", "u16"], help="input sample type (u8 / u16: the fused cpp:61 decode, scale 1/8 and 1/64)")
    ap.add_argument("--oextras", default="0", help="points ADDED to the output frame stride (several: A/B); 256 points = one 4-KiB page: "
                                                   "do write fronts an exact multiple of 8 KiB apart collide in the memory system?")
    ap.add_argument("--splits", default="1", help="sub-launches per step (several: A/B): the frames in `split` groups launched back to back, so "
                                                  "only frames/split write fronts are live at a time")
    ap.add_argument("--ring", type=int, default=1, help="camera-shaped launches: RING x frames distinct input frames (and output slots), every launch "
                                                        "takes the next `frames` of them, so no launch finds its input in the Infinity Cache")
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--h", type=int, default=2160)
    a = ap.parse_args()
    from disparity_to_point_cloud_amd.torch_api import DeviceBatch
    import disparity_to_point_cloud_amd as d2pc
    g = torch.Generator(device="cuda").manual_seed(1)
    NF = a.frames * a.ring
    disp = torch.rand((NF, a.h, a.w), generator=g, device="cuda") * 127.5 + 0.5
    dcode, dscale, esize = {"f32": (0, 1.0, 4), "u8": (1, 0.125, 1), "u16": (2, 1.0 / 64, 2)}[a.dtype]
    if a.holes > 0 and a.blocky:
        m = (torch.rand((NF, (a.h + 63) // 64, (a.w + 63) // 64), generator=g, device="cuda") >= a.holes).float()
        m = m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :a.h, :a.w]
        disp.mul_(m)
    elif a.holes > 0:
        disp.mul_((torch.rand(disp.shape, generator=g, device="cuda") >= a.holes).float())
    if a.dtype == "u8":
        disp = (disp * 2).to(torch.uint8)          # 0 stays 0 (a hole), 1..255
    elif a.dtype == "u16":
        disp = (disp * 64).to(torch.int32).to(torch.uint16)
    cands = []
    # ONE set of buffers for every candidate: kernel time depends on which
    # physical pages a buffer got (+-6 % between allocations of one process)
    W, H, F = a.w, a.h, a.frames
    oaligns = [int(x) for x in a.oaligns.split(",")]
    ooffs = [int(x) for x in a.ooffs.split(",")]
    oextras = [int(x) for x in a.oextras.split(",")]
    splits = [int(x) for x in a.splits.split(",")]
    max_stride = max((W * H + al - 1) // al * al for al in oaligns) + max(oextras)
    pool = torch.empty((NF * max_stride + max(ooffs) + 16, 4), dtype=torch.float32, device="cuda")
    ioffs = [int(x) for x in a.ioffs.split(",")]
    index = torch.empty((NF * max_stride + max(ioffs) + 64,), dtype=torch.int32, device="cuda") if a.idx else None
    counts = torch.zeros((NF,), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    class Cand:
        def __init__(self, ctx, oalign, ooff, ioff=0, oextra=0, split=1):
            self.ctx = ctx
            self.split = split
            self.pos = 0
            self.stride = (W * H + oalign - 1) // oalign * oalign + oextra
            self.out_ptr = pool.data_ptr() + 16 * ooff
            self.idx_ptr = index.data_ptr() + 4 * ioff if index is not None else None
            ctx.reserve(W, H, F)
        def launch(self):
            n = F // self.split
            base = self.pos * F
            self.pos = (self.pos + 1) % a.ring
            for f0 in range(base, base + F, n):
                self.ctx.process_device(disp.data_ptr() + f0 * W * H * esize, dcode, dscale, W, H, W * esize, W * H * esize, n,
                                        self.out_ptr + 16 * f0 * self.stride,
                                        self.idx_ptr + 4 * f0 * self.stride if self.idx_ptr is not None else None, self.stride,
                                        counts.data_ptr() + 4 * f0, stream)

    for lib in a.libs.split(","):
        L = load_variant(lib)
        for mode, border, pxt, bpc, nv, algo, oal, oof, form, tune, iof, oex, spl in itertools.product(a.modes.split(","), a.borders.split(","), a.pxts.split(","), a.bpcs.split(","), a.novecs.split(","), a.algos.split(","), oaligns, ooffs, a.forms.split(","), a.tunes.split(";"), ioffs, oextras, splits):
            m = d2pc.MODE_PARITY if mode == "parity" else d2pc.MODE_COMPACT
            ctx = capi.Context(q=capi.make_q(), border=int(border), mode=m, compact_algo=int(algo))
            if mode == "compact":
                ctx.set_tuning("pxt_compact", int(pxt))
            else:
                ctx.set_tuning("pxt_parity", int(pxt))   # (1, 2: one-shot blocks; 4 / 8 / 16: the tile-walking kernel, --libs exp)
            if a.small is not None:
                ctx.set_tuning("parity_small", a.small)
            if int(form):
                ctx.set_reproject_form(int(form))
            ctx.set_tuning("blocks_per_cu", int(bpc)); ctx.set_tuning("no_vec_rows", int(nv))
            ctx.set_tuning("onepass_blocks_per_cu", a.opbpc)
            for kv in filter(None, tune.split(",")):
                k, v = kv.split("=")
                ctx.set_tuning(k, int(v))
            b = Cand(ctx, oal, oof, iof, oex, spl)
            b.launch(); torch.cuda.synchronize()
            for _ in range(a.ring - 1):
                b.launch()
            torch.cuda.synchronize()
            npts = int(counts.sum().item()) // a.ring
            roi_n = capi.roi_points(W, H, int(border))
            alg = esize * F * roi_n + (20 if a.idx else 16) * npts
            cands.append((f"{lib:8s} {mode:7s} b={border:>2s} pxt={pxt:>2s} bpc={bpc:>2s} novec={nv} algo={algo} oalign={oal} ooff={oof} form={form} {tune}" + (f" ioff={iof}" if len(ioffs) > 1 else "") + (f" oextra={oex}" if len(oextras) > 1 else "") + (f" split={spl}" if len(splits) > 1 else ""), b, alg, []))
    for r in range(a.rounds):
        for label, b, alg, ts in cands:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                b.launch()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / a.iters)
    for label, b, alg, ts in cands:
        ts = np.array(ts[1:]) * 1e3
        print(f"{label}: med {np.median(ts):7.1f} us  min {ts.min():7.1f}  max {ts.max():7.1f}   {alg/np.median(ts)/1e3:7.1f} GB/s", flush=True)

if __name__ == "__main__":
    main()
