"""torch plumbing around d2pc_process_device: device memory and streams come
from torch, the work is done by libd2pc.so's HIP kernels.  No torch op ever
computes a point here."""
import numpy as np
import torch

from . import capi

_T2DT = {torch.float32: capi.DTYPE_F32, torch.uint8: capi.DTYPE_U8, torch.uint16: capi.DTYPE_U16}


class DeviceBatch:
    """Pre-allocated device buffers for a batch of equally sized frames."""

    def __init__(self, ctx: capi.Context, n_frames: int, height: int, width: int, dtype=torch.float32,
                 want_index=False, device="cuda:0", reserve=True):
        cfg = ctx.config()
        self.ctx, self.n_frames, self.height, self.width = ctx, n_frames, height, width
        self.roi_n = capi.roi_points(width, height, cfg.border)
        # frame outputs start on 256-byte boundaries (16 points)
        self.stride = max((self.roi_n + 15) // 16 * 16, 16)
        self.device = torch.device(device)
        self.disp = torch.empty((n_frames, height, width), dtype=dtype, device=self.device)
        self.points = torch.empty((n_frames, self.stride, 4), dtype=torch.float32, device=self.device)
        self.index = (torch.empty((n_frames, self.stride), dtype=torch.int32, device=self.device)
                      if want_index else None)
        self.counts = torch.zeros((n_frames,), dtype=torch.int32, device=self.device)
        if reserve:   # (so that a capture finds its compaction state; False: the first eager launch allocates it)
            ctx.reserve(width, height, n_frames)

    def launch(self, scale=1.0, stream=None):
        """Enqueue one pass over the whole batch on `stream` (default: torch's
        current stream).  Asynchronous."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        d = self.disp
        self.ctx.process_device(
            d.data_ptr(), _T2DT[d.dtype], scale, self.width, self.height, d.stride(1) * d.element_size(),
            d.stride(0) * d.element_size(), self.n_frames, self.points.data_ptr(),
            self.index.data_ptr() if self.index is not None else None, self.stride, self.counts.data_ptr(),
            s.cuda_stream)

    def results(self):
        """Synchronise and copy back: list of (points[, index]) per frame."""
        torch.cuda.synchronize(self.device)
        counts = self.counts.cpu().numpy().astype(np.int64)
        pts = self.points.cpu().numpy()
        idx = self.index.cpu().numpy().view(np.uint32) if self.index is not None else None
        out = []
        for f in range(self.n_frames):
            n = int(counts[f])
            out.append((pts[f, :n].copy(), idx[f, :n].copy() if idx is not None else None))
        return out


def fuse_planes(ctx: capi.Context, planes, rule=capi.FUSE_GRAD_FILTER, crop=(0, 40, 30, 10), want_combined=True,
                stream=None):
    """d2pc_fuse_device on torch uint8 CUDA tensors.

    planes = (depth1, depth2, score1, score2, grad1, grad2), each (H, W) or (F, H, W) with unit
    column stride (row/frame strides are free, so cropped views work).  Returns (fused, combined or
    None) as new tensors; asynchronous on `stream` (default: torch's current stream)."""
    assert len(planes) == 6
    ref = planes[0]
    batched = ref.dim() == 3
    shape = tuple(ref.shape)
    f, h, w = (shape if batched else (1,) + shape)
    l, r, t, b = crop
    desc = capi.fuse_desc_init()
    desc.rule, desc.width, desc.height, desc.n_frames = rule, w, h, f
    desc.crop_left, desc.crop_right, desc.crop_top, desc.crop_bottom = l, r, t, b
    for i, p in enumerate(planes):
        if p is None:
            assert i >= 4 and not want_combined
            continue
        assert p.is_cuda and p.dtype == torch.uint8 and tuple(p.shape) == shape and p.stride(-1) == 1
        desc.planes[i] = p.data_ptr()
        desc.pitch[i] = p.stride(-2)
        desc.frame_stride[i] = p.stride(0) if batched else 0
    oh, ow = max(h - t - b, 0), max(w - l - r, 0)
    fused = torch.empty((f, oh, ow), dtype=torch.uint8, device=ref.device)
    combined = torch.empty((f, h, w), dtype=torch.uint8, device=ref.device) if want_combined else None
    desc.fused = fused.data_ptr() if fused.numel() else ref.new_empty(1).data_ptr()
    desc.fused_pitch, desc.fused_frame_stride = max(ow, 1), max(ow, 1) * oh
    if want_combined:
        desc.combined, desc.combined_pitch, desc.combined_frame_stride = combined.data_ptr(), w, w * h
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    ctx.fuse_device(desc, s.cuda_stream)
    if not batched:
        fused = fused[0]
        combined = combined[0] if want_combined else None
    return fused, combined
