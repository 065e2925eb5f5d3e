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
                 want_index=False, device="cuda:0"):
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
