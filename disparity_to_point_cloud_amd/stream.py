"""Per-rank frame queue: one camera stream / frame
queue per GPU, frames pushed through the pipelined host path
(d2pc_pipeline_*) with several frames in flight.  Nothing crosses GPUs per
frame; ranks only meet for the calibration broadcast and for counters.
Plumbing only: every point is produced by libd2pc.so."""
import time
from collections import deque

import numpy as np

from . import capi, multi_gpu


class RankStream:
    """Feeds host frames of one stream through a context's pipeline, keeping
    `depth` frames in flight, and hands every finished cloud to `on_cloud`
    in submission order."""

    def __init__(self, ctx: capi.Context, depth=3, direct_host_write=True):
        self.ctx, self.depth = ctx, depth
        ctx.pipeline_configure(depth=depth, direct_host_write=direct_host_write)
        self.frames = self.pixels = self.points = 0
        self.busy_ns = 0

    def run(self, frames, on_cloud, scale=1.0, median_ksize=0, want_index=False):
        """frames: iterable of (tag, 2-D numpy image).  on_cloud(tag, points, index)
        receives VIEWS of pinned memory, valid only during the call."""
        t0 = time.perf_counter_ns()
        inflight = deque()
        for tag, img in frames:
            if len(inflight) == self.depth:
                self._collect(on_cloud)
                inflight.popleft()
            self.ctx.pipeline_submit(img, scale=scale, median_ksize=median_ksize, want_index=want_index, tag=tag)
            inflight.append(tag)
            self.frames += 1
            self.pixels += img.shape[0] * img.shape[1]
        while inflight:
            self._collect(on_cloud)
            inflight.popleft()
        self.busy_ns += time.perf_counter_ns() - t0

    def _collect(self, on_cloud):
        pts, idx, tag, slot = self.ctx.pipeline_collect(copy=False)
        self.points += len(pts)
        try:
            on_cloud(tag, pts, idx)
        finally:
            self.ctx.pipeline_release(slot)

    def counters(self):
        return [self.frames, self.pixels, self.points, self.busy_ns]


def run_sharded(ctx: capi.Context, all_frames, on_cloud, **kw):
    """Frame-level data parallelism for ONE shared stream: this rank takes the
    frames multi_gpu.shard_frames assigns to it; returns the job-wide
    {frames, pixels, points, max busy seconds} after a counter all-reduce."""
    rank, _, world = multi_gpu.env_world()
    mine = multi_gpu.shard_frames(len(all_frames), rank, world)
    rs = RankStream(ctx)
    rs.run(((i, all_frames[i]) for i in mine), on_cloud, **kw)
    c = rs.counters()
    tot = multi_gpu.allreduce_sum_counters(c[:3])
    busy = multi_gpu.allreduce_max(c[3] * 1e-9)
    return dict(frames=int(tot[0]), pixels=int(tot[1]), points=int(tot[2]), busy_s=busy)
