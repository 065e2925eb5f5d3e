"""One-process-per-GPU sharding of the path.

Frames are independent units: every rank owns one camera stream / frame queue
and one GPU; nothing crosses GPUs on the per-frame path.  The ONLY exchange is
the calibration: rank 0 broadcasts the 136-byte blob (16 x f64 Q + border +
mode) once at start-up / on recalibration -- RCCL over xGMI when the backend
is "nccl", gloo in the CPU tests.  Reporting uses one all-reduce of counters.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None, share_gpu=False):
    """Initialise torch.distributed from the torchrun environment (no-op for a
    single process).  Returns (rank, local_rank, world_size).  The backend is
    taken from the argument, else D2PC_DIST_BACKEND, else "nccl" (= RCCL) when
    the node exposes GPUs -- decided by COUNTING devices, which does not
    initialise the GPU.  share_gpu: every rank uses device 0 (gloo rehearsal)."""
    rank, local_rank, world = env_world()
    if share_gpu:
        local_rank = 0
    # D2PC_FORCE_DIST=1 initialises the group even for one rank (exercises RCCL on a 1-GPU box)
    if (world > 1 or os.environ.get("D2PC_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("D2PC_DIST_BACKEND") or ("nccl" if torch.cuda.device_count() > 0 else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_blob(blob, src=0) -> bytes:
    """Broadcast the calibration blob from `src`; every rank returns the bytes."""
    n = capi.CALIB_BLOB_BYTES
    if not (dist.is_available() and dist.is_initialized()):
        assert blob is not None and len(blob) == n
        return bytes(blob)
    dev = _comm_device()
    if dist.get_rank() == src:
        assert blob is not None and len(blob) == n
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        t = torch.zeros(n, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().tobytes()


def broadcast_calibration(ctx, src=0):
    """Rank `src` exports its context's calibration; every other rank imports
    it, so all GPUs reproject with bit-identical Q / border / mode."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    blob = ctx.export_calibration() if rank == src else None
    blob = broadcast_blob(blob, src)
    if rank != src:
        ctx.import_calibration(blob)
    return blob


def shard_frames(n_frames_total: int, rank: int, world: int):
    """Frame-level data parallelism for ONE shared stream: frame i goes to
    rank i % world (round-robin keeps per-rank order == arrival order)."""
    return list(range(rank, n_frames_total, world))


def allreduce_max(x: float) -> float:
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_floats(x: float):
    """Every rank's value of x, in rank order."""
    if not dist.is_initialized():
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=_comm_device())
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def backend_name():
    return dist.get_backend() if dist.is_initialized() else None


def allreduce_sum_counters(counters) -> np.ndarray:
    """Sum of per-rank {frames, pixels, points, ns} style counters."""
    t = torch.tensor(list(counters), dtype=torch.int64, device=_comm_device())
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def barrier():
    if dist.is_initialized():
        dist.barrier()
