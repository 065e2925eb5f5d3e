"""Worker for the multi-process tests (launched by torch.distributed.run).
Exits non-zero on any failed check.  argv[1] = "cpu" | "gpu"."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import disparity_to_point_cloud_amd as d2pc  # noqa: E402
from disparity_to_point_cloud_amd import multi_gpu  # noqa: E402


def rccl_one_rank():
    """backend "nccl" (= RCCL) with one rank: the calibration blob goes through an RCCL broadcast on the
    device and must come back bit for bit; the reporting collectives run on RCCL too."""
    import oracle
    from helpers import assert_points_close, synth_disparity

    rank, local_rank, world = multi_gpu.init_distributed()  # default decision: GPUs present => nccl
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    q0 = d2pc.make_q(fx=700.0, fy=690.5, cx=333.25, cy=250.0, baseline=0.043, nx=640, ny=480)
    src = d2pc.Context(device_id=local_rank, q=q0, border=24, mode=d2pc.MODE_COMPACT)
    blob0 = src.export_calibration()
    blob = multi_gpu.broadcast_blob(blob0, src=0)           # host -> device -> ncclBroadcast -> host
    assert blob == blob0 and len(blob) == d2pc.CALIB_BLOB_BYTES
    q, border, mode = d2pc.calib_unpack(blob)
    assert q.tobytes() == q0.tobytes() and np.signbit(q[15]) and (border, mode) == (24, d2pc.MODE_COMPACT)
    assert multi_gpu.broadcast_calibration(src, src=0) == blob0
    assert multi_gpu.allreduce_max(2.5) == 2.5 and multi_gpu.allgather_floats(1.25) == [1.25]
    assert list(multi_gpu.allreduce_sum_counters([3, 4])) == [3, 4]
    dst = d2pc.Context(device_id=local_rank)
    dst.import_calibration(blob)
    disp = synth_disparity(5, 0, 640, 480, "holes")
    gp, gi = dst.process(disp, want_index=True)
    wp, wi = oracle.reproject_compact(disp, q0, border=24)
    assert np.array_equal(gi, wi)
    assert_points_close(gp, wp, max_ulp=1, rel=1e-5, what="rccl-broadcast calibration")
    src.close()
    dst.close()
    multi_gpu.barrier()
    dist.destroy_process_group()
    print("rank 0/1 ok backend=nccl")


def main():
    where = sys.argv[1]
    if where == "rccl1":
        return rccl_one_rank()
    rank, local_rank, world = multi_gpu.init_distributed(backend="gloo")
    assert dist.is_initialized() and world == int(os.environ["WORLD_SIZE"]) and world >= 2

    # C1: rank 0 owns the calibration (a non-default rig, so a rank that kept
    # its own defaults would be caught), every rank must end up bit-identical
    q0 = d2pc.make_q(fx=700.0, fy=690.5, cx=333.25, cy=250.0, baseline=0.043, nx=640, ny=480)
    blob = d2pc.calib_pack(q0, border=24, mode=d2pc.MODE_COMPACT) if rank == 0 else None
    blob = multi_gpu.broadcast_blob(blob, src=0)
    q, border, mode = d2pc.calib_unpack(blob)
    assert q.tobytes() == q0.tobytes(), "Q differs after broadcast"
    assert np.signbit(q[15]) and (border, mode) == (24, d2pc.MODE_COMPACT)
    # every rank holds the same bytes
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).clone()
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    assert all(torch.equal(g, gathered[0]) for g in gathered)

    # frame-level sharding of one stream: disjoint, complete, order-preserving
    mine = multi_gpu.shard_frames(37, rank, world)
    allf = [None] * world
    dist.all_gather_object(allf, mine)
    flat = sorted(sum(allf, []))
    assert flat == list(range(37)) and mine == sorted(mine) and all(f % world == rank for f in mine)

    # reporting collectives
    assert multi_gpu.allreduce_max(float(rank + 1)) == float(world)
    tot = multi_gpu.allreduce_sum_counters([1, 10 * (rank + 1), rank])
    assert list(tot) == [world, 10 * world * (world + 1) // 2, world * (world - 1) // 2]

    if where == "gpu":
        # each rank runs its own frame queue on the GPU with the broadcast
        # calibration; results must equal the single-process oracle
        import oracle
        from helpers import assert_points_close, synth_disparity

        ctx = d2pc.Context(device_id=0)
        ctx.import_calibration(blob)
        for f in mine[:3]:
            disp = synth_disparity(5, f, 640, 480, "holes")
            gp, gi = ctx.process(disp, want_index=True)
            wp, wi = oracle.reproject_compact(disp, q0, border=24)
            assert np.array_equal(gi, wi)
            assert_points_close(gp, wp, max_ulp=1, rel=1e-5, what=f"rank {rank} frame {f}")
        ctx.close()

        # one shared stream of 11 frames sharded over the ranks' frame queues
        # (pipelined host path, several frames in flight per rank)
        from disparity_to_point_cloud_amd.stream import run_sharded

        shared = [synth_disparity(6, i, 320 + 16 * (i % 3), 240, "holes") for i in range(11)]
        ctx2 = d2pc.Context(device_id=0)
        ctx2.import_calibration(blob)
        seen = {}

        def on_cloud(tag, pts, idx):
            seen[tag] = (pts.copy(), idx.copy())

        tot = run_sharded(ctx2, shared, on_cloud, want_index=True)
        ctx2.close()
        assert sorted(seen) == multi_gpu.shard_frames(11, rank, world)
        npts = 0
        for i, (pts, idx) in seen.items():
            wp, wi = oracle.reproject_compact(shared[i], q0, border=24)
            assert np.array_equal(idx, wi)
            assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"sharded frame {i}")
        want_total = sum(len(oracle.reproject_compact(fr, q0, border=24)[0]) for fr in shared)
        assert tot["frames"] == 11 and tot["points"] == want_total, tot
        assert tot["pixels"] == sum(fr.size for fr in shared)
    multi_gpu.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
