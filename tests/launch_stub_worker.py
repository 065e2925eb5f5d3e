"""Stub rank for the bench.py launcher test: joins the group over gloo, does the
barrier + max-over-ranks reduction bench.py does, and rank 0 prints a library-style
banner followed by ONE result line.  argv: --gpus N [--fail-rank R]"""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from disparity_to_point_cloud_amd import multi_gpu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, required=True)
ap.add_argument("--fail-rank", type=int, default=-1)
a = ap.parse_args()
rank, local_rank, world = multi_gpu.init_distributed(backend="gloo")
assert world == a.gpus == int(os.environ["WORLD_SIZE"])
if rank == a.fail_rank:
    raise SystemExit(f"stub rank {rank}: told to fail")
multi_gpu.barrier()
slowest = multi_gpu.allreduce_max(1.0 + rank)
per_rank = multi_gpu.allgather_floats(1.0 + rank)
if rank == 0:
    print("NCCL version 0.0.0 (banner that is not the result line)")
    print(json.dumps({"metric": "stub", "n_gpus": world, "slowest": slowest, "per_rank": per_rank}), flush=True)
multi_gpu.barrier()
dist.destroy_process_group()
