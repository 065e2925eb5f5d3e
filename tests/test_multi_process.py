"""N>1 path: one process per rank over torch.distributed (gloo on CPU here;
RCCL on the GPU node via bench.py).  Covers the calibration broadcast (the
path's only exchange), frame sharding and the reporting collectives."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, where):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), where]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert f"rank {r}/{world} ok" in p.stdout


@pytest.mark.parametrize("world", [2, 3, 8])  # 8 = the node the scaling bench runs on
def test_calibration_broadcast_and_sharding_gloo(world):
    _run(world, "cpu")


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_with_broadcast_calibration():
    _run(2, "gpu")
