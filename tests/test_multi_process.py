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
    # _free_port() closes the socket before torchrun binds the port again: another process can take it in between (seen once in
    # six rounds).  A failed RENDEZVOUS -- and only that -- is tried once more on a new port.
    rendezvous = ("Address already in use", "EADDRINUSE", "errno: 98", "failed to listen", "RendezvousConnectionError",
                  "DistNetworkError")
    if p.returncode != 0 and any(k in p.stderr or k in p.stdout for k in rendezvous):
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
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


# ---- bench.py's own launcher (python bench.py --gpus N starts the N ranks) ----
def _bench_module():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_launcher_spawns_ranks_and_relays_exactly_one_line():
    import io

    bench = _bench_module()
    rc, text = bench.spawn_ranks(3, os.path.join(ROOT, "tests", "launch_stub_worker.py"), ["--gpus", "3"], timeout=300)
    assert rc == 0, text
    out, err = io.StringIO(), io.StringIO()
    rec = bench.relay_json_line(text, out=out, err=err)
    assert rec == {"metric": "stub", "n_gpus": 3, "slowest": 3.0, "per_rank": [1.0, 2.0, 3.0]}
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")          # the contract: ONE JSON line on stdout
    assert "banner" in err.getvalue()                             # everything else goes to stderr


def test_launcher_returns_the_ranks_failure():
    bench = _bench_module()
    rc, text = bench.spawn_ranks(2, os.path.join(ROOT, "tests", "launch_stub_worker.py"),
                                 ["--gpus", "2", "--fail-rank", "1"], timeout=300)
    assert rc != 0
    assert bench.relay_json_line(text, out=open(os.devnull, "w"), err=open(os.devnull, "w")) is None


def _bench(args, env=None, timeout=600):
    e = dict(os.environ if env is None else env, OMP_NUM_THREADS="1")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus2_starts_two_ranks_and_fails_loudly_without_a_gpu():
    """No GPU here: the parent must have started 2 ranks (each says so) and pass their failure on --
    never a silent n_gpus=1 line."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu-marked rehearsal test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env=env)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs a GPU") == 2, p.stderr[-2000:]


def test_bench_rejects_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env=env)
    assert p.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in p.stderr and p.stdout.strip() == ""


@pytest.mark.gpu
def test_rccl_carries_the_calibration_blob_bitwise():
    """RCCL in the suite: the calibration broadcast through backend "nccl" (one rank: a 1-GPU box has no
    second device for a second RCCL rank), the blob checked bit for bit and used for a frame."""
    env = dict(os.environ, D2PC_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "rccl1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "rank 0/1 ok backend=nccl" in p.stdout


@pytest.mark.gpu
def test_bench_gpus2_launches_two_ranks_on_one_gpu_rehearsal():
    """`python bench.py --gpus 2` as the driver types it (plus the rehearsal switches a 1-GPU box needs):
    two ranks come up, barrier, time, reduce, and ONE line with n_gpus=2 comes back through the parent."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "3", "--warmup", "1", "--frames", "4",
                "--no-variants", "--no-cpu"], env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["rehearsal_shared_gpu"] is True
    assert len(rec["roofline"]["kernel_ms_avg_per_rank"]) == 2
    assert rec["config"]["collective_backend"] == "gloo"
