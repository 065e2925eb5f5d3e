"""Seeded synthetic disparity frames for BASELINE.json's configs (host-side
data generation for tests and bench.py; numpy only, no product logic)."""
import numpy as np


def frame_seed(config_id: int, frame_id: int) -> int:
    """seed = 0xD2C00000 + config_id*1000 + frame_id.

    The SEED is the survey's; the GENERATOR behind it is numpy's default (PCG64), because numpy has no 64-bit Mersenne
    twister (`numpy.random.MT19937` is the 32-bit one).  The C++ harness (host/multi_gpu.hpp) draws its frames from
    `std::mt19937_64` as the survey prescribes.  So Python-side and C++-side frames of one seed DIFFER: the two
    harnesses never exchange frames by seed -- tests hand frames to the C++ harness as files (tests/test_host_cpp.py)."""
    return 0xD2C00000 + config_id * 1000 + frame_id


def synth_disparity(config_id: int, frame_id: int, width: int, height: int, kind: str) -> np.ndarray:
    """
      'k8'      d = k/8, k in U{1..255}   (C2: all valid, reference quantisation)
      'uniform' d ~ U(0.5,128)            (C4: all valid)
      'holes'   'uniform' with iid 30 % zeros         (C3)
      'blocky'  'uniform' with 64x64-block holes ~30 % (C3 variant)
      'mono16'  uint16 k*257, k in U{0..255}          (C1)
    """
    rng = np.random.default_rng(frame_seed(config_id, frame_id))
    if kind == "k8":
        return rng.integers(1, 256, size=(height, width)).astype(np.float32) * np.float32(0.125)
    if kind == "mono16":
        return (rng.integers(0, 256, size=(height, width)) * 257).astype(np.uint16)
    d = rng.uniform(0.5, 128.0, size=(height, width)).astype(np.float32)
    if kind == "uniform":
        return d
    if kind == "holes":
        d[rng.random(size=(height, width)) < 0.3] = 0.0
        return d
    if kind == "blocky":
        by, bx = (height + 63) // 64, (width + 63) // 64
        m = rng.random(size=(by, bx)) < 0.3
        m = np.repeat(np.repeat(m, 64, axis=0), 64, axis=1)[:height, :width]
        d[m] = 0.0
        return d
    raise ValueError(kind)
