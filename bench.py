#!/usr/bin/env python3
"""bench.py -- headline benchmark of the disparity -> point-cloud hot path.

Metric (BASELINE.json): Mpixels/s reprojected, device-resident, on the
3840x2160 fp32 disparity stream (configs[3]; configs[4] = one such stream per
GPU).  A "step" is ONE pass of the hot path (one d2pc_process_device call)
over a ring of 16 distinct synthetic 4K frames already resident in HBM
(531 MB in, 2.0 GB out: larger than the 256 MiB Infinity Cache).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N
ranks itself: the parent -- which imports neither torch nor the HIP library, so
it never touches a GPU -- runs `python -m torch.distributed.run` as a CHILD
process (reference src/disparity_to_point_cloud_node.cpp:46-52: one process per
node, here one per GPU), relays rank 0's JSON line and exits with the child's
code.  A rank whose WORLD_SIZE differs from --gpus fails.

Prints ONE JSON line (rank 0) with `roofline` (algorithmic bytes / measured
kernel time vs 8 TB/s HBM peak) and `cpu_baseline` (the oracle's single-thread
restatement of the reference loop timed on this host, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# --------------------------------------------------------------------------
# launcher (stdlib only: runs in a parent that must never initialise the GPU)
# --------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n_ranks, script, script_args, timeout=None, env=None):
    """Start `script` as n_ranks torch.distributed ranks on this node (a child
    process, never an exec) and return (returncode, stdout, stderr-relayed).
    The child's stderr goes straight to ours; its stdout is captured so that
    only the JSON line reaches our stdout."""
    e = dict(os.environ if env is None else env)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + list(script_args)
    p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, text=True, timeout=timeout)
    return p.returncode, p.stdout


def relay_json_line(stdout_text, out=sys.stdout, err=sys.stderr):
    """Print the ranks' ONE JSON result line to `out`, everything else they wrote
    to stdout (library banners) to `err`.  Returns the parsed line or None."""
    found = None
    for line in stdout_text.splitlines():
        s = line.strip()
        rec = None
        if s.startswith("{") and s.endswith("}"):
            try:
                rec = json.loads(s)
            except ValueError:
                rec = None
        if isinstance(rec, dict) and "metric" in rec and found is None:
            found = rec
            print(s, file=out, flush=True)
        elif s:
            print(line, file=err, flush=True)
    return found


def launch_if_needed(argv):
    """--gpus N > 1 outside a torchrun environment: become the launcher."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args(argv)
    if known.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    rc, text = spawn_ranks(known.gpus, os.path.abspath(__file__), argv)
    rec = relay_json_line(text)
    if rc == 0 and (rec is None or rec.get("n_gpus") != known.gpus):
        print(f"bench.py: the {known.gpus} ranks exited 0 without a result line for n_gpus={known.gpus}", file=sys.stderr)
        rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    launch_if_needed(sys.argv[1:])  # before torch / libd2pc.so are imported
    if "WORLD_SIZE" not in os.environ:
        # the all-cores CPU column pins its OpenMP team (round 4's verdict: one unpinned 4-s sample ranged 824-1958 Mpixel/s
        # on 128 cores); the variable must be in place before the first OpenMP runtime of the process initialises
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import disparity_to_point_cloud_amd as d2pc  # noqa: E402
from disparity_to_point_cloud_amd import multi_gpu, telemetry  # noqa: E402
from disparity_to_point_cloud_amd.synth import synth_disparity  # noqa: E402
from disparity_to_point_cloud_amd.torch_api import DeviceBatch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
W4K, H4K = 3840, 2160


def build_id():
    """sha256/12 of the kernel + ABI sources the loaded libd2pc.so was built from (tracked profiles carry the
    same id, so a stale profile is visible in the bench line)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "disparity_to_point_cloud_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:12]


def fill_batch(batch, rank, kind):
    """Ring of distinct seeded frames (config 4), different per rank."""
    for f in range(batch.n_frames):
        fr = synth_disparity(4, rank * batch.n_frames + f, batch.width, batch.height, kind)
        batch.disp[f].copy_(torch.from_numpy(fr))
    torch.cuda.synchronize()


class RingLaunch:
    """Camera-shaped launches: each launch() processes the next `n` frames of `batch`'s ring (inputs, outputs, index and count
    slots all move on), so a launch never finds its frame in the Infinity Cache; fixed=True stays on the first position."""

    def __init__(self, ctx, batch, n, fixed=False):
        assert batch.n_frames % n == 0
        self.ctx, self.b, self.n, self.fixed, self.pos = ctx, batch, n, fixed, 0
        self.positions = batch.n_frames // n
        self.stream = torch.cuda.current_stream(batch.device).cuda_stream

    def launch(self):
        b, f = self.b, self.pos * self.n
        d = b.disp
        fb = d.stride(0) * d.element_size()
        self.ctx.process_device(d.data_ptr() + f * fb, d2pc.DTYPE_F32, 1.0, b.width, b.height, d.stride(1) * d.element_size(), fb,
                                self.n, b.points.data_ptr() + f * b.stride * 16,
                                (b.index.data_ptr() + f * b.stride * 4) if b.index is not None else None, b.stride,
                                b.counts.data_ptr() + f * 4, self.stream)
        if not self.fixed:
            self.pos = (self.pos + 1) % self.positions


def algorithmic_bytes(batch, n_points, with_index):
    """bytes = 4*R_in + 16*P (+4*P with indices), per launch."""
    r_in = batch.n_frames * batch.roi_n
    return 4 * r_in + (20 if with_index else 16) * n_points


def timed_steps(batch, steps, warmup, dist_sync, heat_ms=60.0, first_n=20):
    """The contract's measurement: W untimed warm-up steps, then EXACTLY K timed steps between barrier +
    synchronize pairs.  `batch` only needs a launch() method (one pass of the hot path, asynchronous).

    The warm-up has a TIME floor as well as a count: after the W steps the same launch is repeated (untimed) until
    >= heat_ms of device time have passed since the device was last idle.  After seconds of idleness (frames generated
    on the CPU) this device needs ~25 ms of back-to-back launches to reach its steady clock, and launches 6-25 after an
    idle period run 8-10 % slow (profiles/r03_warmup_ramp.txt, r03_idle_ramp.txt): `--warmup 5 --steps 20` alone
    times exactly that window.  The short gap of the barrier + synchronize in front of the timed region does not
    restart the ramp (r03_idle_ramp.txt, idle <= 0.5 ms).  Returns (wall s, ms per launch over the K steps,
    ms per launch over the first min(first_n, K) of them, warm-up launches actually made, their device ms)."""
    h = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    h[0].record()
    n_probe = max(warmup, 2)
    for _ in range(n_probe):
        batch.launch()
    h[1].record()
    torch.cuda.synchronize()
    spent = h[0].elapsed_time(h[1])
    per = max(spent / n_probe, 1e-3)
    extra = int(min(20000, max(0.0, heat_ms - spent) / per + 0.999)) if heat_ms > 0 else 0
    for _ in range(extra):
        batch.launch()
    h[2].record()
    torch.cuda.synchronize()
    warm_ms = h[0].elapsed_time(h[2])
    # ONE pair of HIP events around the K launches, recorded on the stream the
    # kernels are launched on: average launch duration = elapsed / K (includes
    # the ~1-2 us gaps between back-to-back launches, so it is an upper bound;
    # per-step event pairs add a marker packet per launch and inflate it).  One
    # more marker after the first `first_n` launches when K is larger than that.
    e0, e1, em = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    k_first = min(first_n, steps)
    dist_sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        batch.launch()
        if i + 1 == k_first and k_first < steps:
            em.record()
    e1.record()
    torch.cuda.synchronize()
    dist_sync()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / steps
    ms_first = e0.elapsed_time(em) / k_first if k_first < steps else ms
    return t1 - t0, ms, ms_first, n_probe + extra, warm_ms


def timed_rounds(batch, n, warmup=2, floor_ms=100.0, min_rounds=5, max_rounds=400, heat_ms=30.0):
    """Side measurements and spreads: ROUNDS of `n` back-to-back launches, every round between two HIP events
    on the launch stream, as many rounds as it takes for the GPU time to sum to >= floor_ms (one short probe
    round sizes the run; then everything is enqueued with ONE synchronisation at the end, so no round starts
    on an idle device).  ~heat_ms of untimed launches come first: after seconds of idleness (frames generated on
    the CPU) a kernel takes ~25 ms of back-to-back launches to reach its steady time (profiles/r03_warmup_ramp.txt).
    Returns the per-round averages in ms per launch."""
    for _ in range(warmup):
        batch.launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        batch.launch()
    b.record()
    torch.cuda.synchronize()
    probe = max(a.elapsed_time(b), 1e-3)
    for _ in range(int(min(2000, heat_ms / probe * n))):
        batch.launch()
    rounds = int(min(max_rounds, max(min_rounds, -(-floor_ms // probe))))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(rounds + 1)]
    ev[0].record()
    for r in range(rounds):
        for _ in range(n):
            batch.launch()
        ev[r + 1].record()
    torch.cuda.synchronize()
    return [ev[r].elapsed_time(ev[r + 1]) / n for r in range(rounds)]


def spread(ms):
    """min / median / max of per-round averages (ms per launch)."""
    s = sorted(ms)
    return {"min": round(s[0], 4), "median": round(s[len(s) // 2], 4), "max": round(s[-1], 4), "rounds": len(s)}


def device_calibration(ctx, batch):
    """What THIS device gives kernels that only stream, measured in this run on the benchmark's own 2-GB output
    buffer: a plain 16-B-per-lane fill, and a copy of its first half onto its second (GB/s = bytes read + written)."""
    s = torch.cuda.current_stream().cuda_stream
    nbytes = batch.points.numel() * 4 // 32 * 32
    half = nbytes // 2
    base = batch.points.data_ptr()

    class _Fill:
        def launch(self):
            ctx.membench_fill(base, nbytes, s)

    class _Copy:
        def launch(self):
            ctx.membench_copy(base, base + half, half, s)

    # the persistent fill under a sampler of the computing card's sysfs telemetry (round 5's verdict, item 4a: what distinguishes a
    # 5.0 from a 6.3 TB/s device?  profiles/r06_devclass.txt: nothing these readings show)
    card = telemetry.find_card(pci_address=telemetry.torch_pci_address(torch.cuda.current_device()))
    with telemetry.Sampler(card, 0.004) as smp:
        f = spread(timed_rounds(_Fill(), 5))
    tel, static = smp.summary(), telemetry.static_state(card)
    c = spread(timed_rounds(_Copy(), 5))
    out = {"device_fill_GBs": round(nbytes / (f["median"] * 1e-3) / 1e9, 1),
           "device_copy_GBs": round(2 * half / (c["median"] * 1e-3) / 1e9, 1),
           "device_fill_ms": f, "device_copy_ms": c, "calibration_bytes": nbytes,
           "calibration_what": "k_membench_fill over the bench's output buffer; k_membench_copy of its first half onto "
                               "its second (bytes read + written); medians of >= 100 ms of rounds of 5 launches. "
                               "device_fill / device_copy: persistent blocks, plain stores (the single pass's shape; "
                               "device classes differ by ~15 % on it); *_oneshot_*: one block per 4 KiB, one 16-byte "
                               "access per thread (the headline kernel's shape: the ceiling of a store stream here)"}
    med = lambda k: (tel.get(k) or {}).get("median")
    out.update({"device_fill_sclk_MHz": med("sclk_MHz"), "device_fill_mclk_MHz": med("mclk_MHz"), "device_fill_fclk_MHz": med("dpm_fclk_MHz"),
                "device_fill_power_W": med("metrics_socket_power_W") or med("power_input_W"),
                "device_power_cap_W": static.get("power1_cap_W"),
                "device_partition": (static.get("current_compute_partition", "?") + "/" + static.get("current_memory_partition", "?")) if static else None,
                "device_fill_telemetry": tel if card else "the computing card's sysfs node is not visible"})
    # the headline kernel's launch shape: one-shot blocks, plain and non-temporal
    ctx.set_tuning("membench_blocks_per_cu", 0)
    ctx.set_tuning("membench_unroll", 1)
    for nt in (0, 1):
        ctx.set_tuning("membench_nt", nt)
        f1 = spread(timed_rounds(_Fill(), 5))
        c1 = spread(timed_rounds(_Copy(), 5))
        tag = "nt" if nt else "plain"
        out[f"device_fill_oneshot_{tag}_GBs"] = round(nbytes / (f1["median"] * 1e-3) / 1e9, 1)
        out[f"device_copy_oneshot_{tag}_GBs"] = round(2 * half / (c1["median"] * 1e-3) / 1e9, 1)
    ctx.set_tuning("membench_blocks_per_cu", 8)
    ctx.set_tuning("membench_unroll", 4)
    ctx.set_tuning("membench_nt", 0)
    return out


def compaction_counters(ctx):
    """d2pc_compact_stats since the last reset, per tile: every in-launch hand-off adds to them -- the single pass, the
    resident one-launch forms (k_compact_resident, k_compact_resident_lean) and the tile-fused callback kernels."""
    st = ctx.compact_stats()
    if not st["launches"]:
        return None
    t = max(st["tiles"], 1)
    return {"launches": st["launches"], "tiles_per_launch": st["tiles"] // st["launches"],
            "failed_polls_per_tile": round(st["failed_polls"] / t, 4),
            "wait_us_per_tile": round(st["wait_us"] / t, 4), "timeouts": st["timeouts"],
            "twopass_fallbacks": st["twopass_fallbacks"]}


def shader_clock_GHz(ctx, body, dev, seconds=0.06):
    """Shader clock WHILE `body.launch()` runs: d2pc_clock_probe_device on a second stream -- eight sleeping one-wave blocks, one
    per XCD, count shader cycles against the constant 100 MHz counter -- beside enough launches of the body to cover the
    probe's window; the computing card's socket power is sampled meanwhile.  Returns (median GHz over the XCDs, [per XCD], highest
    socket power sample in W) -- (None, [], power) if no block reported."""
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        body.launch()
    e1.record()
    torch.cuda.synchronize()
    per_ms = max(e0.elapsed_time(e1) / 3, 1e-3)
    n = int(seconds * 1e3 / per_ms) + 4
    lead = max(n // 8, 2)
    card = telemetry.find_card(pci_address=telemetry.torch_pci_address(torch.cuda.current_device()))
    with telemetry.Sampler(card, 0.004) as smp:   # socket power of the computing card while the body runs (the PARITY body: the 1,400 W cap)
        for _ in range(lead):            # the probe starts once the body's launches are well under way ...
            body.launch()
        ctx.clock_probe(out.data_ptr(), int(seconds * 1e6 * 0.6), side.cuda_stream)
        for _ in range(n):               # ... and ends before they run out
            body.launch()
        torch.cuda.synchronize()
    tel = smp.summary()
    power = (tel.get("metrics_socket_power_W") or tel.get("power_input_W") or {}).get("max")
    v = out.cpu().numpy().reshape(8, 2)
    ghz = sorted(float(c) / float(t) * 0.1 for c, t in v if t > 0)
    if not ghz:
        return None, [], power
    return round(ghz[len(ghz) // 2], 3), [round(x, 3) for x in ghz], power


def valu_issue(key, kernel_ms, clock_GHz, build):
    """The callback-body kernels are bound by VALU issue, not by HBM: next to their HBM fraction the line carries the fraction of
    the chip's vector-issue slots they fill = VALU wave-instructions per launch x 2 cycles (a wave64 instruction occupies its
    SIMD-32 for two; fp64 and several integer forms take four: a LOWER bound of the slots really taken) / (1,024 SIMDs x the
    launch's shader cycles).  The cycles are THIS run's: measured time x measured shader clock (shader_clock_GHz).  The
    instruction count is a property of the compiled kernel, read from the tracked counter pass (profiles/valu_issue.json,
    rocprofv3 --pmc SQ_INSTS_VALU) -- and only if that pass ran on the build loaded now; otherwise nothing is reported
    (advisor, round 5: a copied fraction went stale whenever the kernels changed)."""
    prof = os.path.join(ROOT, "profiles", "valu_issue.json")
    try:
        allp = json.load(open(prof))
        t = allp[key]
    except Exception:
        return {}
    if allp.get("build") != build or not clock_GHz:
        return {"valu_issue_frac": None, "valu_issue_note": f"profiles/valu_issue.json is of build {allp.get('build')}, this run of {build}"
                if allp.get("build") != build else "no shader clock measured"}
    cycles = kernel_ms * 1e-3 * clock_GHz * 1e9
    insts = t["valu_wave_insts_per_launch"]
    return {"valu_issue_frac": round(insts * 2.0 / (1024.0 * cycles), 4),
            "simd_cycles_per_valu_wave_instruction": round(1024.0 * cycles / insts, 3),
            "valu_wave_insts_per_launch": insts, "valu_wave_insts_source": t.get("source"),
            "valu_issue_how": "instruction count from the tracked counter pass of this build; cycles = this run's time x this run's clock"}


def cpu_baseline(q, border, budget_s=12.0):
    """The CPU restatement of cpp:63-85 (oracle, kind "port") on the same 4K
    workload, single thread like the reference's ros::spin(); bounded sample."""
    import oracle

    fr = synth_disparity(4, 0, W4K, H4K, "uniform")
    buf = oracle.reproject(fr, q, border=border)  # warm; output buffer reused below
    t0 = time.perf_counter()
    n = 0
    while True:
        oracle.reproject(fr, q, border=border, threads=1, out=buf)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    one = W4K * H4K * n / el / 1e6
    nt = oracle.max_threads()
    oracle.reproject(fr, q, border=border, threads=nt, out=buf)  # start the OpenMP team
    samples = []  # three samples of ~budget/6 each; the median is reported (OMP_PROC_BIND=close, set at start-up)
    m_total, el2_total = 0, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        m = 0
        while True:
            oracle.reproject(fr, q, border=border, threads=nt, out=buf)
            m += 1
            el2 = time.perf_counter() - t0
            if el2 > budget_s / 6:
                break
        samples.append(W4K * H4K * m / el2 / 1e6)
        m_total += m
        el2_total += el2
    allc = sorted(samples)[1]
    m, el2 = m_total, el2_total
    return {
        "value": round(one, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{n} frames of 3840x2160 fp32 (config 4, border {border}), oracle/d2pc_oracle.c FORM_CV24 "
                  f"(gcc -O3 -march=x86-64-v3 -ffp-contract=off), {el:.1f} s",
        "all_cores": {"value": round(allc, 2), "cores": nt,
                      "samples": [round(x, 1) for x in samples], "omp_proc_bind": os.environ.get("OMP_PROC_BIND", ""),
                      "sample": f"median of three samples ({m} frames, OpenMP over rows, {el2:.1f} s in all)"},
    }


def cpu_callback_body(q, border, budget_s=8.0):
    """The WHOLE callback body on the CPU (cpp:55-85: median 11 -> x 1/8 -> reproject + pack), one thread like the reference's
    spinner, a bounded sample of 8-bit 4K frames: the CPU column beside the callback_* lines.  The median is
    oracle.median_u8_fast -- Perreault & Hebert's constant-time sliding histogram, the algorithm cv::medianBlur runs for 8-bit
    images and k > 5 [upstream] (plain C, column stripes, no hand-written SIMD) -- pinned byte for byte to the oracle's checker
    median in tests/test_oracle.py; the reprojection part is the loop cpu_baseline times."""
    import oracle

    rng = np.random.default_rng(0xD2C)
    img = rng.integers(0, 256, size=(H4K, W4K)).astype(np.uint8)
    oracle.reproject(oracle.median_u8_fast(img[:256], 11), q, border=border, scale=0.125)  # warm
    t0 = time.perf_counter()
    tm, frames = 0.0, 0
    while True:
        t1 = time.perf_counter()
        med = oracle.median_u8_fast(img, 11)
        tm += time.perf_counter() - t1
        oracle.reproject(med, q, border=border, scale=0.125, threads=1)
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {"value": round(W4K * H4K * frames / el / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "median_algorithm": "Perreault-Hebert constant-time sliding histogram (oracle.median_u8_fast)",
            "median_share_of_time": round(tm / el, 3),
            "sample": f"{frames} frames of 3840x2160 u8: constant-time median 11 x 11 (Perreault-Hebert, the algorithm of "
                      f"cv::medianBlur for 8-bit k > 5) + x 1/8 + reproject, 1 thread, {el:.1f} s"}


def host_path_rates(q, border):
    """PCIe-inclusive rates of the host entry points for one 4K fp32 frame (never the headline): the
    synchronous d2pc_process on pageable buffers, and the pipelined path (depth 3, pinned staging,
    kernels storing straight into pinned host memory)."""
    import ctypes
    import time
    from disparity_to_point_cloud_amd.capi import FrameDesc
    L = d2pc.load_library()
    fr = synth_disparity(4, 0, W4K, H4K, "uniform")
    cap = d2pc.roi_points(W4K, H4K, border)
    res = {}
    with d2pc.Context(q=q, border=border) as ctx:
        out = np.empty((cap, 4), dtype=np.float32)
        n = ctypes.c_size_t()

        def once():
            st = L.d2pc_process(ctx.handle, fr.ctypes.data, 0, 1.0, W4K, H4K, fr.strides[0], out.ctypes.data, None, cap,
                                ctypes.byref(n))
            assert st == 0, st
        for _ in range(2):
            once()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            once()
            k += 1
        dt = (time.perf_counter() - t0) / k
        res["sync_d2pc_process"] = {"ms_per_frame": round(dt * 1e3, 3), "Mpixels_per_s": round(W4K * H4K / dt / 1e6, 1),
                                    "pcie_GBs": round((fr.nbytes + n.value * 16) / dt / 1e9, 1)}
        # the same call with frame and cloud in pinned memory (d2pc_host_alloc): the kernel reads and writes them in place
        pin_in, pin_out = d2pc.PinnedBuffer((H4K, W4K), np.float32), d2pc.PinnedBuffer((cap, 4), np.float32)
        pin_in.array[:] = fr

        def once_pinned():
            st = L.d2pc_process(ctx.handle, pin_in.array.ctypes.data, 0, 1.0, W4K, H4K, W4K * 4, pin_out.array.ctypes.data,
                                None, cap, ctypes.byref(n))
            assert st == 0, st
        for _ in range(2):
            once_pinned()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            once_pinned()
            k += 1
        dt = (time.perf_counter() - t0) / k
        res["sync_d2pc_process_pinned_io"] = {"ms_per_frame": round(dt * 1e3, 3),
                                              "Mpixels_per_s": round(W4K * H4K / dt / 1e6, 1),
                                              "pcie_GBs": round((fr.nbytes + n.value * 16) / dt / 1e9, 1)}
        pin_in.close()
        pin_out.close()
    with d2pc.Context(q=q, border=border) as ctx:
        ctx.pipeline_configure(depth=3, direct_host_write=True)
        desc = FrameDesc(0, 1.0, W4K, H4K, W4K * 4, 0, 0, 0)
        filled = set()

        def submit():
            hin, slot = ctypes.c_void_p(), ctypes.c_int()
            assert L.d2pc_pipeline_acquire(ctx.handle, ctypes.byref(desc), ctypes.byref(hin), ctypes.byref(slot)) == 0
            if slot.value not in filled:  # a producer decodes straight into the pinned slot; not part of this path
                ctypes.memmove(hin, fr.ctypes.data, fr.nbytes)
                filled.add(slot.value)
            assert L.d2pc_pipeline_submit(ctx.handle, slot.value) == 0

        def collect():
            slot, pts, m = ctypes.c_int(), ctypes.c_void_p(), ctypes.c_size_t()
            assert L.d2pc_pipeline_collect(ctx.handle, ctypes.byref(slot), ctypes.byref(pts), None, ctypes.byref(m), None) == 0
            L.d2pc_pipeline_release(ctx.handle, slot.value)
        submit()
        submit()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            submit()
            collect()
            k += 1
        dt = (time.perf_counter() - t0) / k
        collect()
        collect()
        res["pipelined_direct_host_write"] = {"ms_per_frame": round(dt * 1e3, 3),
                                              "Mpixels_per_s": round(W4K * H4K / dt / 1e6, 1)}
    return res


def workload_string(a):
    """config.workload, at most 120 characters (the driver's record keeps that many): what a step is and how it is warmed."""
    w = (f"config 4: {a.frames}x3840x2160 fp32/step, border {a.border}, {a.mode}; time-floored warm-up >={a.heat_ms:g} ms, "
         f"then {a.steps} timed steps")
    assert len(w) <= 120, len(w)
    return w


# The driver's record keeps the FIRST 21 scalar entries of `roofline` (lists and objects are dropped, strings cut at 120
# characters): these come first, in this order, whatever else the run adds (round 5's verdict, item 1).
ROOFLINE_HEAD = (
    "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch", "kernel_ms_avg",
    "frac_sustained", "device_fill_GBs", "compact_all_valid_frac", "compact_holes_frac", "compact_holes_index_frac",
    "c3_32x1080p_frac", "c4_1frame_compact_us", "c4_2frames_compact_us", "callback_parity_ms", "callback_compact_ms",
    "callback_parity_valu_issue_frac", "callback_parity_clock_GHz")


def order_roofline(r):
    """ROOFLINE_HEAD's keys first (None where this run did not measure one: multi-GPU runs, --no-variants), then the rest in
    the order they were added."""
    out = {k: r.get(k) for k in ROOFLINE_HEAD}
    for k, v in r.items():
        if k not in out:
            out[k] = v
    return out


def driver_view(line, n_scalars=21, max_str=120):
    """What the driver's BENCH_rNN.json `parsed` keeps of a bench line, as observed on rounds 4 and 5: of `roofline` the first
    21 scalar entries, strings cut at 120 characters, nested objects and lists dropped.  tests/test_bench_record.py uses it."""
    def scalars(d, limit):
        out = {}
        for k, v in d.items():
            if isinstance(v, (dict, list, tuple)):
                continue
            if len(out) >= limit:
                break
            out[k] = v[:max_str] if isinstance(v, str) else v
        return out
    view = scalars(line, 10 ** 6)
    for key in ("config", "roofline", "cpu_baseline"):
        if isinstance(line.get(key), dict):
            view[key] = scalars(line[key], n_scalars)
    return view


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=16, help="frames per step (ring size)")
    ap.add_argument("--border", type=int, default=40, help="ROI inset (reference: 40)")
    ap.add_argument("--mode", choices=["parity", "compact"], default="parity")
    ap.add_argument("--heat-ms", type=float, default=60.0,
                    help="time floor of the warm-up: after the W warm-up steps the same launch is repeated, untimed, until "
                         "this much device time has passed (0 = warm up by count only)")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the device calibration and the spread rounds behind the timed region, so that a rocprofv3 "
                         "--stats run sees the contract's launches only (its per-kernel average is then the line's)")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the PCIe-inclusive side measurement (its single-frame and host-memory launches would "
                         "mix into the per-kernel averages of a rocprofv3 --stats run)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="torch.distributed backend for N > 1 (default: nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL ONLY: all ranks use cuda:0 (needs --backend gloo; RCCL refuses two ranks on one "
                         "device).  Exercises the N-rank launch/barrier/report path on a 1-GPU box; the line is "
                         "marked rehearsal_shared_gpu and is not a scaling measurement")
    a = ap.parse_args()
    if a.share_gpu and a.backend != "gloo":
        raise SystemExit("--share-gpu needs --backend gloo")

    # everything that can be refused is refused BEFORE a process group exists (a rank that dies inside
    # init_process_group leaves its peers waiting for the launcher to reap them)
    rank, local_rank, world = multi_gpu.env_world()
    if world != max(a.gpus, 1):
        raise SystemExit(f"bench.py: WORLD_SIZE {world} != --gpus {a.gpus} (start it as `python bench.py --gpus N` or "
                         f"under torch.distributed.run with --nproc-per-node N)")
    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if a.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but this node exposes "
                         f"{torch.cuda.device_count()} GPU(s)")

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 when its first
    # communicator comes up (the GPU boxes export NCCL_DEBUG=VERSION): fd 1 points at stderr until
    # the first collective is through.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, local_rank, world = multi_gpu.init_distributed(backend=a.backend, share_gpu=a.share_gpu)
        multi_gpu.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    mode = d2pc.MODE_PARITY if a.mode == "parity" else d2pc.MODE_COMPACT

    # calibration: rank 0 owns it, everyone else receives the 136-byte blob
    ctx = d2pc.Context(device_id=local_rank, border=a.border, mode=mode)
    if rank == 0:
        ctx.set_q(d2pc.make_q())
    multi_gpu.broadcast_calibration(ctx, src=0)
    q = ctx.get_q()

    batch = DeviceBatch(ctx, a.frames, H4K, W4K, want_index=False, device=dev)
    fill_batch(batch, rank, "uniform")
    batch.launch()
    torch.cuda.synchronize()
    n_points = int(batch.counts.sum().item())

    # The contract's measurement: W warm-up steps (with a time floor: >= --heat-ms of device time since the last idle
    # point, see timed_steps), then K timed steps between barriers.  The rounds after it (kernel_ms_sustained_median)
    # show the sustained time of the same launch; frac and frac_sustained should agree within ~2 %.
    wall, kernel_ms, kernel_ms_first, warm_n, warm_ms = timed_steps(batch, a.steps, a.warmup, multi_gpu.barrier,
                                                                    heat_ms=a.heat_ms)
    # every rank calibrates its device on its own output buffer (plain fill / copy, >= 100 ms each)
    cal = device_calibration(ctx, batch) if not a.no_extras else None
    headline_clock = shader_clock_GHz(ctx, batch, dev) if (not a.no_extras and world == 1) else (None, [], None)
    wall = multi_gpu.allreduce_max(wall)
    kernel_ms_max = multi_gpu.allreduce_max(kernel_ms)
    per_rank_kernel_ms = multi_gpu.allgather_floats(kernel_ms)
    if mode == d2pc.MODE_COMPACT:
        ctx.check_async_error()

    pixels_per_step = a.frames * W4K * H4K
    value = world * pixels_per_step * a.steps / wall / 1e6
    alg = algorithmic_bytes(batch, n_points, False)
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "Mpixels/s reprojected (device-resident)",
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            # <= 120 characters: the driver's record cuts strings there (tests/test_bench_record.py replays the rule)
            "workload": workload_string(a),
            "workload_detail": f"config 4: {a.frames} x 3840x2160 fp32 disparity frames per step, d~U(0.5,128), border {a.border}, "
                               f"mode {a.mode}, one stream per GPU, Q broadcast from rank 0; time-floored warm-up: the {a.warmup} "
                               f"warm-up steps are followed by untimed launches of the same step until >= {a.heat_ms:g} ms of device "
                               f"time have passed (roofline.warmup_launches_actual), then exactly {a.steps} timed steps",
            "frames_per_step": a.frames, "width": W4K, "height": H4K, "border": a.border, "mode": a.mode,
            "points_per_step": n_points,
            "collective_backend": multi_gpu.backend_name(),
            # what the collective backend itself reports (an 8-GPU record shows RCCL saw 8 ranks)
            "ranks_seen_by_backend": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": "k_reproject_pack_small" if mode == d2pc.MODE_PARITY else "k_compact_onepass",
            "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": round(kernel_ms, 4),
            "kernel_ms_avg_max_over_ranks": round(kernel_ms_max, 4),
            "kernel_ms_avg_per_rank": [round(x, 4) for x in per_rank_kernel_ms],
            "read_component_GBs": round(4 * a.frames * batch.roi_n / (kernel_ms * 1e-3) / 1e9, 1),
            # the first min(20, K) launches of the timed region on their own (what `--steps 20` samples)
            "frac_first_20": round(alg / (kernel_ms_first * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernel_ms_first_20": round(kernel_ms_first, 4),
            "warmup_launches_actual": warm_n, "warmup_ms_actual": round(warm_ms, 2), "warmup_heat_floor_ms": a.heat_ms,
        },
    }
    fills = multi_gpu.allgather_floats(cal["device_fill_GBs"] if cal else 0.0)
    if world > 1 and cal:
        out["roofline"]["device_fill_GBs_per_rank"] = [round(x, 1) for x in fills]
        out["roofline"]["device_copy_GBs_rank0"] = cal["device_copy_GBs"]
    if rank == 0 and world == 1 and cal:
        # after the contract's timed region: how much the same launch moves on this device (rounds of `steps`
        # launches until >= 100 ms), next to what the device gives a plain fill / copy
        sp = spread(timed_rounds(batch, a.steps, 0))
        out["roofline"]["kernel_ms_spread"] = sp
        # scalars (nested objects do not survive the driver's parser): the sustained time of the same launch
        out["roofline"]["kernel_ms_sustained_median"] = sp["median"]
        out["roofline"]["kernel_ms_sustained_min"] = sp["min"]
        out["roofline"]["kernel_ms_sustained_max"] = sp["max"]
        out["roofline"]["frac_sustained"] = round(alg / (sp["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["roofline"]["frac_at_min_median_max_ms"] = [round(alg / (sp[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                        for k in ("min", "median", "max")]
        out["roofline"].update(cal)
        out["roofline"]["headline_clock_GHz"] = headline_clock[0]
        out["roofline"]["headline_socket_power_W"] = headline_clock[2]
        out["roofline"]["achieved_over_device_fill"] = round(achieved / cal["device_fill_GBs"], 4)
        out["roofline"]["achieved_over_device_copy"] = round(achieved / cal["device_copy_GBs"], 4)
        # against the best streams of the kernel's own launch shape on this device: a 1:4 read:write stream sits
        # between a copy (1:1) and a fill (0:1)
        best_fill = max(cal["device_fill_oneshot_plain_GBs"], cal["device_fill_oneshot_nt_GBs"])
        best_copy = max(cal["device_copy_oneshot_plain_GBs"], cal["device_copy_oneshot_nt_GBs"])
        out["roofline"]["achieved_over_device_oneshot_fill"] = round(achieved / best_fill, 4)
        out["roofline"]["achieved_over_device_oneshot_copy"] = round(achieved / best_copy, 4)
        if mode == d2pc.MODE_COMPACT:
            out["roofline"]["compaction_counters"] = compaction_counters(ctx)
    if a.share_gpu:
        out["config"]["rehearsal_shared_gpu"] = True
    out["config"]["build"] = build_id()
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            t = json.load(open(prof)).get(f"{a.mode}_border{a.border}_frames{a.frames}")
            if t:
                # copied from the tracked profile of an EARLIER profiler run (separate --pmc passes): not measured
                # by this run
                out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                out["roofline"]["traffic_measured"] = False
                out["roofline"]["traffic_source"] = t.get("source")
                out["roofline"]["traffic_profile_build"] = t.get("build")
        except Exception:
            pass

    if rank == 0 and world == 1 and not a.no_variants:  # side measurements only in the 1-GPU run
        variants = {}
        n_side = max(a.steps // 4, 5)  # launches per round; rounds repeat until >= 100 ms (timed_rounds)
        for name, vmode, vborder, kind, idx, form in (
            ("parity_border0", d2pc.MODE_PARITY, 0, "uniform", False, d2pc.FORM_DEFAULT),
            ("compact_border40_all_valid", d2pc.MODE_COMPACT, 40, "uniform", False, d2pc.FORM_DEFAULT),
            ("compact_border40_30pct_holes", d2pc.MODE_COMPACT, 40, "holes", False, d2pc.FORM_DEFAULT),
            ("compact_border40_30pct_holes_index", d2pc.MODE_COMPACT, 40, "holes", True, d2pc.FORM_DEFAULT),
            ("compact_border40_all_valid_index", d2pc.MODE_COMPACT, 40, "uniform", True, d2pc.FORM_DEFAULT),
            # the headline workload with one OpenCV generation's arithmetic reproduced bit for bit (d2pc_set_reproject_form)
            ("parity_border40_opencv24_bit_for_bit", d2pc.MODE_PARITY, 40, "uniform", False, d2pc.FORM_CV24),
            ("parity_border40_opencv4_bit_for_bit", d2pc.MODE_PARITY, 40, "uniform", False, d2pc.FORM_CV4),
        ):
            c2 = d2pc.Context(device_id=local_rank, border=vborder, mode=vmode, q=q)
            c2.set_reproject_form(form)
            b2 = DeviceBatch(c2, a.frames, H4K, W4K, want_index=idx, device=dev)
            if kind == "uniform":
                b2.disp.copy_(batch.disp)
            else:
                fill_batch(b2, rank, kind)
            b2.launch()
            torch.cuda.synchronize()
            npts = int(b2.counts.sum().item())
            if vmode == d2pc.MODE_COMPACT:
                c2.compact_stats_reset()
            sp = spread(timed_rounds(b2, n_side, 3))
            kms = sp["median"]
            ab = algorithmic_bytes(b2, npts, idx)
            variants[name] = {"Mpixels_per_s": round(pixels_per_step / (kms * 1e-3) / 1e6, 1),
                              "achieved_GBs": round(ab / (kms * 1e-3) / 1e9, 1),
                              "frac": round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "kernel_ms_avg": kms, "kernel_ms_spread": sp, "points_per_step": npts}
            if vmode == d2pc.MODE_COMPACT:
                c2.check_async_error()
                variants[name]["compaction_counters"] = compaction_counters(c2)
            del b2
            c2.close()
        # camera-size COMPACT launches with indices, ~30 % invalid (iid): config 3's geometry as one frame and as a 32-frame batch,
        # ONE 4K frame and TWO.  A camera delivers a NEW frame every time (hpp:77-78): every launch takes the next frame(s) of a
        # ring of distinct frames totalling >= 512 MB of input (16 x 4K, 64 x 1080p), with its own output slot, so that no launch
        # finds its input (or its previous output) in the 256-MiB Infinity Cache; `*_cached_us` is the same launch repeated on
        # ONE frame (round 5's figure: optimistic, the input stays cached).
        W3, H3 = 1920, 1080
        for name, nfr, ring, wv, hv, what in (
                ("compact_1080p_30pct_holes_index_1frame", 1, 64, W3, H3, "config 3 geometry; k_compact_resident (one launch, one resident block per 2048-pixel tile)"),
                ("compact_1080p_30pct_holes_index_32frames", 32, 32, W3, H3, "config 3 geometry; k_compact_onepass"),
                ("compact_4k_30pct_holes_index_1frame", 1, 16, W4K, H4K, "k_compact_resident_lean<32>: one launch, 955 resident blocks of 8192 pixels, disparities in registers between count and scatter"),
                ("compact_4k_30pct_holes_index_2frames", 2, 16, W4K, H4K, "two 4K frames in one call: k_compact_resident_lean<32> twice, back to back")):
            c2 = d2pc.Context(device_id=local_rank, border=40, mode=d2pc.MODE_COMPACT, q=q)
            b2 = DeviceBatch(c2, ring, hv, wv, want_index=True, device=dev, reserve=False)
            c2.reserve(wv, hv, nfr)
            for f in range(ring):
                b2.disp[f].copy_(torch.from_numpy(synth_disparity(3, f, wv, hv, "holes")))
            cam = RingLaunch(c2, b2, nfr)
            for _ in range(cam.positions):
                cam.launch()
            torch.cuda.synchronize()
            npts = int(b2.counts.sum().item()) * nfr // ring   # points per launch, averaged over the ring
            c2.compact_stats_reset()
            sp = spread(timed_rounds(cam, max(a.steps // 2, 20) // cam.positions * cam.positions or cam.positions, 5))
            kms = sp["median"]
            ab = 4 * nfr * b2.roi_n + 20 * npts
            variants[name] = {"Mpixels_per_s": round(nfr * wv * hv / (kms * 1e-3) / 1e6, 1),
                              "achieved_GBs": round(ab / (kms * 1e-3) / 1e9, 1),
                              "frac": round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "ms_per_launch": kms, "ms_per_launch_spread": sp, "points_per_launch": npts,
                              "ring_frames": ring, "ring_input_MB": round(ring * wv * hv * 4 / 1e6, 1),
                              "compaction_counters": compaction_counters(c2), "what": what}
            c2.check_async_error()
            if ring > nfr:   # the same launch on ONE cached position of the ring
                one = RingLaunch(c2, b2, nfr, fixed=True)
                c2.compact_stats_reset()
                spc = spread(timed_rounds(one, max(a.steps // 2, 20), 5))
                variants[name]["cached_ms_per_launch"] = spc["median"]
                variants[name]["cached_ms_per_launch_spread"] = spc
                variants[name]["cached_compaction_counters"] = compaction_counters(c2)
                c2.check_async_error()
            del b2
            c2.close()
        # the whole device-resident callback body (cpp:55-85): 8-bit disparity ->
        # median 11x11 -> x1/8 -> reproject + pack, same 16 x 4K geometry
        c3 = d2pc.Context(device_id=local_rank, border=a.border, mode=d2pc.MODE_PARITY, q=q)
        b3 = DeviceBatch(c3, a.frames, H4K, W4K, dtype=torch.uint8, device=dev)
        raw = torch.randint(0, 256, (a.frames, H4K, W4K), dtype=torch.uint8, device=dev,
                            generator=torch.Generator(device=dev).manual_seed(0xD2C))
        s3 = torch.cuda.current_stream().cuda_stream

        class _Body:  # the entry point a caller uses: d2pc_process_mono_device (k_callback_bs at this size)
            def launch(self):
                c3.process_mono_device(raw.data_ptr(), d2pc.DTYPE_U8, W4K, H4K, W4K, W4K * H4K, a.frames, 11, 0.125,
                                       b3.points.data_ptr(), None, b3.stride, b3.counts.data_ptr(), s3)

        class _TwoLaunches:  # the same work as two kernels: cpp:55-57 for the pixels cpp:70-76 read, then the points
            def launch(self):
                c3.median_roi_device(raw.data_ptr(), W4K, H4K, W4K, W4K * H4K, a.frames, b3.disp.data_ptr(), W4K,
                                     W4K * H4K, 11, s3)
                b3.launch(scale=0.125)

        sp, sp2 = spread(timed_rounds(_Body(), n_side, 3)), spread(timed_rounds(_TwoLaunches(), n_side, 3))
        kms, kms2 = sp["median"], sp2["median"]
        clk, clk_xcd, clk_power = shader_clock_GHz(c3, _Body(), dev)
        ab_cb = a.frames * b3.roi_n * 17   # 1 B read + 16 B written per ROI pixel (the window's halo re-reads come from cache)
        variants["callback_u8_median11_parity_border40"] = {
            "Mpixels_per_s": round(pixels_per_step / (kms * 1e-3) / 1e6, 1), "kernel_ms_avg": kms,
            "achieved_GBs": round(ab_cb / (kms * 1e-3) / 1e9, 1), "frac": round(ab_cb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "clock_GHz": clk, "clock_GHz_per_xcd": clk_xcd, "socket_power_W_max": clk_power,
            **valu_issue("callback_parity", kms, clk, build_id()),
            "kernel_ms_spread": sp, "as_two_launches_ms_spread": sp2,
            "what": "d2pc_process_mono_device: k_callback_bs<11> (bit-sliced median of a tile + its points from LDS) per step",
            "as_two_launches_ms": round(kms2, 4),
            "as_two_launches_what": "k_median_bs_u8<11> over the inset ROI + k_reproject_pack<U8>"}
        b3.disp.copy_(raw)
        sp = spread(timed_rounds(b3, n_side, 3))
        kms = sp["median"]
        ab = a.frames * b3.roi_n * 17
        variants["parity_u8_input_border40"] = {
            "Mpixels_per_s": round(pixels_per_step / (kms * 1e-3) / 1e6, 1),
            "achieved_GBs": round(ab / (kms * 1e-3) / 1e9, 1), "frac": round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernel_ms_avg": kms, "kernel_ms_spread": sp,
            "what": "fused cpp:61 decode: 1 B read + 16 B written per pixel"}
        # the same callback body with a DENSE cloud out (COMPACT): 8-bit frames with ~30 % zero pixels (no match),
        # in 64 x 64 blocks and iid (the 11 x 11 median closes most iid holes) -- one kernel
        # (k_callback_bs_compact) against the filter launch + the single-pass compaction launch
        c4 = d2pc.Context(device_id=local_rank, border=a.border, mode=d2pc.MODE_COMPACT, q=q)
        b4 = DeviceBatch(c4, a.frames, H4K, W4K, dtype=torch.uint8, want_index=True, device=dev)
        gen = torch.Generator(device=dev).manual_seed(0xD2D)
        for hole_kind in ("blocky", "iid"):
            raw4 = raw.clone()
            if hole_kind == "blocky":
                m = torch.rand((a.frames, (H4K + 63) // 64, (W4K + 63) // 64), device=dev, generator=gen) < 0.3
                raw4[m.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)[:, :H4K, :W4K]] = 0
            else:
                raw4[torch.rand(raw4.shape, device=dev, generator=gen) < 0.3] = 0

            class _BodyCompact:
                def launch(self):
                    c4.process_mono_device(raw4.data_ptr(), d2pc.DTYPE_U8, W4K, H4K, W4K, W4K * H4K, a.frames, 11, 0.125,
                                           b4.points.data_ptr(), b4.index.data_ptr(), b4.stride, b4.counts.data_ptr(), s3)

            rec = {}
            for fused in (2, 1, 0):
                c4.set_tuning("callback_fused_compact", fused)
                _BodyCompact().launch()
                torch.cuda.synchronize()
                c4.compact_stats_reset()
                sp = spread(timed_rounds(_BodyCompact(), n_side, 3))
                c4.check_async_error()
                rec[fused] = (sp, compaction_counters(c4))
            npts = int(b4.counts.sum().item())
            kms, kms2 = rec[2][0]["median"], rec[0][0]["median"]
            c4.set_tuning("callback_fused_compact", 2)
            clk, clk_xcd, clk_power = shader_clock_GHz(c4, _BodyCompact(), dev)
            c4.check_async_error()
            ab_cc = a.frames * b4.roi_n * 1 + 20 * npts   # 1 B read per ROI pixel + (16 + 4) B per surviving point
            variants[f"callback_u8_median11_compact_30pct_zero_{hole_kind}"] = {
                "Mpixels_per_s": round(pixels_per_step / (kms * 1e-3) / 1e6, 1), "kernel_ms_avg": kms,
                "achieved_GBs": round(ab_cc / (kms * 1e-3) / 1e9, 1), "frac": round(ab_cc / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "clock_GHz": clk, "clock_GHz_per_xcd": clk_xcd, "socket_power_W_max": clk_power,
                **valu_issue("callback_compact_" + hole_kind, kms, clk, build_id()),
                "kernel_ms_spread": rec[2][0], "points_per_step": npts, "compaction_counters": rec[2][1],
                "what": "d2pc_process_mono_device, COMPACT + indices: k_callback_bs_compact_pipe<11> (persistent blocks: median of "
                        "a tile, then the previous tile's surviving points in row-major order; row counts handed over inside the launch)",
                "as_two_launches_ms": kms2, "as_two_launches_ms_spread": rec[0][0],
                "as_two_launches_what": "k_median_bs_u8<11> over the inset ROI + k_compact_onepass<U8>",
                "speedup_over_two_launches": round(kms2 / kms, 3),
                "one_tile_per_block_form_ms_spread": rec[1][0], "one_tile_per_block_form_counters": rec[1][1]}
            del raw4
        del b4
        c4.close()
        del b3
        c3.close()
        if not a.no_host_path:
            variants["host_path_pcie_inclusive_1x4K_parity"] = host_path_rates(q, a.border)
        out["variants_1gpu"] = variants
        # scalars the driver's parser keeps (nested objects under `parsed` are dropped): the north-star's own kernels
        r = out["roofline"]
        v = variants
        r["compact_all_valid_frac"] = v["compact_border40_all_valid"]["frac"]
        r["compact_holes_frac"] = v["compact_border40_30pct_holes"]["frac"]
        r["compact_holes_index_frac"] = v["compact_border40_30pct_holes_index"]["frac"]
        r["c3_32x1080p_frac"] = v["compact_1080p_30pct_holes_index_32frames"]["frac"]
        # camera-shaped launches on DISTINCT frames (a ring >= 512 MB of input); *_cached_us: the same frame again and again
        r["c4_1frame_compact_us"] = round(v["compact_4k_30pct_holes_index_1frame"]["ms_per_launch"] * 1e3, 2)
        r["c4_2frames_compact_us"] = round(v["compact_4k_30pct_holes_index_2frames"]["ms_per_launch"] * 1e3, 2)
        r["callback_parity_ms"] = v["callback_u8_median11_parity_border40"]["kernel_ms_avg"]
        r["callback_compact_ms"] = v["callback_u8_median11_compact_30pct_zero_blocky"]["kernel_ms_avg"]
        r["callback_parity_valu_issue_frac"] = v["callback_u8_median11_parity_border40"].get("valu_issue_frac")
        r["callback_parity_clock_GHz"] = v["callback_u8_median11_parity_border40"].get("clock_GHz")
        # (beyond the driver's first 21)
        r["compact_all_valid_index_frac"] = v["compact_border40_all_valid_index"]["frac"]
        r["compact_holes_ms"] = v["compact_border40_30pct_holes"]["kernel_ms_avg"]
        r["c3_1frame_us"] = round(v["compact_1080p_30pct_holes_index_1frame"]["ms_per_launch"] * 1e3, 2)
        r["c3_1frame_cached_us"] = round(v["compact_1080p_30pct_holes_index_1frame"]["cached_ms_per_launch"] * 1e3, 2)
        r["c4_1frame_compact_cached_us"] = round(v["compact_4k_30pct_holes_index_1frame"]["cached_ms_per_launch"] * 1e3, 2)
        r["c4_2frames_compact_cached_us"] = round(v["compact_4k_30pct_holes_index_2frames"]["cached_ms_per_launch"] * 1e3, 2)
        r["c4_1frame_wait_us_per_tile"] = (v["compact_4k_30pct_holes_index_1frame"]["compaction_counters"] or {}).get("wait_us_per_tile")
        r["c4_2frames_wait_us_per_tile"] = (v["compact_4k_30pct_holes_index_2frames"]["compaction_counters"] or {}).get("wait_us_per_tile")
        r["parity_u8_frac"] = v["parity_u8_input_border40"]["frac"]
        r["callback_parity_frac"] = v["callback_u8_median11_parity_border40"]["frac"]
        r["callback_compact_frac"] = v["callback_u8_median11_compact_30pct_zero_blocky"]["frac"]
        r["callback_compact_valu_issue_frac"] = v["callback_u8_median11_compact_30pct_zero_blocky"].get("valu_issue_frac")
        r["callback_compact_clock_GHz"] = v["callback_u8_median11_compact_30pct_zero_blocky"].get("clock_GHz")
        r["callback_parity_socket_power_W"] = v["callback_u8_median11_parity_border40"].get("socket_power_W_max")
    if rank == 0 and world == 1 and not a.no_cpu:  # contract: CPU baseline on rank 0 at N=1 only
        out["cpu_baseline"] = cpu_baseline(q, a.border)
        if "variants_1gpu" in out:  # the CPU column of the callback-body lines
            cb = cpu_callback_body(q, a.border)
            out["cpu_baseline"]["callback_body"] = cb
            out["roofline"]["callback_cpu_port_Mpix_s"] = cb["value"]
    out["roofline"] = order_roofline(out["roofline"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    multi_gpu.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
