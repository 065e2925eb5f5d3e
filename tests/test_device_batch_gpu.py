"""GPU tests of the device-resident batched entry point (d2pc_process_device)
at BASELINE.json's full sizes, driven through torch device memory/streams."""
import numpy as np
import pytest

import disparity_to_point_cloud_amd as d2pc
import oracle
from helpers import assert_points_close, synth_disparity, variant_for

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _batch(ctx, frames, want_index, reserve=True):
    from disparity_to_point_cloud_amd.torch_api import DeviceBatch

    n, (h, w) = len(frames), frames[0].shape
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16}[frames[0].dtype]
    b = DeviceBatch(ctx, n, h, w, dtype=tdt, want_index=want_index, reserve=reserve)
    stack = np.stack(frames)
    b.disp.copy_(torch.from_numpy(stack.view(np.int16)).view(tdt) if stack.dtype == np.uint16 else torch.from_numpy(stack))
    b.points.fill_(float("nan"))
    return b


@pytest.mark.parametrize("border", [40, 0])
def test_c4_4k_parity_batch(border):
    """configs[3]: 3840x2160 fp32 stream, PARITY, border 40 and 0."""
    q = d2pc.make_q()
    frames = [synth_disparity(4, f, 3840, 2160, "uniform") for f in range(3)]
    with d2pc.Context(q=q, border=border) as ctx:
        b = _batch(ctx, frames, want_index=False)
        b.launch()
        res = b.results()
    for f, (pts, _) in enumerate(res):
        want = oracle.reproject(frames[f], q, border=border, threads=8)
        assert len(pts) == (3840 - 2 * border) * (2160 - 2 * border)
        assert_points_close(pts, want, max_ulp=1, rel=1e-5, what=f"4K frame {f}")


@pytest.mark.parametrize("algo", [1, 2, 4])
def test_c4_4k_compact_batch(algo):
    q = d2pc.make_q()
    kinds = ["uniform", "holes", "blocky", "holes"]
    frames = [synth_disparity(4, 10 + f, 3840, 2160, k) for f, k in enumerate(kinds)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        b = _batch(ctx, frames, want_index=True)
        for _ in range(3):  # relaunch: state must re-initialise every call
            b.launch()
        res = b.results()
        ctx.check_async_error()
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40)
        assert len(pts) == len(wp), f"frame {f} count"
        assert np.array_equal(idx, wi), f"frame {f} indices"
        assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"4K compact frame {f}")


@pytest.mark.parametrize("chunk_mb,first", [(96, 0), (1, 0), (9, 1), (20, 3), (4096, 0), (17, 65535)])
@pytest.mark.parametrize("dtype", ["f32", "f32_unaligned", "u8", "u16"])
def test_chunked_two_pass_whatever_the_chunking(chunk_mb, first, dtype):
    """compact_algo 4 (k_compact_chunk): launch i scatters chunk i-1 and counts chunk i.  The result may not depend on where
    the batch is cut: one frame per chunk, ragged last chunks, a first chunk longer than the rest, one chunk for everything;
    the 16-byte row loads of the count blocks (f32), their scalar path (an odd width; 8- and 16-bit input)."""
    q = d2pc.make_q(cx=411.3, cy=140.2, nx=823, ny=291)
    n, h, w = 11, 291, 823 if dtype == "f32_unaligned" else 824
    rng = np.random.default_rng(w + chunk_mb)
    if dtype.startswith("f32"):
        frames = [synth_disparity(3, 70 + f, w, h, ["holes", "blocky", "uniform"][f % 3]) for f in range(n)]
        scale = 1.0
    else:
        hi, dt = (256, np.uint8) if dtype == "u8" else (65536, np.uint16)
        frames = [rng.integers(0, hi, size=(h, w)).astype(dt) for _ in range(n)]
        for fr in frames:
            fr[rng.random((h, w)) < 0.3] = 0
        scale = 0.125 if dtype == "u8" else 1.0 / 64
    frames[4][:] = 0          # a frame without a single valid point
    with d2pc.Context(q=q, border=3, mode=d2pc.MODE_COMPACT, compact_algo=4, variant="exp") as ctx:
        ctx.set_tuning("chunk_mb", chunk_mb)
        ctx.set_tuning("chunk_first_frames", first)
        b = _batch(ctx, frames, want_index=True)
        for _ in range(2):
            b.points.fill_(0)
            b.launch(scale=scale)
        res = b.results()
        ctx.check_async_error()
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=3, scale=scale)
        assert len(pts) == len(wp), f"frame {f} count"
        assert np.array_equal(idx, wi), f"frame {f}"
        assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"frame {f}")


@pytest.mark.parametrize("form", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("case", ["f32", "f32_unaligned", "u8", "u16", "general", "sliver", "narrow", "min_disparity", "cv24"])
def test_single_pass_kernel_forms(form, case):
    """The single pass (compact_algo 2) in its kernel forms -- tuning "onepass_form": 2 = the product's (the count phase packs
    the survivors of each 256-pixel run, dense scatter, the launch cleans the state of its successor); experiment build:
    1 = raw tiles in LDS, every pixel decided in both phases (rounds 2-4), 3 = form 2 on 4,096-pixel tiles with 8 worker
    waves, 4 = form 2 with the control wave as the block's loader -- against the oracle: 16-byte row loads and the scalar path (an odd width), 8- and
    16-bit input, a general Q (exact path in the count phase), a Q with tiny W (slivers: runs fall back to the real
    arithmetic), a ROI narrower than a run (coordinates by division), min_disparity, OpenCV 2.4's arithmetic; a frame
    without a single valid point and a ragged last tile in every case."""
    rng = np.random.default_rng(len(case) + form)
    q = d2pc.make_q(cx=411.3, cy=140.2, nx=823, ny=291)
    n, h, w, border, scale, dmin, oform = 9, 291, 824, 3, 1.0, -np.inf, oracle.FORM_CV24
    if case == "f32_unaligned":
        w = 823
    if case == "narrow":
        n, h, w = 5, 1500, 206   # ROI 200 pixels wide: a run of 256 pixels spans two or three rows
    if case in ("u8", "u16"):
        hi, dt = (256, np.uint8) if case == "u8" else (65536, np.uint16)
        frames = [rng.integers(0, hi, size=(h, w)).astype(dt) for _ in range(n)]
        for fr in frames:
            fr[rng.random((h, w)) < 0.3] = 0
        scale = 0.125 if case == "u8" else 1.0 / 64
    else:
        frames = [synth_disparity(3, 170 + f, w, h, ["holes", "blocky", "uniform"][f % 3]) for f in range(n)]
        frames[0][h // 2, 5:11] = [np.nan, np.inf, -np.inf, -1.0, 3.4028235e38, 1e-45]
    frames[2][:] = 0          # a frame without a single valid point
    if case == "general":
        q = rng.uniform(-1, 1, 16)
        q[12:16] = [2e-4, 1e-4, 0.03, 0.7]
        oform = oracle.FORM_CV4
    if case == "sliver":
        q[14] = 1e-36
    if case == "min_disparity":
        dmin = 40.0
    ulp = 0 if case in ("general", "cv24") else 1
    with d2pc.Context(q=q, border=border, mode=d2pc.MODE_COMPACT, compact_algo=2, min_disparity=dmin,
                      variant=variant_for(exp=form != 2)) as ctx:   # (form 2 is the product's; the others: experiment build)
        ctx.set_tuning("onepass_form", form)
        if case == "cv24":
            ctx.set_reproject_form(d2pc.FORM_CV24)
        b = _batch(ctx, frames, want_index=case != "u16")
        for _ in range(2):
            b.points.fill_(0)
            b.launch(scale=scale)
        res = b.results()
        ctx.check_async_error()
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=border, scale=scale, form=oform, min_disparity=dmin)
        assert len(pts) == len(wp), f"frame {f} count"
        if idx is not None:
            assert np.array_equal(idx, wi), f"frame {f}"
        assert_points_close(pts, wp, max_ulp=ulp, rel=1e-5, what=f"frame {f}")


@pytest.mark.parametrize("form", [2, 3, 4, 5, 6, 7])
def test_single_pass_dense_forms_at_4k(form):
    q = d2pc.make_q()
    kinds = ["uniform", "holes", "blocky", "holes", "uniform"]
    frames = [synth_disparity(4, 10 + f, 3840, 2160, k) for f, k in enumerate(kinds)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=2, variant=variant_for(exp=form != 2)) as ctx:
        ctx.set_tuning("onepass_form", form)
        b = _batch(ctx, frames, want_index=True)
        for _ in range(3):
            b.launch()
        res = b.results()
        ctx.check_async_error()
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40)
        assert len(pts) == len(wp), f"frame {f} count"
        assert np.array_equal(idx, wi), f"frame {f} indices"
        assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"4K compact frame {f}")


def test_single_pass_self_cleaning_state_survives_other_uses_of_the_buffer():
    """The dense single pass zeroes the idle half of its state buffer for its successor (no clear kernel in front of eager
    launches).  ONE context, ONE stream, the default routing: big batches (single pass) relaunched, interleaved with a
    camera-size launch (resident blocks: epochs in the same buffer), a mid-size one (two-pass: its partial counts land in
    the same buffer) and a big batch of ANOTHER frame count (other half size) -- whatever ran last, the next single-pass
    launch must find clean state or clear it: every cloud against the oracle."""
    q = d2pc.make_q()
    kinds = ["holes", "blocky", "uniform"]
    pool = [synth_disparity(3, 400 + f, 1920, 1080, kinds[f % 3]) for f in range(30)]
    want = [oracle.reproject_compact(fr, q, border=40) for fr in pool]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        batches = {n: _batch(ctx, pool[:n], want_index=True, reserve=False) for n in (30, 24, 12, 2)}
        ctx.compact_stats_reset()
        seq = [30, 30, 2, 30, 12, 30, 24, 24, 30, 2, 24]   # (12 x 1080p: too many blocks to be resident, too few tiles for the single pass)
        for step, n in enumerate(seq):
            b = batches[n]
            b.points.fill_(0)
            b.index.fill_(0)
            b.counts.fill_(0)
            b.launch()
            res = b.results()
            ctx.check_async_error()
            for f, (pts, idx) in enumerate(res):
                assert np.array_equal(idx, want[f][1]), f"step {step} (n = {n}) frame {f}"
                assert_points_close(pts, want[f][0], max_ulp=1, what=f"step {step} (n = {n}) frame {f}")
        st = ctx.compact_stats()
        # eight single-pass launches (n = 30, 24) and two resident ones (n = 2); the two-pass form (n = 12) hands nothing over
        assert st["launches"] == 10 and st["timeouts"] == 0, st


@pytest.mark.parametrize("big_batch_algo", [2, 4])
def test_default_algorithm_on_a_large_batch(big_batch_algo):
    """compact_algo = 0 picks the big-batch form -- the single pass -- for big launches (>= 4 frames, >= 20,480 tiles):
    30 frames of 1920x1080 (27k tiles) with three validity patterns, relaunched, against the oracle.  (In the experiment
    build the tuning key big_batch_algo = 4 sends such launches through the chunked two-pass instead: second case.)"""
    q = d2pc.make_q()
    kinds = ["holes", "blocky", "uniform"]
    frames = [synth_disparity(3, 40 + f, 1920, 1080, kinds[f % 3]) for f in range(30)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, variant=variant_for(big_batch_algo)) as ctx:
        if big_batch_algo != 2:
            ctx.set_tuning("big_batch_algo", big_batch_algo)
        b = _batch(ctx, frames, want_index=True)
        for _ in range(2):
            b.launch()
        res = b.results()
        ctx.check_async_error()
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40)
        assert np.array_equal(idx, wi), f"frame {f}"
        assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"frame {f}")


def test_batch_u8_native_geometry():
    q = d2pc.make_q()
    rng = np.random.default_rng(21)
    frames = [rng.integers(0, 256, size=(480, 752)).astype(np.uint8) for _ in range(5)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        b = _batch(ctx, frames, want_index=True)
        b.launch(scale=0.125)
        res = b.results()
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40, scale=0.125)
        assert np.array_equal(idx, wi)
        assert_points_close(pts, wp, max_ulp=1)


@pytest.mark.parametrize("form,general", [(0, 0), (24, 0), (4, 0), (24, 1), (4, 1)])
def test_compact_is_idempotent_and_sorted_at_full_size(form, general):
    """Size-independent properties at 4K: indices strictly increasing; the
    points gathered by those indices from a PARITY run equal the COMPACT
    points bit-for-bit; counts equal the number of finite PARITY points.  In every reprojection form
    (d2pc_set_reproject_form) and on both of its routes (specialised kinds / general kernel)."""
    q = d2pc.make_q(cx=1919.37, cy=1079.61, nx=3840, ny=2160)   # a fractional principal point: 2.4's column sum rounds
    frames = [synth_disparity(4, 20, 3840, 2160, "holes")]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as c, d2pc.Context(q=q) as p:
        for ctx in (c, p):
            ctx.set_reproject_form(form)
            ctx.set_test_hook("force_general_q", general)
        bc = _batch(c, frames, want_index=True)
        bp = _batch(p, frames, want_index=False)
        bc.launch()
        bp.launch()
        (cp, ci), = bc.results()
        (pp, _), = bp.results()
    assert np.all(np.diff(ci.astype(np.int64)) > 0)
    keep = np.isfinite(pp[:, :3]).all(axis=1)
    assert keep.sum() == len(cp)
    assert np.array_equal(pp[keep].view(np.uint32), cp.view(np.uint32))
    v, u = np.divmod(ci.astype(np.int64), 3840)
    roi = (v - 40) * 3760 + (u - 40)
    assert np.array_equal(roi, np.nonzero(keep)[0])


@pytest.mark.parametrize("algo", [1, 2, 3, 4])  # count/scan/scatter nodes | memset + single-pass kernel nodes | chunk launches
def test_launch_on_side_stream_and_graph_capture(algo):
    q = d2pc.make_q()
    frames = [synth_disparity(2, f, 640, 480, "holes") for f in range(6)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        b = _batch(ctx, frames, want_index=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            b.launch()
        side.synchronize()
        first = b.results()
        # capture the same call into a hipGraph and replay it
        b.points.fill_(0)
        b.counts.fill_(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            b.launch()
        replays = []
        for _ in range(3):  # EVERY replay must do the whole job: outputs are wiped in between (round 1 replayed
            b.points.fill_(0)  # twice over the same buffers, which hid that the runtime's memset node left stale
            b.index.fill_(0)   # tickets behind and made the second replay a no-op)
            b.counts.fill_(0)
            torch.cuda.synchronize()
            g.replay()
            replays.append(b.results())
            ctx.check_async_error()
    for second in replays:
        for (p1, i1), (p2, i2), fr in zip(first, second, frames):
            wp, wi = oracle.reproject_compact(fr, q, border=40)
            assert np.array_equal(i1, wi) and np.array_equal(i2, wi)
            assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32))
            assert_points_close(p1, wp, max_ulp=1)


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
def test_warm_up_then_capture_on_the_same_stream_without_a_reservation(algo):
    """The flow d2pc.h documents for captures: run the largest batch once, then capture -- on the SAME stream, without
    d2pc_reserve.  Advisor, round 4: since the buffers' completion events are recorded lazily the warm-up left the
    stream's buffer marked busy, the stream could not be asked (it was capturing) and the captured launch failed with
    D2PC_ERR_OUT_OF_MEMORY although the buffer was there.  (The other capture tests warm up on another stream.)"""
    q = d2pc.make_q()
    frames = [synth_disparity(2, 30 + f, 640, 480, "holes") for f in range(5)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        b = _batch(ctx, frames, want_index=True, reserve=False)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            b.launch()          # eager: allocates the state buffer and leaves it bound to s
            b.launch()          # (back to back: no event is recorded in between)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                b.launch()
        for _ in range(2):
            b.points.fill_(0)
            b.index.fill_(0)
            b.counts.fill_(0)
            torch.cuda.synchronize()
            g.replay()
            res = b.results()
            ctx.check_async_error()
            for (pts, idx), fr in zip(res, frames):
                wp, wi = oracle.reproject_compact(fr, q, border=40)
                assert np.array_equal(idx, wi)
                assert_points_close(pts, wp, max_ulp=1)
        # ... and the eager path still works next to the graph (on the same stream it gets a buffer of its own)
        with torch.cuda.stream(s):
            b.launch()
        res = b.results()
        for (pts, idx), fr in zip(res, frames):
            assert np.array_equal(idx, oracle.reproject_compact(fr, q, border=40)[1])


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
def test_two_streams_in_flight_from_one_context_do_not_share_compaction_state(algo):
    """Double buffering: COMPACT launches of ONE context enqueued on two streams overlap on the device.
    Each launch must own its tickets / partial counts / granules (round 1 shared one buffer: points came
    out silently wrong).  Both result sets are checked against the oracle after many overlapping rounds."""
    q = d2pc.make_q()
    fa = [synth_disparity(3, 100 + f, 1920, 1080, "holes") for f in range(6)]
    fb = [synth_disparity(3, 200 + f, 1920, 1080, "blocky") for f in range(6)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        ba, bb = _batch(ctx, fa, want_index=True), _batch(ctx, fb, want_index=True)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(12):  # no synchronisation in between: launches of the two streams interleave freely
            ba.launch(stream=sa)
            bb.launch(stream=sb)
        torch.cuda.synchronize()
        ctx.check_async_error()
        ra, rb = ba.results(), bb.results()
    for frames, res in ((fa, ra), (fb, rb)):
        for f, (pts, idx) in enumerate(res):
            wp, wi = oracle.reproject_compact(frames[f], q, border=40)
            assert np.array_equal(idx, wi), f"frame {f}"
            assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"two-stream frame {f}")


@pytest.mark.parametrize("algo", [1, 2, 4])
def test_graph_replay_survives_a_larger_eager_launch(algo):
    """A captured launch bakes its state pointer into the graph.  A later, LARGER eager launch needs more
    state: it must get a buffer of its own instead of freeing the graph's (round 1: use-after-free)."""
    q = d2pc.make_q()
    small = [synth_disparity(2, f, 640, 480, "holes") for f in range(4)]
    big = [synth_disparity(3, 300 + f, 1920, 1080, "holes") for f in range(8)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=algo, variant=variant_for(algo)) as ctx:
        bs = _batch(ctx, small, want_index=True)       # reserves state for the small batch only
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            bs.launch()
        g.replay()
        torch.cuda.synchronize()
        from disparity_to_point_cloud_amd.torch_api import DeviceBatch

        bb = DeviceBatch.__new__(DeviceBatch)          # a big batch WITHOUT the constructor's d2pc_reserve
        cfg = ctx.config()
        bb.ctx, bb.n_frames, bb.height, bb.width = ctx, len(big), 1080, 1920
        bb.roi_n = d2pc.roi_points(1920, 1080, cfg.border)
        bb.stride = (bb.roi_n + 15) // 16 * 16
        bb.device = torch.device("cuda:0")
        bb.disp = torch.from_numpy(np.stack(big)).cuda()
        bb.points = torch.empty((len(big), bb.stride, 4), dtype=torch.float32, device="cuda")
        bb.index = torch.empty((len(big), bb.stride), dtype=torch.int32, device="cuda")
        bb.counts = torch.zeros((len(big),), dtype=torch.int32, device="cuda")
        bb.launch()
        rbig = bb.results()
        bs.points.fill_(0)
        bs.counts.fill_(0)
        g.replay()                                     # the graph's buffer must still be alive and its own
        bb.launch()                                    # ... while the eager path keeps using the other one
        g.replay()
        rsmall = bs.results()
        ctx.check_async_error()
    for frames, res in ((small, rsmall), (big, rbig)):
        for f, (pts, idx) in enumerate(res):
            wp, wi = oracle.reproject_compact(frames[f], q, border=40)
            assert np.array_equal(idx, wi), f"frame {f}"
            assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"frame {f}")


def test_capture_without_a_reserved_state_buffer_fails_cleanly():
    """Nothing can be allocated during capture: without d2pc_reserve the call must fail with
    OUT_OF_MEMORY (and say what to do) instead of allocating or touching another launch's buffer."""
    q = d2pc.make_q()
    frames = [synth_disparity(2, f, 640, 480, "holes") for f in range(2)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        disp = torch.from_numpy(np.stack(frames)).cuda()
        n = d2pc.roi_points(640, 480, 40)
        pts = torch.empty((2, n, 4), dtype=torch.float32, device="cuda")
        cnt = torch.zeros((2,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        err = None
        with torch.cuda.graph(g):
            try:
                ctx.process_device(disp.data_ptr(), d2pc.DTYPE_F32, 1.0, 640, 480, 640 * 4, 640 * 480 * 4, 2,
                                   pts.data_ptr(), None, n, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
            except d2pc.D2pcError as e:
                err = e
        assert err is not None and "d2pc_reserve" in str(err)


def test_async_error_check_is_per_state_buffer():
    """A big single-pass launch on one stream followed by a small two-pass launch on another: the check must look at
    each buffer's OWN last algorithm (round 1 keyed on the context's last launch and could skip the big one's
    header, or read a stale flag through the small one)."""
    q = d2pc.make_q()
    big = [synth_disparity(3, 500 + f, 1920, 1080, "holes") for f in range(30)]   # 27k tiles: single pass
    small = [synth_disparity(2, 600 + f, 320, 240, "holes") for f in range(2)]    # two-pass
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        bb, bs = _batch(ctx, big, want_index=False), _batch(ctx, small, want_index=True)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(3):
            bb.launch(stream=sa)
            bs.launch(stream=sb)
        torch.cuda.synchronize()
        ctx.check_async_error()
        rs = bs.results()
        counts = bb.counts.cpu().numpy().view(np.uint32)
    assert not np.any(counts == 0xFFFFFFFF)
    for f, (pts, idx) in enumerate(rs):
        wp, wi = oracle.reproject_compact(small[f], q, border=40)
        assert np.array_equal(idx, wi)
        assert_points_close(pts, wp, max_ulp=1)
    assert int(counts[0]) == len(oracle.reproject_compact(big[0], q, border=40)[0])


def test_many_captures_never_starve_eager_launches_and_buffers_can_be_released():
    """Advisor, round 2: a captured buffer was never released, and after 8 captures every COMPACT call of the
    context failed.  Buffers owned by graphs now come on top of the 8 eager ones, and d2pc_release_graph_buffers
    hands them back once the graphs are gone."""
    q = d2pc.make_q()
    frames = [synth_disparity(2, f, 320, 240, "holes") for f in range(2)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        b = _batch(ctx, frames, want_index=True)
        graphs = []
        for _ in range(11):   # more captures than the pool used to hold
            ctx.reserve(320, 240, 2)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                b.launch()
            graphs.append(g)
        b.launch()            # an eager launch still gets a buffer
        eager = b.results()
        for g in graphs:      # every graph kept its own state
            b.points.fill_(0)
            b.counts.fill_(0)
            g.replay()
            rep = b.results()
            for (p1, i1), (p2, i2) in zip(eager, rep):
                assert np.array_equal(i1, i2) and np.array_equal(p1.view(np.uint32), p2.view(np.uint32))
        del graphs, g
        torch.cuda.synchronize()
        ctx.release_graph_buffers()
        b.launch()
        again = b.results()
        ctx.check_async_error()
    for f, (pts, idx) in enumerate(again):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40)
        assert np.array_equal(idx, wi)
        assert_points_close(pts, wp, max_ulp=1)


def test_compact_stats_count_the_single_pass_launches():
    """d2pc_compact_stats: production counters of the single pass (tiles served, failed polls, wait time), folded by
    the next launch's state clear; the two-pass form leaves them alone."""
    q = d2pc.make_q()
    frames = [synth_disparity(3, 700 + f, 1920, 1080, "holes") for f in range(8)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=2) as ctx:
        b = _batch(ctx, frames, want_index=True)
        tiles = 8 * -(-d2pc.roi_points(1920, 1080, 40) // 2048)
        st = ctx.compact_stats()
        assert st["launches"] == 0 and st["tiles"] == 0
        for _ in range(5):
            b.launch()
        torch.cuda.synchronize()
        st = ctx.compact_stats()
        assert st["launches"] == 5 and st["tiles"] == 5 * tiles and st["timeouts"] == 0 and st["twopass_fallbacks"] == 0
        assert st["failed_polls"] < 50 * st["tiles"]
        assert (st["wait_us"] > 0) == (st["failed_polls"] > 0)
        st2 = ctx.compact_stats()            # reading does not consume
        assert st2 == st
        ctx.compact_stats_reset()
        assert ctx.compact_stats()["launches"] == 0
        b.launch()
        torch.cuda.synchronize()
        assert ctx.compact_stats()["tiles"] == tiles
        res = b.results()
    wp, wi = oracle.reproject_compact(frames[3], q, border=40)
    assert np.array_equal(res[3][1], wi)
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=1) as ctx:
        b = _batch(ctx, frames[:2], want_index=False)
        b.launch()
        torch.cuda.synchronize()
        assert ctx.compact_stats()["launches"] == 0


@pytest.mark.parametrize("blocks_per_cu,unroll,nt", [(8, 4, 0), (0, 1, 0), (0, 1, 1), (0, 2, 1), (0, 4, 0), (3, 2, 1)])
def test_membench_kernels_fill_and_copy(blocks_per_cu, unroll, nt):
    """The calibration kernels bench.py times next to the reprojection, in every launch shape it uses (persistent
    blocks / one-shot blocks, plain / non-temporal stores): every byte written / copied, none beyond."""
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        ctx.set_tuning("membench_blocks_per_cu", blocks_per_cu)
        ctx.set_tuning("membench_unroll", unroll)
        ctx.set_tuning("membench_nt", nt)
        n = (1 << 20) + 48
        buf = torch.zeros(2 * n, dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        ctx.membench_fill(buf.data_ptr(), n, s)
        torch.cuda.synchronize()
        host = buf.cpu().numpy()
        want = np.tile(np.array([1.0, 2.0, 3.0, 1.0], dtype=np.float32).view(np.uint8), n // 16)
        assert np.array_equal(host[:n], want) and not host[n:].any()
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
        ctx.membench_copy(src.data_ptr(), buf.data_ptr() + n, n, s)
        torch.cuda.synchronize()
        assert torch.equal(buf[n:], src) and np.array_equal(buf[:n].cpu().numpy(), want)
        with pytest.raises(d2pc.D2pcError):
            ctx.membench_copy(buf.data_ptr(), buf.data_ptr() + 16, n, s)   # overlap
        with pytest.raises(d2pc.D2pcError):
            ctx.set_tuning("membench_unroll", 3)


def test_resident_one_launch_compaction_over_many_relaunches_and_sizes():
    """compact_algo 3 (k_compact_resident): one launch, epochs instead of a state clear.  Relaunched many times on one
    state buffer (every launch must ignore what the previous ones published), on several sizes up to the residency
    limit (beyond it the library falls back by itself); its launches are counted."""
    q = d2pc.make_q()
    for (w, h, n, kind) in ((752, 480, 1, "holes"), (1920, 1080, 1, "holes"), (1920, 1080, 1, "blocky"), (640, 480, 7, "holes"),
                            (2000, 1100, 1, "holes"), (96, 96, 3, "holes"), (3840, 2160, 1, "holes")):
        frames = [synth_disparity(3, 40 + f, w, h, kind) for f in range(n)]
        want = [oracle.reproject_compact(fr, q, border=40) for fr in frames]
        with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=3) as ctx:
            b = _batch(ctx, frames, want_index=True)
            for rep in range(8):
                b.points.fill_(0)
                b.counts.fill_(0)
                b.launch()
                res = b.results()
                ctx.check_async_error()
                for f, (pts, idx) in enumerate(res):
                    assert np.array_equal(idx, want[f][1]), f"{w}x{h} rep {rep} frame {f}"
                    assert_points_close(pts, want[f][0], max_ulp=1, what=f"{w}x{h} rep {rep} frame {f}")
            st = ctx.compact_stats()
            # resident in ordinary 2048-pixel tiles, or -- one or two 4K frames -- in blocks of 32 / 64 pixels per thread
            fits = any(n * -(-d2pc.roi_points(w, h, 40) // (256 * r)) <= 1024 for r in (8, 32, 64))
            assert st["timeouts"] == 0 and st["launches"] == (8 if fits else 0), (w, h, st)


@pytest.mark.parametrize("n,pair", [(1, 0), (2, 0), (2, 1)])
def test_c4_4k_compact_frames_take_one_launch_each(n, pair):
    """A camera delivers frames one at a time: one 4K frame in COMPACT mode used to take two launches and two reads of
    the input (3,819 tiles > the 1,024 resident blocks).  k_compact_resident_lean: 8,192 pixels per block, disparities in
    registers between count and scatter -- ONE launch per frame, default routing, indices and points against the oracle.
    TWO 4K frames in one call: two such launches back to back (round 5; one launch of 16,384-pixel blocks -- tuning
    "resident_pair" = 1, still there -- measured 20 % slower than the two on the driver's device)."""
    q = d2pc.make_q()
    frames = [synth_disparity(4, 30 + f, 3840, 2160, ["holes", "blocky"][f % 2]) for f in range(n)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, variant=variant_for(exp=bool(pair))) as ctx:
        if pair:   # (a closed experiment: the key exists in the experiment build only since round 6)
            ctx.set_tuning("resident_pair", pair)
        b = _batch(ctx, frames, want_index=True)
        ctx.compact_stats_reset()
        for _ in range(3):
            b.points.fill_(0)
            b.launch()
        res = b.results()
        ctx.check_async_error()
        st = ctx.compact_stats()
    launches, r = (3, 64) if pair else (3 * n, 32)
    assert st["launches"] == launches and st["timeouts"] == 0
    assert st["tiles"] == 3 * n * -(-d2pc.roi_points(3840, 2160, 40) // (256 * r))
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40)
        assert len(pts) == len(wp), f"frame {f} count"
        assert np.array_equal(idx, wi), f"frame {f} indices"
        assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"4K compact frame {f}")


def test_a_give_up_in_the_first_frame_of_a_two_frame_call_is_reported():
    """Advisor, round 5 (medium): two 4K frames in one COMPACT call are TWO resident launches with consecutive epochs on one
    state buffer.  A hand-off give-up in frame 0's launch stores epoch E in the header while the buffer remembered only E + 1:
    d2pc_check_async_error answered OK over an incomplete cloud.  The test hook gives frame 0's launch a wait budget of zero
    ticks (its waves give up at the eighth failed look), frame 1's the ordinary one: counts[0] must read 0xFFFFFFFF, frame 1
    must be whole, and the check must fail.  Afterwards the same context, hook off, serves the call again cleanly."""
    q = d2pc.make_q()
    frames = [synth_disparity(4, 60 + f, 3840, 2160, "holes") for f in range(2)]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        b = _batch(ctx, frames, want_index=True)
        b.launch()
        torch.cuda.synchronize()
        ctx.check_async_error()
        ctx.compact_stats_reset()
        ctx.set_test_hook("handoff_spin_ticks_first", 0)
        gave_up = False
        for _ in range(6):   # (a launch in which no wave had to look eight times cannot give up: try again)
            b.launch()
            torch.cuda.synchronize()
            if ctx.compact_stats()["timeouts"] >= 1:
                gave_up = True
                break
        if not gave_up:
            pytest.skip("no wave of frame 0 waited long enough to give up in six launches")
        counts = b.counts.cpu().numpy().view(np.uint32)
        assert counts[0] == 0xFFFFFFFF, counts
        wp1, wi1 = oracle.reproject_compact(frames[1], q, border=40)
        assert counts[1] == len(wi1)                                  # frame 1's launch ran with the ordinary budget
        with pytest.raises(d2pc.D2pcError) as ei:
            ctx.check_async_error()
        assert ei.value.status == 9, ei.value                         # D2PC_ERR_INTERNAL
        ctx.set_test_hook("handoff_spin_ticks_first", -1)
        b.launch()
        res = b.results()
        ctx.check_async_error()                                       # the stale flag of the broken call is not this call's
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=40)
        assert np.array_equal(idx, wi), f"frame {f}"
        assert_points_close(pts, wp, max_ulp=1, rel=1e-5, what=f"frame {f} after the broken call")


@pytest.mark.parametrize("rpxt", [32, 64])
@pytest.mark.parametrize("case", ["f32", "f32_odd", "u8", "u16", "general", "sliver", "cv24", "min_disparity"])
def test_resident_lean_blocks_on_ragged_frames(rpxt, case):
    """The register-resident blocks forced (tuning resident_pxt) onto small ragged frames: last blocks reaching past the
    ROI, rows wrapping inside a wave's run, 8- and 16-bit input, a general Q (the real arithmetic counts), a Q whose W
    lands in the sliver (ditto, per wave), OpenCV 2.4's form, a disparity floor."""
    w, h, border, n = (1001, 333, 3, 3) if case == "f32_odd" else (1000, 700, 7, 2)
    q = d2pc.make_q(cx=w / 2 - 0.3, cy=h / 2 + 0.4, nx=w, ny=h)
    scale, form, dmin = 1.0, oracle.FORM_CV24, -np.inf
    rng = np.random.default_rng(rpxt + len(case))
    if case in ("u8", "u16"):
        hi, dt = (256, np.uint8) if case == "u8" else (65536, np.uint16)
        frames = [rng.integers(0, hi, size=(h, w)).astype(dt) for _ in range(n)]
        for fr in frames:
            fr[rng.random((h, w)) < 0.3] = 0
        scale = 0.125 if case == "u8" else 1.0 / 64
    else:
        frames = [synth_disparity(3, 50 + f, w, h, ["holes", "blocky", "uniform"][f % 3]) for f in range(n)]
        frames[0][h // 2, 5:11] = [np.nan, np.inf, -np.inf, -1.0, 3.4028235e38, 1e-45]
    if case == "general":
        q = rng.uniform(-1, 1, 16)
        q[12:16] = [2e-4, 1e-4, 0.03, 0.7]
        form = oracle.FORM_CV4
    if case == "sliver":
        q[14] = 1e-36
    if case == "min_disparity":
        dmin = 40.0
    ulp = 0 if case in ("general", "cv24") else 1
    with d2pc.Context(q=q, border=border, mode=d2pc.MODE_COMPACT, compact_algo=3, min_disparity=dmin) as ctx:
        ctx.set_tuning("resident_pxt", rpxt)
        if case == "cv24":
            ctx.set_reproject_form(d2pc.FORM_CV24)
        b = _batch(ctx, frames, want_index=True)
        ctx.compact_stats_reset()
        for _ in range(2):
            b.points.fill_(0)
            b.launch(scale=scale)
        res = b.results()
        ctx.check_async_error()
        assert ctx.compact_stats()["launches"] == 2
    for f, (pts, idx) in enumerate(res):
        wp, wi = oracle.reproject_compact(frames[f], q, border=border, scale=scale, form=form, min_disparity=dmin)
        assert len(pts) == len(wp), f"frame {f} count"
        assert np.array_equal(idx, wi), f"frame {f}"
        assert_points_close(pts, wp, max_ulp=ulp, rel=1e-5, what=f"frame {f}")


def test_resident_compaction_alternates_with_the_other_forms_on_one_state_buffer():
    q = d2pc.make_q()
    frames = [synth_disparity(3, 90 + f, 1280, 720, "holes") for f in range(2)]
    want = [oracle.reproject_compact(fr, q, border=40) for fr in frames]
    with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT) as ctx:
        b = _batch(ctx, frames, want_index=True)
        L = d2pc.load_library()
        for rep in range(9):
            ctx._check(L.d2pc_set_tuning(ctx.handle, b"spin_timeout_ms", 2000))
            cfg_algo = (3, 1, 2)[rep % 3]
            # the context's algorithm is fixed at creation: drive the choice through a second context sharing nothing,
            # except every third launch, which goes through THIS context's default (resident for this size)
            if cfg_algo == 3:
                b.points.fill_(0)
                b.counts.fill_(0)
                b.launch()
                res = b.results()
                ctx.check_async_error()
            else:
                with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=cfg_algo) as c2:
                    b2 = _batch(c2, frames, want_index=True)
                    b2.launch()
                    res = b2.results()
                    c2.check_async_error()
            for f, (pts, idx) in enumerate(res):
                assert np.array_equal(idx, want[f][1]), f"rep {rep} frame {f}"
                assert_points_close(pts, want[f][0], max_ulp=1, what=f"rep {rep} frame {f}")
        st = ctx.compact_stats()
        assert st["launches"] == 3 and st["timeouts"] == 0
