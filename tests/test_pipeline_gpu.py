"""GPU tests of the pipelined host path (d2pc_pipeline_*): several frames in
flight on separate streams with pinned staging; results in submission order
and equal to the oracle, in staged and direct-host-write form."""
import numpy as np
import pytest

import disparity_to_point_cloud_amd as d2pc
import oracle
from helpers import assert_points_close, synth_disparity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("mode", [d2pc.MODE_PARITY, d2pc.MODE_COMPACT])
def test_frames_in_flight_come_back_in_order_and_correct(mode, direct):
    q = d2pc.make_q()
    sizes = [(640, 480), (752, 480), (1920, 1080), (640, 480), (97, 131), (1920, 1080), (752, 480)]
    frames = [synth_disparity(6, i, w, h, "holes") for i, (w, h) in enumerate(sizes)]
    with d2pc.Context(q=q, mode=mode) as ctx:
        ctx.pipeline_configure(depth=3, direct_host_write=direct)
        got, nxt = [], 0
        inflight = 0
        while len(got) < len(frames):
            while nxt < len(frames) and inflight < 3:
                ctx.pipeline_submit(frames[nxt], want_index=True, tag=1000 + nxt)
                nxt += 1
                inflight += 1
            p, i, tag, _ = ctx.pipeline_collect()
            inflight -= 1
            got.append((p, i, tag))
    for k, (p, i, tag) in enumerate(got):
        assert tag == 1000 + k, "frames must be collected in submission order"
        if mode == d2pc.MODE_PARITY:
            want = oracle.reproject(frames[k], q, border=40)
            assert_points_close(p, want, max_ulp=1, rel=1e-5, what=f"frame {k}")
        else:
            wp, wi = oracle.reproject_compact(frames[k], q, border=40)
            assert np.array_equal(i, wi)
            assert_points_close(p, wp, max_ulp=1, rel=1e-5, what=f"frame {k}")


def test_pipeline_mono8_with_device_median_and_zero_copy_view():
    q = d2pc.make_q()
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, size=(480, 752)).astype(np.uint8) for _ in range(4)]
    with d2pc.Context(q=q) as ctx:
        ctx.pipeline_configure(depth=2, direct_host_write=True)
        res = []
        ctx.pipeline_submit(imgs[0], scale=0.125, median_ksize=11, tag=0)
        for k in range(1, 4):
            ctx.pipeline_submit(imgs[k], scale=0.125, median_ksize=11, tag=k)
            p, _, tag, slot = ctx.pipeline_collect(copy=False)  # view of the pinned output
            res.append((p.copy(), tag))
            ctx.pipeline_release(slot)
        p, _, tag, slot = ctx.pipeline_collect(copy=False)
        res.append((p.copy(), tag))
        ctx.pipeline_release(slot)
    for k, (p, tag) in enumerate(res):
        assert tag == k
        want = oracle.reproject(oracle.median_u8(imgs[k], 11), q, border=40, scale=0.125)
        assert_points_close(p, want, max_ulp=1, what=f"mono8 frame {k}")


def test_pipeline_misuse():
    q = d2pc.make_q()
    fr = synth_disparity(6, 0, 200, 150, "uniform")
    with d2pc.Context(q=q) as ctx:
        with pytest.raises(d2pc.D2pcError):  # not configured
            ctx.pipeline_submit(fr)
        with pytest.raises(d2pc.D2pcError):
            ctx.pipeline_configure(depth=0)
        ctx.pipeline_configure(depth=2)
        with pytest.raises(d2pc.D2pcError):  # nothing submitted
            ctx.pipeline_collect()
        ctx.pipeline_submit(fr, tag=1)
        ctx.pipeline_submit(fr, tag=2)
        with pytest.raises(d2pc.D2pcError) as e:  # both slots busy
            ctx.pipeline_submit(fr, tag=3)
        assert e.value.status == 4
        with pytest.raises(d2pc.D2pcError):  # frames in flight
            ctx.pipeline_configure(depth=3)
        assert ctx.pipeline_collect()[2] == 1
        ctx.pipeline_submit(fr, tag=3)
        assert ctx.pipeline_collect()[2] == 2
        assert ctx.pipeline_collect()[2] == 3
        with pytest.raises(d2pc.D2pcError):
            ctx.pipeline_submit(fr.astype(np.float32), median_ksize=11)  # median needs u8
        # empty ROI
        ctx.pipeline_submit(np.ones((60, 60), np.float32), tag=9)
        p, _, tag, _ = ctx.pipeline_collect()
        assert p.shape == (0, 4) and tag == 9


def test_border_change_between_acquire_and_submit_is_refused():
    """The slot's buffers are sized when it is acquired; a smaller border afterwards would overflow them."""
    import ctypes
    from disparity_to_point_cloud_amd.capi import FrameDesc
    q = d2pc.make_q()
    fr = synth_disparity(6, 1, 320, 240, "uniform")
    with d2pc.Context(q=q) as ctx:
        ctx.pipeline_configure(depth=2)
        desc = FrameDesc(d2pc.DTYPE_F32, 1.0, 320, 240, 320 * 4, 0, 0, 5)
        host_in, slot = ctypes.c_void_p(), ctypes.c_int()
        assert ctx._L.d2pc_pipeline_acquire(ctx._h, ctypes.byref(desc), ctypes.byref(host_in), ctypes.byref(slot)) == 0
        ctx.set_border(0)                                   # ROI grows from 240x160 to 320x240 points
        assert ctx._L.d2pc_pipeline_submit(ctx._h, slot.value) == 1
        assert b"border changed" in ctx._L.d2pc_last_error(ctx._h)
        assert ctx._L.d2pc_pipeline_release(ctx._h, slot.value) == 0
        ctx.pipeline_submit(fr, tag=6)                      # a fresh acquire sizes for the new border
        p, _, tag, _ = ctx.pipeline_collect()
        assert tag == 6
        assert_points_close(p, oracle.reproject(fr, q, border=0), max_ulp=1)


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("mode", [d2pc.MODE_PARITY, d2pc.MODE_COMPACT])
def test_pipeline_mono16_frames_are_rescaled_on_the_device(mode, direct):
    """DTYPE_MONO16: cpp:50's cv_bridge rescale, the median and the reprojection all run on the slot's stream."""
    q = d2pc.make_q()
    rng = np.random.default_rng(3)
    imgs = [synth_disparity(1, 0, 640, 480, "mono16"), rng.integers(0, 65536, size=(133, 201)).astype(np.uint16),
            rng.integers(0, 65536, size=(480, 752)).astype(np.uint16)]
    ks = [11, 0, 11]
    with d2pc.Context(q=q, mode=mode) as ctx:
        ctx.pipeline_configure(depth=2, direct_host_write=direct)
        got = []
        for i, (img, k) in enumerate(zip(imgs, ks)):
            if i >= 2:
                got.append(ctx.pipeline_collect())
            ctx.pipeline_submit(img, scale=0.125, median_ksize=k, want_index=True, tag=i, mono16=True)
        while len(got) < len(imgs):
            got.append(ctx.pipeline_collect())
        with pytest.raises(d2pc.D2pcError):
            ctx.pipeline_submit(imgs[1], median_ksize=4, mono16=True)
    for i, (p, idx, tag, _) in enumerate(got):
        assert tag == i
        m8 = oracle.mono16_to_mono8(imgs[i])
        filt = oracle.median_u8(m8, ks[i]) if ks[i] else m8
        if mode == d2pc.MODE_PARITY:
            assert_points_close(p, oracle.reproject(filt, q, border=40, scale=0.125), max_ulp=1, what=f"frame {i}")
        else:
            wp, wi = oracle.reproject_compact(filt, q, border=40, scale=0.125)
            assert np.array_equal(idx, wi)
            assert_points_close(p, wp, max_ulp=1, what=f"frame {i}")
