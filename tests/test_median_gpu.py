"""GPU tests of the device median (cv::medianBlur(...,11), cpp:55-57) and of
the fused callback body d2pc_process_mono8 (cpp:55-85) against the oracle."""
import numpy as np
import pytest

import disparity_to_point_cloud_amd as d2pc
import oracle
from helpers import assert_points_close

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _median_gpu(ctx, imgs, k):
    src = torch.from_numpy(np.stack(imgs)).cuda()
    n, h, w = src.shape
    dst = torch.full_like(src, 77)
    ctx.median_device(src.data_ptr(), w, h, w, w * h, n, dst.data_ptr(), w, w * h, k,
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dst.cpu().numpy()


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
@pytest.mark.parametrize("w,h", [(752, 480), (97, 131), (16, 64), (17, 65), (5, 3), (1, 1), (1, 40), (40, 1), (33, 7)])
def test_median_matches_oracle(k, w, h):
    rng = np.random.default_rng(w * 1000 + h + k)
    imgs = [rng.integers(0, 256, size=(h, w)).astype(np.uint8),
            (rng.integers(0, 4, size=(h, w)) * 85).astype(np.uint8)]  # many ties
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        got = _median_gpu(ctx, imgs, k)
    for g, img in zip(got, imgs):
        assert np.array_equal(g, oracle.median_u8(img, k))


def test_median_4k_batch_and_constant_images():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, size=(2160, 3840)).astype(np.uint8)
    b = np.full((2160, 3840), 200, dtype=np.uint8)
    b[1000:1100, 2000:2100] = 0
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        got = _median_gpu(ctx, [a, b], 11)
    assert np.array_equal(got[0], oracle.median_u8(a, 11))
    assert np.array_equal(got[1], oracle.median_u8(b, 11))


def test_median_rejects_bad_arguments():
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        x = torch.zeros((8, 8), dtype=torch.uint8, device="cuda")
        y = torch.zeros_like(x)
        for k in (0, 1, 2, 4, 13):
            with pytest.raises(d2pc.D2pcError) as e:
                ctx.median_device(x.data_ptr(), 8, 8, 8, 64, 1, y.data_ptr(), 8, 64, k)
            assert e.value.status == 1
        with pytest.raises(d2pc.D2pcError):
            ctx.median_device(x.data_ptr(), 8, 8, 8, 64, 1, x.data_ptr(), 8, 64, 3)  # in place
        with pytest.raises(d2pc.D2pcError):
            ctx.median_device(x.data_ptr(), 8, 8, 4, 64, 1, y.data_ptr(), 8, 64, 3)  # stride < width


@pytest.mark.parametrize("mode", [d2pc.MODE_PARITY, d2pc.MODE_COMPACT])
def test_process_mono8_is_the_callback_body(mode):
    """median 11 -> x1/8 -> reproject + pack, native 752x480 geometry."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, size=(480, 752)).astype(np.uint8)
    img[100:180, 300:420] = 0  # an invalid region that survives the median
    q = d2pc.make_q()
    med = oracle.median_u8(img, 11)
    with d2pc.Context(q=q, mode=mode) as ctx:
        got, idx = ctx.process_mono8(img, median_ksize=11, scale=0.125, want_index=True)
        raw, _ = ctx.process_mono8(img, median_ksize=0, scale=0.125, want_index=True)
    if mode == d2pc.MODE_PARITY:
        want = oracle.reproject(med, q, border=40, scale=0.125)
        assert len(got) == 268800
    else:
        want, wi = oracle.reproject_compact(med, q, border=40, scale=0.125)
        assert np.array_equal(idx, wi)
    assert_points_close(got, want, max_ulp=1, rel=1e-5, what="mono8 callback")
    # ksize 0 = no median
    want_raw = (oracle.reproject(img, q, border=40, scale=0.125) if mode == d2pc.MODE_PARITY
                else oracle.reproject_compact(img, q, border=40, scale=0.125)[0])
    assert_points_close(raw, want_raw, max_ulp=1)
