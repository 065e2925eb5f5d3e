"""GPU tests of d2pc_fuse_device: the fusion rule +
combined confidence + 3x3 median + crop of the reference's
src/depth_map_fusion.cpp:113-130, bit-exact against the oracle."""
import numpy as np
import pytest

import disparity_to_point_cloud_amd as d2pc
import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from disparity_to_point_cloud_amd.torch_api import fuse_planes  # noqa: E402


def _planes(rng, h, w, kind="uniform"):
    if kind == "uniform":
        return [rng.integers(0, 256, size=(h, w)).astype(np.uint8) for _ in range(6)]
    # reference-like: smooth-ish depths with black holes, scores near the thresholds 100/125, depths near 230
    d1 = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    d2 = np.clip(d1.astype(np.int32) + rng.integers(-40, 41, size=(h, w)), 0, 255).astype(np.uint8)
    d1[rng.random((h, w)) < 0.1] = 0
    d2[rng.random((h, w)) < 0.1] = 0
    s1 = rng.choice(np.array([0, 1, 19, 20, 49, 50, 99, 100, 101, 124, 125, 126, 255], dtype=np.uint8), size=(h, w))
    s2 = rng.choice(np.array([0, 1, 19, 20, 49, 50, 99, 100, 101, 124, 125, 126, 255], dtype=np.uint8), size=(h, w))
    g1 = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    g2 = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    return [d1, d2, s1, s2, g1, g2]


def _gpu(ctx, planes, **kw):
    dev = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes]
    fused, comb = fuse_planes(ctx, dev, **kw)
    torch.cuda.synchronize()
    return fused.cpu().numpy(), None if comb is None else comb.cpu().numpy()


@pytest.mark.parametrize("rule", range(9))
def test_rule_table_exhaustive_over_distances(rule):
    """Every (dist1, dist2) pair x the scores around every threshold, no median/crop effects:
    a 3x3-constant image makes the median the identity."""
    scores = np.array([0, 1, 2, 19, 20, 21, 49, 50, 99, 100, 101, 124, 125, 126, 229, 230, 255], dtype=np.uint8)
    d = np.arange(256, dtype=np.uint8)
    d1, d2 = np.meshgrid(d, d, indexing="ij")                      # 256 x 256 pairs
    d1 = np.repeat(np.repeat(d1, 3, axis=0), 3, axis=1)            # 768 x 768, constant 3x3 cells
    d2 = np.repeat(np.repeat(d2, 3, axis=0), 3, axis=1)
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        for s1 in scores:
            for s2 in scores[::2] if rule != d2pc.FUSE_GRAD_FILTER else scores:
                planes = [d1, d2, np.full_like(d1, s1), np.full_like(d1, s2), d1, d2]
                got, _ = _gpu(ctx, planes, rule=rule, crop=(0, 0, 0, 0), want_combined=False)
                centre = got[1::3, 1::3]                           # centre of each cell: median of 9 equal values
                want = np.array([[oracle.fuse_pixel(rule, int(a), int(b), int(s1), int(s2)) & 0xFF
                                  for b in (0, 1, 4, 5, 100, 229, 230, 255)] for a in range(256)], dtype=np.uint8)
                assert np.array_equal(centre[:, [0, 1, 4, 5, 100, 229, 230, 255]], want), (rule, s1, s2)
        # and one full 65536-pair table per rule through the oracle's own image path
        planes = [d1, d2, np.full_like(d1, 110), np.full_like(d1, 110), d1, d2]
        got, _ = _gpu(ctx, planes, rule=rule, crop=(0, 0, 0, 0), want_combined=False)
        want, _ = oracle.fuse(planes, rule=rule, crop=(0, 0, 0, 0), want_combined=False)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h", [(465, 465), (480, 480), (248, 8), (249, 9), (247, 7), (496, 16), (497, 17), (1, 1),
                                 (2, 3), (3, 2), (4, 4), (5, 1), (1, 5), (61, 83), (752, 480), (1000, 50)])
@pytest.mark.parametrize("kind", ["uniform", "thresholds"])
def test_fuse_matches_oracle_no_crop(w, h, kind):
    rng = np.random.default_rng(w * 7919 + h + len(kind))
    planes = _planes(rng, h, w, kind)
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        fused, comb = _gpu(ctx, planes, crop=(0, 0, 0, 0))
    want_f, want_c = oracle.fuse(planes, crop=(0, 0, 0, 0))
    assert np.array_equal(comb, want_c)
    assert np.array_equal(fused, want_f)


@pytest.mark.parametrize("crop", [(0, 40, 30, 10), (1, 2, 3, 4), (40, 0, 10, 30), (5, 5, 0, 0), (0, 0, 7, 9),
                                  (100, 100, 0, 0), (0, 465, 0, 0), (0, 0, 465, 0), (232, 233, 232, 233)])
def test_reference_crop_and_others(crop):
    rng = np.random.default_rng(sum(crop))
    planes = _planes(rng, 465, 465, "thresholds")
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        fused, comb = _gpu(ctx, planes, crop=crop)
    want_f, want_c = oracle.fuse(planes, crop=crop)
    assert fused.shape == want_f.shape
    assert np.array_equal(fused, want_f)
    assert np.array_equal(comb, want_c)


def test_batched_views_with_pitches_and_aliasing():
    """Planes are crop-to-square VIEWS of larger images (odd byte offsets, pitch != width), batched,
    and score1 aliases grad1 as in the reference (cpp:77)."""
    rng = np.random.default_rng(5)
    F, H, W = 3, 480, 752
    big = [torch.from_numpy(rng.integers(0, 256, size=(F, H, W)).astype(np.uint8)).cuda() for _ in range(5)]
    x1, y1, n = d2pc.crop_to_square(W, H, -7, 15)
    x2, y2, n2 = d2pc.crop_to_square(W, H, 7, -15, 15)
    assert n == n2 == 465 and (x1, y1) == oracle.crop_to_square(W, H, -7, 15)[:2]
    v = lambda t, x, y: t[:, y:y + n, x:x + n]  # noqa: E731
    d1, s1 = v(big[0], x1, y1), v(big[2], x1, y1)
    d2_, s2 = v(big[1], x2, y2), v(big[3], x2, y2)
    g2 = v(big[4], x2, y2)
    planes = [d1, d2_, s1, s2, s1, g2]  # grad1 IS score1
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        fused, comb = fuse_planes(ctx, planes)
        torch.cuda.synchronize()
    for f in range(F):
        host = [np.ascontiguousarray(p[f].cpu().numpy()) for p in planes]
        want_f, want_c = oracle.fuse(host)
        assert np.array_equal(fused[f].cpu().numpy(), want_f)
        assert np.array_equal(comb[f].cpu().numpy(), want_c)


@pytest.mark.parametrize("rows", [0, 2, 4, 6, 16, 30, 32, 1024])
def test_any_strip_height_gives_the_same_image(rows):
    """The rolling window walks `fuse_rows` rows per wave (two per step, alternating register sets):
    odd heights, heights below one strip and strips of 4k+2 rows all have to land on the same bytes."""
    rng = np.random.default_rng(11)
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        ctx.set_tuning("fuse_rows", rows)
        for h, w in ((301, 500), (31, 249), (2, 9), (33, 4)):
            planes = _planes(rng, h, w, "thresholds")
            fused, comb = _gpu(ctx, planes, crop=(1, 2, 0, 1))
            want_f, want_c = oracle.fuse(planes, crop=(1, 2, 0, 1))
            assert np.array_equal(fused, want_f), (rows, h, w)
            assert np.array_equal(comb, want_c), (rows, h, w)
        with pytest.raises(d2pc.D2pcError):
            ctx.set_tuning("fuse_rows", 1)


def test_full_size_batch():
    rng = np.random.default_rng(12)
    one = [rng.integers(0, 256, size=(1, 2160, 2160)).astype(np.uint8) for _ in range(6)]
    planes = [np.concatenate([p, p[:, ::-1].copy(), p], axis=0) for p in one]   # 3 frames, the middle one flipped
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        fused, comb = _gpu(ctx, planes)
    for f in (0, 1):
        want_f, want_c = oracle.fuse([np.ascontiguousarray(p[f]) for p in planes])
        assert np.array_equal(fused[f], want_f)
        assert np.array_equal(comb[f], want_c)
    assert np.array_equal(fused[2], fused[0]) and np.array_equal(comb[2], comb[0])


def test_without_combined_grads_may_be_null():
    rng = np.random.default_rng(3)
    planes = _planes(rng, 100, 120)
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        dev = [torch.from_numpy(p).cuda() for p in planes[:4]] + [None, None]
        fused, comb = fuse_planes(ctx, dev, want_combined=False)
        torch.cuda.synchronize()
    assert comb is None
    assert np.array_equal(fused.cpu().numpy(), oracle.fuse(planes, want_combined=False)[0])


def test_rejects_bad_arguments():
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        x = [torch.zeros((64, 64), dtype=torch.uint8, device="cuda") for _ in range(6)]
        out = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
        comb = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")

        def desc(**kw):
            d = d2pc.fuse_desc_init()
            d.width = d.height = 64
            d.crop_left = d.crop_right = d.crop_top = d.crop_bottom = 0
            for i in range(6):
                d.planes[i], d.pitch[i] = x[i].data_ptr(), 64
            d.fused, d.fused_pitch = out.data_ptr(), 64
            d.combined, d.combined_pitch = comb.data_ptr(), 64
            for k, v in kw.items():
                setattr(d, k, v)
            return d

        ctx.fuse_device(desc())  # the base descriptor is fine
        for bad, status in ((dict(rule=9), 1), (dict(rule=-1), 1), (dict(width=0), 3), (dict(n_frames=0), 3),
                            (dict(crop_left=65), 3), (dict(crop_top=60, crop_bottom=5), 3), (dict(fused=None), 1),
                            (dict(fused_pitch=63), 3), (dict(combined_pitch=10), 3), (dict(struct_size=8), 1),
                            (dict(fused=x[2].data_ptr()), 1), (dict(combined=x[0].data_ptr()), 1),
                            (dict(combined=out.data_ptr()), 1)):
            with pytest.raises(d2pc.D2pcError) as e:
                ctx.fuse_device(desc(**bad))
            assert e.value.status == status, bad
        d = desc()
        d.planes[1] = None
        with pytest.raises(d2pc.D2pcError):
            ctx.fuse_device(d)
        d = desc()
        d.pitch[3] = 10
        with pytest.raises(d2pc.D2pcError):
            ctx.fuse_device(d)
        torch.cuda.synchronize()


@pytest.mark.parametrize("rows,cols", [(480, 752), (752, 480), (64, 64), (65, 63), (1, 1), (1, 130), (130, 1), (3, 5),
                                       (127, 129), (2160, 3840)])
def test_rotate_cw_matches_oracle(rows, cols):
    rng = np.random.default_rng(rows * 4099 + cols)
    imgs = rng.integers(0, 256, size=(2, rows, cols)).astype(np.uint8)
    src = torch.from_numpy(imgs).cuda()
    dst = torch.full((2, cols, rows), 7, dtype=torch.uint8, device="cuda")
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        ctx.rotate_cw_device(src.data_ptr(), cols, rows, cols, cols * rows, 2, dst.data_ptr(), rows, rows * cols,
                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        for f in range(2):
            assert np.array_equal(got[f], oracle.rotate_cw(imgs[f]))
        # pitched views in and out, single frame
        big = torch.from_numpy(rng.integers(0, 256, size=(rows + 3, cols + 5)).astype(np.uint8)).cuda()
        out = torch.zeros((cols + 2, rows + 9), dtype=torch.uint8, device="cuda")
        view = big[2:2 + rows, 1:1 + cols]
        ctx.rotate_cw_device(view.data_ptr(), cols, rows, big.stride(0), 0, 1, out[1:, 3:].data_ptr(), out.stride(0), 0)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert np.array_equal(o[1:1 + cols, 3:3 + rows], oracle.rotate_cw(np.ascontiguousarray(view.cpu().numpy())))
        assert not o[0].any() and not o[:, :3].any() and not o[:, 3 + rows:].any() and not o[1 + cols:].any()
        for bad in ((0, rows), (cols, 0)):
            with pytest.raises(d2pc.D2pcError):
                ctx.rotate_cw_device(src.data_ptr(), bad[0], bad[1], cols, 0, 1, dst.data_ptr(), rows, 0)
        with pytest.raises(d2pc.D2pcError):
            ctx.rotate_cw_device(src.data_ptr(), cols, rows, cols - 1 if cols > 1 else 0, 0, 1, dst.data_ptr(), rows, 0)
        with pytest.raises(d2pc.D2pcError):
            ctx.rotate_cw_device(src.data_ptr(), cols, rows, cols, 0, 1, src.data_ptr(), rows, 0)


def test_whole_fusion_front_end_on_device():
    """Camera 2's planes rotated and both cameras' planes cropped to square on the device, as DisparityCb1/2 and
    MatchingScoreCb1/2 do (src/depth_map_fusion.cpp:46-62, launch offsets -7/15), then fused."""
    rng = np.random.default_rng(77)
    H, W, ox, oy = 480, 752, -7, 15
    raw = [rng.integers(0, 256, size=(H, W)).astype(np.uint8) for _ in range(4)]   # disp1, disp2, score1, score2
    g = [rng.integers(0, 256, size=(465, 465)).astype(np.uint8) for _ in range(2)]  # the grads: host-filtered
    # oracle chain
    x1, y1, n = oracle.crop_to_square(W, H, ox, oy)
    x2, y2, n2 = oracle.crop_to_square(H, W, -ox, -oy, oy)
    assert n == n2 == 465
    d1 = raw[0][y1:y1 + n, x1:x1 + n]
    s1 = raw[2][y1:y1 + n, x1:x1 + n]
    d2_ = oracle.rotate_cw(raw[1])[y2:y2 + n, x2:x2 + n]
    s2 = oracle.rotate_cw(raw[3])[y2:y2 + n, x2:x2 + n]
    want_f, want_c = oracle.fuse([np.ascontiguousarray(p) for p in (d1, d2_, s1, s2, g[0], g[1])])
    # device chain
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        dev = [torch.from_numpy(r).cuda() for r in raw]
        rot = [torch.empty((W, H), dtype=torch.uint8, device="cuda") for _ in range(2)]
        for src, dst in ((dev[1], rot[0]), (dev[3], rot[1])):
            ctx.rotate_cw_device(src.data_ptr(), W, H, W, 0, 1, dst.data_ptr(), H, 0,
                                 torch.cuda.current_stream().cuda_stream)
        cx1, cy1, cn = d2pc.crop_to_square(W, H, ox, oy)
        cx2, cy2, _ = d2pc.crop_to_square(H, W, -ox, -oy, oy)
        planes = [dev[0][cy1:cy1 + cn, cx1:cx1 + cn], rot[0][cy2:cy2 + cn, cx2:cx2 + cn],
                  dev[2][cy1:cy1 + cn, cx1:cx1 + cn], rot[1][cy2:cy2 + cn, cx2:cx2 + cn],
                  torch.from_numpy(g[0]).cuda(), torch.from_numpy(g[1]).cuda()]
        fused, comb = fuse_planes(ctx, planes)
        torch.cuda.synchronize()
    assert np.array_equal(fused.cpu().numpy(), want_f)
    assert np.array_equal(comb.cpu().numpy(), want_c)


def test_gpu_matches_the_exact_rational_golden_vectors(golden_dir):
    """tests/golden/fusion_rules.npz (make_fusion_golden.py: gradFilter in exact rationals, numpy median)."""
    import os
    g = np.load(os.path.join(golden_dir, "fusion_rules.npz"))
    planes = [np.ascontiguousarray(g[f"image__plane{k}"]) for k in range(6)]
    d = np.arange(256, dtype=np.uint8)
    d1, d2 = np.meshgrid(d, d, indexing="ij")
    d1 = np.repeat(np.repeat(d1, 3, axis=0), 3, axis=1)  # constant 3x3 cells: the median is the identity at the centres
    d2 = np.repeat(np.repeat(d2, 3, axis=0), 3, axis=1)
    with d2pc.Context(q=d2pc.make_q()) as ctx:
        fused, comb = _gpu(ctx, planes)
        assert np.array_equal(fused, g["image__fused"])
        assert np.array_equal(comb, g["image__combined"])
        for s1, s2 in g["score_pairs"]:
            tab, _ = _gpu(ctx, [d1, d2, np.full_like(d1, s1), np.full_like(d1, s2), d1, d2], crop=(0, 0, 0, 0),
                          want_combined=False)
            assert np.array_equal(tab[1::3, 1::3], g[f"grad_filter__{s1}_{s2}"]), (s1, s2)
