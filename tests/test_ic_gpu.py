"""GPU parity: FeatureTracker::trackWithScale (scale-compensated IC refinement)."""
import numpy as np
import pytest

from util import grid_points, image_pair, move_points

pytestmark = pytest.mark.gpu


def _setup(ctx, h, w, seed, motion, margin, step=17, prior_noise=0.8):
    img0, img1 = image_pair(h, w, seed=seed, **motion)
    pts0 = grid_points(h, w, step=step, margin=margin)
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion)
    rng = np.random.default_rng(seed + 100)
    prior = (gt + rng.normal(0, prior_noise, gt.shape)).astype(np.float32)
    scale = np.full(pts0.shape[0], motion.get("scale", 1.0), np.float32)
    scale *= (1 + rng.normal(0, 0.01, scale.shape)).astype(np.float32)
    ctx.set_image(0, img0)
    ctx.set_image(1, img1)
    return img0, img1, pts0, prior, scale, gt


def test_ic_interior_points_match_reference_semantics(ctx, vo, oracle):
    """Points whose taps never leave the image: GPU == oracle(TREE) bit for bit, for both
    border modes (they coincide there), masks == oracle(SEQ) (reference order)."""
    motion = dict(dx=2.2, dy=-1.3, scale=1.04, angle=0.003)
    img0, img1, pts0, prior, scale, gt = _setup(ctx, 300, 420, 3, motion, margin=24)
    ft = vo.FeatureTracker(ctx)
    m_in = np.ones(pts0.shape[0], bool)
    m_in[::9] = False
    for strict in (False, True):
        p, m = ft.trackWithScale(0, 1, pts0, scale, prior, m_in, strict_border=strict)
        rc, pr, mr, tb = oracle.track_with_scale(img0, img1, pts0, scale, prior, m_in, oracle.IC_REFERENCE,
                                                 oracle.SUM_TREE)
        assert rc == 0 and not tb.any()
        assert np.array_equal(m, mr)
        assert np.array_equal(p.view(np.uint32), pr.view(np.uint32)), np.abs(p - pr).max()
    rc, ps, ms, _ = oracle.track_with_scale(img0, img1, pts0, scale, prior, m_in, oracle.IC_REFERENCE,
                                            oracle.SUM_SEQ)
    assert np.array_equal(m, ms)
    assert np.abs(p - ps).max() < 1e-3
    ok = m & m_in
    assert ok.mean() > 0.7
    assert np.abs(p[ok] - gt[ok]).max() < 1.0  # sanity only: IC with a noisy fixed scale is not sub-0.5px
    assert np.array_equal(p[~m_in], prior[~m_in]) and not m[~m_in].any()


@pytest.mark.parametrize("seed", [5, 6])
def test_ic_border_points_masked_semantics(ctx, vo, oracle, seed):
    """Points near / across the border: non-strict GPU == oracle MASKED mode."""
    motion = dict(dx=-3.1, dy=2.4, scale=0.97, angle=-0.004)
    img0, img1, pts0, prior, scale, gt = _setup(ctx, 240, 360, seed, motion, margin=2, step=11)
    extra = np.array([[-5.0, 20.0], [400.0, 100.0], [5.0, 5.0], [354.0, 236.0], [100.0, -30.0]], np.float32)
    pts0 = np.concatenate([pts0, extra])
    prior = np.concatenate([prior, extra + 1.0])
    scale = np.concatenate([scale, np.ones(5, np.float32)])
    ft = vo.FeatureTracker(ctx)
    p, m = ft.trackWithScale(0, 1, pts0, scale, prior, None, strict_border=False)
    rc, pr, mr, tb = oracle.track_with_scale(img0, img1, pts0, scale, prior, None, oracle.IC_MASKED,
                                             oracle.SUM_TREE)
    assert rc == 0 and tb.sum() > 20
    assert np.array_equal(m, mr)
    assert np.array_equal(p.view(np.uint32), pr.view(np.uint32))


@pytest.mark.parametrize("seed,order", [(7, "grid"), (8, "shuffled"), (9, "border_first")])
def test_ic_border_points_strict_reference_semantics(ctx, vo, oracle, seed, order):
    """strict_border: the reference's never-reset tap vectors, replayed on the GPU."""
    motion = dict(dx=1.7, dy=-2.6, scale=1.06, angle=0.002)
    img0, img1, pts0, prior, scale, gt = _setup(ctx, 220, 330, seed, motion, margin=3, step=10)
    rng = np.random.default_rng(seed)
    n = pts0.shape[0]
    if order == "shuffled":
        perm = rng.permutation(n)
    elif order == "border_first":
        d = np.minimum.reduce([pts0[:, 0], pts0[:, 1], 329 - pts0[:, 0], 219 - pts0[:, 1]])
        perm = np.argsort(d, kind="stable")
    else:
        perm = np.arange(n)
    pts0, prior, scale = pts0[perm], prior[perm], scale[perm]
    m_in = rng.random(n) > 0.08
    ft = vo.FeatureTracker(ctx)
    p, m = ft.trackWithScale(0, 1, pts0, scale, prior, m_in, strict_border=True)
    rc, pr, mr, tb = oracle.track_with_scale(img0, img1, pts0, scale, prior, m_in, oracle.IC_REFERENCE,
                                             oracle.SUM_TREE)
    assert rc == 0 and tb.sum() > 30
    assert np.array_equal(m, mr)
    assert np.array_equal(p.view(np.uint32), pr.view(np.uint32))
    # the sequential fallback of the parallel replay gives the same bits
    p2, m2 = ft.trackWithScale(0, 1, pts0, scale, prior, m_in, strict_border=2)
    assert np.array_equal(m2, mr)
    assert np.array_equal(p2.view(np.uint32), pr.view(np.uint32))
    # the two semantics genuinely differ on this input (otherwise the test proves nothing)
    rc, pm, mm, _ = oracle.track_with_scale(img0, img1, pts0, scale, prior, m_in, oracle.IC_MASKED,
                                            oracle.SUM_TREE)
    assert (mm != mr).any() or not np.array_equal(pm, pr)


def test_ic_flat_patch_rejected_and_size_mismatch(ctx, vo, oracle):
    img = np.full((120, 160), 77, np.uint8)
    ctx.set_image(0, img)
    ctx.set_image(1, img)
    ft = vo.FeatureTracker(ctx)
    pts0 = np.array([[60.5, 50.25], [80.0, 70.0]], np.float32)
    p, m = ft.trackWithScale(0, 1, pts0, np.ones(2, np.float32), pts0 + 0.5)
    assert not m.any() and np.array_equal(p, pts0 + 0.5)  # D < 1e-4: not updated
    with pytest.raises(vo.VoError):
        ft.trackWithScale(0, 1, pts0, np.ones(2, np.float32), pts0[:1])
