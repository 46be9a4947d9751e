"""GPU parity: image pyramid + pyramidal LK + FeatureTracker masks vs the CPU oracle.
Integer/byte results (pyramid bytes, status, masks) must be bit-exact; the
tracked positions and errors are compared bit-exactly too, because both sides
accumulate the integer products exactly (see oracle/oracle_klt.c header)."""
import numpy as np
import pytest

from util import grid_points, image_pair, move_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w", [(376, 1241), (480, 752), (251, 333), (48, 64), (33, 47)])
def test_pyramid_bit_exact(ctx, oracle, h, w):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ctx.set_image(0, img)
    ref = oracle.build_pyramid(img, win=3, max_level=6)
    for l, r in enumerate(ref):
        if min(r.shape) < 2 and l > 0:
            break
        g = ctx.get_level(0, l)
        assert g.shape == r.shape
        assert np.array_equal(g, r), f"level {l}"


def _cmp_lk(ctx, vo, oracle, img0, img1, pts0, win, max_level, flags=0, prior=None, max_iter=30,
            eps=0.01, min_eig=1e-4):
    ft = vo.FeatureTracker(ctx)
    ctx.set_image(0, img0)
    ctx.set_image(1, img1)
    lv, p1, st, err = ft.calcOpticalFlowPyrLK(0, 1, pts0, prior, win, max_level, flags, max_iter, eps, min_eig)
    lv_r, p1_r, st_r, err_r = oracle.calc_optical_flow_pyr_lk(img0, img1, pts0, prior, win, max_level, flags,
                                                              max_iter, eps, min_eig)
    assert lv == lv_r
    assert np.array_equal(st, st_r)
    assert np.array_equal(p1.view(np.uint32), p1_r.view(np.uint32)), np.abs(p1 - p1_r).max()
    assert np.array_equal(err.view(np.uint32), err_r.view(np.uint32))
    return p1, st, err


@pytest.mark.parametrize("win", [21, 15, 7, 9, 11, 13, 17, 19, 23, 25, 31])
def test_lk_parity_windows(ctx, vo, oracle, win):
    img0, img1 = image_pair(240, 320, seed=win, dx=2.3, dy=-1.6, scale=1.01, angle=0.004)
    pts0 = grid_points(240, 320, step=23, margin=6)
    _cmp_lk(ctx, vo, oracle, img0, img1, pts0, win, 3)


def test_lk_recovers_motion(ctx, vo, oracle):
    motion = dict(dx=5.2, dy=-3.4, scale=1.0, angle=0.0)
    img0, img1 = image_pair(300, 400, seed=5, **motion)
    pts0 = grid_points(300, 400, step=19, margin=30)
    p1, st, err = _cmp_lk(ctx, vo, oracle, img0, img1, pts0, 21, 3)
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion)
    ok = st.astype(bool)
    assert ok.mean() > 0.95
    assert np.abs(p1[ok] - gt[ok]).max() < 0.15


def test_lk_identical_images_zero_flow(ctx, vo, oracle):
    img0, _ = image_pair(200, 260, seed=9)
    pts0 = grid_points(200, 260, step=21, margin=20)
    p1, st, err = _cmp_lk(ctx, vo, oracle, img0, img0, pts0, 21, 3)
    assert st.all() and np.abs(p1 - pts0).max() < 1e-3 and err.max() == 0.0


def test_lk_initial_flow_borders_and_flat(ctx, vo, oracle):
    motion = dict(dx=-7.5, dy=4.25, scale=0.985, angle=-0.01)
    img0, img1 = image_pair(376, 620, seed=13, **motion)
    img0 = img0.copy(); img1 = img1.copy()
    img0[100:180, 200:330] = 90  # textureless block -> minEig failures
    img1[100:180, 200:330] = 90
    rng = np.random.default_rng(3)
    inside = grid_points(376, 620, step=15, margin=2)
    edge = np.array([[0.2, 0.3], [619.5, 375.2], [-3.0, 50.0], [700.0, 100.0], [310.0, -5.0], [300.0, 420.0],
                     [1.0, 375.0], [618.9, 0.4], [-25.0, -25.0], [10.5, 10.5]], np.float32)
    pts0 = np.concatenate([inside, edge]).astype(np.float32)
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion)
    prior = (gt + rng.normal(0, 1.5, gt.shape)).astype(np.float32)
    prior[::17] += 40.0  # some bad priors
    for flags, pr, me, thr in [(4, prior, 0.0, 0), (0, None, 1e-4, 1e-4), (4, prior, 1e-4, 1e-4)]:
        for max_level in (0, 1, 4, 6):
            _cmp_lk(ctx, vo, oracle, img0, img1, pts0, 21, max_level, flags, pr, 0 if flags else 30,
                    0.0 if flags else 0.01, thr)


def test_lk_kitti_shape_many_points(ctx, vo, oracle):
    motion = dict(dx=3.0, dy=0.5, scale=1.03, angle=0.002)
    img0, img1 = image_pair(376, 1241, seed=21, **motion)
    pts0 = grid_points(376, 1241, step=16, margin=8)
    assert pts0.shape[0] > 1500
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion).astype(np.float32)
    _cmp_lk(ctx, vo, oracle, img0, img1, pts0, 21, 6, 4, gt + 0.7, 0, 0.0, 0.0)


def test_feature_tracker_wrappers(ctx, vo, oracle):
    motion = dict(dx=4.0, dy=-2.0, scale=1.02, angle=0.006)
    img0, img1 = image_pair(300, 420, seed=31, **motion)
    pts0 = grid_points(300, 420, step=13, margin=1)
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion).astype(np.float32)
    rng = np.random.default_rng(8)
    prior = (gt + rng.normal(0, 1.0, gt.shape)).astype(np.float32)
    m_in = rng.random(pts0.shape[0]) > 0.1  # pre-set mask entries are kept (in/out semantics)
    ft = vo.FeatureTracker(ctx)
    ctx.set_image(0, img0)
    ctx.set_image(1, img1)
    win, lvl, te, tb = 21, 4, 12.0, 0.5

    p, m = ft.track(0, 1, pts0, win, lvl, te, m_in)
    rc, pr, mr = oracle.track(img0, img1, pts0, win, lvl, te, m_in)
    assert np.array_equal(m, mr) and np.array_equal(p.view(np.uint32), pr.view(np.uint32))

    p, m = ft.trackWithPrior(0, 1, pts0, win, lvl, te, prior, m_in)
    rc, pr, mr = oracle.track_with_prior(img0, img1, pts0, prior, win, lvl, te, m_in)
    assert np.array_equal(m, mr) and np.array_equal(p.view(np.uint32), pr.view(np.uint32))
    assert m.sum() > 0.5 * m_in.sum()

    p, m = ft.trackBidirection(0, 1, pts0, win, lvl, te, tb, m_in)
    rc, pr, mr = oracle.track_bidirection(img0, img1, pts0, win, lvl, te, tb, m_in)
    assert np.array_equal(m, mr) and np.array_equal(p.view(np.uint32), pr.view(np.uint32))

    p, m = ft.trackBidirectionWithPrior(0, 1, pts0, win, lvl, te, tb, prior, m_in)
    rc, pr, mr = oracle.track_bidirection_with_prior(img0, img1, pts0, prior, win, lvl, te, tb, m_in)
    assert np.array_equal(m, mr) and np.array_equal(p.view(np.uint32), pr.view(np.uint32))


def test_track_empty_and_capacity(ctx, vo):
    ft = vo.FeatureTracker(ctx)
    img0, img1 = image_pair(100, 120, seed=1)
    ctx.set_image(0, img0)
    ctx.set_image(1, img1)
    p, m = ft.track(0, 1, np.zeros((0, 2), np.float32), 21, 3, 10.0)
    assert p.shape == (0, 2) and m.shape == (0,)
    with pytest.raises(vo.VoError):
        ft.track(0, 1, np.zeros((ctx.cfg.max_points + 1, 2), np.float32), 21, 3, 10.0)
    with pytest.raises(vo.VoError):  # reference: throw on pts_track.size() != pts0.size()
        ft.trackWithPrior(0, 1, np.zeros((5, 2), np.float32), 21, 3, 10.0, np.zeros((4, 2), np.float32))
