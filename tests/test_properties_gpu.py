"""Size-independent properties of the device operators (hypothesis, a handful of examples each): what must hold
whatever the oracle says — per-point independence and permutation equivariance of the trackers, metric axioms of
the Hamming distance, sortedness / counts of the compaction, invariance of the GN solve under permutation of its
points, identity remap, shift equivariance of the detector, idempotence of a second BA solve's bookkeeping."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from util import grid_points, image_pair
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import compact_indices

pytestmark = pytest.mark.gpu
FEW = settings(max_examples=8, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@FEW
@given(seed=st.integers(0, 10_000), win=st.sampled_from([15, 21, 31]))
def test_klt_points_are_independent_and_identical_images_do_not_move(ctx, vo, seed, win):
    rng = np.random.default_rng(seed)
    img0, img1 = image_pair(200, 260, seed=seed % 97, dx=rng.uniform(-3, 3), dy=rng.uniform(-3, 3), scale=1.0, angle=0.0)
    pts = grid_points(200, 260, step=21, margin=8, jitter_seed=seed)
    ft = vo.FeatureTracker(ctx)
    ctx.set_image(0, img0)
    ctx.set_image(1, img1)
    lv, p1, status, err = ft.calcOpticalFlowPyrLK(0, 1, pts, None, win, 3, 0, 30, 0.01, 1e-4)
    perm = rng.permutation(pts.shape[0])
    lv2, q1, status2, err2 = ft.calcOpticalFlowPyrLK(0, 1, pts[perm], None, win, 3, 0, 30, 0.01, 1e-4)
    assert np.array_equal(q1, p1[perm]) and np.array_equal(status2, status[perm]) and np.array_equal(err2, err[perm])
    sub = perm[: max(1, perm.size // 3)]  # a subset gives the same answers for its members
    lv3, s1, status3, err3 = ft.calcOpticalFlowPyrLK(0, 1, pts[sub], None, win, 3, 0, 30, 0.01, 1e-4)
    assert np.array_equal(s1, p1[sub]) and np.array_equal(status3, status[sub])
    # identical images: nothing moves (up to the weight quantisation when (x - halfWin) + halfWin rounds, ~1e-4 px)
    ctx.set_image(1, img0)
    lv4, z1, status4, err4 = ft.calcOpticalFlowPyrLK(0, 1, pts, None, win, 3, 0, 30, 0.01, 1e-4)
    ok = status4.astype(bool)
    assert ok.mean() > 0.9 and np.abs(z1[ok] - pts[ok]).max() < 1e-3 and err4[ok].max() < 1e-2


@FEW
@given(seed=st.integers(0, 10_000), na=st.integers(1, 70), nb=st.integers(1, 70))
def test_hamming_is_a_metric(ctx, vo, seed, na, nb):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    fe = vo.FeatureExtractor(ctx)
    dab, dba = fe.descriptorDistance(a, b), fe.descriptorDistance(b, a)
    assert dab.shape == (na, nb) and np.array_equal(dab, dba.T) and dab.max() <= 256
    daa = fe.descriptorDistance(a, a)
    assert np.all(np.diag(daa) == 0) and np.array_equal(daa, daa.T)
    dbb = fe.descriptorDistance(b, b)
    assert np.all(dab[:, :, None] <= dab[:, None, :] + dbb[None, :, :])  # d(a,b) <= d(a,b') + d(b',b)


@FEW
@given(seed=st.integers(0, 10_000), n=st.integers(0, 3000), p=st.floats(0.0, 1.0))
def test_compaction_is_sorted_and_counts(ctx, seed, n, p):
    rng = np.random.default_rng(seed)
    m, alive, tracked = (rng.random(n) < q for q in (p, 0.9, 0.95))
    idx = compact_indices(ctx, m, alive, tracked)
    keep = m & alive & tracked
    assert idx.size == int(keep.sum()) and np.all(np.diff(idx) > 0) and np.array_equal(idx, np.flatnonzero(keep))


@FEW
@given(seed=st.integers(0, 10_000))
def test_gn_is_invariant_under_point_permutation(ctx, vo, seed):
    d = S.two_view_points(n=400, seed=seed % 50 + 1)
    me = vo.MotionEstimator(ctx, True, d["T_lr"])
    ok, T, inl, info = me.poseOnlyBundleAdjustment_Stereo(d["X"], d["pts_l"], d["pts_r"], d["K"], d["K"], d["T_lr"], 3.0,
                                                          np.eye(4, dtype=np.float32))
    perm = np.random.default_rng(seed).permutation(400)
    ok2, T2, inl2, info2 = me.poseOnlyBundleAdjustment_Stereo(d["X"][perm], d["pts_l"][perm], d["pts_r"][perm], d["K"],
                                                              d["K"], d["T_lr"], 3.0, np.eye(4, dtype=np.float32))
    assert ok and ok2 and np.array_equal(inl2, inl[perm])
    assert np.linalg.norm(T - T2) / np.linalg.norm(T) < 1e-5  # only the order of the float sums differs


@FEW
@given(seed=st.integers(0, 10_000), w=st.integers(40, 200), h=st.integers(40, 120))
def test_identity_remap_is_the_identity(vo, seed, w, h):
    c = vo.Context(device=0, max_width=200, max_height=120, max_points=64, n_slots=2, max_level=2)
    try:
        img = np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
        mu, mv = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
        import ctypes as C
        fp = C.POINTER(C.c_float)
        mu, mv = np.ascontiguousarray(mu), np.ascontiguousarray(mv)
        c.check(c.lib.vo_rectify_set_maps(c.handle, 0, mu.ctypes.data_as(fp), mv.ctypes.data_as(fp), w, h))
        c.set_image_rectified(0, img, 0)
        assert np.array_equal(c.get_level(0, 0), img)
        c.set_image(1, img)  # and the rest of the pyramid is the plain one
        assert np.array_equal(c.get_level(0, 1), c.get_level(1, 1))
    finally:
        c.close()


@FEW
@given(seed=st.integers(0, 1000), dx=st.integers(-6, 6), dy=st.integers(-6, 6))
def test_detector_level0_is_shift_equivariant(vo, seed, dx, dy):
    """Level-0 keypoints of an image and of its integer translate correspond (away from the borders; the coarser
    levels resample and need not)."""
    c = vo.Context(device=0, max_width=400, max_height=300, max_points=2048, n_slots=2, max_level=2)
    try:
        rng = np.random.default_rng(seed)
        big = np.clip(np.kron(rng.integers(0, 256, (40, 52)), np.ones((8, 8))) + rng.normal(0, 10, (320, 416)), 0, 255).astype(np.uint8)
        a = np.ascontiguousarray(big[10:310, 8:408])
        b = np.ascontiguousarray(big[10 + dy:310 + dy, 8 + dx:408 + dx])  # b(x, y) = a(x + dx, y + dy)
        fe = vo.FeatureExtractor(c)
        fe.initParams(400, 300, 10, 8, THRES_FAST=20)
        fe.orb.n_levels = 1
        c.set_image(0, a)
        xa, ra, _ = fe.detect(0)
        c.set_image(0, b)
        xb, rb, _ = fe.detect(0)
        sa = {(int(x), int(y)): r for (x, y), r in zip(xa, ra)}
        sb = {(int(x) + dx, int(y) + dy): r for (x, y), r in zip(xb, rb)}
        inner = lambda k: 31 + 6 <= k[0] < 400 - 31 - 6 and 31 + 6 <= k[1] < 300 - 31 - 6
        ka, kb = {k for k in sa if inner(k)}, {k for k in sb if inner(k)}
        assert len(ka) > 10 and ka == kb and all(sa[k] == sb[k] for k in ka)
    finally:
        c.close()


def test_sba_second_solve_continues_from_the_first(ctx, vo):
    """Two 5-iteration solves chained by hand equal one 10-iteration solve (no hidden state between iterations)."""
    from visual_odometry_ros_amd.api import SparseBundleAdjustmentSolver
    p = S.ba_window(n_kf=6, n_points=300, stereo=True, seed=21)
    sol = SparseBundleAdjustmentSolver(ctx, True)
    sol.setStereoCameras(p["K"], p["K"], p["T_lr"])
    sol.setHuberThreshold(0.5)
    lists = (p["opt_index"],), (p["obs_ptr"], p["obs_frame"], p["obs_right"], p["obs_px"])
    ok, T10, X10, e10 = sol.solveForFiniteIterations(10, p["T_jw"], *lists[0], p["X"], *lists[1])
    ok, T5, X5, e5 = sol.solveForFiniteIterations(5, p["T_jw"], *lists[0], p["X"], *lists[1])
    ok, T55, X55, e55 = sol.solveForFiniteIterations(5, T5, *lists[0], X5, *lists[1])
    assert np.array_equal(T55, T10) and np.array_equal(X55, X10) and np.array_equal(np.concatenate([e5, e55]), e10)
