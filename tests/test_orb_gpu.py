"""Keypoint detection of FeatureExtractor::extractORBwithBinning_fast on the device (vo_orb_detect,
vo_extract_orb_with_binning) against oracle/oracle_orb.c: pyramid levels, keypoint coordinates, octaves
bit-exact; Harris responses bit-exact (integer sums, one float expression); the bucketed pixels identical."""
import numpy as np
import pytest

from visual_odometry_ros_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _frame(seed, w=1241, h=376):
    stream = S.StereoStream(width=w, height=h, n_u=8, n_v=4, n_new=8, seed=seed)
    return stream.render_pair(stream.poses(1)[0])[0]


def _check(ctx, vo, oracle, img, thr, **orb):
    fe = vo.FeatureExtractor(ctx)
    fe.initParams(img.shape[1], img.shape[0], 20, 12, THRES_FAST=thr)
    for k, v in orb.items():
        setattr(fe.orb, k, v)
    ctx.set_image(0, img)
    xy, resp, octv = fe.detect(0)
    o = oracle.orb_detect(img, thr, nfeatures=fe.orb.nfeatures, scale_factor=fe.orb.scale_factor,
                          n_levels=fe.orb.n_levels, edge_threshold=fe.orb.edge_threshold, with_levels=True)
    # pyramid
    import ctypes as C
    for l in range(1, fe.orb.n_levels):
        w, h = C.c_int(), C.c_int()
        ctx.check(ctx.lib.vo_orb_get_level(ctx.handle, l, None, C.byref(w), C.byref(h)))
        buf = np.zeros((h.value, w.value), np.uint8)
        ctx.check(ctx.lib.vo_orb_get_level(ctx.handle, l, buf.ctypes.data_as(C.POINTER(C.c_uint8)), None, None))
        assert np.array_equal(buf, o["levels"][l]), f"level {l}"
    assert xy.shape[0] == o["xy"].shape[0]
    assert np.array_equal(octv, o["octave"])
    assert np.array_equal(xy.view(np.uint32), o["xy"].view(np.uint32))
    assert np.array_equal(resp.view(np.uint32), o["response"].view(np.uint32))
    return fe, xy, resp, octv


def test_orb_detect_matches_oracle_kitti_shape(ctx, vo, oracle):
    img = _frame(4)
    fe, xy, resp, octv = _check(ctx, vo, oracle, img, 15)
    assert xy.shape[0] > 1000 and octv.max() >= 4
    # detection + bucketing chained on the device == bucketing of the oracle's keypoints
    fe.suppressCenterBins()
    pts = fe.extractORBwithBinning_fast(0)
    o = oracle.orb_detect(img, 15)
    pts_o, idx_o = oracle.bucket_argmax(o["xy"], o["response"], fe.inv_u_step_, fe.inv_v_step_, 20, 12, fe.weight)
    assert fe.n_detected == o["xy"].shape[0]
    assert np.array_equal(pts.view(np.uint32), pts_o.view(np.uint32)) and 50 < pts.shape[0] <= 240
    # asynchronous variant on the side stream, twice in a row
    for _ in range(2):
        fe.enqueueExtract(0)
        assert np.array_equal(fe.resultExtract(), pts)
    # the same through the host-array entry point
    if xy.shape[0] <= 8000:  # (the shared context's per-point capacity)
        pts2, _ = fe.bucketKeypoints(xy, resp)
        assert np.array_equal(pts2, pts)


def test_orb_detect_quotas_and_thresholds(ctx, vo, oracle):
    """Small nfeatures: both retainBest cuts bite (score histogram cut, radix select on the response)."""
    img = _frame(9)
    fe, xy, resp, octv = _check(ctx, vo, oracle, img, 7, nfeatures=600)
    lw, lh, ls, nper = oracle.orb_level_sizes(img.shape[1], img.shape[0], 1.2, 8, 600)
    counts = np.bincount(octv, minlength=8)
    assert counts[0] >= nper[0] and counts[0] <= nper[0] + 8 and xy.shape[0] < 700
    _check(ctx, vo, oracle, img, 40, nfeatures=10000)       # high threshold: few corners, no cut
    _check(ctx, vo, oracle, img, 10, nfeatures=0)           # nothing may come out
    _check(ctx, vo, oracle, img, 12, n_levels=3, scale_factor=1.5, edge_threshold=16)


def test_orb_detect_other_shapes(vo, oracle):
    c = vo.Context(device=0, max_width=752, max_height=480, max_points=2048, n_slots=2, max_level=4)
    try:
        img = _frame(5, 752, 480)
        _check(c, vo, oracle, img, 20)
        flat = np.full((480, 752), 90, np.uint8)  # no corners at all
        fe = vo.FeatureExtractor(c)
        fe.initParams(752, 480, 20, 12, THRES_FAST=20)
        c.set_image(0, flat)
        xy, resp, octv = fe.detect(0)
        assert xy.shape[0] == 0 and fe.extractORBwithBinning_fast(0).shape[0] == 0
        rng = np.random.default_rng(0)  # white noise: a FAST corner almost everywhere -> capacity error, not a fault
        noise = rng.integers(0, 256, (480, 752), dtype=np.uint8)
        c.set_image(0, noise)
        try:
            xy, resp, octv = fe.detect(0, max_kp=200000)
            o = oracle.orb_detect(noise, 20, max_kp=200000)
            assert np.array_equal(xy, o["xy"]) and np.array_equal(resp, o["response"])
        except vo.VoError as e:
            assert e.code == -8
    finally:
        c.close()


def _table(ctx, vo, oracle, img, thr, nbu, nbv, check_oracle=True, **orb):
    """The per-bin candidate table of the closed step [10] (vo_new_point_candidates_enqueue) by the two tile kernels
    (orb_tile.hpp) and by the per-stage kernels (VO_DBG_STAGED_DETECT): both against the oracle's detection + arg-max per bin
    with every weight 1, and against each other."""
    fe = vo.FeatureExtractor(ctx)
    fe.initParams(img.shape[1], img.shape[0], nbu, nbv, THRES_FAST=thr)
    for k, v in orb.items():
        setattr(fe.orb, k, v)
    fe._bin_params = None
    ctx.set_image(0, img)
    out = []
    for staged in (0, 1, 0):  # (the tile kernels again behind the per-stage ones: they find the counters as those left them)
        ctx.debug_set(ctx.DBG_STAGED_DETECT, staged)
        try:
            fe.enqueueCandidates(0, staged)
            out.append(fe.getCandidates(staged))
        finally:
            ctx.debug_set(ctx.DBG_STAGED_DETECT, 0)
    fe.enqueueCandidates(0, 0)  # and once more: the tile kernels leave their counters and keys zeroed
    out.append(fe.getCandidates(0))
    for xy, has, nd in out[1:]:
        assert nd == out[0][2] and np.array_equal(has, out[0][1]) and np.array_equal(xy.view(np.uint32), out[0][0].view(np.uint32))
    if check_oracle:
        d = oracle.orb_detect(img, thr, nfeatures=fe.orb.nfeatures, scale_factor=fe.orb.scale_factor, n_levels=fe.orb.n_levels,
                              edge_threshold=fe.orb.edge_threshold, max_kp=400000)
        cand, _ = oracle.bucket_argmax(d["xy"], d["response"], fe.inv_u_step_, fe.inv_v_step_, nbu, nbv, np.ones(nbu * nbv, np.int32))
        xy, has, nd = out[0]
        assert nd == d["xy"].shape[0]
        assert int(has.sum()) == cand.shape[0] and np.array_equal(xy[has].view(np.uint32), cand.view(np.uint32))
        assert np.all(xy[~has] == 0)
    return out[0]


def test_candidate_table_by_tile_kernels_kitti_shape(ctx, vo, oracle):
    img = _frame(4)
    xy, has, nd = _table(ctx, vo, oracle, img, 15, 60, 25)
    assert nd > 5000 and has.sum() > 500
    _table(ctx, vo, oracle, _frame(9), 7, 60, 25, nfeatures=600)       # both retainBest cuts bite; 16 candidates per lane
    _table(ctx, vo, oracle, img, 40, 20, 12)                           # few corners, no cut
    _table(ctx, vo, oracle, img, 10, 20, 12, nfeatures=0)              # nothing may come out
    _table(ctx, vo, oracle, img, 12, 20, 12, n_levels=3, scale_factor=1.5, edge_threshold=16)
    _table(ctx, vo, oracle, img, 12, 20, 12, n_levels=12, scale_factor=1.1)


def test_candidate_table_by_tile_kernels_other_shapes(vo, oracle):
    c = vo.Context(device=0, max_width=752, max_height=480, max_points=2048, n_slots=2, max_level=4)
    try:
        _table(c, vo, oracle, _frame(5, 752, 480), 20, 40, 25)
        xy, has, nd = _table(c, vo, oracle, np.full((480, 752), 90, np.uint8), 20, 20, 12)
        assert nd == 0 and not has.any()
        rng = np.random.default_rng(0)  # white noise: levels of more than 16 384 candidates (histogram + radix select inside orb_finish_kernel)
        noise = rng.integers(0, 256, (480, 752), dtype=np.uint8)
        _table(c, vo, oracle, noise, 20, 40, 25)
        _table(c, vo, oracle, _frame(3, 620, 188), 15, 30, 12)
        _table(c, vo, oracle, _frame(3, 333, 251), 15, 10, 8)
    finally:
        c.close()


def test_candidate_table_by_tile_kernels_4k(ctx5, vo, oracle):
    """3840 x 2160 (BASELINE configs[4]): 5440 workgroups of the tile kernel, level lists of several ten thousand candidates."""
    st = S.StereoStream(width=3840, height=2160, K=(718.856 * 3.0, 718.856 * 3.0, 1920.0, 1080.0), n_u=100, n_v=80, seed=2)
    img = st.render_pair(st.poses(1)[0])[0]
    xy, has, nd = _table(ctx5, vo, oracle, img, 15, 100, 80)
    assert has.sum() > 1500
