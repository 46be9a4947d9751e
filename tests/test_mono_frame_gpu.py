"""The steady-state MONO frame (vo_mono_frame_enqueue / _result; mono_vo.cpp:739-963) against the
oracle's vo_ref_mono_frame at BASELINE configs[2]'s shape (752x480, 1000 features, win 15, 5 levels):
stages, counts, pixels and patch scales bit-exact against the oracle in the kernels' summation
order, pose within 1e-4 (relative Frobenius) of the oracle in the reference's order."""
import numpy as np
import pytest

from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import MonoFramePipeline, make_mono_params

pytestmark = pytest.mark.gpu
MONO_K = (458.654, 457.296, 367.215, 248.375)


@pytest.fixture(scope="module")
def mono_ctx(vo):
    c = vo.Context(device=0, max_width=752, max_height=480, max_points=2048, n_slots=3, max_level=5)
    yield c
    c.close()


def _scene(seed, frame=1):
    stream = S.StereoStream(width=752, height=480, K=MONO_K, n_u=40, n_v=25, n_new=50, seed=seed, speed=0.25,
                            margin=6.0)
    poses = stream.poses(frame + 2)
    I0, _, _ = stream.render_pair(poses[frame])
    I1, _, _ = stream.render_pair(poses[frame + 1])
    ts = stream.track_set(frame, poses[frame], poses[frame + 1])
    return I0, I1, ts


def _world(ts, seed):
    """Put the previous camera at a non-trivial world pose: Xw, Tcw_prev, Tcw_prior, dT01_prior."""
    rng = np.random.default_rng(seed)
    w = rng.normal(size=3) * 0.2
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    Twc_prev = np.eye(4)
    Twc_prev[:3, :3], Twc_prev[:3, 3] = R, rng.normal(size=3) * 3.0
    dT01 = ts["dT_prior"].astype(np.float64)
    Xw = (ts["Xp"].astype(np.float64) @ R.T + Twc_prev[:3, 3]).astype(np.float32)
    Tcw_prev = np.linalg.inv(Twc_prev).astype(np.float32)
    Tcw_prior = np.linalg.inv(Twc_prev @ dT01).astype(np.float32)
    return Xw, Tcw_prev, Tcw_prior, dT01.astype(np.float32)


def _compare(g, o, o_seq=None, tol=1e-4):
    assert o["rc"] == 1
    assert np.array_equal(g["stage"], o["stage"])
    assert np.array_equal(g["scale"].view(np.uint32), o["scale"].view(np.uint32))
    assert np.array_equal(g["pts1"].view(np.uint32), o["pts1"].view(np.uint32))
    cg, co = g["counts"], o["counts"]
    for f in ("n_klt", "n_refine", "n_ba", "n_motion", "n_final", "gn_iterations", "need_five_point"):
        assert getattr(cg, f) == getattr(co, f), f
    assert np.linalg.norm(g["dT01"] - o["dT01"]) <= 1e-6 * np.linalg.norm(o["dT01"])
    if o_seq is not None:
        assert np.linalg.norm(g["dT01"] - o_seq["dT01"]) / np.linalg.norm(o_seq["dT01"]) < tol
        assert np.array_equal(g["stage"], o_seq["stage"])


@pytest.mark.parametrize("strict", [1, 0, 2, 3, 4])
def test_mono_frame_matches_oracle(mono_ctx, vo, oracle, strict):
    """strict: 0 masked taps, 1 the replay stream-ordered behind the frame kernel, 2 the sequential replay only, 3 the replay
    as a pool on its own stream next to the frame kernel (joined by the BA launch on the device), 4 = 3 when the previous
    frame replayed something (here: the second frame through the context)."""
    ctx = mono_ctx
    I0, I1, ts = _scene(21)
    pts0 = ts["pts_l0"]
    n = pts0.shape[0]
    assert n == 1000
    Xw, Tcw_prev, Tcw_prior, dT01 = _world(ts, 5)
    rng = np.random.default_rng(7)
    flags = ((rng.random(n) < 0.7).astype(np.uint8) | ((rng.random(n) < 0.8).astype(np.uint8) << 1)).astype(np.uint8)
    flags |= (rng.random(n) < 0.03).astype(np.uint8) << 2  # landmarks that are no longer alive / tracked (landmark.cpp:207)
    args = (752, 480, 15, 5, 20.0, 1.0, 5, 1.0, MONO_K)
    ctx.set_image(0, I0)
    ctx.set_image(1, I1)
    pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=strict)
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    assert (flags & 4).any() and (g["stage"][(flags & 4) != 0] == 0).all()
    border = oracle.IC_REFERENCE if strict else oracle.IC_MASKED
    prm_o = oracle.make_mono_params(*args)
    o = oracle.mono_frame(prm_o, I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512, border, 8)
    o_seq = oracle.mono_frame(prm_o, I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_SEQ, 0, border, 8)
    _compare(g, o, o_seq)
    c = g["counts"]
    assert c.need_five_point == 0 and c.n_klt > 0.7 * n and c.n_refine > 0.6 * n and c.n_ba > 0.4 * n
    assert c.n_final > 0.5 * n
    assert np.linalg.norm(g["dT01"] - ts["dT_true"]) / np.linalg.norm(ts["dT_true"]) < 2e-2
    if strict in (1, 3, 4):
        assert c.n_replayed >= 16  # margin 6 px: some IC windows leave the image (and mode 4 goes concurrent next time)
    # a second frame through the same context: the control block was reset on the device
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g2 = pipe.result()
    _compare(g2, o)
    assert ctx.frame_recoveries() == 0


@pytest.mark.parametrize("win,strict", [(13, 1), (21, 3), (31, 1), (31, 3)])
def test_mono_frame_other_windows(mono_ctx, vo, oracle, win, strict):
    """The other window sizes the mono frame kernel is built for (config/**.yaml use 13, 15, 21; 31 is the largest KLT window
    of the operator tests), stream-ordered and concurrent replay."""
    ctx = mono_ctx
    I0, I1, ts = _scene(23)
    pts0 = ts["pts_l0"]
    n = pts0.shape[0]
    Xw, Tcw_prev, Tcw_prior, dT01 = _world(ts, 4)
    rng = np.random.default_rng(11)
    flags = ((rng.random(n) < 0.7).astype(np.uint8) | ((rng.random(n) < 0.8).astype(np.uint8) << 1)).astype(np.uint8)
    args = (752, 480, win, 4, 20.0, 1.0, 5, 1.0, MONO_K)
    ctx.set_image(0, I0)
    ctx.set_image(1, I1)
    pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=strict)
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    o = oracle.mono_frame(oracle.make_mono_params(*args), I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512,
                          oracle.IC_REFERENCE, 8)
    _compare(g, o)
    assert g["counts"].n_final > 0.4 * n and g["counts"].n_replayed > 0
    assert ctx.frame_recoveries() == 0


def test_mono_frame_join_timeout_is_recovered(vo, oracle):
    """The concurrent replay's device-side join cannot be met (VO_DBG_FAIL_JOIN: the BA launch waits for a count that never
    comes, as under a tool that serialises the queues): the frame is issued again with the stream-ordered replay — same
    results, one recovery, and the context stays in stream order afterwards."""
    ctx = vo.Context(device=0, max_width=752, max_height=480, max_points=2048, n_slots=3, max_level=5)
    try:
        I0, I1, ts = _scene(21)
        pts0 = ts["pts_l0"]
        n = pts0.shape[0]
        Xw, Tcw_prev, Tcw_prior, dT01 = _world(ts, 5)
        rng = np.random.default_rng(7)
        flags = ((rng.random(n) < 0.7).astype(np.uint8) | ((rng.random(n) < 0.8).astype(np.uint8) << 1)).astype(np.uint8)
        args = (752, 480, 15, 5, 20.0, 1.0, 5, 1.0, MONO_K)
        ctx.set_image(0, I0)
        ctx.set_image(1, I1)
        prm_o = oracle.make_mono_params(*args)
        o = oracle.mono_frame(prm_o, I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512, oracle.IC_REFERENCE, 8)
        pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=3)
        ctx.debug_set(ctx.DBG_FAIL_JOIN, 1)
        pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
        g = pipe.result()
        _compare(g, o)
        assert ctx.frame_recoveries() == 1
        ctx.debug_set(ctx.DBG_FAIL_JOIN, 0)
        pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
        _compare(pipe.result(), o)
        assert ctx.frame_recoveries() == 1
    finally:
        ctx.close()


def test_mono_frame_five_point_fallback_and_empty(mono_ctx, vo, oracle):
    ctx = mono_ctx
    I0, I1, ts = _scene(33)
    pts0 = ts["pts_l0"]
    n = pts0.shape[0]
    Xw, Tcw_prev, Tcw_prior, dT01 = _world(ts, 9)
    args = (752, 480, 15, 5, 20.0, 1.0, 5, 1.0, MONO_K)
    ctx.set_image(0, I0)
    ctx.set_image(1, I1)
    pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=1)
    prm_o = oracle.make_mono_params(*args)
    # (a) only 8 landmarks in the BA class -> the BA is not run (mono_vo.cpp:838), 5-point is due
    flags = np.ones(n, np.uint8)
    flags[np.arange(8) * 97 % n] |= 2
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    o = oracle.mono_frame(prm_o, I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512,
                          oracle.IC_REFERENCE, 8)
    _compare(g, o)
    assert g["counts"].need_five_point == 1 and g["stage"].max() == 2
    assert np.array_equal(g["dT01"], dT01.reshape(4, 4))
    # (b) nothing bundled: priors are the previous pixels, scale 1, no BA
    flags[:] = 0
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    o = oracle.mono_frame(prm_o, I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512,
                          oracle.IC_REFERENCE, 8)
    _compare(g, o)
    assert np.all(g["scale"] == 1.0) and g["counts"].n_ba == 0
    # (c) empty track set
    pipe.enqueue(np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.uint8), Tcw_prev,
                 Tcw_prior, dT01)
    g = pipe.result()
    assert g["stage"].size == 0 and g["counts"].need_five_point == 1 and np.array_equal(g["dT01"], dT01.reshape(4, 4))
    # (d) a landmark behind the predicted camera keeps its previous pixel as the prior (camera.cpp:208-213)
    flags[:] = 3
    Xb = Xw.copy()
    Twc_prior = np.linalg.inv(Tcw_prior.astype(np.float64))
    Xb[::50] = (Twc_prior[:3, :3] @ np.array([0.3, -0.2, -2.0]) + Twc_prior[:3, 3]).astype(np.float32)
    pipe.enqueue(pts0, Xb, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    o = oracle.mono_frame(prm_o, I0, I1, pts0, Xb, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512,
                          oracle.IC_REFERENCE, 8)
    _compare(g, o)


def test_mono_frame_rejects_bad_arguments(mono_ctx, vo):
    ctx = mono_ctx
    prm = make_mono_params(752, 480, 17, 5, 20.0, 1.0, 5, 1.0, MONO_K)
    pipe = MonoFramePipeline(ctx, prm)
    with pytest.raises(vo.VoError):
        pipe.enqueue(np.zeros((4, 2), np.float32), np.zeros((4, 3), np.float32), np.zeros(4, np.uint8), np.eye(4),
                     np.eye(4), np.eye(4))


@pytest.mark.parametrize("seed", [3, 8, 15, 29])
def test_mono_frame_strict_replay_other_scenes(mono_ctx, vo, oracle, seed):
    """More scenes with features 4 px from the border: the strict-border replay of the mono frame against the
    oracle's sequential walk (chains of border-touching features differ from scene to scene)."""
    ctx = mono_ctx
    stream = S.StereoStream(width=752, height=480, K=MONO_K, n_u=40, n_v=25, n_new=50, seed=seed, speed=0.25, margin=4.0)
    poses = stream.poses(4)
    I0, _, _ = stream.render_pair(poses[2])
    I1, _, _ = stream.render_pair(poses[3])
    ts = stream.track_set(2, poses[2], poses[3])
    pts0 = ts["pts_l0"]
    n = pts0.shape[0]
    Xw, Tcw_prev, Tcw_prior, dT01 = _world(ts, seed)
    rng = np.random.default_rng(seed)
    flags = ((rng.random(n) < 0.6).astype(np.uint8) | ((rng.random(n) < 0.9).astype(np.uint8) << 1)).astype(np.uint8)
    args = (752, 480, 15, 5, 20.0, 1.0, 5, 1.0, MONO_K)
    ctx.set_image(0, I0)
    ctx.set_image(1, I1)
    pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=1)
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    o = oracle.mono_frame(oracle.make_mono_params(*args), I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01,
                          oracle.SUM_TREE, 512, oracle.IC_REFERENCE, 8)
    _compare(g, o)
    assert g["counts"].n_replayed > 0


def test_mono_frame_window_13(mono_ctx, vo, oracle):
    """window 13 (config/mono/*.yaml use 13 and 15): the other instantiation of the mono frame kernel."""
    ctx = mono_ctx
    I0, I1, ts = _scene(44)
    pts0 = ts["pts_l0"]
    n = pts0.shape[0]
    Xw, Tcw_prev, Tcw_prior, dT01 = _world(ts, 2)
    flags = np.full(n, 3, np.uint8)
    args = (752, 480, 13, 5, 20.0, 1.0, 5, 1.0, MONO_K)
    ctx.set_image(0, I0)
    ctx.set_image(1, I1)
    pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=1)
    pipe.enqueue(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01)
    g = pipe.result()
    o = oracle.mono_frame(oracle.make_mono_params(*args), I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01,
                          oracle.SUM_TREE, 512, oracle.IC_REFERENCE, 8)
    _compare(g, o)
    assert g["counts"].n_final > 0.5 * n


@pytest.mark.parametrize("strict", [1, 0, 3])
def test_mono_frame_closed_new_point_step(vo, oracle, strict):
    """vo_mono_frame_enqueue_closed: the new points a frame reports must be exactly what the reference's sequence gives
    AFTER the frame — updateWeightBin(lmtrack_final.pts1), extractORBwithBinning_fast(I1), trackBidirection(I1, I0, ...)
    (mono_vo.cpp:977-992) — although the device found every bin's best keypoint before the frame and back-tracked all
    of them speculatively inside it. The frame itself is unchanged. Without a BA pose nothing is reported."""
    W, H, win, lvl, nbu, nbv = 752, 480, 15, 5, 40, 25
    ctx = vo.Context(device=0, max_width=W, max_height=H, max_points=2048, n_slots=3, max_level=lvl)
    try:
        I0, I1, ts = _scene(27)
        n = ts["pts_l0"].shape[0]
        rng = np.random.default_rng(3)
        keep = rng.random(n) < 0.75  # a thinned track set: plenty of empty bins
        pts0 = ts["pts_l0"][keep]
        sub = dict(ts, Xp=ts["Xp"][keep])
        Xw, Tcw_prev, Tcw_prior, dT01 = _world(sub, 5)
        m = pts0.shape[0]
        flags = ((rng.random(m) < 0.7).astype(np.uint8) | ((rng.random(m) < 0.8).astype(np.uint8) << 1)).astype(np.uint8)
        args = (W, H, win, lvl, 20.0, 1.0, 5, 1.0, MONO_K)
        ctx.set_image(0, I0)
        ctx.set_image(1, I1)
        fe = vo.FeatureExtractor(ctx)
        fe.initParams(W, H, nbu, nbv, THRES_FAST=15)
        bins = fe.binParams()
        fe.enqueueCandidates(1, 0)
        pipe = MonoFramePipeline(ctx, make_mono_params(*args), strict_border=strict)
        pipe.enqueue_closed(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, bins, 0)
        g = pipe.result()
        border = oracle.IC_REFERENCE if strict else oracle.IC_MASKED
        prm_o = oracle.make_mono_params(*args)
        o = oracle.mono_frame(prm_o, I0, I1, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, oracle.SUM_TREE, 512, border, 8)
        _compare(g, o)
        assert g["counts"].need_five_point == 0
        # the reference's order, on the host, one operator after the other
        final = o["pts1"][o["stage"] == 4]
        us, vs, iu, iv = oracle.weight_bin_init(W, H, nbu, nbv)
        w = oracle.weight_bin_update(final, us, vs, nbu, nbv)
        d = oracle.orb_detect(I1, 15)
        cand, _ = oracle.bucket_argmax(d["xy"], d["response"], iu, iv, nbu, nbv, w)
        rc, p0n, mask = oracle.track_bidirection(I1, I0, cand, win, lvl, 20.0, 1.0, None, 8)
        assert 30 < cand.shape[0] < nbu * nbv
        assert np.array_equal(g["pts1_new"], cand)
        assert np.array_equal(g["mask_new"], mask) and mask.sum() > 10
        assert np.array_equal(g["pts0_new"].view(np.uint32), p0n.view(np.uint32))
        # the same frame again: the table and the control block are reusable
        pipe.enqueue_closed(pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01, bins, 0)
        g2 = pipe.result()
        assert np.array_equal(g2["pts1_new"], cand) and np.array_equal(g2["mask_new"], mask)
        # no pose from the BA (8 landmarks in its class): the 5-point path, and the new points, are the caller's
        f8 = np.ones(m, np.uint8)
        f8[np.arange(8) * 61 % m] |= 2
        pipe.enqueue_closed(pts0, Xw, f8, Tcw_prev, Tcw_prior, dT01, bins, 0)
        g3 = pipe.result()
        assert g3["counts"].need_five_point == 1 and g3["pts1_new"].shape[0] == 0
    finally:
        ctx.close()


def test_mono_closed_step_argument_errors(vo):
    ctx = vo.Context(device=0, max_width=320, max_height=200, max_points=256, n_slots=2, max_level=3)
    try:
        img = (np.random.default_rng(0).random((200, 320)) * 255).astype(np.uint8)
        ctx.set_image(0, img)
        ctx.set_image(1, img)
        fe = vo.FeatureExtractor(ctx)
        fe.initParams(320, 200, 8, 5, THRES_FAST=15)
        bins = fe.binParams()
        pipe = MonoFramePipeline(ctx, make_mono_params(320, 200, 15, 3, 20.0, 1.0, 5, 1.0, (300.0, 300.0, 160.0, 100.0)))
        eye = np.eye(4, dtype=np.float32)
        pts = np.full((4, 2), 100.0, np.float32)
        X = np.ones((4, 3), np.float32)
        fl = np.ones(4, np.uint8)
        with pytest.raises(RuntimeError, match="was not filled"):  # no candidate table yet
            pipe.enqueue_closed(pts, X, fl, eye, eye, eye, bins, 0)
        fe.enqueueCandidates(1, 0)
        with pytest.raises(RuntimeError, match="needs a track set"):
            pipe.enqueue_closed(np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.uint8), eye, eye,
                                eye, bins, 0)
        pipe.enqueue_closed(pts, X, fl, eye, eye, eye, bins, 0)  # and a valid call still goes through
        g = pipe.result()
        assert g["stage"].shape == (4,) and "pts1_new" in g
    finally:
        ctx.close()
