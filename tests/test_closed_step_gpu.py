"""Step [10] closed on the device (vo_new_point_candidates_enqueue + vo_stereo_frame_enqueue_closed): the candidates a
frame reports must be exactly what the reference's sequence gives AFTER the frame — updateWeightBin(lmtrack_final.pts_l1),
extractORBwithBinning_fast(I1_left), trackBidirection(I1_left, I1_right, ...) (stereo_vo.cpp:691-711) — although the
device found the best keypoint of every bin before the frame and tracked all of them speculatively inside it. The oracle
side composes the restated operators in the reference's order, on the host, one after the other."""
import numpy as np
import pytest

from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params

pytestmark = pytest.mark.gpu
GN_T = 512


def _oracle_step10(oracle, fe_cfg, L, R, final_pts, win, max_level, thres_err, thres_bidir):
    n_cols, n_rows, nbu, nbv, thr_fast = fe_cfg
    us, vs, iu, iv = oracle.weight_bin_init(n_cols, n_rows, nbu, nbv)
    w = oracle.weight_bin_update(final_pts, us, vs, nbu, nbv)          # reset + update (feature_extractor.cpp:94-98)
    d = oracle.orb_detect(L, thr_fast)                                    # extractor_orb_->detect (:241)
    kp, resp = d["xy"], d["response"]
    cand, _ = oracle.bucket_argmax(kp, resp, iu, iv, nbu, nbv, w)        # :244-277
    rc, pnr, mask = oracle.track_bidirection(L, R, cand, win, max_level, thres_err, thres_bidir, None, 8)  # stereo_vo.cpp:706
    return cand, pnr, mask


@pytest.mark.parametrize("side_ingest,strict", [(False, True), (True, True), (True, False), (True, 4), (True, 3)])
def test_closed_step10_matches_the_reference_sequence(vo, oracle, side_ingest, strict):
    W, H, win, lvl = 620, 188, 21, 4
    K = tuple(v * 0.5 for v in S.KITTI_K)
    nbu, nbv = 30, 12
    stream = S.StereoStream(width=W, height=H, K=K, n_u=nbu, n_v=nbv, n_new=0, seed=41, margin=5.0 if strict else 14.0)
    prm_g = make_stereo_params(W, H, win, lvl, 80.0, 0.5, 3.0, K, K, stream.T_lr)
    prm_o = oracle.make_stereo_params(W, H, win, lvl, 80.0, 0.5, 3.0, K, K, stream.T_lr)
    ctx = vo.Context(device=0, max_width=W, max_height=H, max_points=1024, n_slots=5, max_level=lvl)
    try:
        ctx.set_ingest_side_stream(side_ingest)
        pipe = StereoFramePipeline(ctx, prm_g, strict_border=strict)
        fe = vo.FeatureExtractor(ctx)
        fe.initParams(W, H, nbu, nbv, THRES_FAST=15)
        bins = fe.binParams()
        poses = stream.poses(4)
        pairs = [stream.render_pair(p)[:2] for p in poses]
        pinned = [(np.ascontiguousarray(L), np.ascontiguousarray(R)) for L, R in pairs]
        slot = dict(P=0, CL=1, CR=2, NL=3, NR=4)
        ctx.set_image(slot["P"], pairs[0][0])
        ctx.set_stereo_pair_host_async(slot["CL"], pinned[1][0].ctypes.data, slot["CR"], pinned[1][1].ctypes.data, W, H, W)
        fe.enqueueCandidates(slot["CL"], 1)
        empty = np.zeros((0, 2), np.float32)
        for k in range(1, 4):
            Lp, (L, R) = pairs[k - 1][0], pairs[k]
            ts = stream.track_set(k - 1, poses[k - 1], poses[k])
            n = ts["pts_l0"].shape[0]
            rng = np.random.default_rng(k)
            keep = rng.random(n) < 0.8  # a thinned track set: plenty of empty bins
            fl = (rng.random(n) >= 0.25).astype(np.uint8)
            sub = {f: ts[f][keep] for f in ("pts_l0", "pts_r0", "Xp")}
            fl = fl[keep]
            if k == 1:  # the table is what the bucketing gives with every weight 1
                xy_t, has_t, nd = fe.getCandidates(1)
                d = oracle.orb_detect(L, 15)
                kp, resp = d["xy"], d["response"]
                assert nd == kp.shape[0] > 500
                cand_all, idx_all = oracle.bucket_argmax(kp, resp, fe.inv_u_step_, fe.inv_v_step_, nbu, nbv,
                                                         np.ones(nbu * nbv, np.int32))
                assert has_t.sum() == cand_all.shape[0] and np.array_equal(xy_t[has_t], cand_all)
            pipe.enqueue_closed(sub["pts_l0"], sub["pts_r0"], sub["Xp"], ts["dT_prior"], bins, k & 1,
                                slots=(slot["P"], slot["CL"], slot["CR"]), lm_flags=fl)
            if k < 3:  # the next pair arrives while this frame runs: ingestion + detection of pair k+1
                ctx.set_stereo_pair_host_async(slot["NL"], pinned[k + 1][0].ctypes.data, slot["NR"],
                                               pinned[k + 1][1].ctypes.data, W, H, W)
                fe.enqueueCandidates(slot["NL"], (k + 1) & 1)
            g = pipe.result()
            border = oracle.IC_REFERENCE if strict else oracle.IC_MASKED
            o = oracle.stereo_frame(prm_o, Lp, L, R, sub["pts_l0"], sub["pts_r0"], sub["Xp"], ts["dT_prior"], empty,
                                    oracle.SUM_TREE, GN_T, border, 8, lm_flags=fl)
            assert o["rc"] == 0
            assert np.array_equal(g["stage"], o["stage"])
            assert np.array_equal(g["pts_l1"].view(np.uint32), o["pts_l1"].view(np.uint32))
            assert np.array_equal(g["pts_r1"].view(np.uint32), o["pts_r1"].view(np.uint32))
            assert np.array_equal(g["dT"].view(np.uint32), o["dT"].astype(np.float32).view(np.uint32))
            final = o["pts_l1"][o["stage"] == 4]
            cand, pnr, mask = _oracle_step10(oracle, (W, H, nbu, nbv, 15), L, R, final, win, lvl, 80.0, 0.5)
            assert 20 < cand.shape[0] < nbu * nbv
            assert np.array_equal(g["pts_new"], cand)
            assert np.array_equal(g["mask_new"], mask)
            assert np.array_equal(g["pts_new_r"].view(np.uint32), pnr.view(np.uint32))
            assert g["counts"].n_new_ok == int(mask.sum()) > 0
            slot["P"], slot["CL"], slot["CR"], slot["NL"], slot["NR"] = (slot["CL"], slot["NL"], slot["NR"], slot["P"],
                                                                           slot["CR"])
    finally:
        ctx.close()


def test_rebuilding_a_slot_under_a_frame_in_flight_is_refused(vo):
    W, H = 320, 200
    stream = S.StereoStream(width=W, height=H, K=(300.0, 300.0, 160.0, 100.0), n_u=8, n_v=5, n_new=4, seed=7)
    ctx = vo.Context(device=0, max_width=W, max_height=H, max_points=256, n_slots=4, max_level=3)
    try:
        ctx.set_ingest_side_stream(True)
        prm = make_stereo_params(W, H, 21, 3, 80.0, 0.5, 3.0, stream.K, stream.K, stream.T_lr)
        pipe = StereoFramePipeline(ctx, prm)
        poses = stream.poses(2)
        L0, _, _ = stream.render_pair(poses[0])
        L1, R1, _ = stream.render_pair(poses[1])
        for s_, im in enumerate((L0, L1, R1)):
            ctx.set_image(s_, im)
        ts = stream.track_set(0, poses[0], poses[1])
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"])
        with pytest.raises(RuntimeError, match="read by the frame in flight"):
            ctx.set_image(1, L0)
        ctx.set_image(3, L0)  # a slot the frame does not read is fine
        g = pipe.result()
        assert g["counts"].n_inlier > 0
        ctx.set_image(1, L0)  # and after the result, so is this one
    finally:
        ctx.close()
