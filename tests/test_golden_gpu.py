"""GPU vs the committed golden fixtures (tests/golden/, made by make_golden.py with the oracle)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_gn_config1_golden(ctx, vo):
    g = np.load(os.path.join(GOLD, "gn_config1.npz"))
    me = vo.MotionEstimator(ctx, True, g["T_lr"])
    ok, T, mask, info = me.poseOnlyBundleAdjustment_Stereo(g["X"], g["pts_l"], g["pts_r"], g["K"], g["K"],
                                                          g["T_lr"], 3.0, np.eye(4, dtype=np.float32))
    assert ok and info.iterations == int(g["iters_tree512"])
    assert np.array_equal(mask, g["mask_tree512"]) and np.array_equal(mask, g["mask_seq"])
    assert np.linalg.norm(T - g["T01_tree512"]) / np.linalg.norm(g["T01_tree512"]) < 1e-6
    assert np.linalg.norm(T - g["T01_seq"]) / np.linalg.norm(g["T01_seq"]) < 1e-4
    me1 = vo.MotionEstimator(ctx)
    ok, R, t, m, inf = me1.poseOnlyBundleAdjustment(g["X"], g["pts_l"], g["K"], 3, np.eye(3), np.zeros(3), 1)
    assert ok and np.array_equal(m, g["mono_mask"])
    assert np.abs(R - g["mono_R"]).max() < 1e-5 and np.abs(t - g["mono_t"]).max() < 1e-4


def test_klt_ic_golden(ctx, vo):
    k = np.load(os.path.join(GOLD, "klt_small.npz"))
    ctx.set_image(0, k["img0"])
    ctx.set_image(1, k["img1"])
    ft = vo.FeatureTracker(ctx)
    lv, p1, st, err = ft.calcOpticalFlowPyrLK(0, 1, k["pts0"], None, 21, 3)
    assert np.array_equal(p1, k["pts1"]) and np.array_equal(st, k["status"]) and np.array_equal(err, k["err"])
    p, m = ft.trackWithScale(0, 1, k["pts0"], k["scale"], k["prior"], None, strict_border=True)
    assert np.array_equal(m, k["ic_mask_tree"]) and np.array_equal(m, k["ic_mask"])
    assert np.array_equal(p, k["ic_pts_tree"])
    assert np.abs(p - k["ic_pts"]).max() < 1e-3


def test_hamming_golden(ctx, vo):
    h = np.load(os.path.join(GOLD, "hamming.npz"))
    fe = vo.FeatureExtractor(ctx)
    assert np.array_equal(fe.descriptorDistance(h["a"], h["b"]), h["dist"])
    bi, bd, sd = fe.match(h["a"], h["b"], 50, 0.6)
    assert np.array_equal(bi, h["best_idx"]) and np.array_equal(bd, h["best_dist"])
    assert np.array_equal(sd, h["second_dist"])
