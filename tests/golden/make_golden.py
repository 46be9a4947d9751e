#!/usr/bin/env python3
"""Generates the fixtures in tests/golden/ with the CPU oracle (oracle/, reference
summation order, reference border semantics). The reference itself has no fixtures
and cannot run here, so these pin the ORACLE's bits: inputs + expected outputs only.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from visual_odometry_ros_amd import synthetic as S  # noqa: E402
from util import grid_points, image_pair, move_points  # noqa: E402


def main():
    # BASELINE config 1: 2-view 500-point synthetic set
    d = S.two_view_points(n=500, seed=1)
    T0 = np.eye(4, dtype=np.float32)
    rc, T, mask, info = O.gn_pose_stereo(d["X"], d["pts_l"], d["pts_r"], d["K"], d["K"], d["T_lr"], 3.0, T0)
    rct, Tt, maskt, infot = O.gn_pose_stereo(d["X"], d["pts_l"], d["pts_r"], d["K"], d["K"], d["T_lr"], 3.0, T0,
                                             O.SUM_TREE, 512)
    rcm, R, t, maskm, infom = O.gn_pose_mono(d["X"], d["pts_l"], d["K"], 3, np.eye(3), np.zeros(3), O.GN_STANDALONE)
    np.savez_compressed(os.path.join(HERE, "gn_config1.npz"), X=d["X"], pts_l=d["pts_l"], pts_r=d["pts_r"],
                        K=d["K"], T_lr=d["T_lr"], T01_seq=T, mask_seq=mask, iters_seq=info.iterations,
                        T01_tree512=Tt, mask_tree512=maskt, iters_tree512=infot.iterations,
                        mono_R=R, mono_t=t, mono_mask=maskm, mono_iters=infom.iterations)
    # small KLT + IC case
    motion = dict(dx=3.3, dy=-2.1, scale=1.02, angle=0.004)
    img0, img1 = image_pair(160, 208, seed=42, **motion)
    pts0 = grid_points(160, 208, step=14, margin=4)
    lv, p1, st, err = O.calc_optical_flow_pyr_lk(img0, img1, pts0, None, 21, 3)
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion).astype(np.float32)
    prior = (gt + 0.5).astype(np.float32)
    scale = np.full(pts0.shape[0], 1.02, np.float32)
    rc, ic_pts, ic_mask, tb = O.track_with_scale(img0, img1, pts0, scale, prior, None, O.IC_REFERENCE, O.SUM_SEQ)
    rc, ic_pts_t, ic_mask_t, _ = O.track_with_scale(img0, img1, pts0, scale, prior, None, O.IC_REFERENCE,
                                                     O.SUM_TREE)
    np.savez_compressed(os.path.join(HERE, "klt_small.npz"), img0=img0, img1=img1, pts0=pts0, pts1=p1, status=st,
                        err=err, prior=prior, scale=scale, ic_pts=ic_pts, ic_mask=ic_mask, ic_pts_tree=ic_pts_t,
                        ic_mask_tree=ic_mask_t, touched=tb)
    a = S.random_descriptors(48, seed=1)
    b = S.random_descriptors(40, seed=2, flip_from=a, flip_bits=18)
    bi, bd, sd = O.hamming_match(a, b, 50, 0.6)
    np.savez_compressed(os.path.join(HERE, "hamming.npz"), a=a, b=b, dist=O.hamming_matrix(a, b), best_idx=bi,
                        best_dist=bd, second_dist=sd)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
