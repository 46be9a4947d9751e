#!/usr/bin/env python3
"""CPU only: the closed loop's CPU restatement in the REFERENCE's summation order (SUM_SEQ) next to the same loop in the
KERNELS' order (SUM_TREE, 512 partials — what the device reproduces bit for bit), both free-running: where do track ids,
flags, keyframe decisions and poses part, and at which gate? (tests/test_stereo_vo_gpu.py: _vs_reference_order runs the
device loop against the SUM_SEQ loop on the GPU box; this script finds and explains the fork without a GPU.)

  python tests/measure/seq_vs_tree.py [--frames 24] [--mono]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stereo(n):
    from oracle import oracle as O
    from oracle.stereo_vo import StereoVORef
    from visual_odometry_ros_amd import synthetic as S
    W, H = S.KITTI_SIZE
    st = S.StereoStream(width=W, height=H, K=S.KITTI_K, n_u=60, n_v=25, seed=2, speed=0.8)
    imgs = [st.render_pair(p)[:2] for p in st.poses(n)]

    def make(sm, tw):
        return StereoVORef(W, H, S.KITTI_K, S.KITTI_K, st.T_lr, 60, 25, thres_fast=15, win=21, max_level=6, kf_trans=1.0, lba=True,
                           sum_mode=sm, tree_width=tw, ic_border=O.IC_REFERENCE, n_threads=8)

    def gate(before, o, i):  # the pose-only BA's inlier measure 0.5 (|rx_l| + |ry_l| + |rx_r| + |ry_r|) at the loop's final pose
        K = np.asarray(S.KITTI_K, np.float64)
        T_pw = np.linalg.inv(before["T_wp"].astype(np.float64))
        Xp = T_pw[:3, :3] @ before["Xw"][i].astype(np.float64) + T_pw[:3, 3]
        T10 = np.linalg.inv(o["dT"].astype(np.float64))
        Xl = T10[:3, :3] @ Xp + T10[:3, 3]
        T_rl = np.linalg.inv(st.T_lr.astype(np.float64))
        Xr = T_rl[:3, :3] @ Xl + T_rl[:3, 3]
        pl, pr = o["pts_l1"][i].astype(np.float64), o["pts_r1"][i].astype(np.float64)
        r = [K[0] * Xl[0] / Xl[2] + K[2] - pl[0], K[1] * Xl[1] / Xl[2] + K[3] - pl[1],
             K[0] * Xr[0] / Xr[2] + K[2] - pr[0], K[1] * Xr[1] / Xr[2] + K[3] - pr[1]]
        return 0.5 * sum(abs(v) for v in r)

    a, b = make(O.SUM_SEQ, 0), make(O.SUM_TREE, 512)
    forked = False
    for k in range(n):
        sa, sb = dict(T_wp=a.T_wp.copy(), Xw=a.Xw.copy()), dict(T_wp=b.T_wp.copy(), Xw=b.Xw.copy())
        ia, ib = a.track(*imgs[k]), b.track(*imgs[k])
        same = np.array_equal(a.ids, b.ids)
        rel = np.linalg.norm(a.T_wp.astype(np.float64) - b.T_wp) / np.linalg.norm(a.T_wp)
        print(f"frame {k:2d}: ids equal {same} ({len(a.ids)} / {len(b.ids)}), flags equal {same and np.array_equal(a.flags, b.flags)}, "
              f"keyframe {ia['keyframe']} / {ib['keyframe']}, pose rel. Frobenius {rel:.2e}", flush=True)
        if not same and not forked and "frame" in ia:
            forked = True
            fa, fb = ia["frame"], ib["frame"]
            d = np.nonzero(fa["stage"] != fb["stage"])[0]
            print(f"  FORK: features {d.tolist()} stages {fa['stage'][d].tolist()} (reference order) / {fb['stage'][d].tolist()} (kernels' order); "
                  f"GN iterations {fa['counts'].gn_iterations} / {fb['counts'].gn_iterations}, BA set {fa['counts'].n_ba} / {fb['counts'].n_ba}")
            for i in d:
                print(f"  feature {i}: BA inlier measure {gate(sa, fa, i):.4f} / {gate(sb, fb, i):.4f} against thres_poseba_error 3.0; refined left "
                      f"pixel differs by {np.abs(fa['pts_l1'][i] - fb['pts_l1'][i]).max():.3f} px, right by {np.abs(fa['pts_r1'][i] - fb['pts_r1'][i]).max():.3f} px")


def mono(n):
    from oracle import oracle as O
    from oracle.mono_vo import MonoVORef
    from test_mono_vo_gpu import MONO_K, TruePoseHook
    from visual_odometry_ros_amd import synthetic as S
    W, H, nu, nv = 752, 480, 40, 25
    st = S.StereoStream(width=W, height=H, K=MONO_K, n_u=nu, n_v=nv, seed=5, speed=0.25)
    poses = st.poses(n)
    imgs = [st.render_pair(p)[0] for p in poses]

    def make(sm, tw, hook):
        return MonoVORef(W, H, MONO_K, nu, nv, hook, thres_fast=15, win=15, max_level=5, thres_err=20.0, thres_bidir=1.0, thres_poseba=5,
                         thres_sampson=1.0, thres_parallax_deg=1.0, kf_trans=2.5, lba=True, sum_mode=sm, tree_width=tw,
                         ic_border=O.IC_REFERENCE, n_threads=8)
    ha, hb = TruePoseHook(poses), TruePoseHook(poses)
    a, b = make(O.SUM_SEQ, 0, ha), make(O.SUM_TREE, 512, hb)
    for k in range(n):
        ha.k = hb.k = k
        ia, ib = a.track(imgs[k]), b.track(imgs[k])
        same = np.array_equal(a.ids, b.ids)
        Ta, Tb = a.frames[k]["T_wc"].astype(np.float64), b.frames[k]["T_wc"].astype(np.float64)
        print(f"frame {k:2d}: ids equal {same} ({len(a.ids)} / {len(b.ids)}), flags equal {same and np.array_equal(a.flags(), b.flags())}, "
              f"keyframe {ia['keyframe']} / {ib['keyframe']}, pose rel. Frobenius {np.linalg.norm(Ta - Tb) / np.linalg.norm(Ta):.2e}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--mono", action="store_true")
    args = ap.parse_args()
    (mono if args.mono else stereo)(args.frames)
