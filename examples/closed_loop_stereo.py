"""A closed-loop stereo VO on the synthetic stream, written the way StereoVO::trackStereoImages
(core/visual_odometry/stereo_vo/stereo_vo.cpp:392-989) is organised — every numeric step is an operator of
libvo_hip.so, the landmark bookkeeping in between is a few numpy arrays on the host:

  first frame   detect + bucket (extractORBwithBinning_fast), trackBidirection left -> right,
                triangulate from the disparity                                   (stereo_vo.cpp:212-330)
  every frame   [3]-[7] prior, KLT l0->l1, patch refinement, KLT l1->r1, stereo pose-only BA: ONE call
                (StereoFramePipeline)                                            (:483-668)
                [8] survivors = BA inliers; their 3-D points re-triangulated in the new camera frame
                [9] updateWeightBin(survivors), extractORBwithBinning_fast       (:691-693)
                [10] trackBidirection l1 -> r1 of the candidates, new landmarks  (:708-760)
With --lba every third frame becomes a stereo keyframe and the window of the last nine is bundle-adjusted
(MotionEstimator::localBundleAdjustmentSparseSolver_Stereo, motion_estimator.cpp:1219-1340): `LocalBA` below is the
numeric part of SparseBAParameters::setPosesAndPoints (sparse_ba_parameters.h:283-440: reference frame = first
keyframe of the window, translations scaled by 1/10, first two keyframes fixed, landmarks seen in at least two
keyframes) and of the solver's finalisation (sparse_bundle_adjustment.cpp:645-745), around vo_sba_solve. The
reference's keyframe selection and landmark graph are not modelled: tracks are (id, pixel, pixel) rows. It exists to show that the operators compose into a working odometry (trajectory against
the renderer's ground truth), not as a benchmark.

usage: python examples/closed_loop_stereo.py [--frames 30]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def triangulate(pl, pr, K, baseline):
    """Rectified pair: depth from the horizontal disparity, point in the left camera frame."""
    fx, fy, cx, cy = K
    d = pl[:, 0] - pr[:, 0]
    ok = d > 0.5
    z = fx * baseline / np.where(ok, d, 1.0)
    X = np.stack([(pl[:, 0] - cx) / fx * z, (pl[:, 1] - cy) / fy * z, z], 1).astype(np.float32)
    return X, ok & (z < 80.0)


class LocalBA:
    """Window of stereo keyframes + the observations of the tracks on them; solve() = one local BA."""
    POSE_SCALE = 10.0   # SparseBAParameters::pose_scale_
    WINDOW, N_FIX = 9, 2  # kitti_00_stereo.yaml:83, motion_estimator.cpp:1245

    def __init__(self, V, ctx, K, T_lr):
        from visual_odometry_ros_amd.api import SparseBundleAdjustmentSolver
        self.K, self.T_lr = np.asarray(K, np.float64), np.asarray(T_lr, np.float64)
        self.solver = SparseBundleAdjustmentSolver(ctx, True)
        T_s = self.T_lr.copy()
        T_s[:3, 3] /= self.POSE_SCALE  # scalingPose(T_stereo_), sparse_ba_parameters.h:295-299
        self.solver.setStereoCameras(self.K, self.K, T_s)
        self.solver.setHuberThreshold(0.5)  # motion_estimator.cpp:1232
        self.kf_pose, self.kf_obs = [], []  # T_wc per keyframe ; {track id: (pl, pr)} per keyframe
        self.Xw = {}                        # track id -> world point

    def add_keyframe(self, T_wc, ids, pl, pr, Xc):
        for i, X in zip(ids, Xc):
            if i not in self.Xw:
                self.Xw[i] = T_wc[:3, :3] @ X + T_wc[:3, 3]
        self.kf_pose.append(T_wc.copy())
        self.kf_obs.append({int(i): (a, b) for i, a, b in zip(ids, pl, pr)})

    def solve(self):
        """Returns (keyframe indices of the window, their poses before, after, avg pixel error per iteration)."""
        win = list(range(max(0, len(self.kf_pose) - self.WINDOW), len(self.kf_pose)))
        if len(win) < 3:  # NUM_MINIMUM_REQUIRED_KEYFRAMES
            return None
        T_ref_w = np.linalg.inv(self.kf_pose[win[0]])  # Tjw_ref_
        T_jw = []
        for k in win:  # changeInvPoseWorldToRef + scalingPose
            T = np.linalg.inv(self.kf_pose[k]) @ self.kf_pose[win[0]]
            T[:3, 3] /= self.POSE_SCALE
            T_jw.append(T)
        seen = {}
        for j, k in enumerate(win):
            for i in self.kf_obs[k]:
                seen.setdefault(i, []).append(j)
        lms = [i for i, js in seen.items() if len(js) >= 2]  # THRES_MINIMUM_SEEN
        if len(lms) < 20:
            return None
        X, obs_ptr, obs_frame, obs_right, obs_px = [], [0], [], [], []
        for i in lms:
            X.append((T_ref_w[:3, :3] @ self.Xw[i] + T_ref_w[:3, 3]) / self.POSE_SCALE)  # warpToRef + scalingPoint
            for j in seen[i]:
                pl, pr = self.kf_obs[win[j]][i]
                obs_frame += [j, j]
                obs_right += [0, 1]
                obs_px += [pl, pr]
            obs_ptr.append(len(obs_frame))
        opt = np.array([-1] * self.N_FIX + list(range(len(win) - self.N_FIX)), np.int32)
        ok, T_new, X_new, err = self.solver.solveForFiniteIterations(
            10, np.stack(T_jw), opt, np.stack(X), np.array(obs_ptr, np.int32), np.array(obs_frame, np.int32),
            np.array(obs_right, np.uint8), np.array(obs_px, np.float64))
        before = [self.kf_pose[k].copy() for k in win]
        for j, k in enumerate(win):  # recoverOriginalScalePose, changeInvPoseRefToWorld, setPose(inverse)
            if opt[j] >= 0:
                T = T_new[j].copy()
                T[:3, 3] *= self.POSE_SCALE
                self.kf_pose[k] = self.kf_pose[win[0]] @ np.linalg.inv(T)
        T_w_ref = before[0]
        for i, x in zip(lms, X_new):  # recoverOriginalScalePoint, warpToWorld
            self.Xw[i] = T_w_ref[:3, :3] @ (x * self.POSE_SCALE) + T_w_ref[:3, 3]
        return win, before, [self.kf_pose[k].copy() for k in win], err


def run(n_frames=30, seed=2, n_bins=(40, 16), thres_fast=15, verbose=False, lba=False, closed=False, strict_border=True):
    """closed: steps [9] + [10] inside the frame operator (enqueueCandidates on the new left image + enqueue_closed; the
    new points come with the frame's result) instead of three operator calls after it — the same odometry, bit for bit."""
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
    V.load()
    stream = S.StereoStream(seed=seed)
    W, H, K, b = stream.width, stream.height, stream.K, stream.baseline
    win, lvl, thr_err, thr_bidir, thr_ba = 21, 6, 80.0, 0.5, 3.0  # config/stereo/kitti_00_stereo.yaml:55-59,74
    poses = stream.poses(n_frames)
    ctx = V.Context(device=0, max_width=W, max_height=H, max_points=4096, n_slots=3, max_level=lvl)
    ft, fe = V.FeatureTracker(ctx), V.FeatureExtractor(ctx)
    fe.initParams(W, H, n_bins[0], n_bins[1], THRES_FAST=thres_fast)
    pipe = StereoFramePipeline(ctx, make_stereo_params(W, H, win, lvl, thr_err, thr_bidir, thr_ba, K, K, stream.T_lr),
                               strict_border=strict_border)
    bins = fe.binParams()
    P, CL, CR = 0, 1, 2  # slots: previous left, current left, current right

    def landmarks_of(cand, pr, m):
        """new stereo landmarks from bucketed keypoints, their right pixels and trackBidirection's mask"""
        if cand.shape[0] == 0:
            return np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32)
        X, ok = triangulate(cand, pr, K, b)
        keep = m & ok & (np.abs(cand[:, 1] - pr[:, 1]) < 2.0)
        return cand[keep], pr[keep], X[keep]

    def new_landmarks(slot_l, slot_r, tracked_pts):
        """steps [9] + [10]: bucketed detections in bins without a tracked point, stereo-matched, triangulated"""
        fe.updateWeightBin(tracked_pts)
        cand = fe.extractORBwithBinning_fast(slot_l)
        if cand.shape[0] == 0:
            return np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32)
        pr, m = ft.trackBidirection(slot_l, slot_r, cand, win, lvl, thr_err, thr_bidir)
        return landmarks_of(cand, pr, m)

    L, R, _ = stream.render_pair(poses[0])
    ctx.set_image(CL, L)
    ctx.set_image(CR, R)
    pts_l, pts_r, X = new_landmarks(CL, CR, np.zeros((0, 2), np.float32))
    ids = np.arange(pts_l.shape[0])
    next_id = ids.size
    ba = LocalBA(V, ctx, K, stream.T_lr) if lba else None
    ba_log = []
    if ba:
        ba.add_keyframe(poses[0], ids, pts_l, pts_r, X)
    T_wc = [poses[0].copy()]
    dT_prev = np.eye(4, dtype=np.float32)
    log = []
    for k in range(1, n_frames):
        L, R, _ = stream.render_pair(poses[k])
        ctx.swap_slots(P, CL)  # the current left image becomes the previous one, its pyramid stays on the device
        ctx.set_image(CL, L)
        ctx.set_image(CR, R)
        if closed:
            fe.enqueueCandidates(CL, k & 1)  # from the image alone: the best keypoint of every bin
            pipe.enqueue_closed(pts_l, pts_r, X, dT_prev, bins, k & 1, slots=(P, CL, CR))
        else:
            pipe.enqueue(pts_l, pts_r, X, dT_prev, np.zeros((0, 2), np.float32), slots=(P, CL, CR))
        r = pipe.result()
        dT = r["dT"].astype(np.float64)
        T_wc.append(T_wc[-1] @ dT)
        inl = r["stage"] >= 4
        pl1, pr1 = r["pts_l1"][inl], r["pts_r1"][inl]
        Xc, ok = triangulate(pl1, pr1, K, b)  # [8]: the survivors' points in the new camera frame
        pl1, pr1, Xc = pl1[ok], pr1[ok], Xc[ok]
        ids = ids[inl][ok]
        if ba and k % 3 == 0:  # a stereo keyframe: the live tracks' observations, then the window's local BA
            ba.add_keyframe(T_wc[-1], ids, pl1, pr1, Xc)
            out = ba.solve()
            if out is not None:
                kf_win, before, after, err = out
                gt = [poses[3 * w] for w in kf_win]
                e0 = float(np.mean([np.linalg.norm(b[:3, 3] - g[:3, 3]) for b, g in zip(before, gt)]))
                e1 = float(np.mean([np.linalg.norm(a_[:3, 3] - g[:3, 3]) for a_, g in zip(after, gt)]))
                ba_log.append(dict(frame=k, keyframes=len(kf_win), err_first=float(err[0]), err_last=float(err[-1]),
                                   kf_pos_err_before=e0, kf_pos_err_after=e1))
                if verbose:
                    print("LBA", ba_log[-1])
                T_wc[-1] = ba.kf_pose[-1].copy()  # the odometry continues from the adjusted keyframe pose
        # [9] + [10] on lmtrack_final (stereo_vo.cpp:691-711: every stage-4 feature, triangulable or not)
        if closed:
            nl, nr, nX = landmarks_of(r["pts_new"], r["pts_new_r"], r["mask_new"])
        else:
            nl, nr, nX = new_landmarks(CL, CR, r["pts_l1"][inl])
        ids = np.concatenate([ids, np.arange(next_id, next_id + nl.shape[0])])
        next_id += nl.shape[0]
        log.append(dict(frame=k, tracked=int(pts_l.shape[0]), inliers=int(inl.sum()), new=int(nl.shape[0]),
                        gn_iterations=int(r["counts"].gn_iterations)))
        if verbose:
            print(log[-1])
        pts_l, pts_r, X = np.concatenate([pl1, nl]), np.concatenate([pr1, nr]), np.concatenate([Xc, nX])
        dT_prev = r["dT"].astype(np.float32)  # constant-velocity prior (stereo_vo.cpp:465-480)
    ctx.close()
    # trajectory error against the renderer's ground truth
    est = np.stack([T[:3, 3] for T in T_wc])
    gt = np.stack([T[:3, 3] for T in poses[:n_frames]])
    path = np.sum(np.linalg.norm(np.diff(gt, axis=0), axis=1))
    rel = [np.linalg.norm(np.linalg.inv(np.linalg.inv(poses[i - 1]) @ poses[i]) @ (np.linalg.inv(T_wc[i - 1]) @ T_wc[i]) - np.eye(4))
           for i in range(1, n_frames)]
    return dict(T_wc=np.stack(T_wc), frames=n_frames, path_m=float(path), end_error_m=float(np.linalg.norm(est[-1] - gt[-1])),
                ate_rmse_m=float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1)))), max_step_error=float(max(rel)),
                mean_tracked=float(np.mean([e["tracked"] for e in log])), mean_inliers=float(np.mean([e["inliers"] for e in log])),
                log=log, lba=ba_log)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--lba", action="store_true", help="stereo keyframe every third frame + local BA of the window")
    ap.add_argument("--closed", action="store_true", help="steps [9] + [10] closed inside the frame operator")
    a = ap.parse_args()
    out = run(a.frames, verbose=a.verbose, lba=a.lba, closed=a.closed)
    out.pop("log")
    out.pop("T_wc")
    print(json.dumps(out))
