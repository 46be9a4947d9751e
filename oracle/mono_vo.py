"""TEST INFRASTRUCTURE ONLY (see vo_oracle.h) — CPU restatement of the closed loop of MonoVO::trackImage
(core/visual_odometry/mono_vo/mono_vo.cpp:496-1194) around the restated operators of oracle/*.c. PARITY UNPINNED (the
reference holds no fixture and cannot be built here).

Follows
  mono_vo.cpp:518             one Frame per image (frame.cpp:22-41: id = frame_counter_++)
  mono_vo.cpp:528-561         the first image: resetWeightBin + extractORBwithBinning_fast, one landmark per pixel, pose I,
                              setPoseDiff10(T_init) with T_init.t = (0, 0, -1)
  mono_vo.cpp:562-696         the second image (initialisation): FeatureTracker::track, the 5-point pose — OpenCV calib3d,
                              out of scope: a CALLER HOOK here —, Sampson gate, observations (added BEFORE the
                              frame's pose is set: their parallax sees the identity), |dt10| = 1, new points back-tracked
                              into I0, reconstruction of every landmark with enough parallax (X0(2) > 0 only)
  mono_vo.cpp:698-1019        steady state: prior + patch scale from bundled landmarks, trackBidirectionWithPrior,
                              trackWithScale, the pose-only BA on the bundled (more than five window keyframes) or
                              triangulated landmarks, mask_motion, Sampson gate, new points (oracle_mono.c:
                              vo_ref_mono_frame); the 5-point fallback (:909-949) through the same hook
  mono_vo.cpp:1022-1157       keyframe rule (keyframes.cpp:47-126), addNewKeyframe (:30-45), reconstruction of landmarks seen
                              on more than two keyframes, localBundleAdjustmentSparseSolver (motion_estimator.cpp:1090-1205,
                              sparse_ba_parameters.h:292-466 in mono mode: at least two window keyframes per landmark),
                              write-back incl. setBundled / setDead (sparse_bundle_adjustment.cpp:624-722)
  landmark.cpp:76-135         addObservationAndRelatedFrame: age, parallax w.r.t. the oldest observation
As in stereo_vo.py the landmarks of the local BA are taken by ascending id (the reference walks an unordered_set).
"""
import numpy as np

from . import oracle as O
from .stereo_vo import LbaError, _inverse_se3_f64, _mul44_f64, _xform_f64

LM_TRIANGULATED, LM_DROPPED, LM_KF_MEMBER, LM_BUNDLED = 1, 2, 4, 8
D2R = np.float32(3.14159265358979323846 / 180.0)


def _se3(R, t):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.asarray(R, np.float32).reshape(3, 3)
    T[:3, 3] = np.asarray(t, np.float32).reshape(3)
    return T


class MonoVORef:
    """One image stream. track(img) = MonoVO::trackImage. `five_point(pts0, pts1) -> (ok, R10, t10, mask)` stands for
    MotionEstimator::calcPose5PointsAlgorithm (motion_estimator.cpp:21-203)."""

    def __init__(self, width, height, K, n_bins_u, n_bins_v, five_point, thres_fast=15, win=15, max_level=5, thres_err=20.0,
                 thres_bidir=1.0, thres_poseba=5, thres_sampson=1.0, thres_parallax_deg=1.0, kf_overlap=0.7, kf_rot_deg=3.0,
                 kf_trans=3.0, kf_window=9, lba=True, ic_border=O.IC_REFERENCE, sum_mode=O.SUM_SEQ, tree_width=0, n_threads=1,
                 undistort_maps=None):
        """undistort_maps = (map_u, map_v) of the camera: flagDoUndistortion (mono_vo.cpp:509-513) — every image goes through
        cam_->undistortImage + convertTo(CV_8UC1) first."""
        self.undistort_maps = undistort_maps
        self.W, self.H, self.K = width, height, np.asarray(K, np.float32)
        self.nu, self.nv, self.thres_fast = n_bins_u, n_bins_v, thres_fast
        self.win, self.max_level = win, max_level
        self.thr = (thres_err, thres_bidir, thres_poseba, thres_sampson)
        self.prm = O.make_mono_params(width, height, win, max_level, thres_err, thres_bidir, thres_poseba, thres_sampson, K)
        self.us, self.vs, self.iu, self.iv = O.weight_bin_init(width, height, n_bins_u, n_bins_v)
        self.thres_parallax = np.float32(np.float32(thres_parallax_deg) * D2R)
        self.kf_overlap = np.float32(kf_overlap)
        self.kf_rot = np.float32(np.float32(kf_rot_deg) * D2R)
        self.kf_trans = np.float32(kf_trans)
        self.kf_window, self.lba = kf_window, lba
        self.border, self.sum_mode, self.tree_width, self.n_threads = ic_border, sum_mode, tree_width, n_threads
        self.five_point = five_point
        self.got_first, self.init_done = False, False
        self.landmark_counter = self.frame_counter = 0
        self.I0 = None
        self.frames = []        # per frame: dict(T_wc, dT01) — T_wc changes when the local BA moves a keyframe
        self.ids = np.zeros(0, np.int32)      # frame_prev_'s related landmarks ...
        self.pts = np.zeros((0, 2), np.float32)  # ... and pixels seen
        self.lm = {}            # every landmark: id -> dict(X, tri, bundled, alive, p_first, f_first, age, last_par, kf_obs)
        self.keyframes = []     # window: dict(serial, frame, ids, pts)
        self.all_keyframes = []
        self.n_keyframes = 0

    # ---- helpers ------------------------------------------------------------------------------------------------
    def _detect_bucket(self, img, weight):
        d = O.orb_detect(img, self.thres_fast)
        pts, _ = O.bucket_argmax(d["xy"], d["response"], self.iu, self.iv, self.nu, self.nv, weight)
        return pts

    def _new_landmark(self, p, frame):
        """Landmark(p, frame): id, first observation (age 1, no parallax yet)."""
        i = self.landmark_counter
        self.landmark_counter += 1
        self.lm[i] = dict(X=np.zeros(3, np.float32), tri=False, bundled=False, alive=True, p_first=np.asarray(p, np.float32).copy(),
                          f_first=frame, age=1, last_par=np.float32(0.0), cos_last=np.float32(1.0), p_last=np.asarray(p, np.float32).copy(),
                          f_last=frame, kf_obs=[])
        return i

    def _add_observation(self, i, p, frame, T_wc_last):
        """addObservationAndRelatedFrame for the second and later observations (landmark.cpp:76-135)."""
        e = self.lm[i]
        e["age"] += 1
        e["p_last"], e["f_last"] = np.asarray(p, np.float32).copy(), frame
        T_cw_first = O.inverse_se3(self.frames[e["f_first"]]["T_wc"])  # getPoseInv(): Tcw_ = inverseSE3_f(Twc_)
        e["last_par"], e["cos_last"] = O.parallax(e["p_first"], p, self.K, T_cw_first, T_wc_last)

    def _flags(self, ids):
        out = np.zeros(len(ids), np.uint8)
        for k, i in enumerate(ids):
            e = self.lm[int(i)]
            out[k] = (LM_TRIANGULATED if e["tri"] else 0) | (0 if e["alive"] else LM_DROPPED) | (LM_BUNDLED if e["bundled"] else 0)
        return out

    def flags(self):
        """The track set's flags as the device keeps them (membership of the last keyframe included)."""
        fl = self._flags(self.ids)
        if self.keyframes:
            member = set(int(i) for i in self.keyframes[-1]["ids"])
            for k, i in enumerate(self.ids):
                if int(i) in member:
                    fl[k] |= LM_KF_MEMBER
        return fl

    def Xw(self):
        return np.stack([self.lm[int(i)]["X"] for i in self.ids]).astype(np.float32) if len(self.ids) else np.zeros((0, 3), np.float32)

    # ---- one call of trackImage ----------------------------------------------------------------------------------
    def track(self, img):
        if self.undistort_maps is not None:
            img = O.remap_linear_u8(img, *self.undistort_maps)
        f = self.frame_counter
        self.frame_counter += 1
        self.frames.append(dict(T_wc=np.eye(4, dtype=np.float32), dT01=np.eye(4, dtype=np.float32)))
        info = dict(frame_id=f, keyframe=False, lba=None, five_point=False)
        if not self.init_done:
            if not self.got_first:
                self._first(img, f, info)
            else:
                self._second(img, f, info)
        else:
            self._steady(img, f, info)
        self._keyframe_step(f, info)
        self.I0 = img
        info["n_tracks"] = int(len(self.ids))
        return info

    def _set_pose_diff10(self, f, dT10):
        self.frames[f]["dT01"] = O.inverse_se3(dT10)  # frame.cpp:50-54

    def _first(self, I1, f, info):  # :528-561
        pts = self._detect_bucket(I1, np.ones(self.nu * self.nv, np.int32))
        self.ids = np.array([self._new_landmark(p, f) for p in pts], np.int32)
        self.pts = pts.astype(np.float32).copy()
        T_init = np.eye(4, dtype=np.float32)
        T_init[2, 3] = -1.0
        self._set_pose_diff10(f, T_init)
        self.got_first = True
        info.update(n_new=len(self.ids), n_final=0)

    def _new_points(self, I1, pts_final, f, ids, pts):
        """:621-658 / :976-1013 — updateWeightBin(lmtrack_final.pts1), extract, trackBidirection(I1, I0), new landmarks
        Landmark(p0_new, frame_prev_) + observation (p1_new, frame_curr)."""
        w = O.weight_bin_update(pts_final, self.us, self.vs, self.nu, self.nv)
        cand = self._detect_bucket(I1, w)
        if cand.shape[0] == 0:
            return ids, pts, cand, np.zeros((0, 2), np.float32), np.zeros(0, bool)
        _, p0n, m = O.track_bidirection(I1, self.I0, cand, self.win, self.max_level, self.thr[0], self.thr[1], None, self.n_threads)
        m = np.asarray(m, bool)
        T_wc = self.frames[f]["T_wc"]
        new_ids = []
        for k in np.nonzero(m)[0]:
            i = self._new_landmark(p0n[k], f - 1)
            self._add_observation(i, cand[k], f, T_wc)
            new_ids.append(i)
        ids = np.concatenate([ids, np.array(new_ids, np.int32)])
        pts = np.concatenate([pts, cand[m].astype(np.float32)])
        return ids, pts, cand, np.asarray(p0n, np.float32), m

    def _second(self, I1, f, info):  # :562-696
        Twc_prev = self.frames[f - 1]["T_wc"]
        alive = np.array([self.lm[int(i)]["alive"] for i in self.ids], bool)
        ids0, pts0 = self.ids[alive], self.pts[alive]
        _, pts1, m = O.track(self.I0, I1, pts0, self.win, self.max_level, self.thr[0], None, self.n_threads)
        m = np.asarray(m, bool)
        ids_k, p0_k, p1_k = ids0[m], pts0[m], np.asarray(pts1, np.float32)[m]
        ok, R10, t10, m5 = self.five_point(p0_k, p1_k)
        if not ok:
            raise RuntimeError("calcPose5PointsAlgorithm() is failed.")
        R10, t10 = np.asarray(R10, np.float32).reshape(3, 3), np.asarray(t10, np.float32).reshape(3)
        F10 = O.fundamental_from_pose(self.K, R10, t10)
        dist = O.sampson_distance(p0_k, p1_k, F10)
        ms = np.asarray(m5, bool) & (dist < np.float32(self.thr[3]))
        ids_f, p1_f = ids_k[ms], p1_k[ms]
        for i, p in zip(ids_f, p1_f):  # (the frame's pose is still the identity here, :602-603 before :611)
            self._add_observation(int(i), p, f, self.frames[f]["T_wc"])
        nrm = np.sqrt(np.float32(np.float32(t10[0] * t10[0]) + np.float32(np.float32(t10[1] * t10[1]) + np.float32(t10[2] * t10[2]))))
        t10 = (t10 / nrm * np.float32(1.0)).astype(np.float32)
        dT10 = _se3(R10, t10)
        dT01 = O.inverse_se3(dT10)
        self.frames[f]["T_wc"] = O.mul44(Twc_prev, dT01)
        self._set_pose_diff10(f, dT10)
        ids, pts, cand, p0n, m_new = self._new_points(I1, p1_f, f, ids_f, p1_f)
        n_rec = 0
        for i in ids:  # :660-687
            e = self.lm[int(i)]
            if not e["tri"] and e["last_par"] >= self.thres_parallax:
                T_w0 = self.frames[e["f_first"]]["T_wc"]
                T_1w = O.inverse_se3(self.frames[e["f_last"]]["T_wc"])
                ok2, X = O.mono_reconstruct(e["p_first"], e["p_last"], T_w0, T_1w, self.K, False)
                if ok2:
                    e["X"], e["tri"] = X, True
                    n_rec += 1
        self.ids, self.pts = ids, pts
        self.init_done = True
        info.update(five_point=True, n_final=int(len(ids_f)), n_new=int(m_new.sum()), n_reconstructed=n_rec, cand=cand, cand0=p0n,
                    mask_new=m_new, dT01=dT01)

    def frame(self, I1, f, sum_mode=None, tree_width=None):
        """:726-963 on the current state: the operator's outputs (no state change) and its inputs."""
        prev = self.frames[f - 1]
        Twc_prev = prev["T_wc"]
        Tcw_prev = O.inverse_se3(Twc_prev)
        dT01_prior = prev["dT01"]
        Tcw_prior = O.inverse_se3(O.mul44(Twc_prev, dT01_prior))
        fl = self._flags(self.ids)
        many = len(self.keyframes) > 5
        op = np.zeros(len(self.ids), np.uint8)
        op |= np.where(fl & LM_BUNDLED, 1, 0).astype(np.uint8)
        op |= np.where(fl & (LM_BUNDLED if many else LM_TRIANGULATED), 2, 0).astype(np.uint8)
        op |= np.where(fl & LM_DROPPED, 4, 0).astype(np.uint8)
        o = O.mono_frame(self.prm, self.I0, I1, self.pts, self.Xw(), op, Tcw_prev, Tcw_prior, dT01_prior,
                         self.sum_mode if sum_mode is None else sum_mode, self.tree_width if tree_width is None else tree_width,
                         self.border, self.n_threads)
        if o["rc"] < 0:
            raise RuntimeError(f"mono frame failed: rc {o['rc']}")
        return o, op

    def _steady(self, I1, f, info):
        o, op = self.frame(I1, f)
        prev = self.frames[f - 1]
        Twc_prev = prev["T_wc"]
        stage = o["stage"]
        if o["counts"].need_five_point:  # :909-949
            sel = stage >= 2
            p0, p1 = self.pts[sel], o["pts1"][sel]
            ok, R10, t10, mm = self.five_point(p0, p1)
            if not ok:
                raise RuntimeError("'calcPose5PointsAlgorithm()' is failed. Terminate the algorithm.")
            R10, t10 = np.asarray(R10, np.float32).reshape(3, 3), np.asarray(t10, np.float32).reshape(3)
            tp = prev["dT01"][:3, 3]
            scale = np.sqrt(np.float32(np.float32(tp[0] * tp[0]) + np.float32(np.float32(tp[1] * tp[1]) + np.float32(tp[2] * tp[2]))))
            nrm = np.sqrt(np.float32(np.float32(t10[0] * t10[0]) + np.float32(np.float32(t10[1] * t10[1]) + np.float32(t10[2] * t10[2]))))
            dT10 = _se3(R10, (np.float32(scale / nrm) * t10).astype(np.float32))
            dT01 = O.inverse_se3(dT10)
            mm = np.asarray(mm, bool)
            F10 = O.fundamental_from_pose(self.K, dT10[:3, :3], dT10[:3, 3])
            dist = O.sampson_distance(p0[mm], p1[mm], F10)
            idx = np.nonzero(sel)[0][mm][dist < np.float32(self.thr[3])]
            surv = np.zeros(len(self.ids), bool)
            surv[idx] = True
            info["five_point"] = True
        else:
            dT01 = o["dT01"].astype(np.float32)
            dT10 = O.inverse_se3(dT01)
            surv = stage == 4
        self.frames[f]["T_wc"] = O.mul44(Twc_prev, dT01)
        self._set_pose_diff10(f, dT10)
        ids_f, p1_f = self.ids[surv], o["pts1"][surv].astype(np.float32)
        T_wc = self.frames[f]["T_wc"]
        for i, p in zip(ids_f, p1_f):
            self._add_observation(int(i), p, f, T_wc)
        ids, pts, cand, p0n, m_new = self._new_points(I1, p1_f, f, ids_f, p1_f)
        info.update(frame=o, op_flags=op, n_in=int(len(self.ids)), n_final=int(len(ids_f)), n_new=int(m_new.sum()), cand=cand, cand0=p0n,
                    mask_new=m_new, dT01=dT01)
        self.ids, self.pts = ids, pts

    # ---- keyframes --------------------------------------------------------------------------------------------------
    def keyframe_rule(self, f):
        """Keyframes::checkUpdateRule, keyframes.cpp:47-126."""
        if not self.keyframes:
            return True, 0
        kf = self.keyframes[-1]
        member = set(int(i) for i in kf["ids"])
        n_tracked = sum(1 for i in self.ids if int(i) in member and self.lm[int(i)]["f_last"] == f)
        ratio = np.float32(n_tracked) / np.float32(len(kf["ids"]))
        if ratio <= self.kf_overlap:
            return True, n_tracked
        T_kw = O.inverse_se3(self.frames[kf["frame"]]["T_wc"])
        dT = O.mul44(T_kw, self.frames[f]["T_wc"])
        cos = np.float32(np.float32(np.float32(np.float32(dT[0, 0] + dT[1, 1]) + dT[2, 2]) - np.float32(1.0)) * np.float32(0.5))
        if cos >= np.float32(0.999999):
            cos = np.float32(0.999999)
        if cos <= np.float32(-0.999999):
            cos = np.float32(-0.999999)
        rot = np.arccos(cos, dtype=np.float32)
        t = dT[:3, 3]
        dtrans = np.sqrt(np.float32(np.float32(t[0] * t[0]) + np.float32(np.float32(t[1] * t[1]) + np.float32(t[2] * t[2]))))
        return bool(rot >= self.kf_rot or dtrans >= self.kf_trans), n_tracked

    def _keyframe_step(self, f, info):  # :1021-1157
        add, n_tracked = self.keyframe_rule(f)
        info["n_kf_tracked"] = n_tracked
        if not add:
            return
        info["keyframe"] = True
        kf = dict(serial=self.n_keyframes, frame=f, ids=self.ids.copy(), pts=self.pts.copy())
        self.n_keyframes += 1
        if len(self.keyframes) == self.kf_window:
            self.keyframes.pop(0)
        self.keyframes.append(kf)
        self.all_keyframes.append(kf)
        for i, p in zip(self.ids, self.pts):  # addObservationAndRelatedKeyframe(lm->getObservations().back(), frame)
            self.lm[int(i)]["kf_obs"].append((kf["serial"], f, p.copy()))
        n_rec = 0
        T_1w = O.inverse_se3(self.frames[f]["T_wc"])
        for i in self.ids:
            e = self.lm[int(i)]
            if e["alive"] and not e["tri"] and e["last_par"] >= self.thres_parallax and len(e["kf_obs"]) > 2:
                _, f0, p0 = e["kf_obs"][0]
                ok, X = O.mono_reconstruct(p0, e["kf_obs"][-1][2], self.frames[f0]["T_wc"], T_1w, self.K, True)
                if ok:
                    e["X"], e["tri"] = X, True
                    n_rec += 1
        info["n_reconstructed_kf"] = n_rec
        if self.lba:
            info["lba"] = self.local_ba()

    def keyframe_stats(self):
        out = []
        for kf in self.all_keyframes:
            X = [self.lm[int(i)]["X"] for i in kf["ids"]]
            out.append((self.frames[kf["frame"]]["T_wc"].copy(), np.stack(X).astype(np.float32) if X else np.zeros((0, 3), np.float32)))
        return out

    # ---- local BA (mono) ---------------------------------------------------------------------------------------------
    def lba_problem(self):
        win = self.keyframes
        if len(win) < 3:
            return None
        POSE_SCALE = 10.0
        inv_scale = 1.0 / POSE_SCALE
        serial_to_j = {kf["serial"]: j for j, kf in enumerate(win)}
        lm_ids = sorted({int(i) for kf in win for i in kf["ids"] if self.lm[int(i)]["tri"] and self.lm[int(i)]["alive"]})
        Twj_ref = self.frames[win[0]["frame"]]["T_wc"].astype(np.float64)
        Twj_ref[3] = (0.0, 0.0, 0.0, 1.0)
        Tjw_ref = _inverse_se3_f64(Twj_ref)
        X, obs_ptr, obs_frame, obs_px, used = [], [0], [], [], []
        for i in lm_ids:
            e = self.lm[i]
            fr, px = [], []
            for serial, _, p in e["kf_obs"]:
                j = serial_to_j.get(serial)
                if j is None:
                    continue
                fr.append(j)
                px.append(p.astype(np.float64))
            if len(fr) < 2:  # THRES_MINIMUM_SEEN
                continue
            X.append(_xform_f64(Tjw_ref, e["X"].astype(np.float64)) * inv_scale)
            obs_frame += fr
            obs_px += px
            obs_ptr.append(len(obs_frame))
            used.append(i)
        if not used:
            return None
        T_jw = []
        for kf in win:
            Tjw = O.inverse_se3(self.frames[kf["frame"]]["T_wc"]).astype(np.float64)
            Tjw[3] = (0.0, 0.0, 0.0, 1.0)
            T = _mul44_f64(Tjw, Twj_ref)
            T[:3, 3] *= inv_scale
            T_jw.append(T)
        opt = np.array([-1, -1] + list(range(len(win) - 2)), np.int32)
        return dict(T_jw=np.stack(T_jw), opt_index=opt, X=np.stack(X), obs_ptr=np.array(obs_ptr, np.int32),
                    obs_frame=np.array(obs_frame, np.int32), obs_right=np.zeros(len(obs_frame), np.uint8), obs_px=np.stack(obs_px),
                    lm_ids=used, Twj_ref=Twj_ref, Tjw_ref=Tjw_ref)

    def local_ba(self):
        p = self.lba_problem()
        if p is None:
            return None
        rc, T_new, X_new, err = O.sba_solve(p["T_jw"], p["opt_index"], p["X"], p["obs_ptr"], p["obs_frame"], p["obs_right"],
                                            p["obs_px"], self.K.astype(np.float64), None, None, 0.5, 10)
        if rc < 0:
            raise LbaError("Local BA NAN!")
        POSE_SCALE = 10.0
        for j, kf in enumerate(self.keyframes):
            if p["opt_index"][j] < 0:
                continue
            T = T_new[j].copy()
            T[:3, 3] *= POSE_SCALE
            Tjw = _mul44_f64(T, p["Tjw_ref"])
            Twj_orig = self.frames[kf["frame"]]["T_wc"].astype(np.float64)
            Twj_orig[3] = (0.0, 0.0, 0.0, 1.0)
            dT = _mul44_f64(Twj_orig, Tjw)
            tn = np.sqrt(dT[0, 3] * dT[0, 3] + (dT[1, 3] * dT[1, 3] + dT[2, 3] * dT[2, 3]))
            if tn > 50:
                raise LbaError("large update!")
            Tf = Tjw.astype(np.float32)
            Tf[3] = (0.0, 0.0, 0.0, 1.0)
            self.frames[kf["frame"]]["T_wc"] = O.inverse_se3(Tf)  # kf->setPose
        for i, x in zip(p["lm_ids"], X_new):
            xw = _xform_f64(p["Twj_ref"], x * POSE_SCALE).astype(np.float32)
            e = self.lm[i]
            e["X"], e["tri"] = xw, True
            nrm = np.sqrt(np.float32(np.float32(xw[0] * xw[0]) + np.float32(np.float32(xw[1] * xw[1]) + np.float32(xw[2] * xw[2]))))
            if nrm <= 3000:
                e["bundled"] = True
            else:
                e["alive"] = False
        return dict(rc=rc, err=err, n_lm=len(p["lm_ids"]), n_obs=int(p["obs_ptr"][-1]))
