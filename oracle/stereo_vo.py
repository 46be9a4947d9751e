"""TEST INFRASTRUCTURE ONLY (see vo_oracle.h) — CPU restatement of the closed loop of StereoVO::trackStereoImages
(core/visual_odometry/stereo_vo/stereo_vo.cpp:392-989) around the restated operators of oracle/*.c: what enters a frame
is what the previous frame left behind. PARITY UNPINNED (the reference holds no fixture and cannot be built here).

Follows
  stereo_vo.cpp:445           StereoFrame per image pair (two Frame ids, left first; frame.cpp:176-180)
  stereo_vo.cpp:465-480       [1] track set of the previous frame, [2] constant-velocity prior T_wp * dT_pc_prev
  stereo_vo.cpp:483-670       [3]-[7] the frame (oracle_misc.c: vo_ref_stereo_frame_world)
  stereo_vo.cpp:640-643       T_wc = T_wp * dT_pc_poBA; setPoseDiff10(dT_pc_poBA.inverse()) -> frame.cpp:50-54: dT01_ =
                              inverseSE3_f(dT10): the next prior is inverseSE3_f(Matrix4f::inverse(dT)), not dT itself
  stereo_vo.cpp:691-711       [10] updateWeightBin(lmtrack_final.pts_l1), extractORBwithBinning_fast, trackBidirection
  stereo_vo.cpp:714-739       new landmarks: mask_new && DLT depths positive, appended to lmtrack_final, NOT triangulated
  stereo_vo.cpp:752           setStereoPtsSeenAndRelatedLandmarks: the next frame's track set
  stereo_vo.cpp:755-797       keyframe rule (keyframes.cpp:217-303) and reconstruction of the first lmtrack_final.n_pts
                              entries — n_pts is the constructor's count (landmark.cpp:314), the landmarks pushed in [10]
                              are not reconstructed at this keyframe
  stereo_vo.cpp:802           localBundleAdjustmentSparseSolver_Stereo (motion_estimator.cpp:1207-1340,
                              sparse_ba_parameters.h:292-466, sparse_bundle_adjustment.cpp:624-722)
  stereo_vo.cpp:842-949       the very first pair
  stereo_vo.cpp:983-985       prev <- curr
One thing cannot be restated: SparseBAParameters walks an std::unordered_set<LandmarkPtr> (sparse_ba_parameters.h:333),
so the order of the landmarks inside the local BA — and with it the last bits of its double-precision sums — depends on
heap addresses in the reference. Here the landmarks are taken by ascending id.
"""
import numpy as np

from . import oracle as O

LM_TRIANGULATED, LM_DROPPED, LM_KF_MEMBER = 1, 2, 4
D2R = np.float32(3.14159265358979323846 / 180.0)  # define_macro.h D2R as the reference multiplies it into a float


class LbaError(RuntimeError):
    pass


class StereoVORef:
    """One image stream. track(L, R) = StereoVO::trackStereoImages; the state is plain arrays so that a test can
    compare every piece of it with the device's."""

    def __init__(self, width, height, Kl, Kr, T_lr, n_bins_u, n_bins_v, thres_fast=15, win=21, max_level=6, thres_err=80.0,
                 thres_bidir=0.5, thres_poseba=3.0, kf_overlap=0.6, kf_rot_deg=15.0, kf_trans=10.0, kf_window=9, lba=True,
                 ic_border=O.IC_REFERENCE, sum_mode=O.SUM_SEQ, tree_width=0, n_threads=1, rectify_maps=None, thres_sampson=60.0):
        """rectify_maps = ((map_u, map_v) of the left camera, (map_u, map_v) of the right one): flagDoUndistortion
        (stereo_vo.cpp:414-421) — every pair is remapped first; Kl / Kr / T_lr are then the rectified camera."""
        self.rectify_maps = rectify_maps
        self.W, self.H = width, height
        self.Kl, self.Kr = np.asarray(Kl, np.float32), np.asarray(Kr, np.float32)
        self.T_lr = np.asarray(T_lr, np.float32).reshape(4, 4)
        self.T_rl = O.inverse_se3(self.T_lr)
        self.nu, self.nv, self.thres_fast = n_bins_u, n_bins_v, thres_fast
        self.win, self.max_level = win, max_level
        self.thr = (thres_err, thres_bidir, thres_poseba)
        self.prm = O.make_stereo_params(width, height, win, max_level, thres_err, thres_bidir, thres_poseba, Kl, Kr, T_lr, thres_sampson)
        self.us, self.vs, self.iu, self.iv = O.weight_bin_init(width, height, n_bins_u, n_bins_v)
        self.kf_overlap = np.float32(kf_overlap)
        self.kf_rot = np.float32(np.float32(kf_rot_deg) * D2R)
        self.kf_trans = np.float32(kf_trans)
        self.kf_window, self.lba = kf_window, lba
        self.border, self.sum_mode, self.tree_width, self.n_threads = ic_border, sum_mode, tree_width, n_threads
        self.first = True
        self.landmark_counter = self.frame_counter = 0
        self.I0 = None
        self.T_wp = np.eye(4, dtype=np.float32)
        self.dT01 = np.eye(4, dtype=np.float32)
        self.ids = np.zeros(0, np.int32)
        self.pts_l, self.pts_r = np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32)
        self.Xw, self.flags = np.zeros((0, 3), np.float32), np.zeros(0, np.uint8)
        self.keyframes = []   # window: dict(serial, frame_id, T_wc, ids, pts_l, pts_r)
        self.all_keyframes = []  # all_stkeyframes_: the same dicts, never dropped
        self.n_keyframes = 0  # all keyframes ever
        self.n_kf_lms = 0
        self.lm = {}          # landmarks seen on a keyframe: id -> dict(X, tri, alive, obs=[(serial, pl, pr)])
        self.poses = []       # T_wc per frame (as stat_.stats_frame keeps it: after the local BA)

    # ---- helpers ------------------------------------------------------------------------------------------------
    def _detect_bucket(self, L, weight):
        d = O.orb_detect(L, self.thres_fast)
        pts, _ = O.bucket_argmax(d["xy"], d["response"], self.iu, self.iv, self.nu, self.nv, weight)
        return pts

    def _new_frame_ids(self):
        a = self.frame_counter
        self.frame_counter += 2
        return a, a + 1

    def _new_landmarks(self, n):
        a = self.landmark_counter
        self.landmark_counter += n
        return np.arange(a, a + n, dtype=np.int32)

    # ---- one call of trackStereoImages -----------------------------------------------------------------------------
    def track(self, L, R):
        if self.rectify_maps is not None:  # rectifyStereoImages + convertTo(CV_8UC1)
            L = O.remap_linear_u8(L, *self.rectify_maps[0])
            R = O.remap_linear_u8(R, *self.rectify_maps[1])
        fid_l, _ = self._new_frame_ids()
        info = dict(frame_id=fid_l, keyframe=False, lba=None)
        if self.first:
            self._first(L, R, info)
        else:
            self._steady(L, R, info)
        self.poses.append(self.T_wp.copy())
        self.I0 = L
        return info

    def _first(self, L, R, info):  # stereo_vo.cpp:842-949
        cand = self._detect_bucket(L, np.ones(self.nu * self.nv, np.int32))  # resetWeightBin
        pr, m = O.track_bidirection(L, R, cand, self.win, self.max_level, self.thr[0], self.thr[1], None, self.n_threads)[1:] \
            if cand.shape[0] else (np.zeros((0, 2), np.float32), np.zeros(0, bool))
        m = np.asarray(m, bool)
        self.pts_l, self.pts_r = cand[m].copy(), np.asarray(pr)[m].copy()
        n = self.pts_l.shape[0]
        self.ids = self._new_landmarks(n)
        X, st = O.keyframe_reconstruct(self.pts_l, self.pts_r, self.T_rl, self.Kl, self.Kr, None, np.zeros((n, 3), np.float32))
        self.Xw = X
        self.flags = np.where(st, LM_TRIANGULATED, 0).astype(np.uint8)
        self.T_wp = np.eye(4, dtype=np.float32)
        self.dT01 = O.inverse_se3(np.eye(4, dtype=np.float32))  # setPoseDiff10(Identity)
        self.first = False
        info.update(n_tracks=n, n_new=n, n_reconstructed=int(st.sum()))

    def frame(self, L, R, sum_mode=None, tree_width=None):
        """[2]-[7] on the current state; returns the operator's outputs (no state change)."""
        T_wc_prior = O.mul44(self.T_wp, self.dT01)
        T_cw_prior = O.inverse_se3(T_wc_prior)
        T_pw = O.inverse_se3(self.T_wp)
        o = O.stereo_frame(self.prm, self.I0, L, R, self.pts_l, self.pts_r, self.Xw, self.dT01, np.zeros((0, 2), np.float32),
                           self.sum_mode if sum_mode is None else sum_mode, self.tree_width if tree_width is None else tree_width,
                           self.border, self.n_threads, lm_flags=(self.flags & 3).astype(np.uint8), T_pw=T_pw,
                           T_cw_prior=T_cw_prior)
        if o["rc"] != 0:
            raise RuntimeError(f"stereo frame failed: rc {o['rc']}")
        return o

    def new_points(self, L, R, pts_final):
        """[10]: (candidates, right pixels, trackBidirection mask, accept mask of :716-725)."""
        w = O.weight_bin_update(pts_final, self.us, self.vs, self.nu, self.nv)
        cand = self._detect_bucket(L, w)
        if cand.shape[0] == 0:
            e2 = np.zeros((0, 2), np.float32)
            return e2, e2, np.zeros(0, bool), np.zeros(0, bool)
        _, pr, m = O.track_bidirection(L, R, cand, self.win, self.max_level, self.thr[0], self.thr[1], None, self.n_threads)
        m = np.asarray(m, bool)
        acc, _ = O.new_landmark_accept(cand, pr, m, self.T_rl, self.Kl, self.Kr)
        return cand, np.asarray(pr), m, acc

    def keyframe_rule(self, n_tracked, T_wc):
        """StereoKeyframes::checkUpdateRule, keyframes.cpp:217-303."""
        if not self.keyframes:
            return True
        ratio = np.float32(n_tracked) / np.float32(self.n_kf_lms)
        if ratio <= self.kf_overlap:
            return True
        T_kw = O.inverse_se3(self.keyframes[-1]["T_wc"])
        dT = O.mul44(T_kw, T_wc)
        cos = np.float32(np.float32(np.float32(np.float32(dT[0, 0] + dT[1, 1]) + dT[2, 2]) - np.float32(1.0)) * np.float32(0.5))
        if float(cos) >= 0.999999:
            cos = np.float32(0.999999)
        if float(cos) <= -0.999999:
            cos = np.float32(-0.999999)
        rot = np.arccos(cos, dtype=np.float32)
        t = dT[:3, 3]
        dtrans = np.sqrt(np.float32(np.float32(t[0] * t[0]) + np.float32(np.float32(t[1] * t[1]) + np.float32(t[2] * t[2]))))
        return bool(rot >= self.kf_rot or dtrans >= self.kf_trans)

    def _steady(self, L, R, info):
        o = self.frame(L, R)
        dT = o["dT"].astype(np.float32)
        T_wc = O.mul44(self.T_wp, dT)
        surv = o["stage"] == 4
        pl, pr = o["pts_l1"][surv], o["pts_r1"][surv]
        ids, Xw, fl = self.ids[surv], self.Xw[surv], self.flags[surv]
        n_surv = int(surv.sum())
        cand, cand_r, m_new, acc = self.new_points(L, R, pl)
        n_acc = int(acc.sum())
        new_ids = self._new_landmarks(n_acc)
        pl, pr = np.concatenate([pl, cand[acc]]), np.concatenate([pr, cand_r[acc]])
        ids = np.concatenate([ids, new_ids])
        Xw = np.concatenate([Xw, np.zeros((n_acc, 3), np.float32)])
        fl = np.concatenate([fl, np.zeros(n_acc, np.uint8)])
        n_tracked = int(((fl[:n_surv] & LM_KF_MEMBER) != 0).sum())
        info.update(frame=o, dT=dT, n_in=int(self.ids.shape[0]), n_surv=n_surv, n_new=n_acc, n_kf_tracked=n_tracked,
                    cand=cand, cand_r=cand_r, mask_new=m_new, accept=acc)
        self.dT01 = O.inverse_se3(O.inverse4x4(dT))
        if self.keyframe_rule(n_tracked, T_wc):
            info["keyframe"] = True
            X2, st = O.keyframe_reconstruct(pl[:n_surv], pr[:n_surv], self.T_rl, self.Kl, self.Kr, T_wc, Xw[:n_surv])
            Xw[:n_surv] = X2
            fl[:n_surv] |= np.where(st, LM_TRIANGULATED, 0).astype(np.uint8)
            fl |= LM_KF_MEMBER
            info["n_reconstructed"] = int(st.sum())
            kf = dict(serial=self.n_keyframes, frame_id=info["frame_id"], T_wc=T_wc.copy(), ids=ids.copy(), pts_l=pl.copy(),
                      pts_r=pr.copy())
            self.n_keyframes += 1
            if len(self.keyframes) == self.kf_window:
                self.keyframes.pop(0)
            self.keyframes.append(kf)
            self.all_keyframes.append(kf)
            self.n_kf_lms = int(ids.shape[0])
            for k in range(ids.shape[0]):  # the landmark table: state as of this keyframe + the observation on it
                e = self.lm.setdefault(int(ids[k]), dict(X=None, tri=False, alive=True, obs=[]))
                e["X"], e["tri"] = Xw[k].copy(), bool(fl[k] & LM_TRIANGULATED)
                e["obs"].append((kf["serial"], pl[k].copy(), pr[k].copy()))
            if self.lba:
                info["lba"] = self.local_ba()
                for k in range(ids.shape[0]):  # what the BA did to the live landmarks
                    e = self.lm[int(ids[k])]
                    Xw[k] = e["X"]
                    if e["tri"]:
                        fl[k] |= LM_TRIANGULATED
                    if not e["alive"]:
                        fl[k] |= LM_DROPPED
                T_wc = self.keyframes[-1]["T_wc"].copy()
        self.ids, self.pts_l, self.pts_r, self.Xw, self.flags = ids, pl, pr, Xw, fl
        self.T_wp = T_wc
        info["n_tracks"] = int(ids.shape[0])

    def keyframe_stats(self):
        """AlgorithmStatistics::stats_keyframe as trackStereoImages refreshes it at every keyframe (stereo_vo.cpp:805-821):
        every keyframe so far with its CURRENT pose and the current 3-D points of its related landmarks."""
        out = []
        for kf in self.all_keyframes:
            X = [self.lm[int(i)]["X"] for i in kf["ids"]]
            out.append((kf["T_wc"].copy(), np.stack(X).astype(np.float32) if X else np.zeros((0, 3), np.float32)))
        return out

    # ---- local BA over the keyframe window ---------------------------------------------------------------------
    def lba_problem(self):
        """SparseBAParameters::setPosesAndPoints (sparse_ba_parameters.h:292-466) as flat arrays; None when the window
        is too short (motion_estimator.cpp:1249)."""
        win = self.keyframes
        if len(win) < 3:
            return None
        POSE_SCALE = 10.0
        inv_scale = 1.0 / POSE_SCALE
        serial_to_j = {kf["serial"]: j for j, kf in enumerate(win)}
        lm_ids = sorted({int(i) for kf in win for i in kf["ids"] if self.lm[int(i)]["tri"] and self.lm[int(i)]["alive"]})
        Twj_ref = win[0]["T_wc"].astype(np.float64)
        Twj_ref[3] = (0.0, 0.0, 0.0, 1.0)
        Tjw_ref = _inverse_se3_f64(Twj_ref)
        X, obs_ptr, obs_frame, obs_right, obs_px, used = [], [0], [], [], [], []
        for i in lm_ids:
            e = self.lm[i]
            fr, rt, px = [], [], []
            for serial, pl, pr in e["obs"]:
                j = serial_to_j.get(serial)
                if j is None:
                    continue
                fr += [j, j]
                rt += [0, 1]
                px += [pl.astype(np.float64), pr.astype(np.float64)]
            if len(fr) < 2:  # THRES_MINIMUM_SEEN
                continue
            x = _xform_f64(Tjw_ref, e["X"].astype(np.float64)) * inv_scale  # warpToRef, scalingPoint
            X.append(x)
            obs_frame += fr
            obs_right += rt
            obs_px += px
            obs_ptr.append(len(obs_frame))
            used.append(i)
        if not used:
            return None
        T_jw = []
        for kf in win:
            Tjw = O.inverse_se3(kf["T_wc"]).astype(np.float64)  # getPoseInv()
            Tjw[3] = (0.0, 0.0, 0.0, 1.0)
            T = _mul44_f64(Tjw, Twj_ref)  # changeInvPoseWorldToRef
            T[:3, 3] *= inv_scale         # scalingPose
            T_jw.append(T)
        opt = np.array([-1, -1] + list(range(len(win) - 2)), np.int32)
        T_s = self.T_lr.astype(np.float64)
        T_s[3] = (0.0, 0.0, 0.0, 1.0)
        T_s[:3, 3] *= inv_scale
        return dict(T_jw=np.stack(T_jw), opt_index=opt, X=np.stack(X), obs_ptr=np.array(obs_ptr, np.int32),
                    obs_frame=np.array(obs_frame, np.int32), obs_right=np.array(obs_right, np.uint8),
                    obs_px=np.stack(obs_px), lm_ids=used, Twj_ref=Twj_ref, Tjw_ref=Tjw_ref, T_lr_scaled=T_s)

    def local_ba(self):
        p = self.lba_problem()
        if p is None:
            return None
        rc, T_new, X_new, err = O.sba_solve(p["T_jw"], p["opt_index"], p["X"], p["obs_ptr"], p["obs_frame"], p["obs_right"],
                                            p["obs_px"], self.Kl.astype(np.float64), self.Kr.astype(np.float64),
                                            p["T_lr_scaled"], 0.5, 10)
        if rc < 0:
            raise LbaError("Local BA NAN!")
        self.lba_finish(p, T_new, X_new)
        return dict(rc=rc, err=err, n_lm=len(p["lm_ids"]), n_obs=int(p["obs_ptr"][-1]))

    def lba_finish(self, p, T_new, X_new):
        """sparse_bundle_adjustment.cpp:624-722: poses and points back into the frames / landmarks."""
        POSE_SCALE = 10.0
        win = self.keyframes
        for j, kf in enumerate(win):
            if p["opt_index"][j] < 0:
                continue
            T = T_new[j].copy()
            T[:3, 3] *= POSE_SCALE                   # recoverOriginalScalePose
            Tjw = _mul44_f64(T, p["Tjw_ref"])        # changeInvPoseRefToWorld
            Twj_orig = kf["T_wc"].astype(np.float64)
            Twj_orig[3] = (0.0, 0.0, 0.0, 1.0)
            dT = _mul44_f64(Twj_orig, Tjw)
            tn = np.sqrt(dT[0, 3] * dT[0, 3] + (dT[1, 3] * dT[1, 3] + dT[2, 3] * dT[2, 3]))
            if tn > 50:
                raise LbaError("large update!")
            Tf = Tjw.astype(np.float32)
            Tf[3] = (0.0, 0.0, 0.0, 1.0)
            kf["T_wc"] = O.inverse_se3(Tf)           # kf->setPose(inverseSE3_f(Tjw_update_float))
        for i, x in zip(p["lm_ids"], X_new):
            xw = _xform_f64(p["Twj_ref"], x * POSE_SCALE).astype(np.float32)  # recoverOriginalScalePoint, warpToWorld
            e = self.lm[i]
            e["X"], e["tri"] = xw, True              # set3DPoint
            nrm = np.sqrt(np.float32(np.float32(xw[0] * xw[0]) + np.float32(np.float32(xw[1] * xw[1]) + np.float32(xw[2] * xw[2]))))
            if not nrm <= 3000:
                e["alive"] = False                   # setDead


def _mul44_f64(A, B):
    """Matrix4d * Matrix4d, column-packet order (left to right over k)."""
    C_ = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            r = A[i, 0] * B[0, j]
            for k in range(1, 4):
                r = A[i, k] * B[k, j] + r
            C_[i, j] = r
    return C_


def _xform_f64(T, X):
    """T.block<3,3>(0,0) * X + T.block<3,1>(0,3) in double, 3-term redux e0 + (e1 + e2)."""
    return np.array([T[i, 0] * X[0] + (T[i, 1] * X[1] + T[i, 2] * X[2]) + T[i, 3] for i in range(3)])


def _inverse_se3_f64(T):
    """geometry::inverseSE3 (geometry_library.cpp:561-567): [R^T, -R^T t]."""
    Ti = np.eye(4)
    Rt = T[:3, :3].T
    Ti[:3, :3] = Rt
    t = T[:3, 3]
    for i in range(3):
        Ti[i, 3] = (-Rt[i, 0]) * t[0] + ((-Rt[i, 1]) * t[1] + (-Rt[i, 2]) * t[2])
    return Ti
