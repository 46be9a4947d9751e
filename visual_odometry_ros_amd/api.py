"""Python host-side mirror of the reference operator interface, on the C ABI.

Class and method names follow the reference
(core/visual_odometry/feature_tracker.h:44-104, motion_estimator.h:117-120,
feature_extractor.h descriptorDistance); argument meaning and error behaviour
match: a reference `throw std::runtime_error` surfaces as VoError, a reference
`return false` as False. numpy arrays stand in for PixelVec / PointVec /
MaskVec; matrices are row-major numpy arrays.

Everything here calls libvo_hip.so. Nothing falls back to numpy or the oracle.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _capi
from ._capi import (FIVE_POINT_FN, BinParams, FrameCounts, GnInfo, MonoCounts, MonoParams, MvoFrameInfo, MvoParams, OrbParams,
                    SbaProblem, StereoParams, SvoFrameInfo, SvoParams, VoConfig, VoError)

KLT_USE_INITIAL_FLOW = 4
GN_CORE, GN_STANDALONE = 0, 1


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class Context:
    """Owns one vo_ctx: a HIP stream, image-pyramid slots and point buffers."""

    def __init__(self, device=0, max_width=1241, max_height=376, max_points=4096, n_slots=4,
                 max_level=6):
        self.lib = _capi.load()
        self._h = C.c_void_p()
        cfg = VoConfig(device, max_width, max_height, max_points, n_slots, max_level)
        rc = self.lib.vo_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self.lib.vo_last_error(self._h if self._h else None).decode()
            if self._h:
                self.lib.vo_destroy(self._h)
                self._h = C.c_void_p()
            raise VoError(rc, msg)
        self.cfg = cfg
        self._children = weakref.WeakSet()  # objects that hold device state of this context (StereoVO): closed before it
        # test / measurement switches (include/vo_hip.h: vo_debug_set). Neither the library nor this mirror reads them from
        # the environment in normal use: only a process started with VO_TEST_SWITCHES=1 (the test-suite's child processes,
        # tools/) has the variables below applied to the contexts it creates, and `debug_switches` says which are active
        # (bench.py reports them).
        self.debug_switches = {}
        if os.environ.get("VO_TEST_SWITCHES") == "1":
            for key, name in ((0, "VO_DEBUG_FAIL_JOIN"), (1, "VO_CONC_GRID"), (2, "VO_SBA_LDS_SOLVE"), (3, "VO_DEBUG_SKIP_DETECT"),
                              (5, "VO_MVO_HOST_ADVANCE"), (6, "VO_STAGED_DETECT")):
                v = os.environ.get(name)
                if v:
                    self.debug_set(key, int(v) if v.lstrip("-").isdigit() else 1)
                    self.debug_switches[name] = v

    DBG_FAIL_JOIN, DBG_CONC_GRID, DBG_SBA_LDS_SOLVE, DBG_SKIP_DETECT, OPT_POLL_YIELD, DBG_MVO_HOST_ADVANCE, DBG_STAGED_DETECT = 0, 1, 2, 3, 4, 5, 6

    def debug_set(self, key, value):
        self.check(self.lib.vo_debug_set(self._h, int(key), int(value)))

    def allocation_count(self):
        """Device + pinned allocations made for this context so far (a steady-state frame makes none)."""
        n = C.c_longlong(0)
        self.check(self.lib.vo_debug_allocation_count(self._h, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            for ch in list(getattr(self, "_children", ())):  # (a StereoVO left open by an exception must not outlive its context)
                ch.close()
            self.lib.vo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, rc):
        if rc < 0:
            raise VoError(rc, self.lib.vo_last_error(self._h).decode())
        return rc

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return self.lib.vo_stream(self._h)

    def synchronize(self):
        self.check(self.lib.vo_synchronize(self._h))

    # ---- images ---------------------------------------------------------
    def set_image(self, slot, img):
        img = _u8(img)
        assert img.ndim == 2
        self.check(self.lib.vo_set_image(self._h, slot, _p(img, C.c_uint8), img.shape[1], img.shape[0],
                                         img.strides[0]))

    def set_image_device(self, slot, dev_ptr, width, height, stride):
        self.check(self.lib.vo_set_image_device(self._h, slot, C.c_void_p(dev_ptr), width, height, stride))

    def set_stereo_pair_device(self, slot_l, ptr_l, slot_r, ptr_r, width, height, stride):
        rc = self.lib.vo_set_stereo_pair_device(self._h, slot_l, ptr_l, slot_r, ptr_r, width, height, stride)
        if rc < 0:
            self.check(rc)

    def set_image_rectified(self, slot, img, cam=0):
        img = _u8(img)
        assert img.ndim == 2
        self.check(self.lib.vo_set_image_rectified(self._h, slot, _p(img, C.c_uint8), img.shape[1], img.shape[0],
                                                   img.strides[0], cam))

    def set_image_rectified_device(self, slot, dev_ptr, width, height, stride, cam=0):
        self.check(self.lib.vo_set_image_rectified_device(self._h, slot, C.c_void_p(dev_ptr), width, height, stride,
                                                          cam))

    def set_stereo_pair_rectified_device(self, slot_l, ptr_l, slot_r, ptr_r, width, height, stride):
        self.check(self.lib.vo_set_stereo_pair_rectified_device(self._h, slot_l, C.c_void_p(ptr_l), slot_r,
                                                                C.c_void_p(ptr_r), width, height, stride))

    def set_pyramid_window_hint(self, win):
        self.check(self.lib.vo_set_pyramid_window_hint(self._h, win))

    def set_ingest_side_stream(self, on=True):
        """Image ingestion (copies + pyramid chains) on the side stream, concurrent with the frame in flight."""
        self.check(self.lib.vo_set_ingest_side_stream(self._h, int(bool(on))))

    def set_stereo_pair_host_async(self, slot_l, ptr_l, slot_r, ptr_r, width, height, stride):
        """Host images (addresses; pinned memory for a true asynchronous copy) without a host synchronisation."""
        rc = self.lib.vo_set_stereo_pair_host_async(self._h, slot_l, ptr_l, slot_r, ptr_r, width, height, stride)
        if rc < 0:
            self.check(rc)

    def frame_recoveries(self):
        """frames re-issued after a device-side join time-out (0 in normal operation)"""
        return self.lib.vo_stereo_frame_recoveries(self._h)

    def swap_slots(self, a, b):
        self.check(self.lib.vo_swap_slots(self._h, a, b))

    def pyramid_levels(self, w, h, win, max_level):
        return self.lib.vo_pyramid_levels(w, h, win, max_level)

    def get_level(self, slot, level):
        w, h = C.c_int(), C.c_int()
        buf = np.zeros((self.cfg.max_height, self.cfg.max_width), np.uint8).reshape(-1)
        self.check(self.lib.vo_get_level(self._h, slot, level, _p(buf, C.c_uint8), C.byref(w), C.byref(h)))
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()

    # ---- profiling --------------------------------------------------------
    def profile_enable(self, max_records):
        self.check(self.lib.vo_profile_enable(self._h, max_records))

    def profile_set_classes(self, mask):
        self.check(self.lib.vo_profile_set_classes(self._h, C.c_uint(mask)))

    def profile_reset(self):
        self.check(self.lib.vo_profile_reset(self._h))

    def profile_get(self, cls):
        n, ms = C.c_int(), C.c_double()
        self.check(self.lib.vo_profile_get(self._h, cls, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class ImageSlots:
    """What stands between the reference's cv::Mat arguments and the context's image slots (the C++ classes keep the same
    table, feature_tracker.h: slot_for): an image is identified by (buffer address, size, caller-supplied frame stamp);
    a slot that already holds it is reused — upload and pyramid once per image and frame instead of once per operator
    call (the reference rebuilds 8 pyramids per stereo frame) — otherwise the least recently used slot takes it.
    stamp None: no identity, always uploaded (the reference's behaviour)."""

    def __init__(self, ctx, slots):
        self.ctx, self.slots = ctx, list(slots)
        self.key = {s: None for s in self.slots}
        self.clock, self.used = 0, {s: 0 for s in self.slots}
        self.uploads = 0

    def get(self, img, stamp=None, avoid=()):
        img = _u8(img)
        k = None if stamp is None else (img.ctypes.data, img.shape, img.strides[0], stamp)
        self.clock += 1
        if k is not None:
            for s in self.slots:
                if self.key[s] == k:
                    self.used[s] = self.clock
                    return s
        s = min((q for q in self.slots if q not in avoid), key=lambda q: self.used[q])
        self.ctx.set_image(s, img)
        self.uploads += 1
        self.key[s], self.used[s] = k, self.clock
        return s


class FeatureTracker:
    """Mirror of the reference FeatureTracker. Images live in context slots
    (set with Context.set_image) instead of cv::Mat arguments; everything else
    keeps the reference argument order."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib

    def calcOpticalFlowPyrLK(self, slot0, slot1, pts0, pts1=None, win=21, max_level=3, flags=0,
                             max_iter=30, eps=0.01, min_eig=1e-4):
        pts0 = _f32(pts0).reshape(-1, 2)
        n = pts0.shape[0]
        p1 = np.zeros((max(n, 1), 2), np.float32)
        if pts1 is not None:
            p1[:n] = _f32(pts1).reshape(-1, 2)
        status = np.zeros(max(n, 1), np.uint8)
        err = np.zeros(max(n, 1), np.float32)
        rc = self.ctx.check(self.lib.vo_klt_track(
            self.ctx.handle, slot0, slot1, _p(pts0), _p(p1), n, win, max_level, flags, max_iter,
            C.c_double(eps), C.c_float(min_eig), _p(status, C.c_uint8), _p(err)))
        return rc, p1[:n], status[:n], err[:n]

    def _run(self, fn, slot0, slot1, pts0, pts_track, mask_valid, win, max_level, extra):
        pts0 = _f32(pts0).reshape(-1, 2)
        n = pts0.shape[0]
        pt = np.zeros((max(n, 1), 2), np.float32)
        if pts_track is not None:
            pts_track = _f32(pts_track).reshape(-1, 2)
            if pts_track.shape[0] != n:
                raise VoError(-4, "pts_track.size() != pts0.size()")
            pt[:n] = pts_track
        m = np.ones(max(n, 1), np.uint8)
        if mask_valid is not None:
            m[:n] = np.asarray(mask_valid, np.uint8)
        self.ctx.check(fn(self.ctx.handle, slot0, slot1, _p(pts0), n, win, max_level, *extra, _p(pt),
                          _p(m, C.c_uint8)))
        return pt[:n], m[:n].astype(bool)

    def track(self, slot0, slot1, pts0, window_size, max_pyr_lvl, thres_err, mask_valid=None):
        return self._run(self.lib.vo_track, slot0, slot1, pts0, None, mask_valid, window_size,
                         max_pyr_lvl, (C.c_float(thres_err),))

    def trackBidirection(self, slot0, slot1, pts0, window_size, max_pyr_lvl, thres_err,
                         thres_bidirection, mask_valid=None):
        return self._run(self.lib.vo_track_bidirection, slot0, slot1, pts0, None, mask_valid,
                         window_size, max_pyr_lvl, (C.c_float(thres_err), C.c_float(thres_bidirection)))

    def trackBidirectionWithPrior(self, slot0, slot1, pts0, window_size, max_pyr_lvl, thres_err,
                                  thres_bidirection, pts_track, mask_valid=None):
        return self._run(self.lib.vo_track_bidirection_with_prior, slot0, slot1, pts0, pts_track,
                         mask_valid, window_size, max_pyr_lvl,
                         (C.c_float(thres_err), C.c_float(thres_bidirection)))

    def trackWithPrior(self, slot0, slot1, pts0, window_size, max_pyr_lvl, thres_err, pts_track,
                       mask_valid=None):
        return self._run(self.lib.vo_track_with_prior, slot0, slot1, pts0, pts_track, mask_valid,
                         window_size, max_pyr_lvl, (C.c_float(thres_err),))

    def calcPrior(self, pts0, Xw, Tw1, K):
        pts0, Xw = _f32(pts0).reshape(-1, 2), _f32(Xw).reshape(-1, 3)
        Tw1, K = _f32(Tw1).reshape(16), _f32(K).reshape(9)
        out = np.zeros_like(pts0)
        self.ctx.check(self.lib.vo_calc_prior(self.ctx.handle, _p(pts0), pts0.shape[0], _p(Xw),
                                              Xw.shape[0], _p(Tw1), _p(K), _p(out)))
        return out

    def trackWithScale(self, slot0, slot1, pts0, scale_est, pts_track, mask_valid=None,
                       strict_border=True):
        pts0 = _f32(pts0).reshape(-1, 2)
        n = pts0.shape[0]
        pt = _f32(pts_track).reshape(-1, 2).copy()
        if pt.shape[0] != n:
            raise VoError(-4, "pts_track.size() != pts0.size()")  # feature_tracker.cpp:282-283
        scale = _f32(scale_est)
        m = np.ones(max(n, 1), np.uint8)
        if mask_valid is not None:
            m[:n] = np.asarray(mask_valid, np.uint8)
        if n == 0:
            return pt, m[:0].astype(bool)
        self.ctx.check(self.lib.vo_track_with_scale(self.ctx.handle, slot0, slot1, _p(pts0), _p(scale),
                                                    n, _p(pt), _p(m, C.c_uint8), int(strict_border)))
        return pt, m[:n].astype(bool)


class MotionEstimator:
    """Mirror of the reference MotionEstimator's pose-only BA entry points."""

    def __init__(self, ctx, is_stereo_mode=False, T_left2right=None):
        self.ctx = ctx
        self.lib = ctx.lib
        self.is_stereo_mode_ = bool(is_stereo_mode)
        self.T_left2right_ = np.eye(4, dtype=np.float32) if T_left2right is None else _f32(T_left2right)

    def poseOnlyBundleAdjustment(self, X, pts1, K, thres_reproj_outlier, R01, t01, variant=GN_CORE):
        X, pts1 = _f32(X).reshape(-1, 3), _f32(pts1).reshape(-1, 2)
        if X.shape[0] != pts1.shape[0]:  # motion_estimator.cpp:669-670
            raise VoError(-4, "In 'poseOnlyBundleAdjustment()': X.size() != pts1.size().")
        n = X.shape[0]
        K = _f32(K)
        R = _f32(R01).reshape(9).copy()
        t = _f32(t01).reshape(3).copy()
        mask = np.zeros(max(n, 1), np.uint8)
        info = GnInfo()
        rc = self.ctx.check(self.lib.vo_gn_pose_mono(
            self.ctx.handle, _p(X), _p(pts1), n, _p(K), int(thres_reproj_outlier), _p(R), _p(t),
            _p(mask, C.c_uint8), variant, C.byref(info)))
        return bool(rc), R.reshape(3, 3), t, mask[:n].astype(bool), info

    def poseOnlyBundleAdjustment_Stereo(self, X, pts_l1, pts_r1, Kl, Kr, T_lr, thres_reproj_outlier, T01):
        if not self.is_stereo_mode_:  # motion_estimator.cpp:866-867
            raise VoError(-1, "In 'poseOnlyBundleAdjustment_Stereo()', is_stereo_mode_ == false")
        X = _f32(X).reshape(-1, 3)
        pts_l1, pts_r1 = _f32(pts_l1).reshape(-1, 2), _f32(pts_r1).reshape(-1, 2)
        if X.shape[0] != pts_l1.shape[0] or X.shape[0] != pts_r1.shape[0]:  # :872-873
            raise VoError(-4, "In 'poseOnlyStereoBundleAdjustment()': size mismatch")
        n = X.shape[0]
        Kl, Kr, T_lr = _f32(Kl), _f32(Kr), _f32(T_lr).reshape(16)
        T = _f32(T01).reshape(16).copy()
        mask = np.zeros(max(n, 1), np.uint8)
        info = GnInfo()
        rc = self.ctx.check(self.lib.vo_gn_pose_stereo(
            self.ctx.handle, _p(X), _p(pts_l1), _p(pts_r1), n, _p(Kl), _p(Kr), _p(T_lr),
            C.c_float(thres_reproj_outlier), _p(T), _p(mask, C.c_uint8), C.byref(info)))
        return bool(rc), T.reshape(4, 4), mask[:n].astype(bool), info


    def setThres1p(self, thres_1p):  # motion_estimator.cpp:655-658 (consumed by the host-side 1-point RANSAC)
        self.thres_1p_ = float(thres_1p)

    def setThres5p(self, thres_5p):  # :660-663
        self.thres_5p_ = float(thres_5p)

    # ---- epipolar gates (motion_estimator.cpp:538-653) ----
    @staticmethod
    def fundamentalFromPose(K, R10, t10):
        """F10 = Kinv^T [t10]x R10 Kinv (motion_estimator.cpp:551-552) in float32, K = (fx, fy, cx, cy).
        3x3 products associate as Eigen's unrolled redux does: e0 + (e1 + e2)."""
        f = np.float32
        K, R, t = _f32(K).reshape(4), _f32(R10).reshape(3, 3), _f32(t10).reshape(3)
        fxi, fyi = f(1.0) / K[0], f(1.0) / K[1]
        Kinv = np.array([[fxi, 0, -K[2] * fxi], [0, fyi, -K[3] * fyi], [0, 0, 1]], f)
        S = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], f)

        def mul(A, B):
            Cm = np.zeros((3, 3), f)
            for i in range(3):
                for j in range(3):
                    Cm[i, j] = f(A[i, 0] * B[0, j]) + f(f(A[i, 1] * B[1, j]) + f(A[i, 2] * B[2, j]))
            return Cm
        return mul(mul(Kinv.T.copy(), mul(S, R)), Kinv)

    def _epi(self, fn, pts0, pts1, F10):
        pts0, pts1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
        if pts0.shape[0] != pts1.shape[0]:  # motion_estimator.cpp:541-542, :575-576, :625-626
            raise VoError(-4, "pts0.size() != pts1.size()")
        n = pts0.shape[0]
        out = np.zeros(max(n, 1), np.float32)
        if n:
            self.ctx.check(fn(self.ctx.handle, _p(pts0), _p(pts1), n, _p(_f32(F10).reshape(9)), _p(out)))
        return out[:n]

    def calcSampsonDistance(self, pts0, pts1, F10=None, K=None, R10=None, t10=None):
        """calcSampsonDistance(pts0, pts1, F10) (:572-599) or, with K/R10/t10, the camera overload (:538-570)."""
        if F10 is None:
            F10 = self.fundamentalFromPose(K, R10, t10)
        return self._epi(self.lib.vo_sampson_distance, pts0, pts1, F10)

    def calcSymmetricEpipolarDistance(self, pts0, pts1, K, R10, t10):
        """motion_estimator.cpp:621-653"""
        return self._epi(self.lib.vo_symmetric_epipolar_distance, pts0, pts1, self.fundamentalFromPose(K, R10, t10))


class FeatureExtractor:
    """descriptorDistance (feature_extractor.cpp:338-357) for whole descriptor sets, and the
    reference-owned bucketing around cv::ORB::detect: WeightBin (feature_extractor.h:58-135) and the
    arg-max-per-bin selection of extractORBwithBinning_fast (feature_extractor.cpp:241-277)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib
        self.n_bins_u_ = self.n_bins_v_ = 0

    def initParams(self, n_cols, n_rows, n_bins_u, n_bins_v, THRES_FAST=15, radius=5):
        """feature_extractor.cpp:26-69 minus cv::ORB::create: WeightBin::init (feature_extractor.h:90-118)."""
        f = np.float32
        self.n_bins_u_, self.n_bins_v_ = int(n_bins_u), int(n_bins_v)
        self.u_step = int(np.floor(f(n_cols) / f(n_bins_u)))
        self.v_step = int(np.floor(f(n_rows) / f(n_bins_v)))
        self.inv_u_step_ = f(1.0) / f(self.u_step)
        self.inv_v_step_ = f(1.0) / f(self.v_step)
        self.weight = np.ones(self.n_bins_u_ * self.n_bins_v_, np.int32)
        # extractor_orb_ settings, feature_extractor.cpp:48-56
        self.orb = OrbParams()
        self.orb.nfeatures, self.orb.scale_factor, self.orb.n_levels = 10000, 1.2, 8
        self.orb.edge_threshold, self.orb.fast_threshold = 31, int(THRES_FAST)

    def detect(self, slot, max_kp=60000):
        """extractor_orb_->detect(img, fts) (feature_extractor.cpp:241) on the image in `slot`:
        (xy, response, octave)."""
        xy = np.zeros((max_kp, 2), np.float32)
        resp = np.zeros(max_kp, np.float32)
        octv = np.zeros(max_kp, np.int32)
        n = C.c_int()
        self.ctx.check(self.lib.vo_orb_detect(self.ctx.handle, slot, C.byref(self.orb), _p(xy), _p(resp),
                                              _p(octv, C.c_int32), max_kp, C.byref(n)))
        return xy[: n.value].copy(), resp[: n.value].copy(), octv[: n.value].copy()

    def extractORBwithBinning_fast(self, slot):
        """feature_extractor.cpp:211-318 with flag_nonmax_ (the branch initParams selects, :37): detection and the
        per-bin arg-max on the device; returns pts_extracted."""
        pts = np.zeros((self.weight.size + 1, 2), np.float32)
        m, nd = C.c_int(), C.c_int()
        self.ctx.check(self.lib.vo_extract_orb_with_binning(
            self.ctx.handle, slot, C.byref(self.orb), C.c_float(self.inv_u_step_), C.c_float(self.inv_v_step_),
            self.n_bins_u_, self.n_bins_v_, _p(self.weight, C.c_int32), _p(pts), C.byref(m), C.byref(nd)))
        self.n_detected = nd.value
        return pts[: m.value].copy()

    def enqueueExtract(self, slot):
        """extractORBwithBinning_fast, asynchronous: runs on the context's side stream (overlaps the frame
        operator of the same image pair); resultExtract() collects pts_extracted."""
        self.ctx.check(self.lib.vo_extract_orb_with_binning_enqueue(
            self.ctx.handle, slot, C.byref(self.orb), C.c_float(self.inv_u_step_), C.c_float(self.inv_v_step_),
            self.n_bins_u_, self.n_bins_v_, _p(np.ascontiguousarray(self.weight, np.int32), C.c_int32)))

    def resultExtract(self):
        if getattr(self, "_pts_buf", None) is None or self._pts_buf.shape[0] < self.weight.size + 1:
            self._pts_buf = np.zeros((self.weight.size + 1, 2), np.float32)
        m, nd = C.c_int(), C.c_int()
        self.ctx.check(self.lib.vo_extract_orb_with_binning_result(self.ctx.handle, _p(self._pts_buf), C.byref(m),
                                                                   C.byref(nd)))
        self.n_detected = nd.value
        return self._pts_buf[: m.value]

    def binParams(self):
        """The bins and detector settings as the closed step [10] takes them (vo_bin_params)."""
        b = BinParams()
        b.n_bins_u, b.n_bins_v, b.u_step, b.v_step = self.n_bins_u_, self.n_bins_v_, self.u_step, self.v_step
        b.inv_u_step, b.inv_v_step = float(self.inv_u_step_), float(self.inv_v_step_)
        b.orb = self.orb
        return b

    def enqueueCandidates(self, slot, table):
        """Detection + best keypoint of EVERY bin for the image in `slot`, on the side stream, into table 0 / 1
        (the image-only part of extractORBwithBinning_fast; StereoFramePipeline.enqueue_closed consumes it)."""
        if getattr(self, "_bin_params", None) is None:
            self._bin_params = self.binParams()
        rc = self.lib.vo_new_point_candidates_enqueue(self.ctx.handle, slot, self._bin_params, table)
        if rc < 0:
            self.ctx.check(rc)

    def getCandidates(self, table):
        """test hook: (xy[n_bins, 2], has[n_bins], n_detected) of a table"""
        nb = self.n_bins_u_ * self.n_bins_v_
        xy, has, nd = np.zeros((nb, 2), np.float32), np.zeros(nb, np.uint8), C.c_int()
        self.ctx.check(self.lib.vo_new_point_candidates_get(self.ctx.handle, table, _p(xy), _p(has, C.c_uint8), C.byref(nd)))
        return xy, has.astype(bool), nd.value

    def resetWeightBin(self):
        self.weight[:] = 1

    def suppressCenterBins(self):
        """feature_extractor.cpp:76-92 (host-side: a fixed pattern of a few dozen bins)"""
        nu, nv = self.n_bins_u_, self.n_bins_v_
        u_cent, v_cent = int(nu * 0.5), int(nv * 0.5)
        wu, wv, wv2 = int(np.float32(0.15) * nu), int(np.float32(0.30) * nv), int(np.float32(0.15) * nv)
        for w in range(-wv, wv + 1):
            v_idx = nu * (w + v_cent - wv2)
            for u in range(-wu, wu + 1):
                self.weight[v_idx + u + u_cent] = 0

    def updateWeightBin(self, fts):
        """feature_extractor.cpp:94-98: reset, then weight 0 for every bin that holds a tracked point"""
        pts = _f32(fts).reshape(-1, 2)
        w = np.zeros(self.weight.size, np.int32)
        self.ctx.check(self.ctx.lib.vo_weight_bin_update(self.ctx.handle, _p(pts), pts.shape[0], self.u_step,
                                                         self.v_step, self.n_bins_u_, self.n_bins_v_,
                                                         _p(w, C.c_int32)))
        self.weight = w
        return w

    def bucketKeypoints(self, kp_xy, kp_response):
        """The flag_nonmax_ branch of extractORBwithBinning_fast (feature_extractor.cpp:241-277) on the
        keypoints cv::ORB::detect returned (positions, responses, detector order)."""
        kp, r = _f32(kp_xy).reshape(-1, 2), _f32(kp_response).reshape(-1)
        if kp.shape[0] != r.shape[0]:
            raise VoError(-4, "keypoint positions / responses differ in length")
        tot = self.weight.size
        out, idx, m = np.zeros((max(tot, 1), 2), np.float32), np.zeros(max(tot, 1), np.int32), C.c_int()
        self.ctx.check(self.ctx.lib.vo_bucket_argmax(
            self.ctx.handle, _p(kp), _p(r), kp.shape[0], C.c_float(self.inv_u_step_), C.c_float(self.inv_v_step_),
            self.n_bins_u_, self.n_bins_v_, _p(np.ascontiguousarray(self.weight, np.int32), C.c_int32), _p(out),
            _p(idx, C.c_int32), C.byref(m)))
        return out[: m.value].copy(), idx[: m.value].copy()

    def descriptorDistance(self, a, b):
        a, b = _u8(a).reshape(-1, 32), _u8(b).reshape(-1, 32)
        out = np.zeros((a.shape[0], b.shape[0]), np.uint16)
        if a.shape[0] and b.shape[0]:
            self.ctx.check(self.lib.vo_orb_hamming(self.ctx.handle, _p(a, C.c_uint8), a.shape[0],
                                                   _p(b, C.c_uint8), b.shape[0], _p(out, C.c_uint16)))
        return out

    def match(self, a, b, th_low=50, ratio=0.6):
        a, b = _u8(a).reshape(-1, 32), _u8(b).reshape(-1, 32)
        na = a.shape[0]
        bi = np.zeros(max(na, 1), np.int32)
        bd = np.zeros(max(na, 1), np.uint16)
        sd = np.zeros(max(na, 1), np.uint16)
        self.ctx.check(self.lib.vo_orb_match(self.ctx.handle, _p(a, C.c_uint8), na, _p(b, C.c_uint8),
                                             b.shape[0], th_low, C.c_float(ratio), _p(bi, C.c_int32),
                                             _p(bd, C.c_uint16), _p(sd, C.c_uint16)))
        return bi[:na], bd[:na], sd[:na]


def compact_indices(ctx, mask, alive=None, tracked=None):
    """Stable compaction of mask && alive && tracked (landmark.cpp:291-332)."""
    mask = _u8(mask)
    n = mask.shape[0]
    idx = np.zeros(max(n, 1), np.int32)
    cnt = C.c_int()
    al = _u8(alive) if alive is not None else None
    tk = _u8(tracked) if tracked is not None else None
    ctx.check(ctx.lib.vo_compact_indices(
        ctx.handle, _p(mask, C.c_uint8), _p(al, C.c_uint8) if al is not None else None,
        _p(tk, C.c_uint8) if tk is not None else None, n, _p(idx, C.c_int32), C.byref(cnt)))
    return idx[: cnt.value].copy()


def se3Exp_f(ctx, xi):
    """geometry::se3Exp_f (geometry_library.cpp:386-440) and inverseSE3_f (:554-560) as the GN kernel evaluates
    them on the device; returns (T, inverse(T)), row-major 4x4."""
    xi = _f32(xi).reshape(6)
    T, Ti = np.zeros(16, np.float32), np.zeros(16, np.float32)
    ctx.check(ctx.lib.vo_se3_exp(ctx.handle, _p(xi), _p(T), _p(Ti)))
    return T.reshape(4, 4), Ti.reshape(4, 4)


def write_trajectory(path, frame_ids, T_wc):
    """The reference's trajectory dump (stereo_vo.cpp:55-115, mono_vo.cpp:64-125): one line per frame,
    `<frame id> r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz`, fixed notation, precision 4."""
    T = np.asarray(T_wc, np.float32).reshape(-1, 4, 4)
    ids = list(frame_ids)
    if len(ids) != T.shape[0]:
        raise ValueError("ids and poses differ in length")
    with open(path, "w") as f:  # (the reference throws "file_dir cannot be opened!" when it cannot)
        for j, Tj in zip(ids, T):
            f.write(str(int(j)) + "".join(" %.4f" % float(v) for v in Tj[:3].reshape(-1)) + "\n")


class TrackIds:
    """Landmark / Frame IDs of ONE image stream and the mask-compaction constructors with their side effect.
    The reference's counters are process-global statics (landmark.h:64, frame.h:53); here they belong to the
    context, so several streams in one process keep the IDs each would have alone."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib

    def reset(self, next_landmark_id=0, next_frame_id=0):
        self.ctx.check(self.lib.vo_ids_reset(self.ctx.handle, int(next_landmark_id), int(next_frame_id)))

    def peek(self):
        a, b = C.c_int32(), C.c_int32()
        self.ctx.check(self.lib.vo_ids_peek(self.ctx.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def newFrames(self, n):
        """n Frame constructions (a StereoFrame is two: left, right — frame.cpp:176-180)."""
        ids = np.zeros(max(n, 1), np.int32)
        self.ctx.check(self.lib.vo_ids_new_frames(self.ctx.handle, int(n), _p(ids, C.c_int32)))
        return ids[:n]

    def newLandmarks(self, accept):
        """One Landmark construction per accepted candidate, in candidate order (stereo_vo.cpp:716-729);
        returns ids with -1 at the rejected candidates."""
        accept = _u8(accept).reshape(-1)
        n = accept.shape[0]
        ids = np.zeros(max(n, 1), np.int32)
        made = C.c_int()
        self.ctx.check(self.lib.vo_ids_new_landmarks(self.ctx.handle, _p(accept, C.c_uint8), n, _p(ids, C.c_int32),
                                                     C.byref(made)))
        return ids[:n]

    def compactTracks(self, mask, alive, tracked, ids):
        """StereoLandmarkTracking(src, mask) / LandmarkTracking(src, mask) (landmark.cpp:291-332, :194-231):
        returns (index_valid, tracked after the constructor's setUntracked() calls, ids of the survivors)."""
        mask, alive = _u8(mask).reshape(-1), _u8(alive).reshape(-1)
        tracked = _u8(tracked).reshape(-1).copy()
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1)
        n = mask.shape[0]
        if not (alive.shape[0] == tracked.shape[0] == ids.shape[0] == n):
            raise VoError(-4, "lmtrack sizes differ from mask.size()")  # landmark.cpp:196-197, :293-295
        idx = np.zeros(max(n, 1), np.int32)
        ids_out = np.zeros(max(n, 1), np.int32)
        cnt = C.c_int()
        self.ctx.check(self.lib.vo_compact_tracks(
            self.ctx.handle, _p(mask, C.c_uint8), _p(alive, C.c_uint8), _p(tracked, C.c_uint8), _p(ids, C.c_int32), n,
            _p(idx, C.c_int32), _p(ids_out, C.c_int32), C.byref(cnt)))
        return idx[:cnt.value].copy(), tracked, ids_out[:cnt.value].copy()


def make_stereo_params(width, height, win, max_level, thres_err, thres_bidir, thres_poseba, Kl, Kr,
                       T_lr, thres_sampson=60.0):
    p = StereoParams()
    p.thres_sampson = thres_sampson  # feature_tracker.thres_sampson (60 in kitti_00_stereo.yaml)
    p.width, p.height, p.win, p.max_level = width, height, win, max_level
    p.thres_err, p.thres_bidirection, p.thres_poseba = thres_err, thres_bidir, thres_poseba
    T = _f32(T_lr).reshape(16)
    for i in range(4):
        p.Kl[i] = float(Kl[i])
        p.Kr[i] = float(Kr[i])
    for i in range(16):
        p.T_lr[i] = float(T[i])
    return p


class StereoFramePipeline:
    """The steady-state stereo frame (stereo_vo.cpp:483-711 operator sequence)
    chained on the device. Slots: 0 = previous left, 1 = current left,
    2 = current right; advance() rotates current-left into previous-left."""

    def __init__(self, ctx, params, strict_border=False):
        self.ctx = ctx
        self.lib = ctx.lib
        self.prm = params
        self.ctx.check(self.lib.vo_stereo_frame_set_strict_border(ctx.handle, int(strict_border)))

    def enqueue(self, pts_l0, pts_r0, Xp, dT_prior, pts_new, slots=(0, 1, 2), lm_flags=None):
        """lm_flags[i] bit 0 = lm->isTriangulated() (stereo_vo.cpp:490, :599); None = every landmark is."""
        pts_l0, pts_r0 = _f32(pts_l0).reshape(-1, 2), _f32(pts_r0).reshape(-1, 2)
        Xp = _f32(Xp).reshape(-1, 3)
        dT = _f32(dT_prior).reshape(16)
        pts_new = _f32(pts_new).reshape(-1, 2)
        self._n, self._nn = pts_l0.shape[0], pts_new.shape[0]
        self._closed = False
        fl = None
        if lm_flags is not None:
            fl = _u8(lm_flags).reshape(-1)
            if fl.shape[0] != self._n:
                raise ValueError("lm_flags.size() != pts_l0.size()")
        self.ctx.check(self.lib.vo_stereo_frame_enqueue(
            self.ctx.handle, C.byref(self.prm), slots[0], slots[1], slots[2], _p(pts_l0), _p(pts_r0),
            _p(Xp), fl.ctypes.data if fl is not None else None, self._n, _p(dT), _p(pts_new), self._nn, 0))

    def enqueue_device(self, d_pts_l0, d_pts_r0, d_Xp, n, dT_prior, d_pts_new, n_new, slots=(0, 1, 2), d_lm_flags=None):
        dT = dT_prior if (isinstance(dT_prior, np.ndarray) and dT_prior.dtype == np.float32 and dT_prior.flags.c_contiguous) \
            else _f32(dT_prior)
        self._n, self._nn = n, n_new
        self._closed = False
        rc = self.lib.vo_stereo_frame_enqueue(self.ctx.handle, self.prm, slots[0], slots[1], slots[2], d_pts_l0, d_pts_r0,
                                              d_Xp, d_lm_flags, n, dT.ctypes.data, d_pts_new, n_new, 1)
        if rc < 0:
            self.ctx.check(rc)

    def enqueue_closed(self, pts_l0, pts_r0, Xp, dT_prior, bins, table, slots=(0, 1, 2), lm_flags=None):
        """The frame with step [10] closed on the device: candidates = best keypoint of every bin of `table`
        (FeatureExtractor.enqueueCandidates on the image in slots[1]), emitted for the bins lmtrack_final leaves
        empty. `bins` = FeatureExtractor.binParams()."""
        pts_l0, pts_r0 = _f32(pts_l0).reshape(-1, 2), _f32(pts_r0).reshape(-1, 2)
        Xp = _f32(Xp).reshape(-1, 3)
        dT = _f32(dT_prior).reshape(16)
        self._n, self._nn = pts_l0.shape[0], bins.n_bins_u * bins.n_bins_v
        self._closed = True
        fl = None
        if lm_flags is not None:
            fl = _u8(lm_flags).reshape(-1)
            if fl.shape[0] != self._n:
                raise ValueError("lm_flags.size() != pts_l0.size()")
        self.ctx.check(self.lib.vo_stereo_frame_enqueue_closed(
            self.ctx.handle, self.prm, slots[0], slots[1], slots[2], pts_l0.ctypes.data, pts_r0.ctypes.data,
            Xp.ctypes.data, fl.ctypes.data if fl is not None else None, self._n, dT.ctypes.data, bins, table, 0))

    def enqueue_closed_device(self, d_pts_l0, d_pts_r0, d_Xp, n, dT_prior, bins, table, slots=(0, 1, 2), d_lm_flags=None):
        dT = dT_prior if (isinstance(dT_prior, np.ndarray) and dT_prior.dtype == np.float32 and dT_prior.flags.c_contiguous) \
            else _f32(dT_prior)
        self._n, self._nn = n, bins.n_bins_u * bins.n_bins_v
        self._closed = True
        rc = self.lib.vo_stereo_frame_enqueue_closed(self.ctx.handle, self.prm, slots[0], slots[1], slots[2], d_pts_l0,
                                                     d_pts_r0, d_Xp, d_lm_flags, n, dT.ctypes.data, bins, table, 1)
        if rc < 0:
            self.ctx.check(rc)

    def _buffers(self, n, nn):
        key = (n, nn)
        if getattr(self, "_buf_key", None) != key:
            self._buf_key = key
            self._b = dict(pts_l1=np.zeros((max(n, 1), 2), np.float32), pts_r1=np.zeros((max(n, 1), 2), np.float32),
                           stage=np.zeros(max(n, 1), np.uint8), dT=np.zeros(16, np.float32),
                           pnr=np.zeros((max(nn, 1), 2), np.float32), mnew=np.zeros(max(nn, 1), np.uint8),
                           pnl=np.zeros((max(nn, 1), 2), np.float32), n_new=C.c_int(), counts=FrameCounts(), gn=GnInfo())
            b = self._b
            self._args = (_p(b["pts_l1"]), _p(b["pts_r1"]), _p(b["stage"], C.c_uint8), _p(b["dT"]), _p(b["pnr"]),
                          _p(b["mnew"], C.c_uint8), C.byref(b["counts"]), C.byref(b["gn"]))
        return self._b

    def result(self, copy=True):
        """Waits for the enqueued frame. copy=False returns views of internal buffers that the
        next result() overwrites (what a frame-by-frame consumer needs; no allocations)."""
        n, nn = self._n, self._nn
        b = self._buffers(n, nn)
        rc = self.lib.vo_stereo_frame_result(self.ctx.handle, *self._args)
        if rc < 0:
            self.ctx.check(rc)
        closed = getattr(self, "_closed", False)
        if closed:  # the candidates are the device's: their number and left pixels come with the result
            self._closed = False
            rc = self.lib.vo_stereo_frame_new_points(self.ctx.handle, b["pnl"].ctypes.data, C.addressof(b["n_new"]))
            if rc < 0:
                self.ctx.check(rc)
            nn = b["n_new"].value
        out = dict(pts_l1=b["pts_l1"][:n], pts_r1=b["pts_r1"][:n], stage=b["stage"][:n], dT=b["dT"].reshape(4, 4),
                   pts_new_r=b["pnr"][:nn], mask_new=b["mnew"][:nn].view(bool), counts=b["counts"], gn=b["gn"])
        if closed:
            out["pts_new"] = b["pnl"][:nn]
        if copy:
            cnt, gn = FrameCounts(), GnInfo()
            C.memmove(C.byref(cnt), C.byref(b["counts"]), C.sizeof(cnt))
            C.memmove(C.byref(gn), C.byref(b["gn"]), C.sizeof(gn))
            out = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in out.items()}
            out["counts"], out["gn"] = cnt, gn
        return out


    def enqueue_closed_world(self, pts_l0, pts_r0, Xw, dT_prior, T_pw, T_cw_prior, bins, table, slots=(0, 1, 2), lm_flags=None):
        """enqueue_closed on the reference's own data flow (stereo_vo.cpp:475-522): Xw = landmarks in the WORLD frame,
        T_pw = previous pose inverse, T_cw_prior = inverseSE3_f(T_wp * dT_pc_prev)."""
        pts_l0, pts_r0 = _f32(pts_l0).reshape(-1, 2), _f32(pts_r0).reshape(-1, 2)
        Xw = _f32(Xw).reshape(-1, 3)
        dT, Tp, Tc = _f32(dT_prior).reshape(16), _f32(T_pw).reshape(16), _f32(T_cw_prior).reshape(16)
        self._n, self._nn = pts_l0.shape[0], bins.n_bins_u * bins.n_bins_v
        self._closed = True
        fl = None
        if lm_flags is not None:
            fl = _u8(lm_flags).reshape(-1)
            if fl.shape[0] != self._n:
                raise ValueError("lm_flags.size() != pts_l0.size()")
        self.ctx.check(self.lib.vo_stereo_frame_enqueue_closed_world(
            self.ctx.handle, self.prm, slots[0], slots[1], slots[2], pts_l0.ctypes.data, pts_r0.ctypes.data,
            Xw.ctypes.data, fl.ctypes.data if fl is not None else None, self._n, dT.ctypes.data, Tp.ctypes.data,
            Tc.ctypes.data, bins, table, 0))


def triangulateDLT(ctx, pts0, pts1, T_10, K0, K1):
    """mapping::triangulateDLT (core/util/triangulate_3d.cpp:91-130) for n pixel pairs: (X0, X1 = R10 X0 + t10)."""
    pts0, pts1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    if pts0.shape[0] != pts1.shape[0]:
        raise VoError(-4, "pts0.size() != pts1.size()")  # triangulate_3d.cpp:9-10
    n = pts0.shape[0]
    X0, X1 = np.zeros((max(n, 1), 3), np.float32), np.zeros((max(n, 1), 3), np.float32)
    T, K0, K1 = _f32(T_10).reshape(16), _f32(K0).reshape(4), _f32(K1).reshape(4)
    ctx.check(ctx.lib.vo_triangulate_dlt(ctx.handle, pts0.ctypes.data, pts1.ctypes.data, n, T.ctypes.data, K0.ctypes.data,
                                         K1.ctypes.data, X0.ctypes.data, X1.ctypes.data))
    return X0[:n], X1[:n]


class StereoVO:
    """StereoVO (core/visual_odometry/stereo_vo/stereo_vo.h:233-249): trackStereoImages with the track set carried on
    the device (include/vo_hip.h: vo_svo_*). The YAML loading of the reference's constructor is the caller's: the
    parameters arrive as numbers. The context needs >= 5 image slots."""

    def __init__(self, ctx, width, height, Kl, Kr, T_lr, n_bins_u, n_bins_v, thres_fastscore=15, window_size=21, max_level=6,
                 thres_error=80.0, thres_bidirection=0.5, thres_poseba_error=3.0, thres_alive_ratio=0.6, thres_rotation=15.0,
                 thres_trans=10.0, n_max_keyframes_in_window=9, strict_border=4, local_ba=True, rectify=False, thres_sampson=60.0):
        """rectify=True is system_flags_.flagDoUndistortion: the context's stereo rectification maps (StereoCamera.
        initStereoCameraToRectify on the same context) are applied to every incoming pair; Kl / Kr / T_lr are then the
        rectified camera and extrinsics (getRectifiedCamera / getRectifiedStereoPoseLeft2Right)."""
        self.ctx, self.lib = ctx, ctx.lib
        fe = FeatureExtractor(ctx)
        fe.initParams(width, height, n_bins_u, n_bins_v, THRES_FAST=thres_fastscore)
        p = SvoParams()
        p.frame = make_stereo_params(width, height, window_size, max_level, thres_error, thres_bidirection, thres_poseba_error,
                                     Kl, Kr, T_lr, thres_sampson)
        p.bins = fe.binParams()
        p.kf_overlap_ratio, p.kf_rotation_deg, p.kf_translation = thres_alive_ratio, thres_rotation, thres_trans
        p.kf_window, p.strict_border, p.local_ba = n_max_keyframes_in_window, int(strict_border), int(bool(local_ba))
        p.rectify = int(bool(rectify))
        self.prm, self.width, self.height = p, width, height
        self._h = C.c_void_p()
        ctx.check(self.lib.vo_svo_create(ctx.handle, C.byref(p), C.byref(self._h)))
        ctx._children.add(self)
        self._info = SvoFrameInfo()
        self.stats_frame = []  # AlgorithmStatistics::FrameStatistics::Twc per frame (stereo_vo.cpp:979-980)

    @classmethod
    def from_yaml(cls, path, device=0, max_points=None, **overrides):
        """StereoVO(mode = "rosbag", directory_intrinsic = path) of the reference (stereo_vo.cpp:15-57, :118-280): the
        object configured by one of its config/stereo/*.yaml files. The context is created here (sized by the file) and
        closed with the object. With flagDoUndistortion the pairs go through the rectification maps and the loop runs on
        the rectified camera (:414-427); without, on the raw cameras and T_lr of the file. `overrides`: keyword
        arguments of the constructor that the file does not know (strict_border, local_ba). `max_points`: capacity of a
        track set (default 2 * bins + 1024; the reference has no such bound — several survivors may share a bin while
        every empty bin adds a landmark — so a caller that sees VO_ERR_CAPACITY raises it)."""
        from . import config as _config
        cfg = _config.load_stereo_config(path)
        cl, cr = cfg["camera"]["left"], cfg["camera"]["right"]
        W, H = cl["width"], cl["height"]
        fe, ft, me, ku = cfg["feature_extractor"], cfg["feature_tracker"], cfg["motion_estimator"], cfg["keyframe_update"]
        cap = int(max_points) if max_points else 2 * fe["n_bins_u"] * fe["n_bins_v"] + 1024
        ctx = Context(device=device, max_width=W, max_height=H, max_points=cap, n_slots=5,
                      max_level=ft["max_level"])
        try:
            Kl, Kr, T_lr, rectify = cl["K"], cr["K"], cfg["T_lr"], False
            if cfg["flagDoUndistortion"]:
                cam = StereoCamera(ctx)
                cam.initParams(W, H, cl["K"], cl["D"], cr["K"], cr["D"])
                cam.setStereoPoseLeft2Right(cfg["T_lr"])
                cam.initStereoCameraToRectify()
                Kl = Kr = cam.getRectifiedCamera()
                T_lr, rectify = cam.getRectifiedStereoPoseLeft2Right(), True
            obj = cls(ctx, W, H, Kl, Kr, T_lr, fe["n_bins_u"], fe["n_bins_v"], thres_fastscore=fe["thres_fastscore"],
                      window_size=ft["window_size"], max_level=ft["max_level"], thres_error=ft["thres_error"],
                      thres_sampson=ft["thres_sampson"],
                      thres_bidirection=ft["thres_bidirection"], thres_poseba_error=me["thres_poseba_error"],
                      thres_alive_ratio=ku["thres_alive_ratio"], thres_rotation=ku["thres_rotation"], thres_trans=ku["thres_trans"],
                      n_max_keyframes_in_window=ku["n_max_keyframes_in_window"], rectify=rectify, **overrides)
        except Exception:
            ctx.close()
            raise
        obj._own_ctx, obj.config = ctx, cfg
        return obj

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vo_svo_destroy(self._h)
            self._h = C.c_void_p()
        own = getattr(self, "_own_ctx", None)
        if own is not None:
            self._own_ctx = None
            own.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ptrs(left, right):
        if isinstance(left, np.ndarray):
            left, right = _u8(left), _u8(right)
            if left.ndim != 2 or left.shape != right.shape:
                raise ValueError("images must be two u8 planes of one size")
            return left, right, left.ctypes.data, right.ctypes.data, left.strides[0], 0
        return left, right, int(left[0]), int(right[0]), int(left[1]), 1  # (device address, stride) pairs

    def _out(self):
        i = self._info
        T = np.array(i.T_wc, np.float32).reshape(4, 4)
        self.stats_frame.append(T)
        return SvoFrameInfo.from_buffer_copy(i)  # (the caller's own copy: a stored info must not change with the next frame)

    def trackStereoImages(self, img_left, img_right, timestamp=0.0):
        """numpy u8 images (host) or (device address, stride) pairs. Returns the frame's SvoFrameInfo."""
        a, b, pl, pr, st, dev = self._ptrs(img_left, img_right)
        self._keep = (a, b)
        rc = self.lib.vo_svo_track(self._h, pl, pr, st, dev, float(timestamp), self._info)
        if rc < 0:
            self.ctx.check(rc)
        return self._out()

    def enqueue(self, img_left, img_right, timestamp=0.0):
        a, b, pl, pr, st, dev = self._ptrs(img_left, img_right)
        self._keep = (a, b)
        rc = self.lib.vo_svo_enqueue(self._h, pl, pr, st, dev, float(timestamp))
        if rc < 0:
            self.ctx.check(rc)

    def prefetch(self, img_left, img_right):
        a, b, pl, pr, st, dev = self._ptrs(img_left, img_right)
        self._keep_next = (a, b)
        rc = self.lib.vo_svo_prefetch(self._h, pl, pr, st, dev)
        if rc < 0:
            self.ctx.check(rc)

    def result(self):
        rc = self.lib.vo_svo_result(self._h, self._info)
        if rc < 0:
            self.ctx.check(rc)
        return self._out()

    def runSequence(self, pairs, k_begin=0, k_end=None):
        """A recorded sequence of (device address, stride) pairs — [((left, stride), (right, stride)), ...] — or of (left, right)
        numpy u8 images of one size and layout (host memory: uploaded one frame ahead, under the frame in flight) through the
        loop inside the library (vo_svo_run): frames k_begin .. k_end - 1 are collected; frame k_end is left in flight for the
        next call (or result()). Returns (list of SvoFrameInfo, numpy array of CLOCK_MONOTONIC stamps)."""
        n = len(pairs)
        k_end = n if k_end is None else int(k_end)
        host = n > 0 and isinstance(pairs[0][0], np.ndarray)
        if host:  # (the arrays themselves are handed to the library: they are kept alive with the list, below)
            if getattr(self, "_seq_ref", None) is not pairs or len(getattr(self, "_seq_host", ())) != n:
                self._seq_host = [(_u8(L), _u8(R)) for L, R in pairs]
                st0 = self._seq_host[0][0].strides[0]
                if any(a.ndim != 2 or a.shape != b.shape or a.strides[0] != st0 or b.strides[0] != st0 for a, b in self._seq_host):
                    raise ValueError("runSequence: host images must be u8 planes of one size and row stride")
            src = [((a.ctypes.data, a.strides[0]), (b.ctypes.data, b.strides[0])) for a, b in self._seq_host]
        else:
            src = pairs
        # The address arrays are kept between calls on the SAME list object (held here, so its id cannot be reused) of the
        # same length, and the entries this call hands to the library (k_begin .. k_end + 1) are compared with them: a list
        # that was changed in place, or another list, rebuilds the arrays.
        L, R = getattr(self, "_seq_L", None), getattr(self, "_seq_R", None)
        fresh = getattr(self, "_seq_ref", None) is not pairs or L is None or len(L) != n
        if not fresh:
            for k in range(max(int(k_begin), 0), min(k_end + 2, n)):
                if L[k] != int(src[k][0][0]) or R[k] != int(src[k][1][0]):
                    fresh = True
                    break
        if fresh:
            self._seq_L = (C.c_void_p * n)(*[int(p[0][0]) for p in src])
            self._seq_R = (C.c_void_p * n)(*[int(p[1][0]) for p in src])
            self._seq_ref = pairs
        self._seq_stride = int(src[0][0][1])
        m = k_end - int(k_begin)
        infos = (SvoFrameInfo * max(m, 1))()
        stamps = np.zeros(max(m, 1), np.float64)
        rc = self.lib.vo_svo_run(self._h, self._seq_L, self._seq_R, n, self._seq_stride, 0 if host else 1, int(k_begin), k_end, infos,
                                 stamps.ctypes.data)
        if rc < 0:
            self.ctx.check(rc)
        out = [SvoFrameInfo.from_buffer_copy(infos[j]) for j in range(m)]
        for i in out:
            self.stats_frame.append(np.array(i.T_wc, np.float32).reshape(4, 4))
        return out, stamps[:m]

    def getTracks(self):
        """The track set the next frame starts from: dict(ids, pts_l, pts_r, Xw, flags)."""
        n = C.c_int()
        self.ctx.check(self.lib.vo_svo_get_tracks(self._h, None, None, None, None, None, 0, C.addressof(n)))
        m = max(n.value, 1)
        ids, pl, pr = np.zeros(m, np.int32), np.zeros((m, 2), np.float32), np.zeros((m, 2), np.float32)
        X, fl = np.zeros((m, 3), np.float32), np.zeros(m, np.uint8)
        self.ctx.check(self.lib.vo_svo_get_tracks(self._h, ids.ctypes.data, pl.ctypes.data, pr.ctypes.data, X.ctypes.data,
                                                  fl.ctypes.data, m, C.addressof(n)))
        k = n.value
        return dict(ids=ids[:k], pts_l=pl[:k], pts_r=pr[:k], Xw=X[:k], flags=fl[:k])

    def getNewPoints(self):
        """Step [10] of the last frame: dict(pts_l, pts_r, mask_new, accept)."""
        nb = self.prm.bins.n_bins_u * self.prm.bins.n_bins_v
        pl, pr = np.zeros((nb, 2), np.float32), np.zeros((nb, 2), np.float32)
        m, a, n = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), C.c_int()
        self.ctx.check(self.lib.vo_svo_get_new_points(self._h, pl.ctypes.data, pr.ctypes.data, m.ctypes.data, a.ctypes.data,
                                                      C.addressof(n)))
        k = n.value
        return dict(pts_l=pl[:k], pts_r=pr[:k], mask_new=m[:k].astype(bool), accept=a[:k].astype(bool))

    def deviceBytes(self):
        """Device memory held for this stream's keyframes (table, ring, pool, the local BA's scratch and arena)."""
        n = C.c_size_t(0)
        self.ctx.check(self.lib.vo_svo_device_bytes(self._h, C.byref(n)))
        return n.value

    def getKeyframes(self):
        """AlgorithmStatistics::stats_keyframe as of now (stereo_vo.cpp:805-821): [(T_wc, mappoints [n][3])] for every
        keyframe so far — current poses, current 3-D points of the related landmarks."""
        n, tot = C.c_int(), C.c_size_t()
        self.ctx.check(self.lib.vo_svo_keyframe_count(self._h, C.addressof(n)))
        nk = n.value
        if nk == 0:
            return []
        T, cnt = np.zeros((nk, 16), np.float32), np.zeros(nk, np.int32)
        self.ctx.check(self.lib.vo_svo_get_keyframes(self._h, T.ctypes.data, cnt.ctypes.data, None, 0, C.addressof(tot)))
        X = np.zeros((max(tot.value, 1), 3), np.float32)
        if tot.value:
            self.ctx.check(self.lib.vo_svo_get_keyframes(self._h, None, None, X.ctypes.data, tot.value, C.addressof(tot)))
        off = np.concatenate([[0], np.cumsum(cnt)])
        return [(T[j].reshape(4, 4).copy(), X[off[j]:off[j + 1]].copy()) for j in range(nk)]

    def getKeyframe(self, j):
        """One keyframe of stats_keyframe: (T_wc, mappoints)."""
        T, m = np.zeros(16, np.float32), C.c_int()
        self.ctx.check(self.lib.vo_svo_get_keyframe(self._h, int(j), T.ctypes.data, None, 0, C.addressof(m)))
        X = np.zeros((max(m.value, 1), 3), np.float32)
        if m.value:
            self.ctx.check(self.lib.vo_svo_get_keyframe(self._h, int(j), None, X.ctypes.data, m.value, C.addressof(m)))
        return T.reshape(4, 4), X[:m.value]

    def getStatistics(self):
        return dict(stats_frame=[T.copy() for T in self.stats_frame], stats_keyframe=self.getKeyframes())


class StereoBatch:
    """S independent stereo streams on one GPU (include/vo_hip.h: vo_batch_*): a context, a StereoVO and a host thread per
    stream inside the library. `svo_params` = StereoVO(...).prm of a template object (or an SvoParams)."""

    def __init__(self, device, n_streams, width, height, max_points, max_level, svo_params):
        self.lib = _capi.load()
        self.n, self.prm = int(n_streams), svo_params
        self.cfg = VoConfig(device, width, height, max_points, 5, max_level)
        self._h = C.c_void_p()
        rc = self.lib.vo_batch_create(C.byref(self.cfg), C.byref(svo_params), self.n, C.byref(self._h))
        if rc < 0:
            raise VoError(rc, "vo_batch_create failed")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vo_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_set(self, key, value):
        if self.lib.vo_batch_debug_set(self._h, int(key), int(value)) < 0:
            raise VoError(-1, "vo_batch_debug_set failed")

    def strict_border(self):
        """The replay arrangement the streams run with (3, 4, 5 become the stream-ordered 1 when S > 1, same results)."""
        return int(self.lib.vo_batch_strict_border(self._h))

    def run(self, left_ptrs, right_ptrs, stride, warmup=0, on_device=True, ids_cap=8192):
        """left_ptrs / right_ptrs: [n_streams][n_frames] addresses. Returns dict(T_wc, ids (list), seconds, wall)."""
        nf = len(left_ptrs[0])
        L = (C.c_void_p * (self.n * nf))(*[p for row in left_ptrs for p in row])
        R = (C.c_void_p * (self.n * nf))(*[p for row in right_ptrs for p in row])
        T = np.zeros((self.n, nf, 4, 4), np.float32)
        ids = np.zeros((self.n, ids_cap), np.int32)
        nid = np.zeros(self.n, np.int32)
        sec = np.zeros(self.n, np.float64)
        wall = C.c_double()
        rc = self.lib.vo_batch_run(self._h, L, R, nf, int(stride), int(bool(on_device)), int(warmup), T.ctypes.data, ids.ctypes.data,
                                   ids_cap, nid.ctypes.data, sec.ctypes.data, C.addressof(wall))
        if rc < 0:
            raise VoError(rc, self.lib.vo_batch_last_error(self._h).decode())
        return dict(T_wc=T, ids=[ids[s, :nid[s]].copy() for s in range(self.n)], seconds=sec, wall=wall.value)


def make_mono_params(width, height, win, max_level, thres_err, thres_bidir, thres_poseba, thres_sampson, K):
    p = MonoParams()
    p.width, p.height, p.win, p.max_level = width, height, win, max_level
    p.thres_err, p.thres_bidirection = thres_err, thres_bidir
    p.thres_poseba, p.thres_sampson = int(thres_poseba), thres_sampson
    for i in range(4):
        p.K[i] = float(K[i])
    return p


class MonoFramePipeline:
    """The steady-state mono frame (mono_vo.cpp:739-963 operator sequence) chained on the
    device: prior + scale, trackBidirectionWithPrior, trackWithScale, pose-only BA on the
    landmarks flagged for it, mask_motion and the Sampson gate. flags[i]: bit 0 =
    lm->isBundled(), bit 1 = member of the class used for the pose-only BA (:800-826)."""

    def __init__(self, ctx, params, strict_border=False):
        self.ctx = ctx
        self.lib = ctx.lib
        self.prm = params
        self.ctx.check(self.lib.vo_stereo_frame_set_strict_border(ctx.handle, int(strict_border)))

    def enqueue(self, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01_prior, slots=(0, 1)):
        pts0 = _f32(pts0).reshape(-1, 2)
        Xw = _f32(Xw).reshape(-1, 3)
        flags = np.ascontiguousarray(flags, np.uint8)
        self._n = pts0.shape[0]
        if Xw.shape[0] != self._n or flags.shape[0] != self._n:
            raise ValueError("pts0 / Xw / flags differ in length")
        self.ctx.check(self.lib.vo_mono_frame_enqueue(
            self.ctx.handle, C.byref(self.prm), slots[0], slots[1], _p(pts0), _p(Xw), _p(flags, C.c_uint8), self._n,
            _p(_f32(Tcw_prev).reshape(16)), _p(_f32(Tcw_prior).reshape(16)), _p(_f32(dT01_prior).reshape(16)), 0))

    def enqueue_device(self, d_pts0, d_Xw, d_flags, n, Tcw_prev, Tcw_prior, dT01_prior, slots=(0, 1)):
        self._n = n
        f = C.POINTER(C.c_float)
        self.ctx.check(self.lib.vo_mono_frame_enqueue(
            self.ctx.handle, C.byref(self.prm), slots[0], slots[1], C.cast(C.c_void_p(d_pts0), f),
            C.cast(C.c_void_p(d_Xw), f), C.cast(C.c_void_p(d_flags), C.POINTER(C.c_uint8)), n,
            _p(_f32(Tcw_prev).reshape(16)), _p(_f32(Tcw_prior).reshape(16)), _p(_f32(dT01_prior).reshape(16)), 1))

    def enqueue_closed(self, pts0, Xw, flags, Tcw_prev, Tcw_prior, dT01_prior, bins, table, slots=(0, 1)):
        """The frame with its new-point step (mono_vo.cpp:977-1001) closed on the device: candidates = best keypoint of
        every bin of `table` (FeatureExtractor.enqueueCandidates on the image in slots[1]), back-tracked inside the frame
        kernel, emitted for the bins lmtrack_final leaves empty. `bins` = FeatureExtractor.binParams()."""
        pts0 = _f32(pts0).reshape(-1, 2)
        Xw = _f32(Xw).reshape(-1, 3)
        flags = np.ascontiguousarray(flags, np.uint8)
        self._n = pts0.shape[0]
        if Xw.shape[0] != self._n or flags.shape[0] != self._n:
            raise ValueError("pts0 / Xw / flags differ in length")
        self._closed_bins = bins.n_bins_u * bins.n_bins_v
        self.ctx.check(self.lib.vo_mono_frame_enqueue_closed(
            self.ctx.handle, C.byref(self.prm), slots[0], slots[1], pts0.ctypes.data, Xw.ctypes.data, flags.ctypes.data,
            self._n, _f32(Tcw_prev).reshape(16).ctypes.data, _f32(Tcw_prior).reshape(16).ctypes.data,
            _f32(dT01_prior).reshape(16).ctypes.data, C.byref(bins), table, 0))

    def enqueue_closed_device(self, d_pts0, d_Xw, d_flags, n, Tcw_prev, Tcw_prior, dT01_prior, bins, table, slots=(0, 1)):
        self._n = n
        self._closed_bins = bins.n_bins_u * bins.n_bins_v
        self.ctx.check(self.lib.vo_mono_frame_enqueue_closed(
            self.ctx.handle, C.byref(self.prm), slots[0], slots[1], d_pts0, d_Xw, d_flags, n,
            _f32(Tcw_prev).reshape(16).ctypes.data, _f32(Tcw_prior).reshape(16).ctypes.data,
            _f32(dT01_prior).reshape(16).ctypes.data, C.byref(bins), table, 1))

    def result(self):
        n = self._n
        pts1 = np.zeros((max(n, 1), 2), np.float32)
        scale = np.zeros(max(n, 1), np.float32)
        stage = np.zeros(max(n, 1), np.uint8)
        dT = np.zeros(16, np.float32)
        cnt, gn = MonoCounts(), GnInfo()
        self.ctx.check(self.lib.vo_mono_frame_result(self.ctx.handle, _p(pts1), _p(scale), _p(stage, C.c_uint8),
                                                     _p(dT), C.byref(cnt), C.byref(gn)))
        out = dict(pts1=pts1[:n], scale=scale[:n], stage=stage[:n], dT01=dT.reshape(4, 4), counts=cnt, gn=gn)
        nb = getattr(self, "_closed_bins", 0)
        if nb:  # the new points of the closed frame: pixels in I1, back-tracked pixels in I0, masks
            self._closed_bins = 0
            p1n, p0n = np.zeros((nb, 2), np.float32), np.zeros((nb, 2), np.float32)
            mn, nn = np.zeros(nb, np.uint8), C.c_int()
            self.ctx.check(self.lib.vo_mono_frame_new_points(self.ctx.handle, p1n.ctypes.data, p0n.ctypes.data,
                                                             mn.ctypes.data, C.addressof(nn)))
            out.update(pts1_new=p1n[:nn.value], pts0_new=p0n[:nn.value], mask_new=mn[:nn.value].view(bool))
        return out


class Camera:
    """Camera (core/visual_odometry/camera.h:22-137): the distortion model, its image-undistortion map
    (generateImageUndistortMaps, camera.cpp:56-90) and undistortImage (camera.cpp:166-183). The map lives
    on the device (`cam` selects which of the context's two map sets); undistortImage delivers straight
    into an image slot's pyramid, including the driver's convertTo(CV_8UC1) (mono_vo.cpp:512)."""

    def __init__(self, ctx, cam=0):
        self.ctx, self.lib, self.cam = ctx, ctx.lib, cam
        self.initialized = False

    def initParams(self, n_cols, n_rows, K, D):
        self.n_cols, self.n_rows = int(n_cols), int(n_rows)
        self.K, self.D = _f32(K).reshape(4), _f32(D).reshape(5)  # fx fy cx cy ; k1 k2 p1 p2 k3
        self.ctx.check(self.lib.vo_rectify_init_mono(self.ctx.handle, self.cam, self.n_cols, self.n_rows, _p(self.K),
                                                     _p(self.D)))
        self.initialized = True

    def fx(self): return float(self.K[0])
    def fy(self): return float(self.K[1])
    def cx(self): return float(self.K[2])
    def cy(self): return float(self.K[3])

    def maps(self):
        return _get_maps(self.ctx, self.cam)

    def undistortImage(self, raw, slot):
        raw = _u8(raw)
        if raw.size == 0 or raw.shape != (self.n_rows, self.n_cols):  # camera.cpp:168-169
            raise VoError(-4, "undistort image: provided image has not the same size as the camera model!")
        self.ctx.set_image_rectified(slot, raw, self.cam)


def _get_maps(ctx, cam):
    w, h = C.c_int(), C.c_int()
    ctx.check(ctx.lib.vo_rectify_get_maps(ctx.handle, cam, None, None, C.byref(w), C.byref(h)))
    mu = np.zeros((h.value, w.value), np.float32)
    mv = np.zeros((h.value, w.value), np.float32)
    ctx.check(ctx.lib.vo_rectify_get_maps(ctx.handle, cam, _p(mu), _p(mv), None, None))
    return mu, mv


class StereoCamera:
    """StereoCamera (core/visual_odometry/camera.h:140-195): setStereoPoseLeft2Right,
    initStereoCameraToRectify (generateStereoImagesUndistortAndRectifyMaps, camera.cpp:364-546),
    rectifyStereoImages (camera.cpp:300-336), the rectified camera and extrinsics."""

    def __init__(self, ctx):
        self.ctx, self.lib = ctx, ctx.lib
        self.is_initialized_to_stereo_rectify_ = False
        self.T_lr = np.eye(4, dtype=np.float32)

    def initParams(self, n_cols, n_rows, Kl, Dl, Kr, Dr):
        self.n_cols, self.n_rows = int(n_cols), int(n_rows)
        self.Kl, self.Dl = _f32(Kl).reshape(4), _f32(Dl).reshape(5)
        self.Kr, self.Dr = _f32(Kr).reshape(4), _f32(Dr).reshape(5)

    def setStereoPoseLeft2Right(self, T_lr):
        self.T_lr = _f32(T_lr).reshape(4, 4)

    def initStereoCameraToRectify(self):
        K_rect, T1, T2 = np.zeros(4, np.float32), np.zeros(16, np.float32), np.zeros(16, np.float32)
        self.ctx.check(self.lib.vo_rectify_init_stereo(
            self.ctx.handle, self.n_cols, self.n_rows, _p(self.Kl), _p(self.Dl), _p(self.Kr), _p(self.Dr),
            _p(_f32(self.T_lr).reshape(16)), _p(K_rect), _p(T1), _p(T2)))
        self.K_rect, self.T_lr_rect, self.T_rl_rect = K_rect, T1.reshape(4, 4), T2.reshape(4, 4)
        self.is_initialized_to_stereo_rectify_ = True

    def _need_init(self, where):
        if not self.is_initialized_to_stereo_rectify_:
            raise VoError(-1, f"In '{where}', is_initialized_to_stereo_rectify_ == false")

    def getRectifiedCamera(self):
        self._need_init("getRectifiedCamera()")
        return self.K_rect

    def getRectifiedStereoPoseLeft2Right(self):
        self._need_init("getRectifiedStereoPoseLeft2Right()")
        return self.T_lr_rect

    def getRectifiedStereoPoseRight2Left(self):
        self._need_init("getRectifiedStereoPoseRight2Left()")
        return self.T_rl_rect

    def maps(self):
        return _get_maps(self.ctx, 0), _get_maps(self.ctx, 1)

    def rectifyStereoImages(self, img_left, img_right, slot_l, slot_r):
        self._need_init("rectifyStereoImages()")
        for im in (img_left, img_right):  # camera.cpp:307, :324
            if np.asarray(im).shape != (self.n_rows, self.n_cols):
                raise VoError(-4, "In 'rectifyStereoImages()': provided image has not the same size as the camera model!")
        self.ctx.set_image_rectified(slot_l, img_left, 0)
        self.ctx.set_image_rectified(slot_r, img_right, 1)


class SparseBundleAdjustmentSolver:
    """SparseBundleAdjustmentSolver (core/visual_odometry/ba_solver/sparse_bundle_adjustment.h:42-158) on the
    device. The reference configures it through setCamera / setStereoCameras, setHuberThreshold and a
    SparseBAParameters object; here the same pieces of information arrive as arrays (see
    include/vo_hip.h, vo_sba_solve) — the landmark / keyframe graph that SparseBAParameters walks stays
    with the caller."""

    POSE_SCALE = 10.0  # SparseBAParameters::pose_scale_ (sparse_ba_parameters.h:255)

    def __init__(self, ctx, is_stereo=False):
        self.ctx, self.lib, self.is_stereo = ctx, ctx.lib, bool(is_stereo)
        self.Kl = self.Kr = None
        self.T_lr = np.eye(4)
        self.thres_huber = 0.0

    def setCamera(self, K):
        if self.is_stereo:
            raise VoError(-1, "In 'SparseBundleAdjustmentSolver::setCamera()': Before call this function, "
                              "'is_stereo' should be set to 'false'.")
        self.Kl = self.Kr = np.asarray(K, np.float64).reshape(4)

    def setStereoCameras(self, Kl, Kr, T_lr_scaled):
        if not self.is_stereo:
            raise VoError(-1, "In 'SparseBundleAdjustmentSolver::setStereoCameras()': Before call this function, "
                              "'is_stereo' should be set to 'true'.")
        self.Kl, self.Kr = np.asarray(Kl, np.float64).reshape(4), np.asarray(Kr, np.float64).reshape(4)
        self.T_lr = np.asarray(T_lr_scaled, np.float64).reshape(4, 4)

    def setHuberThreshold(self, thres_huber):
        self.thres_huber = float(thres_huber)

    def solveForFiniteIterations(self, MAX_ITER, T_jw, opt_index, X, obs_ptr, obs_frame, obs_right, obs_px):
        """Returns (flag_success, T_jw, X, avg_err); raises VoError where the reference throws (NaN)."""
        T = np.ascontiguousarray(T_jw, np.float64).reshape(-1, 16).copy()
        Xo = np.ascontiguousarray(X, np.float64).reshape(-1, 3).copy()
        opt_index = np.ascontiguousarray(opt_index, np.int32)
        obs_ptr, obs_frame = np.ascontiguousarray(obs_ptr, np.int32), np.ascontiguousarray(obs_frame, np.int32)
        obs_right, obs_px = np.ascontiguousarray(obs_right, np.uint8), np.ascontiguousarray(obs_px, np.float64).reshape(-1, 2)
        if obs_ptr.shape[0] != Xo.shape[0] + 1 or opt_index.shape[0] != T.shape[0]:
            raise ValueError("obs_ptr / opt_index do not match the number of landmarks / frames")
        if obs_frame.shape[0] != obs_px.shape[0] or obs_right.shape[0] != obs_px.shape[0]:
            raise ValueError("observation arrays differ in length")
        p = SbaProblem()
        p.n_frames, p.n_points, p.n_obs = T.shape[0], Xo.shape[0], obs_px.shape[0]
        p.n_opt = int(opt_index.max()) + 1 if opt_index.size else 0
        p.stereo, p.max_iter, p.thres_huber = int(self.is_stereo), int(MAX_ITER), self.thres_huber
        for k in range(4):
            p.Kl[k], p.Kr[k] = float(self.Kl[k]), float(self.Kr[k])
        for k in range(16):
            p.T_lr[k] = float(self.T_lr.reshape(16)[k])
        err = np.zeros(max(int(MAX_ITER), 1), np.float64)
        d, i32 = C.c_double, C.c_int32
        rc = self.ctx.check(self.lib.vo_sba_solve(self.ctx.handle, C.byref(p), _p(T, d), _p(opt_index, i32), _p(Xo, d),
                                                  _p(obs_ptr, i32), _p(obs_frame, i32), _p(obs_right, C.c_uint8),
                                                  _p(obs_px, d), _p(err, d)))
        return bool(rc), T.reshape(-1, 4, 4), Xo, err[: int(MAX_ITER)]


class MonoVO:
    """MonoVO (core/visual_odometry/mono_vo/mono_vo.h:235-243, :267): trackImage(img, timestamp) / getStatistics(), the track
    set, the keyframes and the mono local BA carried on the device (vo_mvo_*). `five_point(pts0, pts1) -> (ok, R10, t10,
    mask)` stands for MotionEstimator::calcPose5PointsAlgorithm (OpenCV calib3d: the caller's; called for the second image
    and whenever the pose-only BA yields no pose)."""

    def __init__(self, ctx, width, height, K, n_bins_u, n_bins_v, five_point, thres_fastscore=15, window_size=15, max_level=5,
                 thres_error=20.0, thres_bidirection=1.0, thres_poseba_error=5, thres_sampson=1.0, thres_parallax=1.0,
                 thres_overlap_ratio=0.7, thres_rotation=3.0, thres_translation=3.0, n_max_keyframes_in_window=9, strict_border=4,
                 local_ba=True, rectify=False):
        self.ctx, self.lib = ctx, ctx.lib
        fe = FeatureExtractor(ctx)
        fe.initParams(width, height, n_bins_u, n_bins_v, THRES_FAST=thres_fastscore)
        p = MvoParams()
        p.frame = make_mono_params(width, height, window_size, max_level, thres_error, thres_bidirection, thres_poseba_error,
                                   thres_sampson, K)
        p.bins = fe.binParams()
        p.kf_overlap_ratio, p.kf_rotation_deg, p.kf_translation = thres_overlap_ratio, thres_rotation, thres_translation
        p.kf_window, p.thres_parallax_deg = n_max_keyframes_in_window, thres_parallax
        p.strict_border, p.local_ba, p.rectify = int(strict_border), int(bool(local_ba)), int(bool(rectify))
        self._user_hook = five_point

        def _hook(user, p0, p1, n, Kp, R10, t10, mask):
            a0 = np.ctypeslib.as_array(p0, shape=(max(n, 1), 2))[:n].copy()
            a1 = np.ctypeslib.as_array(p1, shape=(max(n, 1), 2))[:n].copy()
            try:
                ok, R, t, m = self._user_hook(a0, a1)
            except Exception:  # (an exception must not cross the C boundary: the call fails like the reference's throw)
                return 0
            if not ok:
                return 0
            np.ctypeslib.as_array(R10, shape=(9,))[:] = np.asarray(R, np.float32).reshape(9)
            np.ctypeslib.as_array(t10, shape=(3,))[:] = np.asarray(t, np.float32).reshape(3)
            if n:
                np.ctypeslib.as_array(mask, shape=(n,))[:] = np.asarray(m, bool).astype(np.uint8)
            return 1

        self._cb = FIVE_POINT_FN(_hook)  # (kept alive with the object)
        p.five_point = self._cb
        p.five_point_user = None
        self.prm, self.width, self.height = p, width, height
        self._h = C.c_void_p()
        ctx.check(self.lib.vo_mvo_create(ctx.handle, C.byref(p), C.byref(self._h)))
        ctx._children.add(self)
        self._info = MvoFrameInfo()
        self.stats_frame = []

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vo_mvo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ptr(img):
        if isinstance(img, np.ndarray):
            img = _u8(img)
            if img.ndim != 2:
                raise ValueError("the image must be one u8 plane")
            return img, img.ctypes.data, img.strides[0], 0
        return img, int(img[0]), int(img[1]), 1  # (device address, stride)

    def _out(self):
        self.stats_frame.append(np.array(self._info.T_wc, np.float32).reshape(4, 4))
        return MvoFrameInfo.from_buffer_copy(self._info)

    def trackImage(self, img, timestamp=0.0):
        a, p, st, dev = self._ptr(img)
        self._keep = a
        rc = self.lib.vo_mvo_track(self._h, p, st, dev, float(timestamp), self._info)
        if rc < 0:
            self.ctx.check(rc)
        return self._out()

    def enqueue(self, img, timestamp=0.0):
        a, p, st, dev = self._ptr(img)
        self._keep = a
        rc = self.lib.vo_mvo_enqueue(self._h, p, st, dev, float(timestamp))
        if rc < 0:
            self.ctx.check(rc)

    def prefetch(self, img):
        a, p, st, dev = self._ptr(img)
        self._keep_next = a
        rc = self.lib.vo_mvo_prefetch(self._h, p, st, dev)
        if rc < 0:
            self.ctx.check(rc)

    def result(self):
        rc = self.lib.vo_mvo_result(self._h, self._info)
        if rc < 0:
            self.ctx.check(rc)
        return self._out()

    def runSequence(self, images, k_begin=0, k_end=None):
        """A recorded sequence of (device address, stride) images — or of numpy u8 images of one size and layout (host memory) —
        through the loop inside the library (vo_mvo_run): frames k_begin .. k_end - 1 are collected, frame k_end is left in
        flight. Returns (list of MvoFrameInfo, stamps)."""
        n = len(images)
        k_end = n if k_end is None else int(k_end)
        host = n > 0 and isinstance(images[0], np.ndarray)
        if host:
            if getattr(self, "_seq_ref", None) is not images or len(getattr(self, "_seq_host", ())) != n:
                self._seq_host = [_u8(I) for I in images]
                st0 = self._seq_host[0].strides[0]
                if any(a.ndim != 2 or a.shape != self._seq_host[0].shape or a.strides[0] != st0 for a in self._seq_host):
                    raise ValueError("runSequence: host images must be u8 planes of one size and row stride")
            src = [(a.ctypes.data, a.strides[0]) for a in self._seq_host]
        else:
            src = images
        I = getattr(self, "_seq_I", None)  # kept between calls on the same, unchanged list only (as StereoVO.runSequence)
        fresh = getattr(self, "_seq_ref", None) is not images or I is None or len(I) != n
        if not fresh:
            for k in range(max(int(k_begin), 0), min(k_end + 2, n)):
                if I[k] != int(src[k][0]):
                    fresh = True
                    break
        if fresh:
            self._seq_I = (C.c_void_p * n)(*[int(p[0]) for p in src])
            self._seq_ref = images
        self._seq_stride = int(src[0][1])
        m = k_end - int(k_begin)
        infos = (MvoFrameInfo * max(m, 1))()
        stamps = np.zeros(max(m, 1), np.float64)
        rc = self.lib.vo_mvo_run(self._h, self._seq_I, n, self._seq_stride, 0 if host else 1, int(k_begin), k_end, infos, stamps.ctypes.data)
        if rc < 0:
            self.ctx.check(rc)
        out = [MvoFrameInfo.from_buffer_copy(infos[j]) for j in range(m)]
        for i in out:
            self.stats_frame.append(np.array(i.T_wc, np.float32).reshape(4, 4))
        return out, stamps[:m]

    def getTracks(self):
        """frame_prev_'s related landmarks: dict(ids, pts, Xw, flags, age, cos_parallax)."""
        n = C.c_int()
        self.ctx.check(self.lib.vo_mvo_get_tracks(self._h, None, None, None, None, None, None, 0, C.addressof(n)))
        m = max(n.value, 1)
        ids, pts, X = np.zeros(m, np.int32), np.zeros((m, 2), np.float32), np.zeros((m, 3), np.float32)
        fl, age, cp = np.zeros(m, np.uint8), np.zeros(m, np.int32), np.zeros(m, np.float32)
        self.ctx.check(self.lib.vo_mvo_get_tracks(self._h, ids.ctypes.data, pts.ctypes.data, X.ctypes.data, fl.ctypes.data,
                                                  age.ctypes.data, cp.ctypes.data, m, C.addressof(n)))
        k = n.value
        return dict(ids=ids[:k], pts=pts[:k], Xw=X[:k], flags=fl[:k], age=age[:k], cos_parallax=cp[:k])

    def getKeyframes(self):
        """stats_keyframe (mono_vo.cpp:1130-1155): [(T_wc, mappoints)] of every keyframe so far, current values."""
        n, tot = C.c_int(), C.c_size_t()
        self.ctx.check(self.lib.vo_mvo_keyframe_count(self._h, C.addressof(n)))
        nk = n.value
        if nk == 0:
            return []
        T, cnt = np.zeros((nk, 16), np.float32), np.zeros(nk, np.int32)
        self.ctx.check(self.lib.vo_mvo_get_keyframes(self._h, T.ctypes.data, cnt.ctypes.data, None, 0, C.addressof(tot)))
        X = np.zeros((max(tot.value, 1), 3), np.float32)
        if tot.value:
            self.ctx.check(self.lib.vo_mvo_get_keyframes(self._h, None, None, X.ctypes.data, tot.value, C.addressof(tot)))
        off = np.concatenate([[0], np.cumsum(cnt)])
        return [(T[j].reshape(4, 4).copy(), X[off[j]:off[j + 1]].copy()) for j in range(nk)]

    def getStatistics(self):
        """AlgorithmStatistics: stats_frame (pose per frame as it was when the frame returned)."""
        return dict(stats_frame=list(self.stats_frame))
