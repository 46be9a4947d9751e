/*
 * vo_hip.h — C ABI of libvo_hip.so, the MI355X (gfx950) implementation of the
 * per-frame visual-odometry hot path of ChanghyeonKim93/visual_odometry_ros.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types,
 * no exceptions across the ABI. Every entry point names the reference
 * interface it replaces (paths relative to the reference repository root).
 * The C++ classes in visual_odometry_ros_amd/core/visual_odometry/ (same class
 * and method names as the reference) sit directly on top of these calls; see
 * INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *  - Pixels are float pairs (x,y), row i at pts[2*i], pts[2*i+1]
 *    (cv::Point2f layout, core/defines/define_type.h:16).
 *  - 3-D points are float triples (Eigen::Vector3f layout, define_type.h:17).
 *  - Masks are one uint8_t per element (the adapter converts the reference's
 *    bit-packed std::vector<bool>, define_type.h:33).
 *  - 4x4 / 3x3 matrices are ROW-MAJOR here; Eigen::Matrix4f (PoseSE3,
 *    define_type.h:41) is column-major, the adapter transposes explicitly.
 *  - Unless a parameter is documented as a device pointer, pointers are HOST
 *    pointers; the call is synchronous at return (like the reference methods).
 *  - Return value: VO_OK (0) or a negative vo_status. vo_last_error() gives
 *    the message. Codes VO_ERR_NAN_* mirror the reference's
 *    `throw std::runtime_error` sites; VO_ERR_SIZE its size-mismatch throws.
 *  - There is NO CPU fallback: every compute entry point fails with
 *    VO_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef VO_HIP_H_
#define VO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_HIP_ABI_VERSION 3

typedef enum {
  VO_OK = 0,
  VO_ERR_INVALID = -1,    /* bad argument */
  VO_ERR_HIP = -2,        /* HIP runtime error */
  VO_ERR_NO_DEVICE = -3,  /* no usable gfx950 device */
  VO_ERR_SIZE = -4,       /* reference: throw on size mismatch (feature_tracker.cpp:283, motion_estimator.cpp:670,873) */
  VO_ERR_NAN_AXAY = -5,   /* feature_tracker.cpp:414 "ax ay nan" */
  VO_ERR_NAN_PATCH = -6,  /* feature_tracker.cpp:443,447 */
  VO_ERR_NAN_UPDATE = -7, /* feature_tracker.cpp:465 "dtu dtv nan" */
  VO_ERR_CAPACITY = -8,   /* more points / larger image than vo_config allows */
  VO_ERR_GN_FAILED = -9,  /* stereo_vo.cpp:626 "PoseOnlyStereoBA is failed!" */
  VO_ERR_LBA_NAN = -10    /* sparse_bundle_adjustment.cpp:318,:413 "In LBA, pose becomes nan!", :613, :764 "Local BA NAN!" */
} vo_status;

/* cv::OPTFLOW_USE_INITIAL_FLOW */
#define VO_KLT_USE_INITIAL_FLOW 4

/* mono GN variant: core/visual_odometry/motion_estimator.cpp:793-799 (core)
 * vs standalone/motion_estimator/motion_estimator.cpp:135 (standalone) */
#define VO_GN_VARIANT_CORE 0
#define VO_GN_VARIANT_STANDALONE 1

typedef struct vo_ctx vo_ctx;

typedef struct {
  int device;      /* HIP device ordinal */
  int max_width;   /* largest image accepted by vo_set_image */
  int max_height;
  int max_points;  /* capacity of every per-point buffer */
  int n_slots;     /* image slots (each holds an image pyramid), >= 3 for stereo */
  int max_level;   /* deepest pyramid level ever requested (OpenCV maxLevel) */
} vo_config;

typedef struct {
  int iterations;   /* GN iterations executed */
  float err;        /* last err_curr (motion_estimator.cpp:812 / :1042-1043) */
  float delta_err;
  float delta_norm; /* ||delta_xi|| of the last step */
  int cnt_invalid;  /* outliers in the last iteration */
  int is_nan;       /* pose went NaN: reference returns false, pose untouched */
} vo_gn_info;

/* ---- context ------------------------------------------------------------ */
int vo_abi_version(void);
int vo_device_count(void);
int vo_create(const vo_config *cfg, vo_ctx **out);
void vo_destroy(vo_ctx *ctx);
const char *vo_last_error(const vo_ctx *ctx);
/* hipStream_t of the context, as void* (for event timing by the caller). */
void *vo_stream(vo_ctx *ctx);
int vo_synchronize(vo_ctx *ctx);

/* ---- images & pyramids ---------------------------------------------------
 * Replaces what cv::calcOpticalFlowPyrLK does internally on every call
 * (buildOpticalFlowPyramid: pyrDown 5-tap, REFLECT_101 borders; reference call
 * sites core/visual_odometry/feature_tracker.cpp:29,60,69,108,117,186). A slot
 * keeps its pyramid device-resident so the 4 PyrLK calls of a frame and the
 * next frame reuse it. */
int vo_set_image(vo_ctx *ctx, int slot, const uint8_t *host, int width, int height, int stride);
/* same, `dev` is a DEVICE pointer (image already resident in HBM) */
int vo_set_image_device(vo_ctx *ctx, int slot, const void *dev, int width, int height, int stride);
int vo_swap_slots(vo_ctx *ctx, int slot_a, int slot_b);
/* Both images of a stereo pair in one call (one launch per pyramid level for the pair). */
int vo_set_stereo_pair_device(vo_ctx *ctx, int slot_l, const void *dev_l, int slot_r, const void *dev_r,
                              int width, int height, int stride);
/* Image ingestion on the context's SIDE stream (on != 0): vo_set_image* / vo_set_stereo_pair* enqueue their copies
 * and pyramid chains there, concurrent with whatever the main stream runs (the frame in flight); the operators
 * that read a slot wait for it on the device. A slot may then be rebuilt only after the result of every frame that
 * read it has been collected (VO_ERR_INVALID otherwise). Default off: everything on one stream. */
int vo_set_ingest_side_stream(vo_ctx *ctx, int on);
/* Both images of a stereo pair from HOST memory without a host synchronisation (vo_set_image is synchronous):
 * asynchronous H2D + the pyramid chain on the ingest stream. Pinned host buffers give a true asynchronous copy;
 * the buffers must stay untouched until the frame that reads the slots has returned its result. */
int vo_set_stereo_pair_host_async(vo_ctx *ctx, int slot_l, const uint8_t *host_l, int slot_r, const uint8_t *host_r,
                                  int width, int height, int stride);
/* win > 0: build only levels 0..vo_pyramid_levels(w,h,win,max_level) (what PyrLK with that window
 * can use); 0 (default): every level down to vo_config.max_level. */
int vo_set_pyramid_window_hint(vo_ctx *ctx, int win);
/* effective OpenCV maxLevel for (width,height,win,max_level) */
int vo_pyramid_levels(int width, int height, int win, int max_level);
/* test hook: copy unpadded level `level` of a slot back to host (tightly packed) */
int vo_get_level(vo_ctx *ctx, int slot, int level, uint8_t *host, int *width, int *height);

/* ---- pyramidal LK -------------------------------------------------------
 * cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, winSize,
 * maxLevel, criteria, flags, minEigThreshold) on two slots. max_iter <= 0 and
 * eps <= 0 select what the reference's `{}` TermCriteria yields (30, 0.01).
 * pts1 is in/out when flags has VO_KLT_USE_INITIAL_FLOW. */
int vo_klt_track(vo_ctx *ctx, int slot0, int slot1, const float *pts0, float *pts1, int n,
                 int win, int max_level, int flags, int max_iter, double eps, float min_eig_thr,
                 uint8_t *status, float *err);

/* ---- FeatureTracker (core/visual_odometry/feature_tracker.h:44-104) ------ */
/* FeatureTracker::track, feature_tracker.cpp:13-37 */
int vo_track(vo_ctx *ctx, int slot0, int slot1, const float *pts0, int n, int win, int max_level,
             float thres_err, float *pts_track, uint8_t *mask_valid);
/* FeatureTracker::trackBidirection, feature_tracker.cpp:39-86 */
int vo_track_bidirection(vo_ctx *ctx, int slot0, int slot1, const float *pts0, int n, int win,
                         int max_level, float thres_err, float thres_bidirection,
                         float *pts_track, uint8_t *mask_valid);
/* FeatureTracker::trackBidirectionWithPrior, feature_tracker.cpp:88-169 (pts_track in/out) */
int vo_track_bidirection_with_prior(vo_ctx *ctx, int slot0, int slot1, const float *pts0, int n,
                                    int win, int max_level, float thres_err,
                                    float thres_bidirection, float *pts_track,
                                    uint8_t *mask_valid);
/* FeatureTracker::trackWithPrior, feature_tracker.cpp:171-206 (pts_track in/out) */
int vo_track_with_prior(vo_ctx *ctx, int slot0, int slot1, const float *pts0, int n, int win,
                        int max_level, float thres_err, float *pts_track, uint8_t *mask_valid);
/* FeatureTracker::calcPrior, feature_tracker.cpp:208-234 (K row-major 3x3, Tw1 row-major 4x4) */
int vo_calc_prior(vo_ctx *ctx, const float *pts0, int n_pts0, const float *Xw, int n,
                  const float Tw1[16], const float K[9], float *pts1_prior);
/* FeatureTracker::trackWithScale, feature_tracker.cpp:236-504, fused with the
 * cv::Sobel(ksize 3, CV_32F) pair the drivers compute first
 * (stereo_vo.cpp:549-552, mono_vo.cpp:779-782). pts_track and mask_valid are
 * in/out. strict_border != 0 reproduces the reference's never-reset tap masks
 * for points whose taps leave the image; 0 excludes such taps.
 * 1 = parallel fixed-point replay with a sequential fallback, 2 = sequential
 * replay only (validation of the fallback; same results, slower). */
int vo_track_with_scale(vo_ctx *ctx, int slot0, int slot1, const float *pts0,
                        const float *scale_est, int n, float *pts_track, uint8_t *mask_valid,
                        int strict_border);

/* ---- MotionEstimator (core/visual_odometry/motion_estimator.h:117-120) --- */
/* poseOnlyBundleAdjustment, motion_estimator.cpp:665-861. K = (fx,fy,cx,cy).
 * Returns 1 = true, 0 = false (NaN pose, R01/t01 untouched), <0 = error. */
int vo_gn_pose_mono(vo_ctx *ctx, const float *X, const float *pts1, int n, const float K[4],
                    int thres_reproj_outlier, float R01[9], float t01[3], uint8_t *mask_inlier,
                    int variant, vo_gn_info *info);
/* poseOnlyBundleAdjustment_Stereo, motion_estimator.cpp:863-1088. */
int vo_gn_pose_stereo(vo_ctx *ctx, const float *X, const float *pts_l1, const float *pts_r1,
                      int n, const float Kl[4], const float Kr[4], const float T_lr[16],
                      float thres_reproj_outlier, float T01[16], uint8_t *mask_inlier,
                      vo_gn_info *info);

/* geometry::se3Exp_f (core/util/geometry_library.cpp:386-440, incl. the theta < 1e-7 branch) and inverseSE3_f
 * (:554-560) exactly as the GN kernel evaluates them on the device, on their own: T = exp(xi), xi = (v, w);
 * Tinv (may be NULL) = inverseSE3_f(T). Row-major. A test hook: the estimator never needs the host to call it. */
int vo_se3_exp(vo_ctx *ctx, const float xi[6], float T[16], float Tinv[16]);

/* Epipolar gates. MotionEstimator::calcSampsonDistance(pts0, pts1, F10, out)
 * (motion_estimator.cpp:572-599) and calcSymmetricEpipolarDistance (:621-653, its per-point part):
 * F10 row-major 3x3. The camera/pose overloads (:538-570) build F10 = Kinv^T [t10]x R10 Kinv on the
 * host and call these. */
int vo_sampson_distance(vo_ctx *ctx, const float *pts0, const float *pts1, int n, const float F10[9],
                        float *dist);
int vo_symmetric_epipolar_distance(vo_ctx *ctx, const float *pts0, const float *pts1, int n,
                                   const float F10[9], float *dist);

/* ---- FeatureExtractor bucketing (the reference-owned part; cv::ORB::detect stays on the host) */
/* WeightBin::reset + update, feature_extractor.h:120-135: weight[v*n_bins_u+u] = 0 for bins that hold a
 * point (u = floor(x / u_step), only the flattened index is range-tested, as in the reference), else 1. */
int vo_weight_bin_update(vo_ctx *ctx, const float *pts, int n, int u_step, int v_step, int n_bins_u,
                         int n_bins_v, int32_t *weight);
/* The flag_nonmax_ branch of extractORBwithBinning_fast, feature_extractor.cpp:241-277: per bin with
 * weight > 0 the FIRST keypoint (detector order) of largest response; bins ascending. kp_xy / kp_response:
 * the n keypoints cv::ORB::detect returned (n <= vo_config.max_points). idx_out may be NULL. */
int vo_bucket_argmax(vo_ctx *ctx, const float *kp_xy, const float *kp_response, int n, float inv_u_step,
                     float inv_v_step, int n_bins_u, int n_bins_v, const int32_t *weight, float *pts_out,
                     int32_t *idx_out, int *n_out);

/* ---- FeatureExtractor::extractORBwithBinning_fast: detection ----------------------
 * `extractor_orb_->detect(img, fts)` (feature_extractor.cpp:241) on the image held by `slot`. cv::ORB is
 * OpenCV 4 (not in the reference tree); its detection pipeline — 8-level INTER_LINEAR_EXACT pyramid, FAST-9/16
 * with non-max suppression, runByImageBorder, retainBest on the FAST score, HarrisResponses, retainBest on the
 * Harris response, pt *= scale — is restated (oracle/oracle_orb.c says what could not be verified here).
 * Keypoints come back as level-0 pixel coordinates, Harris response and octave, ordered by level and then
 * raster order (cv's own order is unspecified). Orientation is not computed (the reference does not read it). */
typedef struct {
  int nfeatures;        /* setMaxFeatures(10000), feature_extractor.cpp:48 */
  double scale_factor;  /* 1.2 */
  int n_levels;         /* 8 */
  int edge_threshold;   /* 31 */
  int fast_threshold;   /* THRES_FAST */
} vo_orb_params;
int vo_orb_detect(vo_ctx *ctx, int slot, const vo_orb_params *prm, float *kp_xy, float *kp_response,
                  int32_t *kp_octave, int max_kp, int *n_out);
/* test hook: pyramid level >= 1 of the last detection, tightly packed */
int vo_orb_get_level(vo_ctx *ctx, int level, uint8_t *host, int *width, int *height);
/* detection + the flag_nonmax_ bucketing of feature_extractor.cpp:244-277 (vo_bucket_argmax) chained on the
 * device: only the bucketed pixels are copied back. n_detected (may be NULL) = number of keypoints before bucketing. */
int vo_extract_orb_with_binning(vo_ctx *ctx, int slot, const vo_orb_params *prm, float inv_u_step, float inv_v_step,
                                int n_bins_u, int n_bins_v, const int32_t *weight, float *pts_out, int *n_out,
                                int *n_detected);

/* The same, asynchronous: the kernels run on the context's side stream behind the last pyramid build (vo_set_image*),
 * so that the detection of an image overlaps the frame operator working on it. One detection in flight per
 * context; the image slot must stay untouched until _result() has returned. */
int vo_extract_orb_with_binning_enqueue(vo_ctx *ctx, int slot, const vo_orb_params *prm, float inv_u_step,
                                        float inv_v_step, int n_bins_u, int n_bins_v, const int32_t *weight);
int vo_extract_orb_with_binning_result(vo_ctx *ctx, float *pts_out, int *n_out, int *n_detected);

/* ---- FeatureExtractor::descriptorDistance (feature_extractor.cpp:338-357) - */
/* all-pairs 256-bit Hamming distance, dist is na x nb row-major */
int vo_orb_hamming(vo_ctx *ctx, const uint8_t *a, int na, const uint8_t *b, int nb,
                   uint16_t *dist);
/* nearest / second-nearest + the test/test_orbmatching.cpp:87-137 accept rule */
int vo_orb_match(vo_ctx *ctx, const uint8_t *a, int na, const uint8_t *b, int nb, int th_low,
                 float ratio, int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist);

/* ---- landmark mask compaction (landmark.cpp:291-332, :194-231) ----------- */
/* stable compaction indices of mask && alive && tracked; returns count in *n_out */
int vo_compact_indices(vo_ctx *ctx, const uint8_t *mask, const uint8_t *alive,
                       const uint8_t *tracked, int n, int32_t *index_valid, int *n_out);

/* ---- track IDs (landmark.h:64, landmark.cpp:6,:29; frame.h:53, frame.cpp:15,:35) ----------------
 * The reference numbers landmarks and frames with two process-global counters (inline static
 * Landmark::landmark_counter_, Frame::frame_counter_). Here every vo_ctx — one image stream — owns its pair, so the
 * streams of a batch-of-sequences process get the IDs each would get in a process of its own.
 * Both start at 0 (vo_create). */
int vo_ids_reset(vo_ctx *ctx, int32_t next_landmark_id, int32_t next_frame_id);
int vo_ids_peek(const vo_ctx *ctx, int32_t *next_landmark_id, int32_t *next_frame_id);
/* n Frame constructions in order: ids[i] = frame_counter_++. StereoFrame(cam_l, cam_r, t) is two of them, left
 * first (frame.cpp:176-180). */
int vo_ids_new_frames(vo_ctx *ctx, int n, int32_t *ids);
/* Landmark constructions for the candidates i = 0..n-1 in order, where accept[i] != 0 (NULL = all), e.g. step [10]'s
 * `mask_new[i] && Xl(2) > 0 && Xr(2) > 0` (stereo_vo.cpp:716-729): ids[i] = landmark_counter_++, -1 where rejected. */
int vo_ids_new_landmarks(vo_ctx *ctx, const uint8_t *accept, int n, int32_t *ids, int *n_created);
/* StereoLandmarkTracking(src, mask) / LandmarkTracking(src, mask) with their side effect (landmark.cpp:291-332,
 * :194-231): index_valid = stable compaction of mask && alive && tracked (as vo_compact_indices), every other
 * landmark is setUntracked() — tracked[] is in/out —, and ids_out[k] = ids[index_valid[k]] (both may be NULL). */
int vo_compact_tracks(vo_ctx *ctx, const uint8_t *mask, const uint8_t *alive, uint8_t *tracked, const int32_t *ids,
                      int n, int32_t *index_valid, int32_t *ids_out, int *n_out);

/* ---- steady-state stereo frame -------------------------------------------
 * The operator sequence of StereoVO::trackStereoImages, steps [3]-[7] and the
 * tracking part of [10] (core/visual_odometry/stereo_vo/stereo_vo.cpp:483-711),
 * chained on the device with no host round trip. */
typedef struct {
  int width, height;
  int win, max_level;
  float thres_err, thres_bidirection, thres_poseba;
  float Kl[4], Kr[4];
  float T_lr[16];
  float thres_sampson;  /* feature_tracker.thres_sampson (stereo_vo.cpp:245, :395): step [7] keeps a feature whose "distance" —
                           100 for pts_l1.y > 660, else 0 (:653-668: the epipolar distance itself is commented out in the reference) —
                           is below it. 60 in kitti_00_stereo.yaml, 0.5 in most others: the same gate; above 100 it never fires */
} vo_stereo_params;

typedef struct {
  int n_l0l1, n_refine, n_l1r1, n_inlier, n_new_ok;
  int gn_iterations;
  int n_replayed; /* strict border: features whose refinement was replayed (their IC window left the image) */
  int n_ba;       /* size of the pose-only BA set: survivors of [5] whose landmark is triangulated (stereo_vo.cpp:599) */
} vo_frame_counts;

/* flags[i] of a tracked feature in the stereo frame: bit 0 = lm->isTriangulated(); bit 1 = the landmark is no
 * longer alive or tracked (!(lm->isAlive() && lm->isTracked())): the first compaction of the frame drops it
 * whatever the tracker says (landmark.cpp:305), so it never reaches trackWithScale or the BA. */
#define VO_LM_TRIANGULATED 1
#define VO_LM_DROPPED 2
/* flags[i] in the mono frame: bit 0 = lm->isBundled(), bit 1 = member of the pose-only BA class, bit 2 = no longer
 * alive or tracked (landmark.cpp:207) */
#define VO_MONO_LM_BUNDLED 1
#define VO_MONO_LM_BA_CLASS 2
#define VO_MONO_LM_DROPPED 4

/* strict != 0: the frame's trackWithScale step replays border-touching points
 * with the reference's never-reset tap state (same as vo_track_with_scale's
 * strict_border; 2 = sequential replay only, for validation). Default 0.
 * 3 (stereo and mono frames) = the parallel replay runs on a stream of its own NEXT TO the frame kernel, as a pool of
 * resident workgroups that pick the border-touching features up as the frame kernel lists them; a dependency chain
 * starts when its members are past their first refinement instead of behind the frame kernel's last wavefront. Joined
 * on the device (no HIP event). The pool costs the frame kernel room, so it pays only when there is something to
 * replay. It needs the two queues to really run concurrently: under a tool that serialises kernels across queues
 * (rocprofv3 --pmc) its bounded waits run out; the frame is then issued again with the stream-ordered replay and the
 * context stays on it (vo_stereo_frame_recoveries).
 * 5 (stereo frame only; the mono frame takes 1) = the stream-ordered replay of mode 1, but on the replay stream behind a one-wavefront gate that
 * waits for the frame kernel's last pass 1: its (normally idle) launches run under the frame kernel's tail instead of
 * between it and the BA launch. Same conditions as 3.
 * 4 (stereo and mono frames) = 3 or 1, chosen per frame: 3 when the previous frame replayed at least 16 features and the
 * frame kernel is small enough (at most 4096 features + candidates) to leave the chip mostly idle in its second half.
 * Every non-zero value gives the same results (DESIGN.md §4.3 has the measurements). */
int vo_stereo_frame_set_strict_border(vo_ctx *ctx, int strict);

/* Frames of this context that vo_stereo_frame_result had to issue again because the device-side join of modes 3 / 5
 * timed out (the replay stream did not run next to the frame kernel: a serialising tool, a shared GPU). Such a frame
 * is re-run with the stream-ordered replay (its inputs are still on the device) and delivers the normal results; the
 * context then stays on the stream-ordered arrangement. 0 in normal operation. */
int vo_stereo_frame_recoveries(const vo_ctx *ctx);

/* ---- test / measurement hooks of one context ----------------------------------------------------------------
 * The library reads no switch from the environment (VO_SVO_TRACE, which only prints timings, is the exception); a test or
 * a measurement script sets these per context. Every value defaults to 0 = normal operation.
 *   VO_DBG_FAIL_JOIN      != 0: the device-side join of strict-border modes 3 / 5 waits for a count that never comes
 *                         (as under a tool that serialises the queues) — exercises the re-issue path
 *   VO_DBG_CONC_GRID      > 0: workgroups of the concurrent replay's pool (a pool smaller than the list)
 *   VO_DBG_SBA_LDS_SOLVE  != 0: the local BA's general dense solve (matrix in LDS) for every window size
 *   VO_DBG_SKIP_DETECT    != 0: a candidate table filled twice keeps its content (what a frame costs without the
 *                         detection under it; the results are those of a stale table)
 *   VO_OPT_POLL_YIELD     != 0: the result polls (a pinned word written last by the frame's BA launch / the local BA) give the
 *                         CPU up between looks (sched_yield) instead of spinning — for hosts where the ranks or streams
 *                         outnumber the idle cores (8 ranks on a few cores: a spinning rank keeps another one's launches waiting)
 *   VO_DBG_STAGED_DETECT  != 0: the per-bin candidate table of the closed step [10] by the per-stage detector kernels (19
 *                         launches) instead of the two tile kernels — the same table, bit for bit (tests compare the two)
 *   VO_DBG_MVO_HOST_ADVANCE != 0: MonoVO launches a frame's advance step (the next track set) from vo_mvo_result, behind the
 *                         frame's result, instead of chaining it behind the BA launch in vo_mvo_enqueue (measurement) */
enum { VO_DBG_FAIL_JOIN = 0, VO_DBG_CONC_GRID = 1, VO_DBG_SBA_LDS_SOLVE = 2, VO_DBG_SKIP_DETECT = 3, VO_OPT_POLL_YIELD = 4,
       VO_DBG_MVO_HOST_ADVANCE = 5, VO_DBG_STAGED_DETECT = 6, VO_DBG_COUNT = 8 };
int vo_debug_set(vo_ctx *ctx, int key, int value);
/* Device and pinned-host allocations made on behalf of this context so far (vo_create included). A steady-state frame —
 * keyframes and their local BA included — makes none: tests/test_stereo_vo_gpu.py asserts it. */
int vo_debug_allocation_count(const vo_ctx *ctx, long long *count);

/* Asynchronous: enqueues one frame on the context stream. slot_l0 must hold the
 * previous left pyramid, slot_l1 / slot_r1 the current pair. Track-set inputs
 * are DEVICE pointers when `inputs_on_device` != 0, else host pointers.
 * flags[i] bit 0 (VO_LM_TRIANGULATED) = lm->isTriangulated() of feature i; NULL = every landmark is
 * triangulated. New stereo landmarks stay untriangulated until the next keyframe (stereo_vo.cpp:736, :794), so a
 * real track set is a mix: an untriangulated feature takes pts_l0 / pts_r0 as prior and patch scale 1
 * (stereo_vo.cpp:490, :515-519), goes through [4], [4-1], [5] like every other, stays out of the pose-only BA
 * (:599) with mask_motion = true (:582) and meets the y > 660 gate of [7] (:653-668). Xp[i] (the landmark in the
 * previous left camera frame, T_pw * X) is read only where the bit is set. Bit 1 (VO_LM_DROPPED): see above.
 * One frame in flight per context (VO_ERR_INVALID otherwise). */
int vo_stereo_frame_enqueue(vo_ctx *ctx, const vo_stereo_params *prm, int slot_l0, int slot_l1,
                            int slot_r1, const float *pts_l0, const float *pts_r0,
                            const float *Xp, const uint8_t *flags, int n, const float dT_prior[16],
                            const float *pts_new, int n_new, int inputs_on_device);
/* ---- step [10] closed on the device ------------------------------------------------------------------------
 * In the reference the new-point candidates of a frame are extractORBwithBinning_fast(I1_left) AFTER
 * updateWeightBin(lmtrack_final.pts_l1) (stereo_vo.cpp:691-693), i.e. they depend on this frame's BA. What depends on
 * the BA is only WHICH bins ask for a point; the detection and the best keypoint of each bin depend on the image
 * alone. So: vo_new_point_candidates_enqueue runs cv::ORB::detect's restatement and the per-bin arg-max for EVERY bin
 * as soon as the image is in its slot (side stream, next to the previous frame), vo_stereo_frame_enqueue_closed
 * tracks every bin's candidate bidirectionally next to the features (speculatively), and the BA launch's epilogue
 * applies updateWeightBin and emits, bins ascending, exactly the candidates the reference would have extracted and
 * tracked — no host round trip and no dependent launch behind the BA. Two tables (0 / 1) so that the detection of
 * pair k+1 can run while frame k is in flight. */
typedef struct {
  int n_bins_u, n_bins_v;        /* FeatureExtractor::initParams */
  int u_step, v_step;            /* WeightBin::init, feature_extractor.h:103-104: (int)floor(n_cols / n_bins_u), ... */
  float inv_u_step, inv_v_step;  /* :106-107 */
  vo_orb_params orb;
} vo_bin_params;
int vo_new_point_candidates_enqueue(vo_ctx *ctx, int slot, const vo_bin_params *bins, int table);
/* test hook: waits for the table; xy[n_bins][2], has[n_bins] (either may be NULL), keypoints detected */
int vo_new_point_candidates_get(vo_ctx *ctx, int table, float *xy, uint8_t *has, int *n_detected);
/* vo_stereo_frame_enqueue with the candidates taken from `table` (which must hold the detection of the image in
 * slot_l1). n must be > 0 and prm->win one of 13, 15, 21, 31. Results: vo_stereo_frame_result (pts_new_r, mask_new:
 * room for n_bins entries; counts->n_new_ok) and vo_stereo_frame_new_points. */
int vo_stereo_frame_enqueue_closed(vo_ctx *ctx, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                                   const float *pts_l0, const float *pts_r0, const float *Xp, const uint8_t *flags,
                                   int n, const float dT_prior[16], const vo_bin_params *bins, int table,
                                   int inputs_on_device);
/* The closed frame on the reference's own data flow (stereo_vo.cpp:475-522, :595-613): Xw holds the landmarks in the
 * WORLD frame (lm->get3DPoint()), T_pw = stframe_prev->getLeft()->getPoseInv(), T_cw_prior = inverseSE3_f(T_wp *
 * dT_pc_prev), dT_prior = dT_pc_prev (the BA's initial value, :585). X_l1 = T_cw_prior X, the patch scale is
 * (T_pw X)(2) / X_l1(2) and the BA takes Xp = T_pw X, each in Eigen's evaluation order of `R * X + t`. */
int vo_stereo_frame_enqueue_closed_world(vo_ctx *ctx, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                                         const float *pts_l0, const float *pts_r0, const float *Xw, const uint8_t *flags,
                                         int n, const float dT_prior[16], const float T_pw[16], const float T_cw_prior[16],
                                         const vo_bin_params *bins, int table, int inputs_on_device);
/* after vo_stereo_frame_result of a closed frame: the candidates' left pixels (pts_l1_new; room for n_bins) */
int vo_stereo_frame_new_points(vo_ctx *ctx, float *pts_new, int *n_new);

/* Waits for the last enqueued frame and copies its results to host buffers
 * (any of which may be NULL). stage[i] = number of gates point i passed (0..4):
 * 1 [4], 2 [4-1], 3 [5], 4 = in lmtrack_final (BA inlier or untriangulated, and past the gate of [7]). */
int vo_stereo_frame_result(vo_ctx *ctx, float *pts_l1, float *pts_r1, uint8_t *stage,
                           float dT[16], float *pts_new_r, uint8_t *mask_new,
                           vo_frame_counts *counts, vo_gn_info *gn);

/* ---- StereoVO: the closed loop around the frame ---------------------------------------------------------------
 * StereoVO::trackStereoImages (core/visual_odometry/stereo_vo/stereo_vo.cpp:392-989, class surface stereo_vo.h:233-249)
 * with the track set carried ON THE DEVICE from frame to frame: what enters frame k+1 is what frame k left behind —
 * lmtrack_final (the stage-4 survivors in index order, :670) followed by the new landmarks of step [10] (:714-739:
 * trackBidirection mask and both DLT depths positive, mapping::triangulateDLT = core/util/triangulate_3d.cpp:91-130
 * with Eigen's 4x4 JacobiSVD restated), landmark ids from the context's counter (landmark.cpp:29), the previous pose
 * and the constant-velocity prior (:475-480, frame.cpp:50-54), the keyframe rule (keyframes.cpp:217-303) and, at a
 * keyframe, the reconstruction of lmtrack_final (:763-797) and the local bundle adjustment over the keyframe window
 * (:802 -> motion_estimator.cpp:1207-1340, sparse_ba_parameters.h:292-466, sparse_bundle_adjustment.cpp:624-722).
 * The first pair is the reference's initialisation (:842-949). The host sees one small result block per frame.
 * Needs a context with >= 5 image slots (previous left, current pair, next pair) and prm.frame.win in {13,15,21,31}. */
typedef struct vo_svo vo_svo;
typedef struct {
  vo_stereo_params frame;   /* Camera.*, feature_tracker.*, motion_estimator.thres_poseba_error */
  vo_bin_params bins;       /* feature_extractor.* as FeatureExtractor::initParams derives them */
  float kf_overlap_ratio;   /* keyframe_update.thres_alive_ratio   (StereoKeyframes::setThresOverlapRatio) */
  float kf_rotation_deg;    /* keyframe_update.thres_rotation      (degrees; the reference multiplies by D2R) */
  float kf_translation;     /* keyframe_update.thres_trans */
  int kf_window;            /* keyframe_update.n_max_keyframes_in_window (with local_ba: at most 16) */
  int strict_border;        /* vo_stereo_frame_set_strict_border */
  int local_ba;             /* != 0: localBundleAdjustmentSparseSolver_Stereo at every keyframe (the reference's behaviour);
                               the window may span at most 2^24 landmark ids — the distance from the oldest landmark
                               still tracked at the window's oldest keyframe to the newest, ~80 000 frames of a track
                               that never dies (VO_ERR_CAPACITY beyond). Landmark table, keyframe window and the BA
                               problem live on the device, allocated by vo_svo_create (vo_svo_device_bytes: ~32 MB at
                               configs[1]); the table doubles with the ids handed out (17 B per landmark ever created,
                               like the reference's all_landmarks_; beyond 2^24 the oldest read as the origin) */
  int rectify;              /* != 0: flagDoUndistortion (stereo_vo.cpp:414-427) — every incoming pair goes through the
                               context's stereo rectification maps (vo_rectify_init_stereo / vo_rectify_set_maps first) on
                               its way into the pyramids; frame.Kl / Kr / T_lr are then the RECTIFIED camera and extrinsics */
} vo_svo_params;
typedef struct {
  int frame_id;             /* id of the left Frame of this pair (the right one is frame_id + 1, frame.cpp:176-180) */
  int is_first, is_keyframe, lba_ran;
  int n_tracks_in;          /* size of the track set that entered the frame */
  int n_final;              /* lmtrack_final.n_pts: stage-4 survivors */
  int n_new;                /* landmarks created in step [10] */
  int n_tracks_out;         /* n_final + n_new: the next frame's track set */
  int n_kf_tracked;         /* survivors that belong to the last keyframe (numerator of the overlap ratio) */
  int n_new_candidates;     /* candidates step [10] extracted (bins left empty that hold a keypoint) */
  vo_frame_counts counts;
  vo_gn_info gn;
  float dT[16];             /* dT_pc_poBA */
  float T_wc[16];           /* pose of the left camera after this call (after the local BA at a keyframe) */
  double lba_err_first, lba_err_last; /* average pixel error of the local BA's first / last iteration */
  int lba_landmarks, lba_observations;
} vo_svo_frame_info;
int vo_svo_create(vo_ctx *ctx, const vo_svo_params *prm, vo_svo **out);
void vo_svo_destroy(vo_svo *svo);
/* StereoVO::trackStereoImages(img_left, img_right, timestamp): synchronous. Images: tightly addressed u8, `stride`
 * bytes per row; device pointers when on_device != 0, else host pointers (pinned memory gives a true asynchronous
 * copy). Errors: VO_ERR_GN_FAILED = the reference's throw "PoseOnlyStereoBA is failed!" (:626), VO_ERR_LBA_NAN. */
int vo_svo_track(vo_svo *svo, const void *left, const void *right, int stride, int on_device, double timestamp,
                 vo_svo_frame_info *info);
/* The same in two halves, so that a caller that already holds the NEXT pair (a recorded sequence) can hand it over
 * while this frame is in flight: vo_svo_enqueue(k); vo_svo_prefetch(k+1); vo_svo_result(k). The prefetch builds the
 * pair's pyramids and the per-bin candidate table on the side stream; the following vo_svo_enqueue / vo_svo_track with
 * the SAME two pointers uses them (other pointers: the prefetch is dropped and the pair is ingested as usual). Host
 * buffers of a prefetched pair must stay untouched until that pair's frame has returned. */
int vo_svo_enqueue(vo_svo *svo, const void *left, const void *right, int stride, int on_device, double timestamp);
int vo_svo_prefetch(vo_svo *svo, const void *left, const void *right, int stride, int on_device);
int vo_svo_result(vo_svo *svo, vo_svo_frame_info *info);
/* A recorded sequence through the same three calls, the loop on this side of the ABI: collects frames k_begin .. k_end - 1
 * of the n_total pairs left[k] / right[k] — vo_svo_result(k), then at once vo_svo_enqueue(k + 1) and vo_svo_prefetch(k + 2).
 * k_begin == 0 starts the sequence; otherwise frame k_begin is the one a previous call left in flight, and frame k_end is
 * in flight on return (when there is one; collect it with vo_svo_result or the next vo_svo_run). infos (may be NULL):
 * [k_end - k_begin]; stamps (may be NULL): CLOCK_MONOTONIC seconds at which each frame's result was in hand. */
int vo_svo_run(vo_svo *svo, const void *const *left, const void *const *right, int n_total, int stride, int on_device,
               int k_begin, int k_end, vo_svo_frame_info *infos, double *stamps);
/* The track set the next frame will start from (stframe_prev_'s pts seen + related landmarks): ids, left / right
 * pixels, world points, flags (VO_LM_TRIANGULATED, VO_LM_DROPPED, VO_LM_KF_MEMBER). Any pointer may be NULL; *n
 * receives the size. A device-to-host copy with a synchronisation: a test / inspection hook, not part of the loop. */
#define VO_LM_KF_MEMBER 4
int vo_svo_get_tracks(vo_svo *svo, int32_t *ids, float *pts_l, float *pts_r, float *Xw, uint8_t *flags, int cap, int *n);
/* step [10] of the last frame: the extracted candidates (left / right pixels, trackBidirection mask) and which of them
 * became landmarks (stereo_vo.cpp:716-725). Capacity n_bins each. */
int vo_svo_get_new_points(vo_svo *svo, float *pts_l, float *pts_r, uint8_t *mask_new, uint8_t *accept, int *n);
/* mapping::triangulateDLT (triangulate_3d.cpp:91-130) for n pixel pairs on the device: X0 (first camera), X1 = R10 X0 +
 * t10. T_10 row-major 4x4 (e.g. T_rl); K0 / K1 = fx, fy, cx, cy. */
int vo_triangulate_dlt(vo_ctx *ctx, const float *pts0, const float *pts1, int n, const float T_10[16], const float K0[4],
                       const float K1[4], float *X0, float *X1);

/* ---- batch of sequences on one device ---------------------------------------------------------------------
 * S independent stereo streams on ONE GPU, each with its own context (HIP streams, slots, landmark / frame id counters:
 * SURVEY F11), its own StereoVO and its own host thread inside the library. (BASELINE's batch mode proper is one stream
 * per GPU; a single sequential stream is latency-bound and leaves most of an MI355X idle.) The streams share nothing.
 * vo_batch_run: stream s tracks the pairs left[s * n_frames + k], right[...] (k = 0..n_frames-1; device pointers when
 * on_device != 0) in order, pair k+1 handed over while frame k is in flight; the first `warmup` frames of every stream are
 * untimed and all streams start the timed part together. Outputs (each may be NULL): T_wc [n_streams][n_frames][16],
 * last_ids [n_streams][ids_cap] + n_ids [n_streams] (every stream's final track-set ids), seconds [n_streams] (timed
 * wall time per stream), *wall (first start to last end). */
typedef struct vo_batch vo_batch;
int vo_batch_create(const vo_config *cfg, const vo_svo_params *prm, int n_streams, vo_batch **out);
void vo_batch_destroy(vo_batch *batch);
const char *vo_batch_last_error(const vo_batch *batch);
/* vo_debug_set on every stream's context (measurement switches, e.g. VO_DBG_SKIP_DETECT) */
int vo_batch_debug_set(vo_batch *batch, int key, int value);
/* The strict-border arrangement the batch's streams run with: prm->strict_border, except that the arrangements with a
 * replay next to the frame kernel (3, 4, 5) become the stream-ordered one (1, same results) when n_streams > 1. */
int vo_batch_strict_border(const vo_batch *batch);
int vo_batch_run(vo_batch *batch, const void *const *left, const void *const *right, int n_frames, int stride, int on_device,
                 int warmup, float *T_wc, int32_t *last_ids, int ids_cap, int *n_ids, double *seconds, double *wall);

/* AlgorithmStatistics::stats_keyframe as trackStereoImages refreshes it at every keyframe (stereo_vo.cpp:805-821; read by
 * the ROS 2 node for its trajectory and map-point topics, ros2/visual_odometry/stereo_vo_ros2.cpp:141-166): every
 * keyframe so far, j = 0 .. count-1, with its CURRENT pose and the current 3-D points of its related landmarks (the local
 * BA keeps changing both). mappoints may be NULL (count only); cap = room for that many points. */
int vo_svo_keyframe_count(vo_svo *svo, int *n_keyframes);
int vo_svo_get_keyframe(vo_svo *svo, int j, float T_wc[16], float *mappoints, int cap, int *n_points);
/* all of them at once (what trackStereoImages rewrites at every keyframe): T_wc [n_keyframes][16], n_points [n_keyframes],
 * mappoints [total][3] in keyframe order; any of the three may be NULL; *total_points = sum of n_points */
int vo_svo_get_keyframes(vo_svo *svo, float *T_wc, int32_t *n_points, float *mappoints, size_t cap_points, size_t *total_points);
/* Device memory this StereoVO holds for its keyframes (landmark table, keyframe ring, keyframe pool, the local BA's window
 * scratch and arena): ~32 MB at BASELINE configs[1] (max_points 4024, window of 9), all of it allocated by vo_svo_create. */
int vo_svo_device_bytes(const vo_svo *svo, size_t *bytes);

/* ---- undistortion / stereo rectification in front of the trackers ----------
 * core/visual_odometry/camera.cpp. A context holds the maps of two cameras
 * (cam 0 = left or the mono camera, cam 1 = right), device-resident. */
/* Camera::generateImageUndistortMaps (camera.cpp:56-90): K = fx,fy,cx,cy ; D = k1,k2,p1,p2,k3 */
int vo_rectify_init_mono(vo_ctx *ctx, int cam, int width, int height, const float K[4], const float D[5]);
/* StereoCamera::generateStereoImagesUndistortAndRectifyMaps (camera.cpp:364-546): maps of cam 0 and 1;
 * returns the rectified camera K_rect = f,f,cu,cv (getRectifiedCamera) and the rectified extrinsics
 * (getRectifiedStereoPoseLeft2Right / Right2Left; row-major 4x4; T_rl_rect may be NULL) */
int vo_rectify_init_stereo(vo_ctx *ctx, int width, int height, const float Kl[4], const float Dl[5],
                           const float Kr[4], const float Dr[5], const float T_lr[16], float K_rect[4],
                           float T_lr_rect[16], float T_rl_rect[16]);
/* caller-made maps (what cv::remap takes as map1 / map2, CV_32FC1, width x height, tightly packed) */
int vo_rectify_set_maps(vo_ctx *ctx, int cam, const float *map_u, const float *map_v, int width, int height);
int vo_rectify_get_maps(vo_ctx *ctx, int cam, float *map_u, float *map_v, int *width, int *height);
/* Camera::undistortImage / StereoCamera::rectifyStereoImages (camera.cpp:166-183, :300-336) +
 * convertTo(CV_8UC1) (stereo_vo.cpp:420-421, mono_vo.cpp:512), fused into the pyramid build of the slot:
 * vo_set_image* with the raw image and the camera whose map applies. The image must have the map's
 * size (VO_ERR_SIZE otherwise, as the reference throws). */
int vo_set_image_rectified(vo_ctx *ctx, int slot, const uint8_t *host, int width, int height, int stride, int cam);
int vo_set_image_rectified_device(vo_ctx *ctx, int slot, const void *dev, int width, int height, int stride,
                                  int cam);
/* left image through cam 0's map, right image through cam 1's, one launch chain */
int vo_set_stereo_pair_rectified_device(vo_ctx *ctx, int slot_l, const void *dev_l, int slot_r, const void *dev_r,
                                        int width, int height, int stride);

/* ---- steady-state mono frame ----------------------------------------------
 * The operator sequence of MonoVO::trackImage
 * (core/visual_odometry/mono_vo/mono_vo.cpp:739-963): prior pixel + patch scale
 * (:739-761), trackBidirectionWithPrior (:768), trackWithScale (:779-788), the
 * selection of the pose-only BA set (:799-826), poseOnlyBundleAdjustment
 * (:856-867), mask_motion (:872-879) and the Sampson gate (:954-963), chained on
 * the device. The 5-point fallback (:905-935, OpenCV calib3d) stays with the
 * caller: counts.need_five_point says when it is due (fewer than 11 BA points,
 * or the BA failed); stages then stop at 2 and dT01 is the prior. */
typedef struct {
  int width, height;
  int win, max_level;
  float thres_err, thres_bidirection;
  int thres_poseba;      /* int, as the reference's poseOnlyBundleAdjustment takes it */
  float thres_sampson;
  float K[4];
} vo_mono_params;

typedef struct {
  int n_klt, n_refine, n_ba, n_motion, n_final;
  int gn_iterations;
  int need_five_point;
  int n_replayed;
} vo_mono_counts;

/* flags[i]: bit 0 = lm->isBundled() (prior pixel and patch scale come from the
 * 3-D point), bit 1 = the landmark is in the class this frame hands to the
 * pose-only BA (mono_vo.cpp:800-826), bit 2 = VO_MONO_LM_DROPPED (fails the first
 * compaction, mono_vo.cpp:773 -> landmark.cpp:207). Xw is read only where bit 0 or 1 is set.
 * Tcw_prev = inverse pose of the previous frame, Tcw_prior = inverse of the
 * predicted current pose, dT01_prior = predicted motion (all row-major 4x4).
 * The strict-border mode is the context's (vo_stereo_frame_set_strict_border). */
int vo_mono_frame_enqueue(vo_ctx *ctx, const vo_mono_params *prm, int slot0, int slot1,
                          const float *pts0, const float *Xw, const uint8_t *flags, int n,
                          const float Tcw_prev[16], const float Tcw_prior[16],
                          const float dT01_prior[16], int inputs_on_device);
/* stage[i] = gates passed: 1 tracked, 2 refined, 3 motion inlier (or not in the
 * BA set), 4 passed the Sampson gate. pts1 = refined pixel (stage >= 2), else the
 * forward KLT result. */
int vo_mono_frame_result(vo_ctx *ctx, float *pts1, float *scale, uint8_t *stage, float dT01[16],
                         vo_mono_counts *counts, vo_gn_info *gn);
/* The mono frame with its new-point step closed on the device (MonoVO::trackImage, mono_vo.cpp:977-1001:
 * extractor_->updateWeightBin(lmtrack_final.pts1), extractORBwithBinning_fast(I1, ...), trackBidirection(I1, I0, ...)):
 * `table` holds the best keypoint of every bin of the image in slot1 (vo_new_point_candidates_enqueue, from the image
 * alone, any time before), the frame kernel back-tracks every bin's candidate, and the BA launch's epilogue emits the
 * candidates of the bins that lmtrack_final (stage 4) leaves empty — the reference's result in the reference's order.
 * When the BA gives no pose (counts.need_five_point) no new points are reported: that path is the caller's. */
int vo_mono_frame_enqueue_closed(vo_ctx *ctx, const vo_mono_params *prm, int slot0, int slot1,
                                 const float *pts0, const float *Xw, const uint8_t *flags, int n,
                                 const float Tcw_prev[16], const float Tcw_prior[16],
                                 const float dT01_prior[16], const vo_bin_params *bins, int table,
                                 int inputs_on_device);
/* After vo_mono_frame_result of a closed frame: pts1_new = the new points' pixels in I1 (bins ascending), pts0_new =
 * their back-tracked pixels in I0, mask_new = trackBidirection's mask. Capacity: n_bins_u * n_bins_v each. */
int vo_mono_frame_new_points(vo_ctx *ctx, float *pts1_new, float *pts0_new, uint8_t *mask_new, int *n_new);

/* ---- MonoVO: the closed loop around the mono frame -------------------------------------------------------------
 * MonoVO::trackImage (core/visual_odometry/mono_vo/mono_vo.cpp:496-1194, class surface mono_vo.h:235-243, :267) with the
 * track set carried ON THE DEVICE: per landmark the pixel seen, id, world point, flags (VO_LM_TRIANGULATED | VO_LM_DROPPED |
 * VO_LM_KF_MEMBER | VO_LM_BUNDLED), its first observation and frame, age, the parallax of its newest observation
 * (landmark.cpp:76-135), its first keyframe observation and their number. The first image creates one landmark per bucketed
 * keypoint (:528-561); the second is the initialisation (:562-696); then the steady state (:698-1019: prior and patch scale
 * from bundled landmarks, trackBidirectionWithPrior, trackWithScale, pose-only BA on the bundled / triangulated
 * landmarks, Sampson gate, new points back-tracked into the previous image); at the end of every call the keyframe rule
 * (keyframes.cpp:47-126), and at a keyframe addNewKeyframe (:30-45), the reconstruction of landmarks seen on more than
 * two keyframes (:1032-1076) and the mono local BA (motion_estimator.cpp:1090-1205, sparse_ba_parameters.h:292-466 with
 * one observation per keyframe and at least two window keyframes per landmark; setBundled / setDead on the way back).
 * The 5-point / essential-matrix pose (MotionEstimator::calcPose5PointsAlgorithm, motion_estimator.cpp:21-203: OpenCV
 * calib3d, out of scope per SURVEY §2) is a CALLER HOOK: called at the initialisation and whenever the pose-only BA
 * yields no pose (:909-949). pts0 / pts1: n pixel pairs (previous, current image); K = fx, fy, cx, cy; outputs R10
 * (row-major 3x3), t10 (any length; the driver rescales it as the reference does) and mask[n] (inliers). Return non-zero
 * on success; 0 ends the call with VO_ERR_GN_FAILED (the reference throws).
 * Needs a context with >= 3 image slots and prm.frame.win in {13, 15, 21, 31}. */
#define VO_LM_BUNDLED 8
typedef int (*vo_five_point_fn)(void *user, const float *pts0, const float *pts1, int n, const float K[4], float R10[9],
                                float t10[3], uint8_t *mask);
typedef struct vo_mvo vo_mvo;
typedef struct {
  vo_mono_params frame;     /* Camera.*, feature_tracker.* (incl. thres_sampson), motion_estimator.thres_poseba_error */
  vo_bin_params bins;       /* feature_extractor.* as FeatureExtractor::initParams derives them */
  float kf_overlap_ratio;   /* keyframe_update.thres_overlap_ratio */
  float kf_rotation_deg;    /* keyframe_update.thres_rotation (degrees) */
  float kf_translation;     /* keyframe_update.thres_translation */
  int kf_window;            /* keyframe_update.n_max_keyframes_in_window (at most 16) */
  float thres_parallax_deg; /* map_update.thres_parallax (degrees; the reference multiplies by D2R) */
  int strict_border;        /* trackWithScale's never-reset tap state: 0 masked taps; 1..4 reference-exact — 1 the replay stream-ordered
                               behind the frame kernel, 2 sequential replay only (validation), 3 the replay next to the frame
                               kernel joined on the device, 4 = 3 or 1 per frame (vo_stereo_frame_set_strict_border); same results */
  int local_ba;             /* != 0: localBundleAdjustmentSparseSolver at every keyframe */
  int rectify;              /* != 0: flagDoUndistortion (mono_vo.cpp:509-513) — images go through camera 0's undistortion
                               map (vo_rectify_init_mono first) on their way into the pyramid */
  vo_five_point_fn five_point;
  void *five_point_user;
} vo_mvo_params;
typedef struct {
  int frame_id;             /* Frame id of this image (the context's frame counter) */
  int is_first, is_init, is_keyframe, lba_ran, used_five_point;
  int n_tracks_in, n_final, n_new, n_tracks_out, n_kf_tracked, n_reconstructed;
  vo_mono_counts counts;
  vo_gn_info gn;
  float dT01[16];           /* motion previous -> current camera as the frame used it */
  float T_wc[16];           /* pose after this call (after the local BA at a keyframe) */
  double lba_err_first, lba_err_last;
  int lba_landmarks, lba_observations;
} vo_mvo_frame_info;
int vo_mvo_create(vo_ctx *ctx, const vo_mvo_params *prm, vo_mvo **out);
void vo_mvo_destroy(vo_mvo *mvo);
/* MonoVO::trackImage(img, timestamp): synchronous. Image: u8, `stride` bytes per row, device pointer when on_device != 0. */
int vo_mvo_track(vo_mvo *mvo, const void *img, int stride, int on_device, double timestamp, vo_mvo_frame_info *info);
/* A recorded sequence through vo_mvo_result(k) / vo_mvo_enqueue(k + 1) / vo_mvo_prefetch(k + 2), the loop on this side of the ABI
 * (see vo_svo_run). The 5-point hook is called from inside as usual. */
int vo_mvo_run(vo_mvo *mvo, const void *const *img, int n_total, int stride, int on_device, int k_begin, int k_end,
               vo_mvo_frame_info *infos, double *stamps);
/* the same in halves, and the next image handed over early (pyramid + per-bin candidate table on the side stream). A host image is
 * uploaded asynchronously: its buffer must stay untouched until that image's frame has returned (as for vo_svo_prefetch). A refused
 * vo_mvo_enqueue leaves the driver as it was (the image can be handed over again); an error return of vo_mvo_result ENDS the
 * stream, as the reference's throw ends its node (mono_vo.cpp:909-949): the frame is no longer in flight and the track set / pose
 * are where the failing step left them. */
int vo_mvo_enqueue(vo_mvo *mvo, const void *img, int stride, int on_device, double timestamp);
int vo_mvo_prefetch(vo_mvo *mvo, const void *img, int stride, int on_device);
int vo_mvo_result(vo_mvo *mvo, vo_mvo_frame_info *info);
/* frame_prev_'s related landmarks (test / inspection hook): any pointer may be NULL. cos_parallax: cosine of
 * Landmark::getLastParallax() (2 = no second observation yet) */
int vo_mvo_get_tracks(vo_mvo *mvo, int32_t *ids, float *pts, float *Xw, uint8_t *flags, int32_t *age, float *cos_parallax, int cap,
                      int *n);
/* stats_keyframe (mono_vo.cpp:1130-1155), as vo_svo_keyframe_count / vo_svo_get_keyframes */
int vo_mvo_keyframe_count(vo_mvo *mvo, int *n_keyframes);
int vo_mvo_get_keyframes(vo_mvo *mvo, float *T_wc, int32_t *n_points, float *mappoints, size_t cap_points, size_t *total_points);

/* ---- sparse local bundle adjustment -----------------------------------------
 * SparseBundleAdjustmentSolver::solveForFiniteIterations
 * (core/visual_odometry/ba_solver/sparse_bundle_adjustment.cpp:150-643; called from
 * MotionEstimator::localBundleAdjustmentSparseSolver[_Stereo], motion_estimator.cpp:1090-1340, with
 * MAX_ITER 10 and THRES_HUBER 0.5). The problem is what SparseBAParameters holds after
 * setPosesAndPoints (sparse_ba_parameters.h:283-440), as flat arrays, in double:
 *   T_jw      n_frames x 16   poses (row-major 4x4) of the frames that carry one - mono keyframes or the LEFT
 *                             frames of stereo keyframes - in the window's reference frame, translations scaled
 *   opt_index n_frames        index in [0, n_opt) of the pose in the optimisation, or -1 when it is fixed
 *   X         n_points x 3    landmarks (same frame and scale)
 *   obs_ptr   n_points + 1    observations of landmark i are obs_ptr[i] .. obs_ptr[i+1]-1, in the order of
 *                             LandmarkBA::kfs_seen
 *   obs_frame / obs_right / obs_px   frame index (the left frame for a right-image observation), seen in the
 *                             right image, pixel
 * T_jw (optimised poses) and X are updated in place. avg_err (max_iter doubles, may be NULL) receives the
 * average pixel error the reference prints per iteration. Returns 1 = flag_success, 0 = average error of the
 * last iteration above 1 px, VO_ERR_LBA_NAN where the reference throws. n_opt <= 20. */
typedef struct {
  int n_frames, n_opt, n_points, n_obs;
  int stereo;
  int max_iter;
  double Kl[4], Kr[4];
  double T_lr[16]; /* stereo pose left -> right as SparseBAParameters::getStereoPose gives it (scaled) */
  double thres_huber;
} vo_sba_problem;
int vo_sba_solve(vo_ctx *ctx, const vo_sba_problem *prm, double *T_jw, const int32_t *opt_index, double *X,
                 const int32_t *obs_ptr, const int32_t *obs_frame, const uint8_t *obs_right,
                 const double *obs_px, double *avg_err);

/* ---- kernel timing (HIP events on the context stream) --------------------- */
enum { VO_K_PYRAMID = 0, VO_K_KLT = 1, VO_K_IC = 2, VO_K_GN = 3, VO_K_HAMMING = 4, VO_K_AUX = 5, VO_K_COUNT = 6 };
int vo_profile_enable(vo_ctx *ctx, int max_records);
int vo_profile_reset(vo_ctx *ctx);
/* restrict the event brackets to a set of kernel classes: mask = OR of (1 << VO_K_*); 0 = all */
int vo_profile_set_classes(vo_ctx *ctx, unsigned mask);
/* after vo_synchronize(): launches and summed device time (ms) of one kernel class */
int vo_profile_get(vo_ctx *ctx, int kernel_class, int *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H_ */
