/*
 * vo_oracle.h — CPU parity oracle for the per-frame VO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker (never as the thing measured or shipped).
 *
 * PARITY UNPINNED: the reference (ChanghyeonKim93/visual_odometry_ros) holds
 * no golden vector, known-answer test or fixture for this path (its test/
 * directory is four print-only programs), and it cannot be compiled in the
 * build container (Eigen3 and OpenCV 4 are absent).  This file is a plain-C
 * restatement that follows the reference text function by function; every
 * function cites the reference file:line it restates.  The KLT core
 * (cv::calcOpticalFlowPyrLK) is a third-party dependency that is NOT in the
 * reference tree (OpenCV 4.x, modules/video/src/lkpyramid.cpp, pinned only as
 * "find_package(OpenCV 4 REQUIRED)" in core/CMakeLists.txt:12); its published
 * algorithm is restated here and anchored on the reference's call sites
 * (core/visual_odometry/feature_tracker.cpp:29,60,69,108,117,186).
 *
 * All matrices crossing this interface are ROW-MAJOR float[16]/float[9].
 */
#ifndef VO_ORACLE_H_
#define VO_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Summation order for the genuinely-floating-point reductions (GN, IC).
 *  SEQ : the reference's order (one accumulator, index ascending).
 *  TREE: the order the wavefront kernels use — T strided serial partials
 *        (partial t takes elements t, t+T, ...) combined by a balanced binary
 *        tree over the partials in natural order (adjacent pairs first). */
#define VO_SUM_SEQ 0
#define VO_SUM_TREE 1

/* Mono GN variant: core adds the weighted w*ry^2 for the y row
 * when the Huber weight is active; standalone adds the unweighted ry^2. */
#define VO_GN_VARIANT_CORE 0
#define VO_GN_VARIANT_STANDALONE 1

/* IC border semantics.
 *  REFERENCE: reproduce the reference exactly, including the tap mask / value
 *             vectors that are never reset between points and iterations.
 *  MASKED   : the evident intent — a tap outside the image is excluded from
 *             that evaluation's sums. Identical to REFERENCE for every point
 *             whose taps all stay inside the image in every evaluation. */
#define VO_IC_BORDER_REFERENCE 0
#define VO_IC_BORDER_MASKED 1

/* OpenCV flag (cv::OPTFLOW_USE_INITIAL_FLOW == 4). */
#define VO_KLT_USE_INITIAL_FLOW 4

typedef struct {
  int iterations;   /* GN iterations executed (reference `iter`+1 at break) */
  float err;        /* last err_curr */
  float delta_err;  /* last |err_curr - err_prev| */
  float delta_norm; /* last ||delta_xi|| */
  int cnt_invalid;  /* outliers counted in the last iteration */
  int is_nan;       /* 1 if the pose went NaN (reference returns false) */
} vo_ref_gn_info;

/* ---- geometry helpers -------------------------------------------------- */
void vo_ref_se3_exp(const float xi[6], float T[16]);
void vo_ref_inverse_se3(const float T[16], float Tinv[16]);
void vo_ref_inverse4x4(const float T[16], float Tinv[16]);
int vo_ref_ldlt6_solve(const float A[36], const float b[6], float x[6]);

/* ---- Gauss-Newton pose-only BA ----------------------------------------- */
int vo_ref_gn_pose_mono(const float *X, const float *pts1, int n,
                        const float K[4], int thres_reproj_outlier,
                        float R01[9], float t01[3], uint8_t *mask_inlier,
                        int variant, int sum_mode, int tree_width,
                        vo_ref_gn_info *info);
int vo_ref_gn_pose_stereo(const float *X, const float *pts_l1,
                          const float *pts_r1, int n, const float Kl[4],
                          const float Kr[4], const float T_lr[16],
                          float thres_reproj_outlier, float T01[16],
                          uint8_t *mask_inlier, int sum_mode, int tree_width,
                          vo_ref_gn_info *info);

/* ---- image pyramid (OpenCV buildOpticalFlowPyramid semantics) ---------- */
int vo_ref_pyramid_levels(int w, int h, int win, int max_level);
void vo_ref_level_size(int w, int h, int level, int *lw, int *lh);
void vo_ref_pyr_down(const uint8_t *src, int sw, int sh, int sstride,
                     uint8_t *dst, int dstride);
void vo_ref_scharr(const uint8_t *src, int w, int h, int sstride,
                   int16_t *dxy /* w*h*2 interleaved */);
void vo_ref_sobel3(const uint8_t *src, int w, int h, int sstride, float *du,
                   float *dv);

/* ---- pyramidal LK (cv::calcOpticalFlowPyrLK semantics) ------------------ */
int vo_ref_calc_optical_flow_pyr_lk(const uint8_t *img0, const uint8_t *img1,
                                    int w, int h, int stride, const float *pts0,
                                    float *pts1, int n, int win, int max_level,
                                    int flags, int max_iter, double eps,
                                    float min_eig_thr, uint8_t *status,
                                    float *err, int n_threads);

/* ---- FeatureTracker front-ends (masks) ---------------------------------- */
int vo_ref_track(const uint8_t *img0, const uint8_t *img1, int w, int h,
                 int stride, const float *pts0, int n, int win, int max_level,
                 float thres_err, float *pts_track, uint8_t *mask,
                 int n_threads);
int vo_ref_track_bidirection(const uint8_t *img0, const uint8_t *img1, int w,
                             int h, int stride, const float *pts0, int n,
                             int win, int max_level, float thres_err,
                             float thres_bidirection, float *pts_track,
                             uint8_t *mask, int n_threads);
int vo_ref_track_bidirection_with_prior(const uint8_t *img0,
                                        const uint8_t *img1, int w, int h,
                                        int stride, const float *pts0, int n,
                                        int win, int max_level, float thres_err,
                                        float thres_bidirection,
                                        float *pts_track, uint8_t *mask,
                                        int n_threads);
int vo_ref_track_with_prior(const uint8_t *img0, const uint8_t *img1, int w,
                            int h, int stride, const float *pts0, int n,
                            int win, int max_level, float thres_err,
                            float *pts_track, uint8_t *mask, int n_threads);
void vo_ref_calc_prior(const float *pts0, int n_pts0, const float *Xw, int n,
                       const float Tw1[16], const float K[9],
                       float *pts1_prior);

/* ---- scale-compensated inverse-compositional refinement ---------------- */
int vo_ref_track_with_scale(const uint8_t *img0, const uint8_t *img1, int w,
                            int h, int stride, const float *pts0,
                            const float *scale_est, int n, float *pts_track,
                            uint8_t *mask, int border_mode, int sum_mode,
                            uint8_t *touched_border /* optional, n */);

/* ---- ORB descriptor distance ------------------------------------------- */
int vo_ref_descriptor_distance(const uint8_t *a, const uint8_t *b);
void vo_ref_hamming_matrix(const uint8_t *a, int na, const uint8_t *b, int nb,
                           uint16_t *dist);
void vo_ref_hamming_match(const uint8_t *a, int na, const uint8_t *b, int nb,
                          int th_low, float ratio, int32_t *best_idx,
                          uint16_t *best_dist, uint16_t *second_dist);

/* ---- landmark mask compaction / track ids ------------------------------ */
int vo_ref_compact_indices(const uint8_t *mask, const uint8_t *alive,
                           const uint8_t *tracked, int n, int32_t *index_valid,
                           uint8_t *tracked_out);

/* ---- steady-state stereo frame (operator sequence of
 *      core/visual_odometry/stereo_vo/stereo_vo.cpp:465-740) ------------- */
typedef struct {
  int width, height;
  int win, max_level;
  float thres_err, thres_bidirection, thres_poseba;
  float Kl[4], Kr[4];
  float T_lr[16];
  float thres_sampson; /* feature_tracker.thres_sampson: the y > 660 gate of [7] (stereo_vo.cpp:653-668) */
} vo_ref_stereo_params;

typedef struct {
  int n_l0l1, n_refine, n_l1r1, n_inlier, n_new_ok;
  int gn_iterations;
  int n_ba; /* size of the pose-only BA set (triangulated survivors of [5], stereo_vo.cpp:599) */
} vo_ref_frame_counts;

int vo_ref_stereo_frame(const vo_ref_stereo_params *prm, const uint8_t *I0l,
                        const uint8_t *I1l, const uint8_t *I1r, int stride,
                        const float *pts_l0, const float *Xp /* prev-cam */,
                        const uint8_t *lm_flags /* bit 0 = isTriangulated(); NULL = all */,
                        int n, const float dT_prior[16] /* T_pc prior */,
                        const float *pts_new, int n_new, int sum_mode,
                        int tree_width, int ic_border_mode, int n_threads,
                        /* outputs, all sized n (or n_new) */
                        float *pts_l1, float *pts_r1, uint8_t *stage_mask,
                        float dT_out[16], float *pts_new_r, uint8_t *mask_new,
                        vo_ref_frame_counts *counts);

/* The same frame on the reference's own data flow (stereo_vo.cpp:475-522, :595-613): landmarks in the WORLD frame,
 * T_pw = pose inverse of the previous left frame, T_cw_prior = inverseSE3_f(T_wp * dT_pc_prev); dT_prior = dT_pc_prev. */
int vo_ref_stereo_frame_world(const vo_ref_stereo_params *prm, const uint8_t *I0l, const uint8_t *I1l, const uint8_t *I1r,
                              int stride, const float *pts_l0, const float *Xw, const uint8_t *lm_flags, int n,
                              const float dT_prior[16], const float T_pw[16], const float T_cw_prior[16],
                              const float *pts_new, int n_new, int sum_mode, int tree_width, int ic_border_mode,
                              int n_threads, float *pts_l1, float *pts_r1, uint8_t *stage_mask, float dT_out[16],
                              float *pts_new_r, uint8_t *mask_new, vo_ref_frame_counts *counts);

/* ---- the loop around the frame (oracle_vo.c) ---- */
void vo_ref_mul44(const float A[16], const float B[16], float C[16]);
void vo_ref_xform_eig(const float T[16], const float X[3], float Y[3]);
int vo_ref_jacobi_svd4(const float M[16], float V[16], float sv[4]);
void vo_ref_dlt_projection(const float K1[4], const float R10[9], const float t10[3], float P10[12]);
int vo_ref_triangulate_dlt(const float pt0[2], const float pt1[2], const float R10[9], const float t10[3],
                           const float K0[4], const float K1[4], float X0[3], float X1[3]);
int vo_ref_new_landmark_accept(const float *pts_l, const float *pts_r, const uint8_t *mask_new, int n,
                               const float T_rl[16], const float Kl[4], const float Kr[4], uint8_t *accept, float *Xl_out);
int vo_ref_keyframe_reconstruct(const float *pts_l, const float *pts_r, int n, const float T_rl[16], const float Kl[4],
                                const float Kr[4], const float *T_wc, float *Xw, uint8_t *set);
/* MonoVO: landmark.cpp:100-121 (parallax of the newest observation w.r.t. the oldest), mono_vo.cpp:669-686 / :1041-1073
 * (reconstruction of a landmark from its first and last observation) */
float vo_ref_parallax(const float p0[2], const float p1[2], const float K[4], const float T_cw_first[16],
                      const float T_wc_last[16], float *cos_out);
int vo_ref_mono_reconstruct(const float pt0[2], const float pt1[2], const float T_w0[16], const float T_1w[16],
                            const float K[4], int keyframe_rule, float Xw[3]);

/* per-stage wall clock (ms) of the last vo_ref_stereo_frame call: priors, KLT l0->l1, trackWithScale, KLT l1->r1, BA,
 * gates + compactions, new-point tracking */
void vo_ref_stereo_frame_stage_ms(double out[8]);

/* epipolar gates: motion_estimator.cpp:572-599 (Sampson), :621-653 (symmetric epipolar), :551-552 (F10) */
void vo_ref_sampson_distance(const float *pts0, const float *pts1, int n, const float F10[9], float *dist);
void vo_ref_symmetric_epipolar_distance(const float *pts0, const float *pts1, int n, const float F10[9],
                                        float *dist);
void vo_ref_fundamental_from_pose(const float K[4], const float R10[9], const float t10[3], float F10[9]);

/* feature bucketing: feature_extractor.h:90-135 (WeightBin), feature_extractor.cpp:241-277 (arg-max per bin) */
void vo_ref_weight_bin_init(int n_cols, int n_rows, int n_bins_u, int n_bins_v, int *u_step, int *v_step,
                            float *inv_u_step, float *inv_v_step);
void vo_ref_weight_bin_update(const float *pts, int n, int u_step, int v_step, int n_bins_u, int n_bins_v,
                              int32_t *weight);
int vo_ref_bucket_argmax(const float *kp_xy, const float *kp_response, int n, float inv_u_step, float inv_v_step,
                         int n_bins_u, int n_bins_v, const int32_t *weight, float *pts_out, int32_t *idx_out);

/* ---- steady-state mono frame (oracle_mono.c; mono_vo.cpp:739-963) ---- */
typedef struct {
  int width, height, win, max_level;
  float thres_err, thres_bidirection;
  int thres_poseba; /* the reference's parameter is int-typed (motion_estimator.h:117) */
  float thres_sampson;
  float K[4];
} vo_ref_mono_params;
typedef struct {
  int n_klt, n_refine, n_ba, n_motion, n_final, gn_iterations, need_five_point;
} vo_ref_mono_counts;
int vo_ref_mono_frame(const vo_ref_mono_params *prm, const uint8_t *I0, const uint8_t *I1, int stride,
                      const float *pts0, const float *Xw, const uint8_t *flags, int n, const float Tcw_prev[16],
                      const float Tcw_prior[16], const float dT01_prior[16], int sum_mode, int tree_width,
                      int ic_border_mode, int n_threads, float *pts1, float *scale, uint8_t *stage,
                      float dT01_out[16], vo_ref_mono_counts *counts);

/* ---- image ingestion with flagDoUndistortion (oracle_rectify.c) ---- */
/* Camera::generateImageUndistortMaps, camera.cpp:56-90. K = fx,fy,cx,cy ; D = k1,k2,p1,p2,k3 */
void vo_ref_image_undistort_maps(int n_cols, int n_rows, const float K[4], const float D[5], float *map_u,
                                 float *map_v);
/* pixel-independent part of StereoCamera::generateStereoImagesUndistortAndRectifyMaps (camera.cpp:364-432,
 * :530-535): M = R_0n * K_rect^-1, R_l0, R_r0 (row-major 3x3), K_rect = f,f,cu,cv, T_lr_rect row-major 4x4 */
void vo_ref_stereo_rectify_setup(int n_cols, int n_rows, const float Kl[4], const float Kr[4], const float T_lr[16],
                                 float M[9], float R_l0[9], float R_r0[9], float K_rect[4], float T_lr_rect[16]);
/* the whole of camera.cpp:364-546 */
void vo_ref_stereo_rectify_maps(int n_cols, int n_rows, const float Kl[4], const float Dl[5], const float Kr[4],
                                const float Dr[5], const float T_lr[16], float *map_lu, float *map_lv, float *map_ru,
                                float *map_rv, float K_rect[4], float T_lr_rect[16]);
/* convertTo(CV_32FC1) -> cv::remap(float maps, INTER_LINEAR, BORDER_CONSTANT 0) -> convertTo(CV_8UC1)
 * (camera.cpp:166-183, :300-336 ; stereo_vo.cpp:420-421 ; mono_vo.cpp:512). dst is dw x dh, tightly packed. */
void vo_ref_remap_linear_u8(const uint8_t *src, int w, int h, int stride, const float *map_u, const float *map_v,
                            int dw, int dh, uint8_t *dst);

/* ---- sparse local bundle adjustment (oracle_sba.c) ---- */
typedef struct {
  int n_frames; /* frames that carry a pose (mono frames / left frames of stereo keyframes) */
  int n_opt;    /* optimised poses; opt_index[f] in [0, n_opt) or -1 */
  int n_points, n_obs;
  int stereo;
  int max_iter;
  double Kl[4], Kr[4];
  double T_lr[16]; /* stereo pose left -> right as the solver sees it (scaled), row-major */
  double thres_huber;
} vo_ref_sba_dims;
/* Eigen::LDLT restated in double: m n x n row-major (destroyed), B n x nrhs row-major in/out */
int vo_ref_ldlt_solve_f64(int n, double *m, int nrhs, double *B);
void vo_ref_se3_exp_f64(const double xi[6], double T[16]);
void vo_ref_se3_log_f64(const double T[16], double xi[6]);
void vo_ref_sba_pose_update(double T[16], const double x[6]);
void vo_ref_sba_linearize(const vo_ref_sba_dims *d, const double T_jw[16], const double X[3], const double px[2],
                          int right, double r[2], double *w, double R[6], double Q[12]);
/* SparseBundleAdjustmentSolver::solveForFiniteIterations (sparse_bundle_adjustment.cpp:150-643).
 * T_jw: n_frames x 16 in/out; X: n_points x 3 in/out; observations of landmark i are
 * obs_ptr[i] .. obs_ptr[i+1]-1 in the order of lmba.kfs_seen: obs_frame = index of the (left) frame,
 * obs_right = seen in that keyframe's right image, obs_px = pixel. avg_err[max_iter] (may be NULL).
 * returns 1 = flag_success, 0 = average error above 1 px, -1 = NaN (the reference throws) */
int vo_ref_sba_solve(const vo_ref_sba_dims *d, double *T_jw, const int *opt_index, double *X, const int *obs_ptr,
                     const int *obs_frame, const uint8_t *obs_right, const double *obs_px, double *avg_err);

/* ---- cv::ORB::detect as FeatureExtractor configures it (oracle_orb.c; OpenCV restated, see its header) ---- */
void vo_ref_resize_linear_exact_u8(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dw, int dh);
void vo_ref_fast_score_image(const uint8_t *img, int w, int h, int stride, int threshold, uint8_t *score);
void vo_ref_orb_level_sizes(int w, int h, double scale_factor, int n_levels, int nfeatures, int *lw, int *lh,
                            float *lscale, int *nper);
int vo_ref_orb_detect(const uint8_t *img, int w, int h, int stride, int nfeatures, double scale_factor, int n_levels,
                      int edge_threshold, int fast_threshold, float *kp_xy, float *kp_response, int32_t *kp_octave,
                      int max_kp, uint8_t *levels_out);

#ifdef __cplusplus
}
#endif
#endif /* VO_ORACLE_H_ */
