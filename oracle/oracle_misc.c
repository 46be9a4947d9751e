/*
 * oracle_misc.c — ORB descriptor distance, landmark mask compaction, and the
 * steady-state stereo frame operator sequence.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h). PARITY UNPINNED.
 * Follows:
 *   core/visual_odometry/feature_extractor.cpp:338-357   (descriptorDistance)
 *   test/test_orbmatching.cpp:87-137                     (NN + ratio matcher skeleton)
 *   core/visual_odometry/landmark.cpp:291-332, :194-231  (mask compaction ctors)
 *   core/visual_odometry/stereo_vo/stereo_vo.cpp:465-740 (steady-state frame)
 *   core/visual_odometry/camera.cpp:208-229              (projectToPixel, inImage)
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* feature_extractor.cpp:338-357: 8 x (xor, SWAR popcount) over 32 bytes */
int vo_ref_descriptor_distance(const uint8_t *a, const uint8_t *b) {
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

void vo_ref_hamming_matrix(const uint8_t *a, int na, const uint8_t *b, int nb,
                           uint16_t *dist) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      dist[(size_t)i * nb + j] = (uint16_t)vo_ref_descriptor_distance(a + 32 * i, b + 32 * j);
}

/* test/test_orbmatching.cpp:87-137 policy: for every query the nearest and
 * second-nearest train descriptor (first index wins ties), accepted iff
 * best <= th_low and best < ratio * second. best_idx = -1 when rejected. */
void vo_ref_hamming_match(const uint8_t *a, int na, const uint8_t *b, int nb,
                          int th_low, float ratio, int32_t *best_idx,
                          uint16_t *best_dist, uint16_t *second_dist) {
  for (int i = 0; i < na; ++i) {
    int bd = 256, bd2 = 256, bi = -1;
    for (int j = 0; j < nb; ++j) {
      int d = vo_ref_descriptor_distance(a + 32 * i, b + 32 * j);
      if (d < bd) {
        bd2 = bd;
        bd = d;
        bi = j;
      } else if (d < bd2) {
        bd2 = d;
      }
    }
    best_dist[i] = (uint16_t)bd;
    second_dist[i] = (uint16_t)bd2;
    if (bi >= 0 && bd <= th_low && (float)bd < ratio * (float)bd2)
      best_idx[i] = bi;
    else
      best_idx[i] = -1;
  }
}

/* landmark.cpp:291-332: stable compaction by mask && alive && tracked; every
 * rejected landmark is setUntracked(). Returns the survivor count. */
int vo_ref_compact_indices(const uint8_t *mask, const uint8_t *alive,
                           const uint8_t *tracked, int n, int32_t *index_valid,
                           uint8_t *tracked_out) {
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    int al = alive ? alive[i] : 1, tr = tracked ? tracked[i] : 1;
    if (mask[i] && al && tr) {
      index_valid[cnt++] = i;
      if (tracked_out) tracked_out[i] = 1;
    } else if (tracked_out)
      tracked_out[i] = 0;
  }
  return cnt;
}

/* camera.cpp:208-213 */
static void project(const float K[4], const float X[3], float *px, float *py) {
  const float invz = 1.0f / X[2];
  *px = K[0] * X[0] * invz + K[2];
  *py = K[1] * X[1] * invz + K[3];
}
/* camera.cpp:220-229 */
static int in_image(float x, float y, int n_cols, int n_rows) {
  const float offset = 3.0f;
  return !(x < offset || y < offset || x >= n_cols - offset || y >= n_rows - offset);
}
static void xform(const float T[16], const float X[3], float Y[3]) {
  for (int r = 0; r < 3; ++r)
    Y[r] = ((T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1]) + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
}

/*
 * Steady-state stereo frame, expressed in the previous left-camera frame
 * (the reference carries world-frame landmarks and T_wp; with Xp = T_pw X and
 * dT = T_pw T_wc the operator sequence is the same):
 *   [3] prior pixels + patch scale            stereo_vo.cpp:483-522
 *   [4] trackWithPrior  I0_L -> I1_L          :531-538
 *   [4-1] Sobel + trackWithScale              :549-558
 *   [5] trackWithPrior  I1_L -> I1_R          :564-571
 *   [6] poseOnlyBundleAdjustment_Stereo       :595-646
 *   [7] the y>660 gate                        :653-670 (thres_sampson = 60)
 *   [10] trackBidirection for new points      :706-711
 * stage_mask[i] = number of gates point i passed (0..4; 4 = survived all).
 * pts_r0 is taken as pts_l0 when a prior falls outside the image only through
 * the caller: here the prior for such points is (pts_l0, pts_l0 - disparity)
 * — the caller passes pts_r0 through pts_r1 (in/out).
 */
/* wall-clock split of the last vo_ref_stereo_frame call (bench.py's cpu_baseline reports it per stage):
 * [0] priors  [1] trackWithPrior l0->l1 (builds both pyramids, as every cv::calcOpticalFlowPyrLK call does)
 * [2] trackWithScale (incl. the convertTo / Sobel images)  [3] trackWithPrior l1->r1  [4] pose-only BA
 * [5] gates + compactions  [6] trackBidirection of the new points */
#include <omp.h>
static double g_stage_ms[8];
void vo_ref_stereo_frame_stage_ms(double out[8]) { memcpy(out, g_stage_ms, sizeof(g_stage_ms)); }
#define STAGE_T(k)                                 \
  do {                                             \
    const double now_ = omp_get_wtime();           \
    g_stage_ms[k] += 1e3 * (now_ - t_stage);       \
    t_stage = now_;                                \
  } while (0)

/* world != 0: the reference's own data flow (stereo_vo.cpp:475-522, :595-613) — Xp holds the landmarks in the WORLD
 * frame, T_pw = stframe_prev->getLeft()->getPoseInv(), T_cw_prior = inverseSE3_f(T_wp * dT_pc_prev); X_l1 = T_cw_prior X,
 * X_l0 = T_pw X (patch scale and the BA's Xp), with Eigen's evaluation order of `R * X + t` (vo_ref_xform_eig). */
static int stereo_frame_impl(const vo_ref_stereo_params *prm, const uint8_t *I0l,
                        const uint8_t *I1l, const uint8_t *I1r, int stride,
                        const float *pts_l0, const float *Xp, const uint8_t *lm_flags, int n,
                        const float dT_prior[16], const float *T_pw, const float *T_cw_prior, const float *pts_new,
                        int n_new, int sum_mode, int tree_width,
                        int ic_border_mode, int n_threads, float *pts_l1,
                        float *pts_r1, uint8_t *stage_mask, float dT_out[16],
                        float *pts_new_r, uint8_t *mask_new,
                        vo_ref_frame_counts *counts) {
  const int W = prm->width, H = prm->height;
  const int world = T_pw != NULL;
  float T_rl[16], T_cp[16];
  vo_ref_inverse_se3(prm->T_lr, T_rl);
  vo_ref_inverse_se3(dT_prior, T_cp);
  float *scale = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
  uint8_t *m = (uint8_t *)malloc((size_t)n + 1);
  float *a0 = (float *)malloc(sizeof(float) * 2 * ((size_t)n + 1));
  float *a1 = (float *)malloc(sizeof(float) * 2 * ((size_t)n + 1));
  float *a2 = (float *)malloc(sizeof(float) * 2 * ((size_t)n + 1));
  float *aX = (float *)malloc(sizeof(float) * 3 * ((size_t)n + 1));
  float *as = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  memset(counts, 0, sizeof(*counts));
  memset(g_stage_ms, 0, sizeof(g_stage_ms));
  double t_stage = omp_get_wtime();
  /* [3] priors; pts_r1 holds pts_r0 on entry. lm_flags[i] bit 0 = lm->isTriangulated() (stereo_vo.cpp:490);
   * NULL = every landmark is triangulated */
  for (int i = 0; i < n; ++i) {
    stage_mask[i] = 0;
    idx[i] = i;
    if (lm_flags && !(lm_flags[i] & 1)) { /* :515-519: prior = previous pixels, scale_tmp stays 1 */
      scale[i] = 1.0f;
      pts_l1[2 * i] = pts_l0[2 * i];
      pts_l1[2 * i + 1] = pts_l0[2 * i + 1];
      continue;
    }
    float Xl1[3], Xr1[3];
    if (world) {
      float Xl0[3];
      vo_ref_xform_eig(T_cw_prior, Xp + 3 * i, Xl1); /* :493 */
      vo_ref_xform_eig(T_rl, Xl1, Xr1);              /* :494 */
      vo_ref_xform_eig(T_pw, Xp + 3 * i, Xl0);       /* :497 */
      scale[i] = Xl0[2] / Xl1[2];
    } else {
      xform(T_cp, Xp + 3 * i, Xl1);
      xform(T_rl, Xl1, Xr1);
      scale[i] = Xp[3 * i + 2] / Xl1[2];
    }
    float plx, ply, prx, pry;
    project(prm->Kl, Xl1, &plx, &ply);
    project(prm->Kr, Xr1, &prx, &pry);
    if (!in_image(plx, ply, W, H) || !in_image(prx, pry, W, H) || Xl1[2] < 0.1 || Xr1[2] < 0.1) {
      pts_l1[2 * i] = pts_l0[2 * i];
      pts_l1[2 * i + 1] = pts_l0[2 * i + 1];
      /* pts_r1 keeps pts_r0 */
    } else {
      pts_l1[2 * i] = plx;
      pts_l1[2 * i + 1] = ply;
      pts_r1[2 * i] = prx;
      pts_r1[2 * i + 1] = pry;
    }
  }
  int cur = n;
  STAGE_T(0);
  /* [4] l0 -> l1 */
  for (int i = 0; i < n; ++i) m[i] = 1;
  vo_ref_track_with_prior(I0l, I1l, W, H, stride, pts_l0, n, prm->win, prm->max_level,
                          prm->thres_err, pts_l1, m, n_threads);
  STAGE_T(1);
  /* StereoLandmarkTracking(lmtrack_prev, mask_l0l1), landmark.cpp:305: mask && isAlive() && isTracked();
   * lm_flags bit 1 = the landmark is no longer alive / tracked */
  if (lm_flags)
    for (int i = 0; i < n; ++i)
      if (lm_flags[i] & 2) m[i] = 0;
  int c = 0;
  for (int i = 0; i < cur; ++i)
    if (m[i]) {
      stage_mask[idx[i]] = 1;
      idx[c] = idx[i];
      ++c;
    }
  cur = c;
  counts->n_l0l1 = cur;
  /* [4-1] refine on the compacted set */
  for (int i = 0; i < cur; ++i) {
    int o = idx[i];
    a0[2 * i] = pts_l0[2 * o];
    a0[2 * i + 1] = pts_l0[2 * o + 1];
    a1[2 * i] = pts_l1[2 * o];
    a1[2 * i + 1] = pts_l1[2 * o + 1];
    as[i] = scale[o];
    m[i] = 1;
  }
  STAGE_T(5);
  int rc = vo_ref_track_with_scale(I0l, I1l, W, H, stride, a0, as, cur, a1, m, ic_border_mode,
                                   sum_mode, NULL);
  STAGE_T(2);
  if (rc < 0) goto fail;
  c = 0;
  for (int i = 0; i < cur; ++i) {
    int o = idx[i];
    pts_l1[2 * o] = a1[2 * i];
    pts_l1[2 * o + 1] = a1[2 * i + 1];
    if (m[i]) {
      stage_mask[o] = 2;
      idx[c++] = o;
    }
  }
  cur = c;
  counts->n_refine = cur;
  /* [5] l1 -> r1 */
  for (int i = 0; i < cur; ++i) {
    int o = idx[i];
    a0[2 * i] = pts_l1[2 * o];
    a0[2 * i + 1] = pts_l1[2 * o + 1];
    a1[2 * i] = pts_r1[2 * o];
    a1[2 * i + 1] = pts_r1[2 * o + 1];
    m[i] = 1;
  }
  STAGE_T(5);
  vo_ref_track_with_prior(I1l, I1r, W, H, stride, a0, cur, prm->win, prm->max_level,
                          prm->thres_err, a1, m, n_threads);
  STAGE_T(3);
  c = 0;
  for (int i = 0; i < cur; ++i) {
    int o = idx[i];
    pts_r1[2 * o] = a1[2 * i];
    pts_r1[2 * o + 1] = a1[2 * i + 1];
    if (m[i]) {
      stage_mask[o] = 3;
      idx[c++] = o;
    }
  }
  cur = c;
  counts->n_l1r1 = cur;
  /* [6] stereo pose-only BA on the triangulated survivors (:595-613); mask_motion starts true (:582) and is
   * overwritten only at index_poBA (:631-638) */
  int nba = 0;
  for (int i = 0; i < cur; ++i) {
    int o = idx[i];
    if (lm_flags && !(lm_flags[o] & 1)) continue;
    a0[2 * nba] = pts_l1[2 * o];
    a0[2 * nba + 1] = pts_l1[2 * o + 1];
    a2[2 * nba] = pts_r1[2 * o];
    a2[2 * nba + 1] = pts_r1[2 * o + 1];
    if (world) {
      vo_ref_xform_eig(T_pw, Xp + 3 * o, aX + 3 * nba); /* :605 Xp = T_pw * X */
    } else {
      aX[3 * nba] = Xp[3 * o];
      aX[3 * nba + 1] = Xp[3 * o + 1];
      aX[3 * nba + 2] = Xp[3 * o + 2];
    }
    ++nba;
  }
  counts->n_ba = nba;
  memcpy(dT_out, dT_prior, sizeof(float) * 16);
  vo_ref_gn_info gi;
  STAGE_T(5);
  rc = vo_ref_gn_pose_stereo(aX, a0, a2, nba, prm->Kl, prm->Kr, prm->T_lr, prm->thres_poseba,
                             dT_out, m, sum_mode, tree_width, &gi);
  STAGE_T(4);
  if (rc <= 0) {
    rc = -6; /* reference throws "PoseOnlyStereoBA is failed!" (:626) */
    goto fail;
  }
  counts->gn_iterations = gi.iterations;
  /* [7] y > 660 gate (THRES_SAMPSON = feature_tracker.thres_sampson, 60 in kitti_00_stereo.yaml) on lmtrack_motion_ok (:653-668) */
  c = 0;
  nba = 0;
  for (int i = 0; i < cur; ++i) {
    int o = idx[i];
    int motion_ok = 1;
    if (!(lm_flags && !(lm_flags[o] & 1))) motion_ok = m[nba++];
    float d = pts_l1[2 * o + 1] > 660 ? 100.f : 0.f;
    if (motion_ok && d < prm->thres_sampson) {
      stage_mask[o] = 4;
      ++c;
    }
  }
  counts->n_inlier = c;
  STAGE_T(5);
  /* [10] new points */
  if (n_new > 0) {
    for (int i = 0; i < n_new; ++i) mask_new[i] = 1;
    vo_ref_track_bidirection(I1l, I1r, W, H, stride, pts_new, n_new, prm->win, prm->max_level,
                             prm->thres_err, prm->thres_bidirection, pts_new_r, mask_new,
                             n_threads);
    for (int i = 0; i < n_new; ++i) counts->n_new_ok += mask_new[i];
  }
  STAGE_T(6);
  rc = 0;
fail:
  free(scale);
  free(idx);
  free(m);
  free(a0);
  free(a1);
  free(a2);
  free(aX);
  free(as);
  return rc;
}

int vo_ref_stereo_frame(const vo_ref_stereo_params *prm, const uint8_t *I0l, const uint8_t *I1l, const uint8_t *I1r,
                        int stride, const float *pts_l0, const float *Xp, const uint8_t *lm_flags, int n,
                        const float dT_prior[16], const float *pts_new, int n_new, int sum_mode, int tree_width,
                        int ic_border_mode, int n_threads, float *pts_l1, float *pts_r1, uint8_t *stage_mask,
                        float dT_out[16], float *pts_new_r, uint8_t *mask_new, vo_ref_frame_counts *counts) {
  return stereo_frame_impl(prm, I0l, I1l, I1r, stride, pts_l0, Xp, lm_flags, n, dT_prior, NULL, NULL, pts_new, n_new,
                           sum_mode, tree_width, ic_border_mode, n_threads, pts_l1, pts_r1, stage_mask, dT_out, pts_new_r,
                           mask_new, counts);
}

int vo_ref_stereo_frame_world(const vo_ref_stereo_params *prm, const uint8_t *I0l, const uint8_t *I1l, const uint8_t *I1r,
                              int stride, const float *pts_l0, const float *Xw, const uint8_t *lm_flags, int n,
                              const float dT_prior[16], const float T_pw[16], const float T_cw_prior[16],
                              const float *pts_new, int n_new, int sum_mode, int tree_width, int ic_border_mode,
                              int n_threads, float *pts_l1, float *pts_r1, uint8_t *stage_mask, float dT_out[16],
                              float *pts_new_r, uint8_t *mask_new, vo_ref_frame_counts *counts) {
  if (!T_pw || !T_cw_prior) return -1;
  return stereo_frame_impl(prm, I0l, I1l, I1r, stride, pts_l0, Xw, lm_flags, n, dT_prior, T_pw, T_cw_prior, pts_new, n_new,
                           sum_mode, tree_width, ic_border_mode, n_threads, pts_l1, pts_r1, stage_mask, dT_out, pts_new_r,
                           mask_new, counts);
}

/* ---- epipolar gates ---------------------------------------------------
 * MotionEstimator::calcSampsonDistance (core/visual_odometry/motion_estimator.cpp:572-599) and
 * calcSymmetricEpipolarDistance (:621-653), the per-point part on a given F10 (row-major).
 * Eigen evaluates the 3x3 * 3x1 products and the 1x3 * 3x1 dot coefficient-wise as an unrolled
 * 3-term redux, which associates as e0 + (e1 + e2); the hand-written sums are left to right. */
static inline float eig_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);
}
static void epi_terms(const float *F, float x0, float y0, float x1, float y1, float *Fp0, float *Ftp1, float *num) {
  for (int i = 0; i < 3; ++i) Fp0[i] = eig_dot3(F[i * 3 + 0], x0, F[i * 3 + 1], y0, F[i * 3 + 2], 1.0f);
  for (int i = 0; i < 3; ++i) Ftp1[i] = eig_dot3(F[0 * 3 + i], x1, F[1 * 3 + i], y1, F[2 * 3 + i], 1.0f);
  *num = eig_dot3(x1, Fp0[0], y1, Fp0[1], 1.0f, Fp0[2]);
}
void vo_ref_sampson_distance(const float *pts0, const float *pts1, int n, const float F10[9], float *dist) {
  for (int i = 0; i < n; ++i) {
    float a[3], b[3], num;
    epi_terms(F10, pts0[2 * i], pts0[2 * i + 1], pts1[2 * i], pts1[2 * i + 1], a, b, &num);
    num *= num;
    const float den = ((a[0] * a[0] + a[1] * a[1]) + b[0] * b[0]) + b[1] * b[1];
    dist[i] = num / den;
  }
}
void vo_ref_symmetric_epipolar_distance(const float *pts0, const float *pts1, int n, const float F10[9],
                                        float *dist) {
  for (int i = 0; i < n; ++i) {
    float a[3], b[3], num;
    epi_terms(F10, pts0[2 * i], pts0[2 * i + 1], pts1[2 * i], pts1[2 * i + 1], a, b, &num);
    num = fabsf(num);
    const float den = 1.0f / sqrtf(a[0] * a[0] + a[1] * a[1]) + 1.0f / sqrtf(b[0] * b[0] + b[1] * b[1]);
    dist[i] = num * den;
  }
}
/* F10 = Kinv^T * (skew(t10) * R10) * Kinv (motion_estimator.cpp:551-552), 3x3 products as Eigen
 * evaluates them (same 3-term redux); K = (fx, fy, cx, cy), Kinv as Camera::Kinv (camera.cpp) */
static void mat3_mul(const float *A, const float *B, float *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = eig_dot3(A[i * 3 + 0], B[0 * 3 + j], A[i * 3 + 1], B[1 * 3 + j], A[i * 3 + 2], B[2 * 3 + j]);
}
void vo_ref_fundamental_from_pose(const float K[4], const float R10[9], const float t10[3], float F10[9]) {
  const float fxi = 1.0f / K[0], fyi = 1.0f / K[1];
  const float Kinv[9] = {fxi, 0.0f, -K[2] * fxi, 0.0f, fyi, -K[3] * fyi, 0.0f, 0.0f, 1.0f};
  const float KinvT[9] = {Kinv[0], Kinv[3], Kinv[6], Kinv[1], Kinv[4], Kinv[7], Kinv[2], Kinv[5], Kinv[8]};
  const float S[9] = {0.0f, -t10[2], t10[1], t10[2], 0.0f, -t10[0], -t10[1], t10[0], 0.0f};
  float E[9], T[9];
  mat3_mul(S, R10, E);
  mat3_mul(KinvT, E, T);
  mat3_mul(T, Kinv, F10);
}

/* ---- feature bucketing (the reference-owned part; cv::ORB::detect stays third-party) ----
 * WeightBin::init / reset / update (core/visual_odometry/feature_extractor.h:90-135) and the
 * flag_nonmax_ branch of FeatureExtractor::extractORBwithBinning_fast (feature_extractor.cpp:241-277). */
void vo_ref_weight_bin_init(int n_cols, int n_rows, int n_bins_u, int n_bins_v, int *u_step, int *v_step,
                            float *inv_u_step, float *inv_v_step) {
  *u_step = (int)floor((float)n_cols / (float)n_bins_u); /* :108-109 */
  *v_step = (int)floor((float)n_rows / (float)n_bins_v);
  *inv_u_step = 1.0f / (float)*u_step;                   /* :111-112 */
  *inv_v_step = 1.0f / (float)*v_step;
}
/* reset() then update(pts): a bin that holds a tracked point gets weight 0. The reference tests only the
 * flattened index (:130), so a point right of the last column lands in the next row: reproduced. */
void vo_ref_weight_bin_update(const float *pts, int n, int u_step, int v_step, int n_bins_u, int n_bins_v,
                              int32_t *weight) {
  const int total = n_bins_u * n_bins_v;
  for (int i = 0; i < total; ++i) weight[i] = 1;
  for (int i = 0; i < n; ++i) {
    const int u_idx = (int)floor((float)pts[2 * i] / (float)u_step);
    const int v_idx = (int)floor((float)pts[2 * i + 1] / (float)v_step);
    const int bin_idx = v_idx * n_bins_u + u_idx;
    if (bin_idx >= 0 && bin_idx < total) weight[bin_idx] = 0;
  }
}
/* one keypoint per wanted bin: the FIRST keypoint (detector order) with the largest response, bins in
 * ascending order. Returns the number of points; idx_out[k] = index of the chosen keypoint. */
int vo_ref_bucket_argmax(const float *kp_xy, const float *kp_response, int n, float inv_u_step, float inv_v_step,
                         int n_bins_u, int n_bins_v, const int32_t *weight, float *pts_out, int32_t *idx_out) {
  const int total = n_bins_u * n_bins_v;
  int *index_max = (int *)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
  float *max_score = (float *)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1));
  for (int j = 0; j < total; ++j) {
    index_max[j] = -1;
    max_score[j] = -1.0f;
  }
  for (int i = 0; i < n; ++i) {
    const unsigned int u = (unsigned int)(int)floor(kp_xy[2 * i] * inv_u_step);
    const unsigned int v = (unsigned int)(int)floor(kp_xy[2 * i + 1] * inv_v_step);
    if (u >= (unsigned int)n_bins_u || v >= (unsigned int)n_bins_v) continue; /* (the reference reads weight[] first: UB) */
    const int bin_idx = (int)(v * (unsigned int)n_bins_u + u);
    if (weight[bin_idx] == 0) continue;
    if (max_score[bin_idx] < kp_response[i]) {
      index_max[bin_idx] = i;
      max_score[bin_idx] = kp_response[i];
    }
  }
  int m = 0;
  for (int j = 0; j < total; ++j)
    if (index_max[j] > -1 && weight[j] > 0) {
      pts_out[2 * m] = kp_xy[2 * index_max[j]];
      pts_out[2 * m + 1] = kp_xy[2 * index_max[j] + 1];
      idx_out[m] = index_max[j];
      ++m;
    }
  free(index_max);
  free(max_score);
  return m;
}
