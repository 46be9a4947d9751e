/* oracle_mono.c — CPU restatement (TEST INFRASTRUCTURE, parity unpinned — see vo_oracle.h) of the
 * steady-state operator sequence of MonoVO::trackImage (core/visual_odometry/mono_vo/mono_vo.cpp):
 *   prior pixels + patch scale            :739-761
 *   trackBidirectionWithPrior I0 -> I1    :768-770, LandmarkTracking(src, mask) :773
 *   Sobel + trackWithScale                :779-786, compaction :788
 *   index_ba selection (depth > 0.1)      :799-826
 *   poseOnlyBundleAdjustment (core)       :856-867, mask_motion :872-879
 *   Sampson distance gate                 :954-963
 * expressed on plain arrays. What the landmark graph decides on the host comes in as flags:
 * bit 0 = lm->isBundled() (prior and scale from the 3-D point), bit 1 = the landmark belongs to the
 * class the frame uses for pose-only BA (isBundled() with more than 5 keyframes, else
 * isTriangulated(), :800-826). The 5-point fallback (:905-935, OpenCV calib3d) is not part of the
 * path: when the BA has too few points or fails, need_five_point is set and the frame stops after
 * the refinement. stage[i] = number of gates feature i passed: 1 tracked, 2 refined, 3 motion
 * inlier (or not part of the BA), 4 passed the Sampson gate. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

static void xform3(const float T[16], const float X[3], float Y[3]) {
  for (int r = 0; r < 3; ++r)
    Y[r] = ((T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1]) + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
}

int vo_ref_mono_frame(const vo_ref_mono_params *prm, const uint8_t *I0, const uint8_t *I1, int stride,
                      const float *pts0, const float *Xw, const uint8_t *flags, int n, const float Tcw_prev[16],
                      const float Tcw_prior[16], const float dT01_prior[16], int sum_mode, int tree_width,
                      int ic_border_mode, int n_threads, float *pts1, float *scale, uint8_t *stage,
                      float dT01_out[16], vo_ref_mono_counts *counts) {
  const int W = prm->width, H = prm->height;
  const size_t N = (size_t)n + 1;
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * N), *idx_ba = (int32_t *)malloc(sizeof(int32_t) * N);
  uint8_t *m = (uint8_t *)malloc(N), *motion = (uint8_t *)malloc(N);
  float *a0 = (float *)malloc(sizeof(float) * 2 * N), *a1 = (float *)malloc(sizeof(float) * 2 * N);
  float *as = (float *)malloc(sizeof(float) * N), *aX = (float *)malloc(sizeof(float) * 3 * N);
  float *Xp = (float *)malloc(sizeof(float) * 3 * N), *dist = (float *)malloc(sizeof(float) * N);
  int rc = 1;
  memset(counts, 0, sizeof(*counts));
  memcpy(dT01_out, dT01_prior, sizeof(float) * 16);
  /* prior + scale (:739-761) */
  for (int i = 0; i < n; ++i) {
    float patch_scale = 1.0f;
    pts1[2 * i] = pts0[2 * i];
    pts1[2 * i + 1] = pts0[2 * i + 1];
    Xp[3 * i] = Xp[3 * i + 1] = Xp[3 * i + 2] = 0.0f;
    if (flags[i] & 3) xform3(Tcw_prev, Xw + 3 * i, Xp + 3 * i);
    if (flags[i] & 1) {
      float Xc[3];
      xform3(Tcw_prior, Xw + 3 * i, Xc);
      patch_scale = Xp[3 * i + 2] / Xc[2];
      if (Xc[2] > 0) { /* cam_->projectToPixel, camera.cpp:208-213 */
        const float invz = 1.0f / Xc[2];
        pts1[2 * i] = prm->K[0] * Xc[0] * invz + prm->K[2];
        pts1[2 * i + 1] = prm->K[1] * Xc[1] * invz + prm->K[3];
      }
    }
    scale[i] = patch_scale;
    stage[i] = 0;
    m[i] = 1;
  }
  /* trackBidirectionWithPrior (:768) */
  vo_ref_track_bidirection_with_prior(I0, I1, W, H, stride, pts0, n, prm->win, prm->max_level, prm->thres_err,
                                      prm->thres_bidirection, pts1, m, n_threads);
  int cur = 0;
  for (int i = 0; i < n; ++i)
    if (m[i] && !(flags[i] & 4)) { /* LandmarkTracking(lmtrack_prev, mask_track), landmark.cpp:207; bit 2 = not alive / tracked */
      stage[i] = 1;
      idx[cur++] = i;
    }
  counts->n_klt = cur;
  /* trackWithScale on the survivors (:779-788) */
  for (int i = 0; i < cur; ++i) {
    const int o = idx[i];
    a0[2 * i] = pts0[2 * o];
    a0[2 * i + 1] = pts0[2 * o + 1];
    a1[2 * i] = pts1[2 * o];
    a1[2 * i + 1] = pts1[2 * o + 1];
    as[i] = scale[o];
    m[i] = 1;
  }
  rc = vo_ref_track_with_scale(I0, I1, W, H, stride, a0, as, cur, a1, m, ic_border_mode, sum_mode, NULL);
  if (rc < 0) goto done;
  rc = 1;
  int c = 0;
  for (int i = 0; i < cur; ++i) {
    const int o = idx[i];
    pts1[2 * o] = a1[2 * i];
    pts1[2 * o + 1] = a1[2 * i + 1];
    if (m[i]) {
      stage[o] = 2;
      idx[c++] = o;
    }
  }
  cur = c;
  counts->n_refine = cur;
  /* index_ba (:799-826) and pose-only BA (:838-867) */
  int n_ba = 0;
  for (int i = 0; i < cur; ++i) {
    const int o = idx[i];
    motion[i] = 1;
    if ((flags[o] & 2) && Xp[3 * o + 2] > 0.1f) {
      idx_ba[n_ba] = i;
      aX[3 * n_ba] = Xp[3 * o];
      aX[3 * n_ba + 1] = Xp[3 * o + 1];
      aX[3 * n_ba + 2] = Xp[3 * o + 2];
      a0[2 * n_ba] = pts1[2 * o];
      a0[2 * n_ba + 1] = pts1[2 * o + 1];
      ++n_ba;
    }
  }
  counts->n_ba = n_ba;
  int ok = 0;
  float R01[9], t01[3];
  if (n_ba > 10) {
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) R01[r * 3 + q] = dT01_prior[r * 4 + q];
      t01[r] = dT01_prior[r * 4 + 3];
    }
    vo_ref_gn_info gi;
    ok = vo_ref_gn_pose_mono(aX, a0, n_ba, prm->K, prm->thres_poseba, R01, t01, m, VO_GN_VARIANT_CORE, sum_mode,
                             tree_width, &gi) > 0;
    counts->gn_iterations = gi.iterations;
  }
  if (!ok) {
    counts->need_five_point = 1; /* :905: calcPose5PointsAlgorithm on the host */
    goto done;
  }
  for (int i = 0; i < n_ba; ++i) motion[idx_ba[i]] = m[i];
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) dT01_out[r * 4 + q] = R01[r * 3 + q];
    dT01_out[r * 4 + 3] = t01[r];
  }
  dT01_out[12] = dT01_out[13] = dT01_out[14] = 0.0f;
  dT01_out[15] = 1.0f;
  /* LandmarkTracking(lmtrack_scaleok, mask_motion) (:951), Sampson gate (:954-963) */
  {
    float dT10[16], R10[9], t10[3], F10[9];
    vo_ref_inverse_se3(dT01_out, dT10);
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) R10[r * 3 + q] = dT10[r * 4 + q];
      t10[r] = dT10[r * 4 + 3];
    }
    vo_ref_fundamental_from_pose(prm->K, R10, t10, F10);
    c = 0;
    for (int i = 0; i < cur; ++i)
      if (motion[i]) {
        const int o = idx[i];
        stage[o] = 3;
        a0[2 * c] = pts0[2 * o];
        a0[2 * c + 1] = pts0[2 * o + 1];
        a1[2 * c] = pts1[2 * o];
        a1[2 * c + 1] = pts1[2 * o + 1];
        idx[c++] = o;
      }
    counts->n_motion = c;
    vo_ref_sampson_distance(a0, a1, c, F10, dist);
    int f = 0;
    for (int i = 0; i < c; ++i)
      if (dist[i] < prm->thres_sampson) {
        stage[idx[i]] = 4;
        ++f;
      }
    counts->n_final = f;
  }
done:
  free(idx); free(idx_ba); free(m); free(motion); free(a0); free(a1); free(as); free(aX); free(Xp); free(dist);
  return rc;
}
