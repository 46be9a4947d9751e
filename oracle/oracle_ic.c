/*
 * oracle_ic.c — CPU restatement of FeatureTracker::trackWithScale, the
 * reference-native scale-compensated inverse-compositional patch refinement.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h). PARITY UNPINNED.
 * Follows:
 *   core/visual_odometry/feature_tracker.cpp:236-504     (trackWithScale)
 *   core/util/image_processing.cpp:79-118                (interpImageSameRatio)
 *   core/util/image_processing.cpp:268-331               (interpImage3SameRatio)
 *   cv::Sobel(..., CV_32FC1, ksize 3) as called at stereo_vo.cpp:551-552
 *
 * REFERENCE border mode reproduces the reference exactly: mask_I0 / mask_I1 /
 * I0_patt / du0_patt / dv0_patt / I1_patt are allocated once before the point
 * loop (feature_tracker.cpp:324-333) and the samplers' resize(n, false / -2.0f)
 * never resets a same-size vector, so a tap that is outside the image in the
 * current evaluation keeps the mask bit and the VALUE written by the most
 * recent earlier evaluation (any point, any iteration) in which that tap was
 * inside the image.  MASKED mode clears the masks before each evaluation.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IC_HALF 11
#define IC_LEN (2 * IC_HALF + 1)
#define IC_MAXELEM (IC_LEN * IC_LEN)

#define IC_TREE_W 64 /* ic_refine_kernel: one wavefront per point, tap j -> lane j mod 64 (ascending j), tree over the 64 lanes */

/* diagnostic: dependency depth of the never-reset tap state (REFERENCE mode): depth(P) = 1 + max depth of
 * the points whose stale tap values P actually read; untouched points have depth 0 */
static int g_depth_max, g_depth_hist[64];
void vo_ref_ic_depth_hist(int *out64, int *maxd) {
  for (int i = 0; i < 64; ++i) out64[i] = g_depth_hist[i];
  *maxd = g_depth_max;
  memset(g_depth_hist, 0, sizeof(g_depth_hist));
  g_depth_max = 0;
}

/* diagnostic: histogram of IC iterations executed per processed point since the last reset */
static int g_iter_hist[32];
void vo_ref_ic_iter_hist(int *out31, int reset) {
  for (int i = 0; i < 31; ++i) out31[i] = g_iter_hist[i];
  if (reset) memset(g_iter_hist, 0, sizeof(g_iter_hist));
}

static float tree_w(const float *part) {
  float v[IC_TREE_W];
  memcpy(v, part, sizeof(v));
  for (int n = 1; n < IC_TREE_W; n <<= 1)
    for (int i = 0; i < IC_TREE_W; i += 2 * n) v[i] = v[i] + v[i + n];
  return v[0];
}

/* Sum of term[j] over j in [0,n) where use[j]!=0.
 * SEQ: j ascending. TREE: IC_TREE_W strided partials (term j -> partial j mod IC_TREE_W, ascending j),
 * then a balanced binary tree over the partials (adjacent pairs first). */
static float masked_sum(const float *term, const uint8_t *use, int n, int mode) {
  if (mode == VO_SUM_SEQ) {
    float s = 0.0f;
    for (int j = 0; j < n; ++j)
      if (use[j]) s += term[j];
    return s;
  }
  float part[IC_TREE_W];
  for (int l = 0; l < IC_TREE_W; ++l) {
    float s = 0.0f;
    for (int j = l; j < n; j += IC_TREE_W)
      if (use[j]) s += term[j];
    part[l] = s;
  }
  return tree_w(part);
}

static inline float bilin(const float I1, const float I2, const float I3,
                          const float I4, float ax, float ay, float axay) {
  /* image_processing.cpp:115 / :313, evaluated left to right */
  return ((axay * (((I1 - I2) - I3) + I4) + ax * (-I1 + I2)) + ay * (-I1 + I3)) + I1;
}

int vo_ref_track_with_scale(const uint8_t *img0, const uint8_t *img1, int w,
                            int h, int stride, const float *pts0,
                            const float *scale_est, int n, float *pts_track,
                            uint8_t *mask, int border_mode, int sum_mode,
                            uint8_t *touched_border) {
  const int MAX_ITER = 30;
  const float EPS_ERR_RATE = 1e-3;
  const float EPS_UPDATE = 1e-4;
  const float minEigThreshold = 1e-4;
  const int n_cols = w, n_rows = h;

  /* img.convertTo(CV_32FC1) and cv::Sobel on the previous image */
  float *I0 = (float *)malloc(sizeof(float) * (size_t)w * h);
  float *I1 = (float *)malloc(sizeof(float) * (size_t)w * h);
  float *dU = (float *)malloc(sizeof(float) * (size_t)w * h);
  float *dV = (float *)malloc(sizeof(float) * (size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      I0[y * w + x] = (float)img0[y * stride + x];
      I1[y * w + x] = (float)img1[y * stride + x];
    }
  vo_ref_sobel3(img0, w, h, stride, dU, dV);

  /* checkerboard pattern, feature_tracker.cpp:308-320 */
  float patt_x[IC_MAXELEM], patt_y[IC_MAXELEM];
  int n_elem = 0;
  for (int v = 0; v < IC_LEN; ++v)
    for (int u = !(v & 0x01); u < IC_LEN; u += 2) {
      patt_x[n_elem] = (float)(u - IC_HALF);
      patt_y[n_elem] = (float)(v - IC_HALF);
      ++n_elem;
    }

  /* containers allocated ONCE (:324-333); zero / false initialised */
  float I0_patt[IC_MAXELEM] = {0}, du0_patt[IC_MAXELEM] = {0}, dv0_patt[IC_MAXELEM] = {0};
  float I1_patt[IC_MAXELEM] = {0};
  uint8_t mask_I0[IC_MAXELEM] = {0}, mask_I1[IC_MAXELEM] = {0};
  float term[IC_MAXELEM];
  uint8_t use[IC_MAXELEM];
  float patt_sx[IC_MAXELEM], patt_sy[IC_MAXELEM];
  int rc = 0;
  int w0[IC_MAXELEM], w1[IC_MAXELEM]; /* diagnostics: last writer point of each tap */
  int *depth = (int *)calloc((size_t)n + 1, sizeof(int));
  for (int j = 0; j < IC_MAXELEM; ++j) w0[j] = w1[j] = -1;

  for (int i = 0; i < n; ++i) {
    if (touched_border) touched_border[i] = 0;
    if (!mask[i]) continue;
    const float pt0x = pts0[2 * i], pt0y = pts0[2 * i + 1];
    const float pt1x = pts_track[2 * i], pt1y = pts_track[2 * i + 1];
    const float scale = scale_est[i];
    for (int j = 0; j < n_elem; ++j) {
      patt_sx[j] = patt_x[j] * scale;
      patt_sy[j] = patt_y[j] * scale;
    }
    float ax = (float)((double)pt0x - floor((double)pt0x));
    float ay = (float)((double)pt0y - floor((double)pt0y));
    float axay = ax * ay;
    if (ax < 0 || ax > 1 || ay < 0 || ay > 1) {
      mask[i] = 0;
      continue;
    }
    /* interpImage3SameRatio (image_processing.cpp:268-331) */
    if (border_mode == VO_IC_BORDER_MASKED) memset(mask_I0, 0, sizeof(mask_I0));
    for (int j = 0; j < n_elem; ++j) {
      float uc = pt0x + patt_x[j], vc = pt0y + patt_y[j];
      int u0 = (int)uc, v0 = (int)vc;
      if (u0 < 1 || u0 >= n_cols - 2 || v0 < 1 || v0 >= n_rows - 2) {
        if (touched_border) touched_border[i] = 1;
        if (mask_I0[j] && w0[j] >= 0 && w0[j] != i && depth[w0[j]] + 1 > depth[i]) depth[i] = depth[w0[j]] + 1;
        continue;
      }
      w0[j] = i;
      int idx = v0 * n_cols + u0;
      I0_patt[j] = bilin(I0[idx], I0[idx + 1], I0[idx + n_cols], I0[idx + n_cols + 1], ax, ay, axay);
      du0_patt[j] = bilin(dU[idx], dU[idx + 1], dU[idx + n_cols], dU[idx + n_cols + 1], ax, ay, axay);
      dv0_patt[j] = bilin(dV[idx], dV[idx + 1], dV[idx + n_cols], dV[idx + n_cols + 1], ax, ay, axay);
      mask_I0[j] = 1;
    }
    for (int j = 0; j < n_elem; ++j) term[j] = du0_patt[j] * du0_patt[j];
    float A11 = masked_sum(term, mask_I0, n_elem, sum_mode);
    for (int j = 0; j < n_elem; ++j) term[j] = du0_patt[j] * dv0_patt[j];
    float A12 = masked_sum(term, mask_I0, n_elem, sum_mode);
    for (int j = 0; j < n_elem; ++j) term[j] = dv0_patt[j] * dv0_patt[j];
    float A22 = masked_sum(term, mask_I0, n_elem, sum_mode);
    float D = A11 * A22 - A12 * A12;
    if (D < minEigThreshold) {
      mask[i] = 0;
      continue;
    }
    float invD = (float)(1.0 / (double)D);
    float iD_A11 = A11 * invD, iD_A12 = A12 * invD, iD_A22 = A22 * invD;

    float err_curr = 0, err_prev = 1e12;
    float tx = pt1x - pt0x, ty = pt1y - pt0y;
    int iters_done = 0;
    for (int iter = 0; iter < MAX_ITER; ++iter) {
      iters_done = iter + 1;
      float pux = pt0x + tx, puy = pt0y + ty;
      ax = (float)((double)pux - floor((double)pux));
      ay = (float)((double)puy - floor((double)puy));
      axay = ax * ay;
      if (ax < 0 || ax > 1 || ay < 0 || ay > 1) {
        mask[i] = 0;
        break;
      }
      if (isnan(ax + ay)) {
        rc = -3; /* reference throws "ax ay nan" */
        goto done;
      }
      /* interpImageSameRatio (image_processing.cpp:79-118): float compares */
      if (border_mode == VO_IC_BORDER_MASKED) memset(mask_I1, 0, sizeof(mask_I1));
      for (int j = 0; j < n_elem; ++j) {
        float uc = pux + patt_sx[j], vc = puy + patt_sy[j];
        if (uc < 1 || uc >= (float)(n_cols - 2) || vc < 1 || vc >= (float)(n_rows - 2)) {
          if (touched_border) touched_border[i] = 1;
          if (mask_I1[j] && mask_I0[j] && w1[j] >= 0 && w1[j] != i && depth[w1[j]] + 1 > depth[i]) depth[i] = depth[w1[j]] + 1;
          continue;
        }
        w1[j] = i;
        int u0 = (int)uc, v0 = (int)vc;
        int idx = v0 * n_cols + u0;
        I1_patt[j] = bilin(I1[idx], I1[idx + 1], I1[idx + n_cols], I1[idx + n_cols + 1], ax, ay, axay);
        mask_I1[j] = 1;
      }
      int cnt_valid = 0;
      for (int j = 0; j < n_elem; ++j) {
        use[j] = (mask_I0[j] && mask_I1[j]);
        cnt_valid += use[j];
        if (use[j] && (isnan(I0_patt[j]) || isnan(I1_patt[j]) || isnan(du0_patt[j]) ||
                       isnan(dv0_patt[j]))) {
          rc = -4; /* reference throws */
          goto done;
        }
      }
      for (int j = 0; j < n_elem; ++j) term[j] = du0_patt[j] * (I1_patt[j] - I0_patt[j]);
      float b1 = masked_sum(term, use, n_elem, sum_mode);
      for (int j = 0; j < n_elem; ++j) term[j] = dv0_patt[j] * (I1_patt[j] - I0_patt[j]);
      float b2 = masked_sum(term, use, n_elem, sum_mode);
      for (int j = 0; j < n_elem; ++j) {
        float r = I1_patt[j] - I0_patt[j];
        term[j] = r * r;
      }
      err_curr = masked_sum(term, use, n_elem, sum_mode);

      float dtu = (-iD_A22 * b1 + iD_A12 * b2);
      float dtv = (iD_A12 * b1 - iD_A11 * b2);
      if (isnan(dtu + dtv)) {
        rc = -5; /* reference throws "dtu dtv nan" */
        goto done;
      }
      tx += dtu;
      ty += dtv;
      err_curr /= (float)cnt_valid;
      err_curr = sqrtf(err_curr);
      float err_rate = fabsf(err_prev - err_curr) / err_prev;
      float dt_norm = dtu * dtu + dtv * dtv;
      if (iter > 1) {
        if (err_rate <= EPS_ERR_RATE || dt_norm <= EPS_UPDATE) break;
      }
      err_prev = err_curr;
    }
    g_iter_hist[iters_done > 30 ? 30 : iters_done]++;
    g_depth_hist[depth[i] > 63 ? 63 : depth[i]]++;
    if (depth[i] > g_depth_max) g_depth_max = depth[i];
    if (isnan(err_curr)) {
      mask[i] = 0;
    } else if (err_curr <= 30) {
      pts_track[2 * i] = pt0x + tx;
      pts_track[2 * i + 1] = pt0y + ty;
      mask[i] = 1;
    } else
      mask[i] = 0;
  }
done:
  free(depth);
  free(I0);
  free(I1);
  free(dU);
  free(dV);
  return rc;
}
