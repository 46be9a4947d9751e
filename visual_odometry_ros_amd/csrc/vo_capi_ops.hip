// vo_capi_ops.hip — extern "C" entry points (include/vo_hip.h) for
// trackWithScale, calcPrior, the ORB Hamming operators and mask compaction.
// Host pointers in, host pointers out, synchronous at return.
#include "vo_internal.hpp"
#include "vo_kernels.hpp"

#define H2D(dst, src, bytes) VO_CHECK_HIP(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, c->stream))
#define D2H(dst, src, bytes) VO_CHECK_HIP(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream))
#define SYNC() VO_CHECK_HIP(c, hipStreamSynchronize(c->stream))

static int check_n(vo_ctx *c, int n) {
  if (n < 0) VO_FAIL(c, VO_ERR_INVALID, "negative point count");
  if (n > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "n=%d exceeds vo_config.max_points=%d", n, c->cfg.max_points);
  return VO_OK;
}

static int take_flags(vo_ctx *c) {
  int flags = 0;
  D2H(&flags, c->d_flags, sizeof(int));
  SYNC();
  if (!flags) return VO_OK;
  VO_CHECK_HIP(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
  SYNC();
  if (flags & 1) VO_FAIL(c, VO_ERR_NAN_AXAY, "ax ay nan");
  if (flags & 2) VO_FAIL(c, VO_ERR_NAN_PATCH, "I0 I1 / du0 dv0 nan");
  VO_FAIL(c, VO_ERR_NAN_UPDATE, "dtu dtv nan");
}

// FeatureTracker::trackWithScale, feature_tracker.cpp:236-504
extern "C" int vo_track_with_scale(vo_ctx *c, int slot0, int slot1, const float *pts0, const float *scale_est,
                                   int n, float *pts_track, uint8_t *mask_valid, int strict_border) {
  if (!c || !pts0 || !scale_est || !pts_track || !mask_valid) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  if (n == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  const size_t pb = sizeof(float) * 2 * (size_t)n;
  H2D(c->d_pts0, pts0, pb);
  H2D(c->d_pts1, pts_track, pb);
  H2D(c->d_scale, scale_est, sizeof(float) * (size_t)n);
  H2D(c->d_mask, mask_valid, (size_t)n);
  // d_pts1 = prior (in), d_pts2 = refined (out), d_mask2 = touched, d_status = class, d_pts3 = last pt_update
  rc = vo_ic_enqueue(c, slot0, slot1, c->d_pts0, c->d_scale, c->d_pts1, c->d_pts2, c->d_mask, c->d_mask,
                     c->d_mask2, c->d_status, c->d_pts3, n, nullptr, nullptr, strict_border != 0);
  if (rc < 0) return rc;
  if (strict_border) {
    rc = vo_ic_strict_enqueue(c, slot0, slot1, c->d_pts0, c->d_scale, c->d_pts1, c->d_pts2, c->d_mask,
                              c->d_mask2, c->d_status, c->d_pts3, n, nullptr, nullptr, strict_border == 2);
    if (rc < 0) return rc;
  }
  D2H(pts_track, c->d_pts2, pb);
  D2H(mask_valid, c->d_mask, (size_t)n);
  rc = take_flags(c);
  return rc;
}

// general 4x4 inverse by cofactors: Eigen Matrix4f::inverse() at feature_tracker.cpp:215
static void inverse4x4(const float m[16], float inv[16]) {
  float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
  float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
  float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
  float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
  float det = ((s0 * c5 - s1 * c4) + s2 * c3 + s3 * c2 - s4 * c1) + s5 * c0;
  float id = 1.0f / det;
  inv[0] = ((m[5] * c5 - m[6] * c4) + m[7] * c3) * id;
  inv[1] = ((-m[1] * c5 + m[2] * c4) - m[3] * c3) * id;
  inv[2] = ((m[13] * s5 - m[14] * s4) + m[15] * s3) * id;
  inv[3] = ((-m[9] * s5 + m[10] * s4) - m[11] * s3) * id;
  inv[4] = ((-m[4] * c5 + m[6] * c2) - m[7] * c1) * id;
  inv[5] = ((m[0] * c5 - m[2] * c2) + m[3] * c1) * id;
  inv[6] = ((-m[12] * s5 + m[14] * s2) - m[15] * s1) * id;
  inv[7] = ((m[8] * s5 - m[10] * s2) + m[11] * s1) * id;
  inv[8] = ((m[4] * c4 - m[5] * c2) + m[7] * c0) * id;
  inv[9] = ((-m[0] * c4 + m[1] * c2) - m[3] * c0) * id;
  inv[10] = ((m[12] * s4 - m[13] * s2) + m[15] * s0) * id;
  inv[11] = ((-m[8] * s4 + m[9] * s2) - m[11] * s0) * id;
  inv[12] = ((-m[4] * c3 + m[5] * c1) - m[6] * c0) * id;
  inv[13] = ((m[0] * c3 - m[1] * c1) + m[2] * c0) * id;
  inv[14] = ((-m[12] * s3 + m[13] * s1) - m[14] * s0) * id;
  inv[15] = ((m[8] * s3 - m[9] * s1) + m[10] * s0) * id;
}

// FeatureTracker::calcPrior, feature_tracker.cpp:208-234
extern "C" int vo_calc_prior(vo_ctx *c, const float *pts0, int n_pts0, const float *Xw, int n, const float Tw1[16],
                             const float K[9], float *pts1_prior) {
  if (!c || !pts0 || !Xw || !Tw1 || !K || !pts1_prior) return VO_ERR_INVALID;
  int rc = check_n(c, n_pts0);
  if (rc) return rc;
  rc = check_n(c, n);
  if (rc) return rc;
  if (n > n_pts0) VO_FAIL(c, VO_ERR_SIZE, "calcPrior: Xw.size() > pts0.size()");
  if (n_pts0 == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  H2D(c->d_pts0, pts0, sizeof(float) * 2 * (size_t)n_pts0);
  if (n) H2D(c->d_X, Xw, sizeof(float) * 3 * (size_t)n);
  float T1w[16];
  inverse4x4(Tw1, T1w);
  rc = vo_calc_prior_enqueue(c, c->d_pts0, n_pts0, c->d_X, n, T1w, K, c->d_pts1);
  if (rc < 0) return rc;
  D2H(pts1_prior, c->d_pts1, sizeof(float) * 2 * (size_t)n_pts0);
  SYNC();
  return VO_OK;
}

// MotionEstimator::calcSampsonDistance (F10 overload, motion_estimator.cpp:572-599) and
// calcSymmetricEpipolarDistance (:621-653)
static int epi_distance(vo_ctx *c, int mode, const float *pts0, const float *pts1, int n, const float F10[9],
                        float *dist) {
  if (!c || !pts0 || !pts1 || !F10 || !dist) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  if (n == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  H2D(c->d_pts0, pts0, sizeof(float) * 2 * (size_t)n);
  H2D(c->d_pts1, pts1, sizeof(float) * 2 * (size_t)n);
  rc = vo_epi_distance_enqueue(c, mode, c->d_pts0, c->d_pts1, n, F10, c->d_err);
  if (rc < 0) return rc;
  D2H(dist, c->d_err, sizeof(float) * (size_t)n);
  SYNC();
  return VO_OK;
}
extern "C" int vo_sampson_distance(vo_ctx *c, const float *pts0, const float *pts1, int n, const float F10[9],
                                   float *dist) {
  return epi_distance(c, 0, pts0, pts1, n, F10, dist);
}
extern "C" int vo_symmetric_epipolar_distance(vo_ctx *c, const float *pts0, const float *pts1, int n,
                                              const float F10[9], float *dist) {
  return epi_distance(c, 1, pts0, pts1, n, F10, dist);
}

// WeightBin::reset + update (feature_extractor.h:120-135): weight[j] = 0 for bins holding a point
extern "C" int vo_weight_bin_update(vo_ctx *c, const float *pts, int n, int u_step, int v_step, int n_bins_u,
                                    int n_bins_v, int32_t *weight) {
  if (!c || (n > 0 && !pts) || !weight || u_step <= 0 || v_step <= 0 || n_bins_u <= 0 || n_bins_v <= 0)
    return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  rc = check_n(c, n_bins_u * n_bins_v);
  if (rc) return rc;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (n) H2D(c->d_pts0, pts, sizeof(float) * 2 * (size_t)n);
  rc = vo_weight_bin_update_enqueue(c, c->d_pts0, n, u_step, v_step, n_bins_u, n_bins_v, c->d_idx);
  if (rc < 0) return rc;
  D2H(weight, c->d_idx, sizeof(int32_t) * (size_t)(n_bins_u * n_bins_v));
  SYNC();
  return VO_OK;
}

// the bucketing of FeatureExtractor::extractORBwithBinning_fast (feature_extractor.cpp:241-277) on the
// keypoints (position, response) cv::ORB::detect returned, in detector order
extern "C" int vo_bucket_argmax(vo_ctx *c, const float *kp_xy, const float *kp_response, int n, float inv_u_step,
                                float inv_v_step, int n_bins_u, int n_bins_v, const int32_t *weight, float *pts_out,
                                int32_t *idx_out, int *n_out) {
  if (!c || (n > 0 && (!kp_xy || !kp_response)) || !weight || !pts_out || !n_out || n_bins_u <= 0 || n_bins_v <= 0)
    return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  const int total = n_bins_u * n_bins_v;
  rc = check_n(c, total);
  if (rc) return rc;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (n) {
    H2D(c->d_pts0, kp_xy, sizeof(float) * 2 * (size_t)n);
    H2D(c->d_err, kp_response, sizeof(float) * (size_t)n);
  }
  H2D(c->d_idx, weight, sizeof(int32_t) * (size_t)total);
  rc = vo_bucket_argmax_enqueue(c, c->d_pts0, c->d_err, n, inv_u_step, inv_v_step, n_bins_u, n_bins_v, c->d_idx,
                                (unsigned long long *)c->d_pts2, c->d_pts3, (int32_t *)c->d_pts1, c->d_count);
  if (rc < 0) return rc;
  int m = 0;
  D2H(&m, c->d_count, sizeof(int));
  SYNC();
  if (m > 0) {
    D2H(pts_out, c->d_pts3, sizeof(float) * 2 * (size_t)m);
    if (idx_out) D2H(idx_out, c->d_pts1, sizeof(int32_t) * (size_t)m);
    SYNC();
  }
  *n_out = m;
  return VO_OK;
}

static int ensure_desc(vo_ctx *c, int na, int nb, bool need_dist) {
  const size_t need = (size_t)(na > nb ? na : nb) * 32;
  if (need > c->desc_cap) {
    if (c->d_desc_a) (void)hipFree(c->d_desc_a);
    if (c->d_desc_b) (void)hipFree(c->d_desc_b);
    c->d_desc_a = c->d_desc_b = nullptr;
    c->desc_cap = 0;
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&c->d_desc_a, need));
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&c->d_desc_b, need));
    c->desc_cap = need;
  }
  if (need_dist) {
    const size_t nd = (size_t)na * (size_t)nb * sizeof(uint16_t);
    if (nd > c->dist_cap) {
      if (c->d_dist) (void)hipFree(c->d_dist);
      c->d_dist = nullptr;
      c->dist_cap = 0;
      VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&c->d_dist, nd));
      c->dist_cap = nd;
    }
  }
  return VO_OK;
}

// FeatureExtractor::descriptorDistance over two descriptor sets
extern "C" int vo_orb_hamming(vo_ctx *c, const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *dist) {
  if (!c || !a || !b || !dist || na < 0 || nb < 0) return VO_ERR_INVALID;
  if (na == 0 || nb == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = ensure_desc(c, na, nb, true);
  if (rc) return rc;
  H2D(c->d_desc_a, a, (size_t)na * 32);
  H2D(c->d_desc_b, b, (size_t)nb * 32);
  rc = vo_hamming_enqueue(c, c->d_desc_a, na, c->d_desc_b, nb, c->d_dist);
  if (rc < 0) return rc;
  D2H(dist, c->d_dist, (size_t)na * nb * sizeof(uint16_t));
  SYNC();
  return VO_OK;
}

extern "C" int vo_orb_match(vo_ctx *c, const uint8_t *a, int na, const uint8_t *b, int nb, int th_low, float ratio,
                            int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist) {
  if (!c || !a || (!b && nb > 0) || !best_idx || !best_dist || !second_dist || na < 0 || nb < 0)
    return VO_ERR_INVALID;
  if (na == 0) return VO_OK;
  int rc = check_n(c, na);
  if (rc) return rc;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  rc = ensure_desc(c, na, nb > 0 ? nb : 1, false);
  if (rc) return rc;
  H2D(c->d_desc_a, a, (size_t)na * 32);
  if (nb > 0) H2D(c->d_desc_b, b, (size_t)nb * 32);
  // outputs reuse per-point buffers: idx (int32), err (as 2 x uint16 planes)
  uint16_t *d_bd = (uint16_t *)c->d_err, *d_sd = (uint16_t *)c->d_err2;
  rc = vo_match_enqueue(c, c->d_desc_a, na, c->d_desc_b, nb, th_low, ratio, c->d_idx, d_bd, d_sd);
  if (rc < 0) return rc;
  D2H(best_idx, c->d_idx, sizeof(int32_t) * (size_t)na);
  D2H(best_dist, d_bd, sizeof(uint16_t) * (size_t)na);
  D2H(second_dist, d_sd, sizeof(uint16_t) * (size_t)na);
  SYNC();
  return VO_OK;
}

// StereoLandmarkTracking(src, mask) / LandmarkTracking(src, mask): landmark.cpp:291-332, :194-231
extern "C" int vo_compact_indices(vo_ctx *c, const uint8_t *mask, const uint8_t *alive, const uint8_t *tracked,
                                  int n, int32_t *index_valid, int *n_out) {
  if (!c || !mask || !index_valid || !n_out) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  *n_out = 0;
  if (n == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  H2D(c->d_mask, mask, (size_t)n);
  if (alive) H2D(c->d_status, alive, (size_t)n);
  if (tracked) H2D(c->d_status2, tracked, (size_t)n);
  CompactArgsHost h;
  h.mask = c->d_mask;
  h.alive = alive ? c->d_status : nullptr;
  h.tracked = tracked ? c->d_status2 : nullptr;
  h.n = n;
  h.index_valid = c->d_idx;
  h.d_n_out = c->d_count;
  rc = vo_compact_enqueue(c, h);
  if (rc < 0) return rc;
  int cnt = 0;
  D2H(&cnt, c->d_count, sizeof(int));
  SYNC();
  if (cnt > 0) {
    D2H(index_valid, c->d_idx, sizeof(int32_t) * (size_t)cnt);
    SYNC();
  }
  *n_out = cnt;
  return VO_OK;
}

// ---- track IDs -----------------------------------------------------------------------------------
// Landmark::landmark_counter_ (landmark.h:64; id_(landmark_counter_++) in both constructors, landmark.cpp:6, :29) and
// Frame::frame_counter_ (frame.h:53; id_ = frame_counter_++, frame.cpp:15, :35) as state of the context.
extern "C" int vo_ids_reset(vo_ctx *c, int32_t next_landmark_id, int32_t next_frame_id) {
  if (!c || next_landmark_id < 0 || next_frame_id < 0) return VO_ERR_INVALID;
  c->next_landmark_id = next_landmark_id;
  c->next_frame_id = next_frame_id;
  return VO_OK;
}

extern "C" int vo_ids_peek(const vo_ctx *c, int32_t *next_landmark_id, int32_t *next_frame_id) {
  if (!c) return VO_ERR_INVALID;
  if (next_landmark_id) *next_landmark_id = c->next_landmark_id;
  if (next_frame_id) *next_frame_id = c->next_frame_id;
  return VO_OK;
}

extern "C" int vo_ids_new_frames(vo_ctx *c, int n, int32_t *ids) {
  if (!c || n < 0 || (n > 0 && !ids)) return VO_ERR_INVALID;
  if (c->next_frame_id > INT32_MAX - n) VO_FAIL(c, VO_ERR_CAPACITY, "frame id counter would overflow");
  for (int i = 0; i < n; ++i) ids[i] = c->next_frame_id++;
  return VO_OK;
}

extern "C" int vo_ids_new_landmarks(vo_ctx *c, const uint8_t *accept, int n, int32_t *ids, int *n_created) {
  if (!c || n < 0 || (n > 0 && !ids)) return VO_ERR_INVALID;
  if (c->next_landmark_id > INT32_MAX - n) VO_FAIL(c, VO_ERR_CAPACITY, "landmark id counter would overflow");
  int made = 0;
  for (int i = 0; i < n; ++i) {
    if (!accept || accept[i]) {
      ids[i] = c->next_landmark_id++;
      ++made;
    } else {
      ids[i] = -1;
    }
  }
  if (n_created) *n_created = made;
  return VO_OK;
}

extern "C" int vo_compact_tracks(vo_ctx *c, const uint8_t *mask, const uint8_t *alive, uint8_t *tracked,
                                 const int32_t *ids, int n, int32_t *index_valid, int32_t *ids_out, int *n_out) {
  if (!c || !mask || !tracked || !index_valid || !n_out) return VO_ERR_INVALID;
  int rc = vo_compact_indices(c, mask, alive, tracked, n, index_valid, n_out);  // the stable compaction, on the device
  if (rc < 0) return rc;
  // the constructor's `else lms[i]->setUntracked()` (landmark.cpp:211-212, :309-310): index_valid ascends
  int k = 0;
  for (int i = 0; i < n; ++i) {
    if (k < *n_out && index_valid[k] == i) {
      if (ids && ids_out) ids_out[k] = ids[i];
      ++k;
    } else {
      tracked[i] = 0;
    }
  }
  return VO_OK;
}
