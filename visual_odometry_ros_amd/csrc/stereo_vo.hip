// stereo_vo.hip — the closed loop of StereoVO::trackStereoImages around the frame operator, track set on the device.
//
// Reference (paths relative to the reference repository root):
//   core/visual_odometry/stereo_vo/stereo_vo.cpp:392-989   trackStereoImages
//     :465-480   [1] the previous frame's track set, [2] constant-velocity prior T_wp * dT_pc_prev
//     :483-670   [3]-[7]                                      -> frame_fused.hip / gn_pose.hip (world-frame landmarks)
//     :691-711   [10] updateWeightBin / extract / trackBidirection -> closed on the device (orb_detect.hip, np_emit.hpp)
//     :714-739   new landmarks: mask_new && Xl(2) > 0 && Xr(2) > 0      -> svo_advance_kernel
//     :752       setStereoPtsSeenAndRelatedLandmarks: the next track set -> svo_advance_kernel
//     :755-797   keyframe rule + reconstruction of lmtrack_final        -> host (rule) + svo_keyframe_kernel
//     :802       local bundle adjustment                                -> svo_local_ba (sba.hip)
//     :842-949   the very first pair
//   core/util/triangulate_3d.cpp:91-130   mapping::triangulateDLT (Eigen::JacobiSVD<MatrixXf>, ComputeFullV, restated)
//   core/visual_odometry/keyframes.cpp:185-303  addNewStereoKeyframe, checkUpdateRule
//   core/visual_odometry/frame.cpp:44-54, :176-196  setPose / setPoseDiff10 / StereoFrame ids
//   core/visual_odometry/landmark.cpp:28-52   Landmark id = landmark_counter_++ (per context here)
//
// Data on the device: two track sets (current / next) of {left pixel, right pixel, world point, flags, id}; a frame reads
// one and svo_advance_kernel writes the other behind the BA launch: survivors (stage 4) in index order, then the
// accepted new points in bin order with ids id_base + rank. The host receives one 48-byte block per frame (counts) next to
// the frame operator's result block, chains the pose (two 4x4 products and two inverses in the reference's order), applies
// the keyframe rule and, at a keyframe, enqueues the reconstruction kernel on the next track set.
#include "frame_state.hpp"
#include "vo_kernels.hpp"

#include <math.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <vector>

int vo_frame_enqueue_impl(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                          const float *pts_l0, const float *pts_r0, const float *Xp, const uint8_t *flags, int n,
                          const float dT_prior[16], const float *pts_new, int n_new, int inputs_on_device,
                          const vo_bin_params *bp, int table, const float *T_pw, const float *T_cw_prior);
int vo_frame_set_advance(vo_ctx *c, const VoAdvArgs *adv);          // frame_pipeline.hip
int vo_frame_set_deferred_detection(vo_ctx *c, int issued);

#define RC(x)                \
  do {                       \
    int _rc = (x);           \
    if (_rc < 0) return _rc; \
  } while (0)

#include "stereo_vo.hpp"

#ifdef GN_STAMP
extern "C" int vo_debug_gn_stamps(vo_ctx *c, long long out[12]);
#endif

__global__ void svo_dlt_kernel(SvoCam cam, const float *p0, const float *p1, int n, float *X0, float *X1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a[3], b[3];
  svo_triangulate(cam, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], a, b);
  for (int k = 0; k < 3; ++k) {
    X0[3 * i + k] = a[k];
    if (X1) X1[3 * i + k] = b[k];
  }
}

// keyframe: reconstruction of the first n_surv entries of the (next) track set, stereo_vo.cpp:763-797 (first frame:
// :907-941, T_wc absent: Xworld = Xl), and every entry becomes a member of the new keyframe (keyframes.cpp:200-215)
struct SvoKfArgs {
  SvoTrackSet ts;
  int n_surv, n_all;
  SvoCam cam;
  float Kl[4], Kr[4];
  int member;  // every entry becomes a member of this (new) keyframe
  int has_T;
  float T_wc[12];
  int *n_recon;  // may be null
};
__global__ void svo_keyframe_kernel(SvoKfArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_all) return;
  uint8_t fl = a.ts.flags[i] | (a.member ? VO_LM_KF_MEMBER : 0);
  if (i < a.n_surv) {
    const float pl[2] = {a.ts.pts_l[2 * i], a.ts.pts_l[2 * i + 1]}, pr[2] = {a.ts.pts_r[2 * i], a.ts.pts_r[2 * i + 1]};
    float Xl[3], Xr[3];
    svo_triangulate(a.cam, pl[0], pl[1], pr[0], pr[1], Xl, Xr);
    // reprojection error in both images at most 1 px (camera.cpp:208-213 projectToPixel)
    float iz = 1.0f / Xl[2];
    float dx = pl[0] - (a.Kl[0] * Xl[0] * iz + a.Kl[2]), dy = pl[1] - (a.Kl[1] * Xl[1] * iz + a.Kl[3]);
    bool ok = !(dx * dx + dy * dy > 1.0f);
    iz = 1.0f / Xr[2];
    dx = pr[0] - (a.Kr[0] * Xr[0] * iz + a.Kr[2]);
    dy = pr[1] - (a.Kr[1] * Xr[1] * iz + a.Kr[3]);
    ok = ok && !(dx * dx + dy * dy > 1.0f);
    ok = ok && Xl[2] > 0 && Xr[2] > 0;
    if (ok) {
      float Xw[3] = {Xl[0], Xl[1], Xl[2]};
      if (a.has_T) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Xw[r] = (a.T_wc[r * 4 + 0] * Xl[0] + (a.T_wc[r * 4 + 1] * Xl[1] + a.T_wc[r * 4 + 2] * Xl[2])) + a.T_wc[r * 4 + 3];
      }
      a.ts.Xw[3 * i] = Xw[0];
      a.ts.Xw[3 * i + 1] = Xw[1];
      a.ts.Xw[3 * i + 2] = Xw[2];
      fl |= VO_LM_TRIANGULATED;
      if (a.n_recon) atomicAdd(a.n_recon, 1);
    }
  }
  a.ts.flags[i] = fl;
}

// ---- host: small fixed-size algebra in the reference's (Eigen's) evaluation order -------------------------------------
void svo_mul44(const float A[16], const float B[16], float C[16]) {  // Matrix4f * Matrix4f: res = a0 b0; res = a_k b_k + res
  float R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float r = A[i * 4 + 0] * B[0 * 4 + j];
      for (int k = 1; k < 4; ++k) r = A[i * 4 + k] * B[k * 4 + j] + r;
      R[i * 4 + j] = r;
    }
  memcpy(C, R, sizeof(R));
}
void svo_inv_se3(const float T[16], float Ti[16]) {  // geometry::inverseSE3_f
  float Rt[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = T[j * 4 + i];
  const float t[3] = {T[3], T[7], T[11]};
  float R[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i * 4 + j] = Rt[i * 3 + j];
    R[i * 4 + 3] = ((-Rt[i * 3 + 0]) * t[0] + (-Rt[i * 3 + 1]) * t[1]) + (-Rt[i * 3 + 2]) * t[2];
  }
  R[12] = R[13] = R[14] = 0;
  R[15] = 1;
  memcpy(Ti, R, sizeof(R));
}
static void svo_inv44(const float m[16], float inv[16]) {  // Matrix4f::inverse() by cofactors (as gn_pose.hip, oracle_gn.c)
  float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
  float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
  float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
  float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
  float det = ((s0 * c5 - s1 * c4) + s2 * c3 + s3 * c2 - s4 * c1) + s5 * c0;
  float id = 1.0f / det;
  float r[16];
  r[0] = ((m[5] * c5 - m[6] * c4) + m[7] * c3) * id;
  r[1] = ((-m[1] * c5 + m[2] * c4) - m[3] * c3) * id;
  r[2] = ((m[13] * s5 - m[14] * s4) + m[15] * s3) * id;
  r[3] = ((-m[9] * s5 + m[10] * s4) - m[11] * s3) * id;
  r[4] = ((-m[4] * c5 + m[6] * c2) - m[7] * c1) * id;
  r[5] = ((m[0] * c5 - m[2] * c2) + m[3] * c1) * id;
  r[6] = ((-m[12] * s5 + m[14] * s2) - m[15] * s1) * id;
  r[7] = ((m[8] * s5 - m[10] * s2) + m[11] * s1) * id;
  r[8] = ((m[4] * c4 - m[5] * c2) + m[7] * c0) * id;
  r[9] = ((-m[0] * c4 + m[1] * c2) - m[3] * c0) * id;
  r[10] = ((m[12] * s4 - m[13] * s2) + m[15] * s0) * id;
  r[11] = ((-m[8] * s4 + m[9] * s2) - m[11] * s0) * id;
  r[12] = ((-m[4] * c3 + m[5] * c1) - m[6] * c0) * id;
  r[13] = ((m[0] * c3 - m[1] * c1) + m[2] * c0) * id;
  r[14] = ((-m[12] * s3 + m[13] * s1) - m[14] * s0) * id;
  r[15] = ((m[8] * s3 - m[9] * s1) + m[10] * s0) * id;
  memcpy(inv, r, sizeof(r));
}
static inline float dot3e(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

static void svo_make_cam(const float T_10[16], const float K0[4], const float K1[4], SvoCam *cam) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) cam->R10[i * 3 + j] = T_10[i * 4 + j];
    cam->t10[i] = T_10[i * 4 + 3];
  }
  // P10 << cam1->K() * R10, cam1->K() * t10 (triangulate_3d.cpp:104)
  const float Km[9] = {K1[0], 0.0f, K1[2], 0.0f, K1[1], K1[3], 0.0f, 0.0f, 1.0f};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      cam->P10[i * 4 + j] = dot3e(Km[i * 3 + 0], cam->R10[0 * 3 + j], Km[i * 3 + 1], cam->R10[1 * 3 + j], Km[i * 3 + 2], cam->R10[2 * 3 + j]);
    cam->P10[i * 4 + 3] = dot3e(Km[i * 3 + 0], cam->t10[0], Km[i * 3 + 1], cam->t10[1], Km[i * 3 + 2], cam->t10[2]);
  }
  memcpy(cam->K0, K0, sizeof(cam->K0));
  memcpy(cam->K1, K1, sizeof(cam->K1));
}

extern "C" int vo_triangulate_dlt(vo_ctx *c, const float *pts0, const float *pts1, int n, const float T_10[16],
                                  const float K0[4], const float K1[4], float *X0, float *X1) {
  if (!c || !pts0 || !pts1 || !T_10 || !K0 || !K1 || !X0 || n < 0) return VO_ERR_INVALID;
  if (n > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "n=%d exceeds vo_config.max_points=%d", n, c->cfg.max_points);
  if (n == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  SvoCam cam;
  svo_make_cam(T_10, K0, K1, &cam);
  hipStream_t s = c->stream;
  VO_CHECK_HIP(c, hipMemcpyAsync(c->d_pts0, pts0, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
  VO_CHECK_HIP(c, hipMemcpyAsync(c->d_pts1, pts1, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
  float *dX0 = c->d_X, *dX1 = X1 ? c->d_X2 : nullptr;
  hipLaunchKernelGGL(svo_dlt_kernel, dim3((n + 63) / 64), dim3(64), 0, s, cam, c->d_pts0, c->d_pts1, n, dX0, dX1);
  VO_CHECK_HIP(c, hipGetLastError());
  VO_CHECK_HIP(c, hipMemcpyAsync(X0, dX0, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, s));
  if (X1) VO_CHECK_HIP(c, hipMemcpyAsync(X1, dX1, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, s));
  VO_CHECK_HIP(c, hipStreamSynchronize(s));
  return VO_OK;
}

// ---- host: the driver -------------------------------------------------------------------------------------------------
static int svo_alloc_ts(vo_ctx *c, SvoTrackSet *t, int cap) {
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->pts_l, sizeof(float) * 2 * cap));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->pts_r, sizeof(float) * 2 * cap));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->Xw, sizeof(float) * 3 * cap));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->flags, (size_t)cap));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->ids, sizeof(int32_t) * cap));
  return VO_OK;
}
static void svo_free_ts(SvoTrackSet *t) {
  void *b[] = {t->pts_l, t->pts_r, t->Xw, t->flags, t->ids};
  for (void *p : b)
    if (p) (void)hipFree(p);
  memset(t, 0, sizeof(*t));
}

extern "C" int vo_svo_create(vo_ctx *c, const vo_svo_params *prm, vo_svo **out) {
  if (!c || !prm || !out) return VO_ERR_INVALID;
  *out = nullptr;
  if (c->cfg.n_slots < 5) VO_FAIL(c, VO_ERR_INVALID, "StereoVO needs a context with at least 5 image slots");
  if (!vo_frame_fused_supported(prm->frame.win))
    VO_FAIL(c, VO_ERR_INVALID, "StereoVO needs a window the fused frame kernel is built for (13, 15, 21, 31)");
  const int bins = prm->bins.n_bins_u * prm->bins.n_bins_v;
  if (bins <= 0 || bins > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "%d bins exceed vo_config.max_points=%d", bins, c->cfg.max_points);
  if (prm->kf_window < 1) VO_FAIL(c, VO_ERR_INVALID, "kf_window must be at least 1");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_svo *s = new vo_svo();
  s->c = c;
  s->prm = *prm;
  s->cap = c->cfg.max_points;
  int rc = VO_OK;
  for (int k = 0; k < 2 && rc == VO_OK; ++k) rc = svo_alloc_ts(c, &s->ts[k], s->cap);
  if (rc == VO_OK && vo_dev_malloc(c, (void **)&s->d_accept, (size_t)s->cap) != hipSuccess) rc = VO_ERR_HIP;
  if (rc == VO_OK && vo_dev_malloc(c, (void **)&s->d_acc_bin, sizeof(int) * (size_t)s->cap) != hipSuccess) rc = VO_ERR_HIP;
  if (rc == VO_OK && vo_dev_malloc(c, (void **)&s->d_hdr, sizeof(SvoHdr)) != hipSuccess) rc = VO_ERR_HIP;
  if (rc == VO_OK && vo_host_malloc(c, (void **)&s->h_hdr, sizeof(SvoHdr), hipHostMallocDefault) != hipSuccess) rc = VO_ERR_HIP;
  if (rc != VO_OK) {
    vo_svo_destroy(s);
    VO_FAIL(c, rc, "StereoVO: device allocation failed");
  }
  memset(s->h_hdr, 0, sizeof(SvoHdr));
  float T_rl[16];
  svo_inv_se3(prm->frame.T_lr, T_rl);
  memcpy(s->T_rl, T_rl, sizeof(T_rl));
  svo_make_cam(T_rl, prm->frame.Kl, prm->frame.Kr, &s->cam);
  for (int i = 0; i < 16; ++i) s->T_wp[i] = s->dT01[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  s->kf_rot = prm->kf_rotation_deg * (float)(3.14159265358979323846 / 180.0);  // thres_rotation * D2R
  s->first = true;
  for (int k = 0; k < 5; ++k) s->slot[k] = k;
  // the keyframes' device storage (landmark table, keyframe ring, the local BA's scratch and arena) is made here, so that a
  // failure surfaces at construction and no keyframe ever allocates
  rc = vo_svo_lba_init(s);
  if (rc >= 0) rc = vo_stereo_frame_set_strict_border(c, prm->strict_border);
  if (rc >= 0) rc = vo_set_pyramid_window_hint(c, prm->frame.win);
  if (rc >= 0) rc = vo_set_ingest_side_stream(c, 1);
  if (rc < 0) {  // (the context's error text is the failing call's)
    vo_svo_destroy(s);
    return rc;
  }
  *out = s;
  return VO_OK;
}

extern "C" void vo_svo_destroy(vo_svo *s) {
  if (!s) return;
  if (s->c) (void)hipSetDevice(s->c->device);
  if (s->c && s->c->frame) memset(&s->c->frame->adv_next, 0, sizeof(s->c->frame->adv_next));  // (points into the track sets freed below)
  for (int k = 0; k < 2; ++k) svo_free_ts(&s->ts[k]);
  if (s->d_accept) (void)hipFree(s->d_accept);
  if (s->d_acc_bin) (void)hipFree(s->d_acc_bin);
  if (s->d_hdr) (void)hipFree(s->d_hdr);
  if (s->h_hdr) (void)hipHostFree(s->h_hdr);
  vo_svo_lba_free(s);
  delete s;
}

enum { S_P = 0, S_CL = 1, S_CR = 2, S_NL = 3, S_NR = 4 };

// VO_SVO_TRACE=1: where the host's time goes per frame (vo_svo::ht, averages on stderr every 200 frames)
static double svo_now();
static const bool g_trace = getenv("VO_SVO_TRACE") != nullptr;  // (read once, never written: the accumulators are per StereoVO)

// the pair into the "next" slots + its candidate table (side stream); detect = false: the images and pyramids only (the
// synchronous call defers the detection to the frame's enqueue, where it runs next to the features' tracking)
static int svo_ingest(vo_svo *s, const void *left, const void *right, int stride, int on_device, bool detect = true) {
  vo_ctx *c = s->c;
  const int W = s->prm.frame.width, H = s->prm.frame.height;
  // A pair that comes WITH its frame (no look-ahead: detect == false) is ingested on the main stream: the frame kernel, which
  // is enqueued right behind it, then needs no cross-queue wait for the pyramids (an event pair between two queues costs
  // ~15-30 us on this stack); the detector on the side stream is the one that waits for the slot's event — it has the slack.
  struct IngestHere {
    vo_ctx *c;
    int keep;
    IngestHere(vo_ctx *ctx, bool main_stream) : c(ctx), keep(ctx->ingest_side) {
      if (main_stream) c->ingest_side = 0;
    }
    ~IngestHere() { c->ingest_side = keep; }
  } here(c, !detect);
  if (s->prm.rectify) {
    // flagDoUndistortion (stereo_vo.cpp:414-421): rectifyStereoImages + convertTo(CV_8UC1), fused into the pyramid build
    if (on_device) {
      RC(vo_set_stereo_pair_rectified_device(c, s->slot[S_NL], left, s->slot[S_NR], right, W, H, stride));
    } else {
      RC(vo_set_image_rectified(c, s->slot[S_NL], (const uint8_t *)left, W, H, stride, 0));
      RC(vo_set_image_rectified(c, s->slot[S_NR], (const uint8_t *)right, W, H, stride, 1));
    }
  } else if (on_device) {
    RC(vo_set_stereo_pair_device(c, s->slot[S_NL], left, s->slot[S_NR], right, W, H, stride));
  } else {
    RC(vo_set_stereo_pair_host_async(c, s->slot[S_NL], (const uint8_t *)left, s->slot[S_NR], (const uint8_t *)right, W, H, stride));
  }
  if (detect) RC(vo_new_point_candidates_enqueue(c, s->slot[S_NL], &s->prm.bins, s->tab_next));
  return VO_OK;
}

extern "C" int vo_svo_prefetch(vo_svo *s, const void *left, const void *right, int stride, int on_device) {
  if (!s || !left || !right) return VO_ERR_INVALID;
  const double t_in = g_trace ? svo_now() : 0.0;
  VO_CHECK_HIP(s->c, hipSetDevice(s->c->device));
  RC(svo_ingest(s, left, right, stride, on_device));
  if (g_trace) s->ht.acc[2] += svo_now() - t_in;
  s->pre_l = left;
  s->pre_r = right;
  s->prefetched = true;
  return VO_OK;
}

static int svo_first_frame(vo_svo *s);

extern "C" int vo_svo_enqueue(vo_svo *s, const void *left, const void *right, int stride, int on_device, double timestamp) {
  if (!s || !left || !right) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "a frame is already in flight: call vo_svo_result first");
  const double t_in = g_trace ? svo_now() : 0.0;
  if (g_trace && s->ht.t_ret > 0) s->ht.acc[0] += t_in - s->ht.t_ret;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  (void)timestamp;
  // what can be refused is refused before any state changes (an empty track set is the reference's throw at :626)
  if (!s->first && s->n <= 0) VO_FAIL(c, VO_ERR_GN_FAILED, "the track set is empty: PoseOnlyStereoBA is failed!");
  // the pair goes into the NEXT slots and the next candidate table: nothing the loop reads changes if this fails
  // No pair handed over early (trackStereoImages as the reference's caller uses it): the images and pyramids go in now, the
  // keypoint detection is deferred into the frame's enqueue, where it runs on the side stream NEXT TO the features' tracking
  // and the candidates follow as a launch of their own (frame_pipeline.hip: vo_frame_set_deferred_detection)
  bool deferred = false, issued = false;
  if (!(s->prefetched && s->pre_l == left && s->pre_r == right)) {
    deferred = !s->first && s->c->ingest_side;
    if (deferred && !s->prm.rectify && s->n > 0 && vo_orb_cand_table(c, s->tab_next)) {
      // the detector needs the left IMAGE, not its pyramid: it starts now, on the side stream — from the caller's device image at
      // once, from the slot's staging plane as soon as a host image's upload is queued — while the main stream builds the pyramids
      if (on_device) {
        const int rc = vo_new_point_candidates_enqueue_image(c, (const uint8_t *)left, stride, s->prm.frame.width, s->prm.frame.height,
                                                             &s->prm.bins, s->tab_next);
        if (rc < 0) return rc;
        issued = rc == VO_OK;
      } else {
        c->early_bins = &s->prm.bins;
        c->early_table = s->tab_next;
        c->early_issued = 0;
      }
    }
    const int rc_in = svo_ingest(s, left, right, stride, on_device, !deferred);
    if (c->early_bins) {
      issued = c->early_issued != 0;
      c->early_bins = nullptr;
    }
    if (rc_in < 0) {
      if (issued && on_device) (void)hipStreamSynchronize(c->stream2);  // (as in undo below)
      return rc_in;
    }
  }
  // from here on the driver's state moves; every error return below puts it back (slots, tables, both id counters), so
  // that the caller can hand the pair over again — or another one — and track against the right previous image
  struct {
    int slot[5], tab_cur, tab_next, frame_id;
    int32_t next_frame_id, next_landmark_id;
  } keep;
  memcpy(keep.slot, s->slot, sizeof(keep.slot));
  keep.tab_cur = s->tab_cur;
  keep.tab_next = s->tab_next;
  keep.frame_id = s->frame_id;
  keep.next_frame_id = c->next_frame_id;
  keep.next_landmark_id = c->next_landmark_id;
  auto undo = [&](int rc) {
    memcpy(s->slot, keep.slot, sizeof(keep.slot));
    s->tab_cur = keep.tab_cur;
    s->tab_next = keep.tab_next;
    s->frame_id = keep.frame_id;
    c->next_frame_id = keep.next_frame_id;
    c->next_landmark_id = keep.next_landmark_id;
    s->pending = false;
    s->pending_first = false;
    if (issued) (void)hipStreamSynchronize(c->stream2);  // (the detector reads the CALLER's image: it has finished before we say "refused")
    return rc;
  };
  s->prefetched = false;
  {  // prev <- curr (stereo_vo.cpp:983-985), curr <- the pair just ingested; the two freed slots take the next pair
    const int p = s->slot[S_P], cl = s->slot[S_CL], cr = s->slot[S_CR], nl = s->slot[S_NL], nr = s->slot[S_NR];
    s->slot[S_P] = cl;
    s->slot[S_CL] = nl;
    s->slot[S_CR] = nr;
    s->slot[S_NL] = p;
    s->slot[S_NR] = cr;
    s->tab_cur = s->tab_next;
    s->tab_next ^= 1;
  }
  // StereoFrame(cam_left, cam_right, timestamp): two Frame ids, left first (frame.cpp:176-180)
  s->frame_id = c->next_frame_id;
  c->next_frame_id += 2;
  if (s->first) {
    const int rc = svo_first_frame(s);
    if (rc < 0) return undo(rc);
    s->pending = true;
    s->pending_first = true;
    return VO_OK;
  }
  // [2] T_wc_prior = T_wp * dT_pc_prev; T_cw_prior = inverseSE3_f(T_wc_prior); T_pw = getPoseInv() (stereo_vo.cpp:475-480)
  float T_wc_prior[16], T_cw_prior[16], T_pw[16];
  svo_mul44(s->T_wp, s->dT01, T_wc_prior);
  svo_inv_se3(T_wc_prior, T_cw_prior);
  svo_inv_se3(s->T_wp, T_pw);
  const SvoTrackSet &t = s->ts[s->cur];
  {  // the BA launch of this frame leaves the next track set behind (gn_pose.hip: DLT workers + epilogue)
    VoAdvArgs a;
    memset(&a, 0, sizeof(a));
    a.cur = t;
    a.nxt = s->ts[s->cur ^ 1];
    a.acc_bin = s->d_acc_bin;
    a.accept = s->d_accept;
    a.cam = s->cam;
    a.id_base = c->next_landmark_id;
    a.cap = s->cap;
    a.hdr_dev = s->d_hdr;
    a.hdr_host = s->h_hdr;
    int rc = vo_frame_set_advance(c, &a);
    if (rc >= 0 && deferred) rc = vo_frame_set_deferred_detection(c, issued ? 1 : 0);
    if (rc < 0) return undo(rc);
  }
  int rc = vo_frame_enqueue_impl(c, &s->prm.frame, s->slot[S_P], s->slot[S_CL], s->slot[S_CR], t.pts_l, t.pts_r, t.Xw, t.flags,
                                 s->n, s->dT01, nullptr, 0, 1, &s->prm.bins, s->tab_cur, T_pw, T_cw_prior);
  if (rc < 0) return undo(rc);
  s->pending = true;
  s->pending_first = false;
  if (g_trace) s->ht.acc[1] += svo_now() - t_in;
  return VO_OK;
}

// stereo_vo.cpp:842-949 — once per stream, composed of the operators (synchronous)
static int svo_first_frame(vo_svo *s) {
  vo_ctx *c = s->c;
  const vo_svo_params &p = s->prm;
  const int bins = p.bins.n_bins_u * p.bins.n_bins_v;
  std::vector<float> xy(2 * (size_t)bins), cand, pr;
  std::vector<uint8_t> has(bins), m;
  // resetWeightBin + extractORBwithBinning_fast: every bin that holds a keypoint, bins ascending = the table
  RC(vo_new_point_candidates_get(c, s->tab_cur, xy.data(), has.data(), nullptr));
  for (int j = 0; j < bins; ++j)
    if (has[j]) {
      cand.push_back(xy[2 * j]);
      cand.push_back(xy[2 * j + 1]);
    }
  const int nc = (int)(cand.size() / 2);
  pr.assign(2 * (size_t)std::max(nc, 1), 0.f);
  m.assign((size_t)std::max(nc, 1), 1);
  if (nc > 0)
    RC(vo_track_bidirection(c, s->slot[S_CL], s->slot[S_CR], cand.data(), nc, p.frame.win, p.frame.max_level, p.frame.thres_err,
                            p.frame.thres_bidirection, pr.data(), m.data()));
  std::vector<float> pl2, pr2;
  for (int j = 0; j < nc; ++j)
    if (m[j]) {
      pl2.insert(pl2.end(), {cand[2 * j], cand[2 * j + 1]});
      pr2.insert(pr2.end(), {pr[2 * j], pr[2 * j + 1]});
    }
  const int n = (int)(pl2.size() / 2);
  if (n > s->cap) VO_FAIL(c, VO_ERR_CAPACITY, "%d initial landmarks exceed vo_config.max_points=%d", n, s->cap);
  std::vector<int32_t> ids(std::max(n, 1));
  for (int i = 0; i < n; ++i) ids[i] = c->next_landmark_id + i;  // Landmark(pt_left, frame): id = landmark_counter_++
  c->next_landmark_id += n;
  SvoTrackSet &t = s->ts[s->cur];
  hipStream_t st = c->stream;
  if (n > 0) {
    VO_CHECK_HIP(c, hipMemcpyAsync(t.pts_l, pl2.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.pts_r, pr2.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.ids, ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemsetAsync(t.Xw, 0, sizeof(float) * 3 * n, st));
    VO_CHECK_HIP(c, hipMemsetAsync(t.flags, 0, (size_t)n, st));
    SvoKfArgs k;
    memset(&k, 0, sizeof(k));
    k.ts = t;
    k.n_surv = n;
    k.n_all = n;
    k.cam = s->cam;
    memcpy(k.Kl, p.frame.Kl, sizeof(k.Kl));
    memcpy(k.Kr, p.frame.Kr, sizeof(k.Kr));
    k.has_T = 0;
    k.member = 0;  // the first frame is not a keyframe (no checkUpdateRule on this branch)
    hipLaunchKernelGGL(svo_keyframe_kernel, dim3((n + 255) / 256), dim3(256), 0, st, k);
    VO_CHECK_HIP(c, hipGetLastError());
  }
  VO_CHECK_HIP(c, hipStreamSynchronize(st));
  s->n = n;
  s->first_n_cand = nc;
  // stframe_curr->setStereoPoseByLeft(Identity, T_lr); setPoseDiff10(Identity) -> dT01_ = inverseSE3_f(Identity)
  float I[16];
  for (int i = 0; i < 16; ++i) I[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  memcpy(s->T_wp, I, sizeof(I));
  svo_inv_se3(I, s->dT01);
  s->first = false;
  return VO_OK;
}

static double svo_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// StereoKeyframes::checkUpdateRule, keyframes.cpp:217-303
static bool svo_keyframe_rule(const vo_svo *s, int n_tracked, const float T_wc[16]) {
  if (s->keyframes.empty()) return true;
  const float ratio = (float)n_tracked / (float)s->n_kf_lms;
  if (ratio <= s->prm.kf_overlap_ratio) return true;
  float T_kw[16], dT[16];
  svo_inv_se3(s->keyframes.back().T_wc, T_kw);  // getPoseInv() of the last keyframe (setPose: Tcw_ = inverseSE3_f(Twc_))
  svo_mul44(T_kw, T_wc, dT);
  float costheta = (((dT[0] + dT[5]) + dT[10]) - 1.0f) * 0.5f;
  if ((double)costheta >= 0.999999) costheta = (float)0.999999;
  if ((double)costheta <= -0.999999) costheta = (float)-0.999999;
  const float rot = acosf(costheta);
  const float dtrans = sqrtf(dT[3] * dT[3] + (dT[7] * dT[7] + dT[11] * dT[11]));
  return rot >= s->kf_rot || dtrans >= s->prm.kf_translation;
}

extern "C" int vo_svo_result(vo_svo *s, vo_svo_frame_info *info) {
  if (!s || !s->pending) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_svo_frame_info I;
  memset(&I, 0, sizeof(I));
  I.frame_id = s->frame_id;
  s->pending = false;
  if (s->pending_first) {
    I.is_first = 1;
    I.n_new = I.n_tracks_out = s->n;
    I.n_new_candidates = s->first_n_cand;
    memcpy(I.T_wc, s->T_wp, sizeof(I.T_wc));
    for (int i = 0; i < 16; ++i) I.dT[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (info) *info = I;
    return VO_OK;
  }
  // the BA launch's epilogue wrote the loop's counts (pinned block) in front of the frame's sequence word
  float dT[16];
  const double t_in = g_trace ? svo_now() : 0.0;
  int rc = vo_stereo_frame_result(c, nullptr, nullptr, nullptr, dT, nullptr, nullptr, &I.counts, &I.gn);
  if (rc < 0) return rc;
  const double t_seen = g_trace ? svo_now() : 0.0;
#ifdef GN_STAMP  // measurement build: phases of the BA launch in the loop (10 ns ticks), averaged under VO_SVO_TRACE
  {
    double *acc = s->gn_acc;
    int &nacc = s->gn_nacc;
    if (g_trace) {
      long long t[12];
      if (vo_debug_gn_stamps(c, t) == VO_OK) {
        const int order[9] = {0, 5, 1, 2, 3, 6, 7, 8, 4};  // start, joined, prologue, loads, iterations, stage+DLT wait, emit, track set, copy
        for (int k = 0; k < 8; ++k) acc[k] += 0.01 * (double)(t[order[k + 1]] - t[order[k]]);
        if (++nacc % 100 == 0)
          fprintf(stderr, "[gn] per launch (us): join wait %.1f  prologue %.1f  loads %.1f  iterations %.1f  stage marks + DLT wait %.1f  "
                          "step [10] emission %.1f  next track set %.1f  copy-out + sequence word %.1f\n",
                  acc[0] / nacc, acc[1] / nacc, acc[2] / nacc, acc[3] / nacc, acc[4] / nacc, acc[5] / nacc, acc[6] / nacc, acc[7] / nacc);
      }
    }
  }
#endif
  const SvoHdr h = *s->h_hdr;
  if (h.overflow) VO_FAIL(c, VO_ERR_CAPACITY, "the next track set (%d) exceeds vo_config.max_points=%d", h.n_next, s->cap);
  I.n_tracks_in = s->n;
  I.n_final = h.n_surv;
  I.n_new = h.n_new;
  I.n_tracks_out = h.n_next;
  I.n_kf_tracked = h.n_kf_tracked;
  I.n_new_candidates = h.n_emit;
  memcpy(I.dT, dT, sizeof(dT));
  c->next_landmark_id += h.n_new;
  // T_wc = T_wp * dT_pc_poBA (:640); setPoseDiff10(dT.inverse()) -> dT01_ = inverseSE3_f(dT10) (:643, frame.cpp:50-54)
  float T_wc[16], dT10[16];
  svo_mul44(s->T_wp, dT, T_wc);
  svo_inv44(dT, dT10);
  svo_inv_se3(dT10, s->dT01);
  const int nxt = s->cur ^ 1;
  if (svo_keyframe_rule(s, h.n_kf_tracked, T_wc)) {
    I.is_keyframe = 1;
    SvoKfArgs k;
    memset(&k, 0, sizeof(k));
    k.ts = s->ts[nxt];
    k.n_surv = h.n_surv;  // lmtrack_final.n_pts: the landmarks pushed in [10] are not reconstructed at this keyframe
    k.n_all = h.n_next;
    k.cam = s->cam;
    memcpy(k.Kl, s->prm.frame.Kl, sizeof(k.Kl));
    memcpy(k.Kr, s->prm.frame.Kr, sizeof(k.Kr));
    k.has_T = 1;
    k.member = 1;
    memcpy(k.T_wc, T_wc, sizeof(k.T_wc));
    if (h.n_next > 0) {
      hipLaunchKernelGGL(svo_keyframe_kernel, dim3((h.n_next + 255) / 256), dim3(256), 0, c->stream, k);
      VO_CHECK_HIP(c, hipGetLastError());
    }
    SvoKeyframe kf;
    kf.serial = s->n_keyframes++;
    kf.frame_id = s->frame_id;
    memcpy(kf.T_wc, T_wc, sizeof(T_wc));
    if ((int)s->keyframes.size() == s->prm.kf_window) s->keyframes.erase(s->keyframes.begin());
    s->keyframes.push_back(kf);
    s->n_kf_lms = h.n_next;
    {
      s->cur = nxt;  // (the keyframe bookkeeping and the local BA read and update the track set the next frame starts from)
      s->n = h.n_next;
      rc = vo_svo_local_ba(s, &I, h.id_min);
      if (rc < 0) return rc;
      memcpy(T_wc, s->keyframes.back().T_wc, sizeof(T_wc));
    }
  }
  s->cur = nxt;
  s->n = h.n_next;
  memcpy(s->T_wp, T_wc, sizeof(T_wc));
  memcpy(I.T_wc, T_wc, sizeof(T_wc));
  if (info) *info = I;
  if (g_trace) {
    s->ht.t_ret = svo_now();
    ++s->ht.n_all;
    if (!I.is_keyframe) {
      s->ht.acc[3] += t_seen - t_in;
      s->ht.acc[4] += s->ht.t_ret - t_seen;
      ++s->ht.n;
    }
    if (s->ht.n > 0 && s->ht.n % 200 == 0 && !I.is_keyframe)
      fprintf(stderr, "[svo host] per frame (us): caller between result and enqueue %.1f  enqueue %.1f  prefetch %.1f  waiting in result %.1f  "
                      "rest of result (ordinary frames) %.1f\n",
              1e6 * s->ht.acc[0] / s->ht.n_all, 1e6 * s->ht.acc[1] / s->ht.n_all, 1e6 * s->ht.acc[2] / s->ht.n_all, 1e6 * s->ht.acc[3] / s->ht.n,
              1e6 * s->ht.acc[4] / s->ht.n);
  }
  return VO_OK;
}

extern "C" int vo_svo_track(vo_svo *s, const void *left, const void *right, int stride, int on_device, double timestamp,
                            vo_svo_frame_info *info) {
  RC(vo_svo_enqueue(s, left, right, stride, on_device, timestamp));
  return vo_svo_result(s, info);
}

// A recorded sequence through the loop, driven from here (the reference's caller is compiled code too: a ROS node's
// callback; a caller that holds the whole sequence hands every pair over one frame early). Collects frames k_begin ..
// k_end - 1 of the n_total pairs: frame k's result, then at once vo_svo_enqueue(k + 1) and vo_svo_prefetch(k + 2) — the
// caller's per-frame work (here: one copy of the info block and a time stamp) runs under the next frame. k_begin == 0
// starts the sequence (nothing may be in flight); otherwise frame k_begin is the one a previous call left in flight; on
// return frame k_end is in flight (when there is one).
extern "C" int vo_svo_run(vo_svo *s, const void *const *left, const void *const *right, int n_total, int stride, int on_device,
                          int k_begin, int k_end, vo_svo_frame_info *infos, double *stamps) {
  if (!s || !left || !right || n_total <= 0 || k_begin < 0 || k_end > n_total || k_begin >= k_end) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (k_begin == 0) {
    if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "a frame is already in flight: call vo_svo_result first");
    RC(vo_svo_enqueue(s, left[0], right[0], stride, on_device, 0.0));
    if (n_total > 1) RC(vo_svo_prefetch(s, left[1], right[1], stride, on_device));
  } else if (!s->pending) {
    VO_FAIL(c, VO_ERR_INVALID, "vo_svo_run: frame %d is not in flight", k_begin);
  }
  vo_svo_frame_info info;
  for (int k = k_begin; k < k_end; ++k) {
    RC(vo_svo_result(s, &info));
    if (stamps) {
      timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      stamps[k - k_begin] = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    if (k + 1 < n_total) {
      RC(vo_svo_enqueue(s, left[k + 1], right[k + 1], stride, on_device, 0.1 * (k + 1)));
      if (k + 2 < n_total) RC(vo_svo_prefetch(s, left[k + 2], right[k + 2], stride, on_device));
    }
    if (infos) infos[k - k_begin] = info;
  }
  return VO_OK;
}

extern "C" int vo_svo_get_tracks(vo_svo *s, int32_t *ids, float *pts_l, float *pts_r, float *Xw, uint8_t *flags, int cap,
                                 int *n) {
  if (!s || !n) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_svo_result first");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  *n = s->n;
  if (s->n > cap && (ids || pts_l || pts_r || Xw || flags)) VO_FAIL(c, VO_ERR_CAPACITY, "%d tracks, room for %d", s->n, cap);
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  const SvoTrackSet &t = s->ts[s->cur];
  const size_t m = (size_t)s->n;
  if (m == 0) return VO_OK;
  if (ids) VO_CHECK_HIP(c, hipMemcpy(ids, t.ids, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
  if (pts_l) VO_CHECK_HIP(c, hipMemcpy(pts_l, t.pts_l, sizeof(float) * 2 * m, hipMemcpyDeviceToHost));
  if (pts_r) VO_CHECK_HIP(c, hipMemcpy(pts_r, t.pts_r, sizeof(float) * 2 * m, hipMemcpyDeviceToHost));
  if (Xw) VO_CHECK_HIP(c, hipMemcpy(Xw, t.Xw, sizeof(float) * 3 * m, hipMemcpyDeviceToHost));
  if (flags) VO_CHECK_HIP(c, hipMemcpy(flags, t.flags, m, hipMemcpyDeviceToHost));
  return VO_OK;
}

extern "C" int vo_svo_get_new_points(vo_svo *s, float *pts_l, float *pts_r, uint8_t *mask_new, uint8_t *accept, int *n) {
  if (!s || !n) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_svo_result first");
  if (!c->frame || !c->frame->closed) VO_FAIL(c, VO_ERR_INVALID, "no steady-state frame has run yet");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_frame_state *f = c->frame;
  const int m = s->h_hdr->n_emit;
  *n = m;
  if (m <= 0) return VO_OK;
  if (pts_l) memcpy(pts_l, f->res_host + f->off_newl, sizeof(float) * 2 * (size_t)m);
  if (pts_r) memcpy(pts_r, f->res_host + f->off_newr, sizeof(float) * 2 * (size_t)m);
  if (mask_new) memcpy(mask_new, f->res_host + f->off_mnew, (size_t)m);
  if (accept) VO_CHECK_HIP(c, hipMemcpy(accept, s->d_accept, (size_t)m, hipMemcpyDeviceToHost));
  return VO_OK;
}
