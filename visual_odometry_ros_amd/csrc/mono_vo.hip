// mono_vo.hip — the closed loop of MonoVO::trackImage around the mono frame operator, track set on the device.
//
// Reference (paths relative to the reference repository root):
//   core/visual_odometry/mono_vo/mono_vo.cpp:496-1194   trackImage
//     :528-561    the first image: extract, one landmark per pixel, pose I, setPoseDiff10(T_init), T_init.t = (0, 0, -1)
//     :562-696    the second image: FeatureTracker::track, 5-point pose (OpenCV calib3d: a CALLER HOOK), Sampson
//                 gate, |dt10| = 1, new points back-tracked into I0, reconstruction of landmarks with enough parallax
//     :698-1019   steady state                                      -> frame_mono.hip / gn_pose.hip (vo_mono_frame_*)
//     :909-949    the 5-point fallback                              -> the same hook, host-driven operator calls
//     :1022-1157  keyframe rule, addNewKeyframe, reconstruction of landmarks seen on more than two keyframes, local BA
//   core/visual_odometry/landmark.cpp:76-135   addObservationAndRelatedFrame: age, parallax w.r.t. the oldest observation
//   core/visual_odometry/keyframes.cpp:30-126  addNewKeyframe, checkUpdateRule
//   core/visual_odometry/motion_estimator.cpp:1090-1205 + ba_solver/sparse_ba_parameters.h:292-466 (mono mode) +
//   ba_solver/sparse_bundle_adjustment.cpp:624-722                   -> stereo_vo_lba.hip (one observation per keyframe) + sba.hip
//
// Data on the device. Two track sets (current / next), one entry per landmark of frame_prev_: the pixel seen, the id, the
// world point, flags (triangulated | dead | member of the last keyframe | bundled), and what Landmark keeps for its own
// decisions: the FIRST observation (pixel + frame), the age, the parallax of the newest observation (as its cosine: the
// reference compares acosf(c) with a threshold — the host turns that threshold into the largest float c that passes
// its own acosf, so the device never evaluates acosf), and the first observation on a keyframe + their number. A ring of
// frame poses (T_wc and T_cw, 2^14 frames) serves "related_frames_.front()->getPoseInv()"; the poses of window keyframes
// are refreshed in it after every local BA. The advance step builds the next set behind every frame (survivors in index
// order, then the new landmarks: `mvo_advance_scan_kernel` + `mvo_advance_work_kernel`), `mvo_keyframe_kernel` does addNewKeyframe + reconstruction, the landmark table / keyframe
// ring / local BA are stereo_vo_lba.hip's in mono mode. The host chains the pose, applies the keyframe rule and calls
// the hook.
#include "frame_state.hpp"
#include "vo_kernels.hpp"

#include <math.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <vector>

#include "mvo_device.hpp"
#include "stereo_vo.hpp"

int vo_mono_frame_set_advance(vo_ctx *c, const MvoAdvArgs *adv);  // frame_mono.hip
int vo_frame_set_deferred_detection(vo_ctx *c, int issued);         // frame_pipeline.hip
int vo_mono_frame_set_track_flags(vo_ctx *c, int mode);

#define RC(x)                \
  do {                       \
    int _rc = (x);           \
    if (_rc < 0) return _rc; \
  } while (0)

// mapping::triangulateDLT for one camera with T10 = T1w * Tw0 (mono_vo.cpp:672-678, :1046-1052); true + Xworld = Tw0 * X0 when
// the landmark is reconstructed (keyframe_rule: both reprojections within 1 px and both depths positive, :1054-1073;
// otherwise X0(2) > 0, :680)
__device__ bool mvo_reconstruct(float p0x, float p0y, float p1x, float p1y, const float *Tw0, const float *T1w, const float K[4],
                                bool keyframe_rule, float (&Xw)[3]) {
  float T10[12];
  mvo_mul34(T1w, Tw0, T10);
  SvoCam cam;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) cam.R10[i * 3 + j] = T10[i * 4 + j];
    cam.t10[i] = T10[i * 4 + 3];
  }
  const float Km[9] = {K[0], 0.0f, K[2], 0.0f, K[1], K[3], 0.0f, 0.0f, 1.0f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      cam.P10[i * 4 + j] = mvo_dot3(Km[i * 3 + 0], cam.R10[0 * 3 + j], Km[i * 3 + 1], cam.R10[1 * 3 + j], Km[i * 3 + 2], cam.R10[2 * 3 + j]);
    cam.P10[i * 4 + 3] = mvo_dot3(Km[i * 3 + 0], cam.t10[0], Km[i * 3 + 1], cam.t10[1], Km[i * 3 + 2], cam.t10[2]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) cam.K0[k] = cam.K1[k] = K[k];
  float X0[3], X1[3];
  svo_triangulate(cam, p0x, p0y, p1x, p1y, X0, X1);
  if (keyframe_rule) {
    float iz = 1.0f / X0[2];
    float dx = p0x - (K[0] * X0[0] * iz + K[2]), dy = p0y - (K[1] * X0[1] * iz + K[3]);
    if (dx * dx + dy * dy > 1.0f) return false;
    iz = 1.0f / X1[2];
    dx = p1x - (K[0] * X1[0] * iz + K[2]);
    dy = p1y - (K[1] * X1[1] * iz + K[3]);
    if (dx * dx + dy * dy > 1.0f) return false;
    if (!(X0[2] > 0 && X1[2] > 0)) return false;
  } else if (!(X0[2] > 0)) {
    return false;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) Xw[r] = (Tw0[r * 4 + 0] * X0[0] + (Tw0[r * 4 + 1] * X0[1] + Tw0[r * 4 + 2] * X0[2])) + Tw0[r * 4 + 3];
  return true;
}

// The advance step as launches of its own (initialisation, 5-point fallback, VO_DBG_MVO_HOST_ADVANCE; a steady-state frame
// has it in its BA launch's epilogue: mvo_advance_body, mvo_device.hpp). Two launches: the ORDER (one workgroup: which entries
// survive and where they go — a scan — the counts and the sequence word the host waits for) and the WORK (one lane per entry,
// any number of workgroups: the copies and the parallaxes, nothing serial).
__global__ __launch_bounds__(1024) void mvo_advance_scan_kernel(MvoAdvArgs a) {
  __shared__ int s_w[16];
  __shared__ int s_kft[16];
  __shared__ int s_idmin, s_old;
  const int tid = threadIdx.x;
  if (tid == 0) s_old = 0;
  if (tid < 16) {  // the frame table's entry of this frame (T_wc | T_cw)
    a.frameT[(size_t)(a.f & (MVO_FRAME_RING - 1)) * 32 + tid] = a.T_wc[tid];
    a.frameT[(size_t)(a.f & (MVO_FRAME_RING - 1)) * 32 + 16 + tid] = a.T_cw[tid];
  }
  if (tid == 0) s_idmin = a.id_base;
  __syncthreads();
  int base = 0, kft = 0;
  for (int c0 = 0; c0 < a.n; c0 += 1024) {
    const int k = c0 + tid;
    const int ok = (k < a.n && a.stage[k] == 4) ? 1 : 0;
    int total;
    const int pos = base + mvo_block_scan<16>(ok, s_w, total);
    if (k < a.n) a.pos_s[k] = (ok && pos < a.cap) ? pos : -1;
    if (ok) {
      kft += (a.cur.t.flags[k] & VO_LM_KF_MEMBER) ? 1 : 0;
      if (pos == 0) s_idmin = a.cur.t.ids[k];
      if (a.f - a.cur.f_first[k] >= MVO_FRAME_RING) s_old = 1;  // (as mvo_advance_body)
    }
    base += total;
  }
  const int n_surv = base;
  for (int c0 = 0; c0 < a.m; c0 += 1024) {
    const int j = c0 + tid;
    const int ok = (j < a.m && a.mnew[j]) ? 1 : 0;
    int total;
    const int r = base + mvo_block_scan<16>(ok, s_w, total);
    if (j < a.m) a.pos_n[j] = (ok && r < a.cap) ? r : -1;
    base += total;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) kft += __shfl_down(kft, off);
  if ((tid & 63) == 0) s_kft[tid >> 6] = kft;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += s_kft[k];
    MvoHdr h;
    h.n_surv = n_surv;
    h.n_new = base - n_surv;
    h.n_next = base < a.cap ? base : a.cap;
    h.n_kf_tracked = t;
    h.overflow = (base > a.cap ? 1 : 0) | (s_old ? 2 : 0);
    h.n_recon = 0;
    h.pad = 0;
    h.seq = 0;
    h.id_min = s_idmin;  // (the next set's first id; the first new id when nothing survived)
    *a.hdr_dev = h;
    *a.hdr_host = h;
    __threadfence_system();
    __hip_atomic_store(&a.hdr_host->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void mvo_advance_work_kernel(MvoAdvArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < a.n) {
    const int pos = a.pos_s[g];
    if (pos >= 0) {
      float T[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) T[k] = a.T_obs[k];
      mvo_put_survivor(a, g, pos, a.pts1[2 * g], a.pts1[2 * g + 1], T);
    }
    return;
  }
  const int j = g - a.n;
  if (j >= a.m) return;
  const int r = a.pos_n[j];
  if (r >= 0) {
    float T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = a.T_wc[k];
    mvo_put_new(a, r, a.id_base + (r - a.hdr_dev->n_surv), a.cand0[2 * j], a.cand0[2 * j + 1], a.cand1[2 * j], a.cand1[2 * j + 1], T);
  }
}

// reconstruction at the initialisation (mono_vo.cpp:660-687): every landmark of lmtrack_final that is not triangulated and
// whose newest observation has enough parallax — from its first and last observation and their frames' poses
struct MvoRecArgs {
  MvoSet s;
  int n, f;
  float K[4], T_cw[16], cos_thres;
  const float *frameT;
  int *n_recon;
};
__global__ void mvo_init_reconstruct_kernel(MvoRecArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const uint8_t fl = a.s.t.flags[k];
  if ((fl & VO_LM_TRIANGULATED) || !(a.s.cos_last[k] <= a.cos_thres)) return;
  float Xw[3];
  const float *Tw0 = a.frameT + (size_t)(a.s.f_first[k] & (MVO_FRAME_RING - 1)) * 32;
  if (mvo_reconstruct(a.s.p_first[2 * k], a.s.p_first[2 * k + 1], a.s.t.pts_l[2 * k], a.s.t.pts_l[2 * k + 1], Tw0, a.T_cw, a.K, false, Xw)) {
    a.s.t.Xw[3 * k] = Xw[0];
    a.s.t.Xw[3 * k + 1] = Xw[1];
    a.s.t.Xw[3 * k + 2] = Xw[2];
    a.s.t.flags[k] = fl | VO_LM_TRIANGULATED;
    if (a.n_recon) atomicAdd(a.n_recon, 1);
  }
}

// a new keyframe (keyframes.cpp:30-45 addNewKeyframe: every related landmark gets the observation on it; mono_vo.cpp:
// 1032-1076: landmarks that are alive, not triangulated, with enough parallax and seen on MORE than two keyframes are
// reconstructed from their first and last keyframe observation)
__global__ void mvo_keyframe_kernel(MvoRecArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  uint8_t fl = a.s.t.flags[k] | VO_LM_KF_MEMBER;
  const int nk = a.s.n_kf[k] + 1;
  a.s.n_kf[k] = nk;
  const float px = a.s.t.pts_l[2 * k], py = a.s.t.pts_l[2 * k + 1];
  float p0x = a.s.p_kf_first[2 * k], p0y = a.s.p_kf_first[2 * k + 1];
  int f0 = a.s.kf_first[k];
  if (nk == 1) {
    a.s.p_kf_first[2 * k] = p0x = px;
    a.s.p_kf_first[2 * k + 1] = p0y = py;
    a.s.kf_first[k] = f0 = a.f;
  }
  if (!(fl & VO_LM_DROPPED) && !(fl & VO_LM_TRIANGULATED) && a.s.cos_last[k] <= a.cos_thres && nk > 2) {
    float Xw[3];
    // (the first keyframe's CURRENT pose: the local BA may have moved it; frameT is refreshed after every solve)
    const float *Tw0 = a.frameT + (size_t)(f0 & (MVO_FRAME_RING - 1)) * 32;
    if (mvo_reconstruct(p0x, p0y, px, py, Tw0, a.T_cw, a.K, true, Xw)) {
      a.s.t.Xw[3 * k] = Xw[0];
      a.s.t.Xw[3 * k + 1] = Xw[1];
      a.s.t.Xw[3 * k + 2] = Xw[2];
      fl |= VO_LM_TRIANGULATED;
      if (a.n_recon) atomicAdd(a.n_recon, 1);
    }
  }
  a.s.t.flags[k] = fl;
}

// the local BA moved keyframes: their entries of the frame table
struct MvoPoseArgs {
  int n, f[16];
  float T[16][32];
};
__global__ void mvo_pose_put_kernel(MvoPoseArgs a, float *frameT) {
  const int j = blockIdx.x, t = threadIdx.x;
  if (j < a.n && t < 32) frameT[(size_t)(a.f[j] & (MVO_FRAME_RING - 1)) * 32 + t] = a.T[j][t];
}

// ---- host -------------------------------------------------------------------------------------------------------------
struct vo_mvo {
  vo_ctx *c = nullptr;
  vo_mvo_params prm;
  int cap = 0;
  vo_svo core;  // keyframe window, landmark table, keyframe ring, local BA (stereo_vo_lba.hip in mono mode)
  MvoSet ts[2] = {};
  int cur = 0, n = 0;
  float *d_frameT = nullptr;
  MvoHdr *d_hdr = nullptr, *h_hdr = nullptr;
  uint32_t seq = 0;
  int *d_nrec = nullptr;
  int *d_pos = nullptr;  // [2 cap] the advance step's scratch
  // uploads of the host-driven paths (initialisation, 5-point fallback)
  uint8_t *d_stage = nullptr, *d_mnew = nullptr;
  float *d_pts1 = nullptr, *d_cand1 = nullptr, *d_cand0 = nullptr;
  bool got_first = false, init_done = false, pending = false, prefetched = false;
  const void *pre = nullptr;
  int pend_kind = 0;  // 0 first image, 1 initialisation, 2 steady state
  bool chained = false;  // the advance step of the frame in flight is part of its BA launch
  int slot[3] = {0, 1, 2};  // previous, current, next
  int tab_cur = 0, tab_next = 0;
  int f = -1;        // index of the current frame (0, 1, ...) — Frame ids come from the context's counter
  int frame_id = 0;
  float T_wp[16], dT01[16];  // frame_prev_: pose, getPoseDiff01()
  std::vector<int> kf_frame;  // frame index of every window keyframe (parallel to core.keyframes)
  float cos_thres = -2.0f;
  vo_mvo_frame_info info;
};

extern void svo_mul44(const float A[16], const float B[16], float C[16]);
extern void svo_inv_se3(const float T[16], float Ti[16]);

static void mvo_eye(float T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}
static inline float dot3e(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }
// F10 = Kinv^T [t10]x R10 Kinv (motion_estimator.cpp:551-552), Eigen's product order (as mono_gate.hpp)
static void mvo_fundamental(const float K[4], const float R10[9], const float t10[3], float F[9]) {
  const float fxi = 1.0f / K[0], fyi = 1.0f / K[1];
  const float Kinv[9] = {fxi, 0.0f, -K[2] * fxi, 0.0f, fyi, -K[3] * fyi, 0.0f, 0.0f, 1.0f};
  const float KinvT[9] = {Kinv[0], Kinv[3], Kinv[6], Kinv[1], Kinv[4], Kinv[7], Kinv[2], Kinv[5], Kinv[8]};
  const float Sx[9] = {0.0f, -t10[2], t10[1], t10[2], 0.0f, -t10[0], -t10[1], t10[0], 0.0f};
  float E[9], T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) E[i * 3 + j] = dot3e(Sx[i * 3 + 0], R10[0 * 3 + j], Sx[i * 3 + 1], R10[1 * 3 + j], Sx[i * 3 + 2], R10[2 * 3 + j]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = dot3e(KinvT[i * 3 + 0], E[0 * 3 + j], KinvT[i * 3 + 1], E[1 * 3 + j], KinvT[i * 3 + 2], E[2 * 3 + j]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) F[i * 3 + j] = dot3e(T[i * 3 + 0], Kinv[0 * 3 + j], T[i * 3 + 1], Kinv[1 * 3 + j], T[i * 3 + 2], Kinv[2 * 3 + j]);
}
// the largest float c with acosf(c) >= thres (this host's acosf, the one the reference's comparison would use): the device
// compares cosines. acosf falls monotonically, so a bisection over the ordered float patterns finds it.
static float mvo_cos_threshold(float thres) {
  if (!(thres > 0.0f)) return 1.0f;
  auto key = [](float v) {
    int32_t b;
    memcpy(&b, &v, 4);
    return b < 0 ? (int64_t)INT32_MIN - (int64_t)b : (int64_t)b;  // monotone in v
  };
  auto unkey = [](int64_t k) {
    int32_t b = k < 0 ? (int32_t)((int64_t)INT32_MIN - k) : (int32_t)k;
    float v;
    memcpy(&v, &b, 4);
    return v;
  };
  int64_t lo = key(-1.0f), hi = key(1.0f);  // acosf(lo) = pi >= thres (thres <= pi), acosf(hi) = 0 < thres
  if (!(acosf(-1.0f) >= thres)) return -2.0f;
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (acosf(unkey(mid)) >= thres)
      lo = mid;
    else
      hi = mid;
  }
  return unkey(lo);
}

static int mvo_alloc_set(vo_ctx *c, MvoSet *t, int cap) {
  const size_t n = (size_t)cap;
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->t.pts_l, sizeof(float) * 2 * n));
  t->t.pts_r = t->t.pts_l;
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->t.Xw, sizeof(float) * 3 * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->t.flags, n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->t.ids, sizeof(int32_t) * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->p_first, sizeof(float) * 2 * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->f_first, sizeof(int32_t) * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->age, sizeof(int32_t) * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->cos_last, sizeof(float) * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->n_kf, sizeof(int32_t) * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->p_kf_first, sizeof(float) * 2 * n));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->kf_first, sizeof(int32_t) * n));
  return VO_OK;
}
static void mvo_free_set(MvoSet *t) {
  void *b[] = {t->t.pts_l, t->t.Xw, t->t.flags, t->t.ids, t->p_first, t->f_first, t->age, t->cos_last, t->n_kf, t->p_kf_first, t->kf_first};
  for (void *p : b)
    if (p) (void)hipFree(p);
  memset(t, 0, sizeof(*t));
}

extern "C" void vo_mvo_destroy(vo_mvo *s) {
  if (!s) return;
  if (s->c) (void)hipSetDevice(s->c->device);
  if (s->c) (void)hipStreamSynchronize(s->c->stream);
  for (int k = 0; k < 2; ++k) mvo_free_set(&s->ts[k]);
  void *b[] = {s->d_frameT, s->d_hdr, s->d_nrec, s->d_pos, s->d_stage, s->d_mnew, s->d_pts1, s->d_cand1, s->d_cand0};
  for (void *p : b)
    if (p) (void)hipFree(p);
  if (s->h_hdr) (void)hipHostFree(s->h_hdr);
  vo_svo_lba_free(&s->core);
  delete s;
}

extern "C" int vo_mvo_create(vo_ctx *c, const vo_mvo_params *prm, vo_mvo **out) {
  if (!c || !prm || !out) return VO_ERR_INVALID;
  *out = nullptr;
  if (c->cfg.n_slots < 3) VO_FAIL(c, VO_ERR_INVALID, "MonoVO needs a context with at least 3 image slots");
  if (!prm->five_point) VO_FAIL(c, VO_ERR_INVALID, "MonoVO needs the 5-point pose hook (calcPose5PointsAlgorithm is the caller's)");
  const int w = prm->frame.win;
  if (w != 13 && w != 15 && w != 21 && w != 31) VO_FAIL(c, VO_ERR_INVALID, "MonoVO needs a window the mono frame kernel is built for (13, 15, 21, 31)");
  const int bins = prm->bins.n_bins_u * prm->bins.n_bins_v;
  if (bins <= 0 || bins > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "%d bins exceed vo_config.max_points=%d", bins, c->cfg.max_points);
  if (prm->kf_window < 1) VO_FAIL(c, VO_ERR_INVALID, "kf_window must be at least 1");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_mvo *s = new vo_mvo();
  s->c = c;
  s->prm = *prm;
  s->cap = c->cfg.max_points;
  memset(&s->info, 0, sizeof(s->info));
  int rc = VO_OK;
  for (int k = 0; k < 2 && rc == VO_OK; ++k) rc = mvo_alloc_set(c, &s->ts[k], s->cap);
  auto dm = [&](void **p, size_t bytes) {
    if (rc == VO_OK && vo_dev_malloc(c, p, bytes) != hipSuccess) rc = VO_ERR_HIP;
  };
  dm((void **)&s->d_frameT, sizeof(float) * 32 * (size_t)MVO_FRAME_RING);
  dm((void **)&s->d_hdr, sizeof(MvoHdr));
  dm((void **)&s->d_nrec, 64);
  dm((void **)&s->d_pos, sizeof(int) * 2 * (size_t)s->cap);
  dm((void **)&s->d_stage, (size_t)s->cap);
  dm((void **)&s->d_mnew, (size_t)s->cap);
  dm((void **)&s->d_pts1, sizeof(float) * 2 * (size_t)s->cap);
  dm((void **)&s->d_cand1, sizeof(float) * 2 * (size_t)s->cap);
  dm((void **)&s->d_cand0, sizeof(float) * 2 * (size_t)s->cap);
  if (rc == VO_OK && vo_host_malloc(c, (void **)&s->h_hdr, sizeof(MvoHdr), hipHostMallocDefault) != hipSuccess) rc = VO_ERR_HIP;
  if (rc == VO_OK) {
    memset(s->h_hdr, 0, sizeof(MvoHdr));
    // the keyframe storage: a vo_svo in mono mode (one observation per keyframe entry, bundled flags)
    vo_svo &k = s->core;
    k.c = c;
    memset(&k.prm, 0, sizeof(k.prm));
    k.prm.kf_window = prm->kf_window;
    k.prm.local_ba = prm->local_ba;
    for (int i = 0; i < 4; ++i) k.prm.frame.Kl[i] = k.prm.frame.Kr[i] = prm->frame.K[i];
    mvo_eye(k.prm.frame.T_lr);
    k.cap = s->cap;
    k.mono = 1;
    k.ts[0] = s->ts[0].t;
    k.ts[1] = s->ts[1].t;
    rc = vo_svo_lba_init(&k);
  }
  if (rc >= 0) rc = vo_stereo_frame_set_strict_border(c, prm->strict_border);
  if (rc >= 0) rc = vo_set_pyramid_window_hint(c, prm->frame.win);
  if (rc >= 0) rc = vo_set_ingest_side_stream(c, 1);
  if (rc < 0) {
    if (rc == VO_ERR_HIP && !c->err[0]) snprintf(c->err, sizeof(c->err), "MonoVO: device allocation failed");
    vo_mvo_destroy(s);
    return rc;
  }
  mvo_eye(s->T_wp);
  mvo_eye(s->dT01);
  const float thres = prm->thres_parallax_deg * (float)(3.14159265358979323846 / 180.0);  // map_update.thres_parallax * D2R
  s->cos_thres = mvo_cos_threshold(thres);
  *out = s;
  return VO_OK;
}

// detect = false (MonoVO's synchronous call in the steady state): the image and its pyramid only, on the MAIN stream — the frame
// kernel behind it needs no cross-queue wait — and the detection is the frame enqueue's (vo_frame_set_deferred_detection)
static int mvo_ingest(vo_mvo *s, const void *img, int stride, int on_device, bool detect = true) {
  vo_ctx *c = s->c;
  const int W = s->prm.frame.width, H = s->prm.frame.height, slot = s->slot[2];
  struct IngestHere {
    vo_ctx *c;
    int keep;
    IngestHere(vo_ctx *ctx, bool main_stream) : c(ctx), keep(ctx->ingest_side) {
      if (main_stream) c->ingest_side = 0;
    }
    ~IngestHere() { c->ingest_side = keep; }
  } here(c, !detect);
  if (s->prm.rectify) {  // flagDoUndistortion (mono_vo.cpp:509-513): undistortImage + convertTo(CV_8UC1), fused into the pyramid build
    if (on_device)
      RC(vo_set_image_rectified_device(c, slot, img, W, H, stride, 0));
    else
      RC(vo_set_image_rectified(c, slot, (const uint8_t *)img, W, H, stride, 0));
  } else if (on_device) {
    RC(vo_set_image_device(c, slot, img, W, H, stride));
  } else {
    RC(vo_set_image_host_async(c, slot, (const uint8_t *)img, W, H, stride));
  }
  if (detect) RC(vo_new_point_candidates_enqueue(c, slot, &s->prm.bins, s->tab_next));
  return VO_OK;
}

extern "C" int vo_mvo_prefetch(vo_mvo *s, const void *img, int stride, int on_device) {
  if (!s || !img) return VO_ERR_INVALID;
  VO_CHECK_HIP(s->c, hipSetDevice(s->c->device));
  RC(mvo_ingest(s, img, stride, on_device));
  s->pre = img;
  s->prefetched = true;
  return VO_OK;
}

// wait for the advance kernel's counts (pinned block, sequence word last)
static int mvo_wait_hdr(vo_mvo *s) {
  vo_ctx *c = s->c;
  volatile const uint32_t *seqp = &s->h_hdr->seq;
  timespec p0;
  clock_gettime(CLOCK_MONOTONIC, &p0);
  for (int spin = 0;; ++spin) {
    if (*seqp == s->seq) break;
    if ((spin & 255) == 255) {
      timespec p1;
      clock_gettime(CLOCK_MONOTONIC, &p1);
      if ((p1.tv_sec - p0.tv_sec) * 1e9 + (p1.tv_nsec - p0.tv_nsec) > 2e7) {
        VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        if (*seqp != s->seq) VO_FAIL(c, VO_ERR_HIP, "MonoVO: the track-set advance of the frame did not report (sequence word %u, expected %u)", *seqp, s->seq);
        break;
      }
    }
    if (c->dbg[VO_OPT_POLL_YIELD]) sched_yield();
#if defined(__x86_64__)
    else __builtin_ia32_pause();
#endif
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return VO_OK;
}

// the next track set behind a frame; stage / pts1 / new points are DEVICE arrays. What does not depend on who runs the step:
static void mvo_advance_args(vo_mvo *s, MvoAdvArgs *a, const uint8_t *stage, const float *pts1, const float *cand1, const float *cand0,
                             const uint8_t *mnew, int m, int f) {
  memset(a, 0, sizeof(*a));
  a->cur = s->ts[s->cur];
  a->nxt = s->ts[s->cur ^ 1];
  a->n = s->n;
  a->cap = s->cap;
  a->stage = stage;
  a->pts1 = pts1;
  a->cand1 = cand1;
  a->cand0 = cand0;
  a->mnew = mnew;
  a->m = m;
  a->id_base = s->c->next_landmark_id;
  a->f = f;
  memcpy(a->K, s->prm.frame.K, sizeof(a->K));
  a->frameT = s->d_frameT;
  a->hdr_dev = s->d_hdr;
  a->hdr_host = s->h_hdr;
  s->seq = s->seq + 1 == 0 ? 1 : s->seq + 1;
  a->seq = s->seq;
  a->pos_s = s->d_pos;
  a->pos_n = s->d_pos + s->cap;
}
// ... as launches of its own (initialisation, 5-point fallback). T_obs: what the survivors' observation sees as the frame's pose.
static int mvo_advance_launch(vo_mvo *s, const uint8_t *stage, const float *pts1, const float *cand1, const float *cand0,
                              const uint8_t *mnew, int m, const float *T_obs, const float *T_wc) {
  vo_ctx *c = s->c;
  MvoAdvArgs a;
  mvo_advance_args(s, &a, stage, pts1, cand1, cand0, mnew, m, s->f);
  memcpy(a.T_obs, T_obs, sizeof(a.T_obs));
  memcpy(a.T_wc, T_wc, sizeof(a.T_wc));
  svo_inv_se3(T_wc, a.T_cw);
  hipLaunchKernelGGL(mvo_advance_scan_kernel, dim3(1), dim3(1024), 0, c->stream, a);
  hipLaunchKernelGGL(mvo_advance_work_kernel, dim3((unsigned)((s->n + m + 255) / 256 + 1)), dim3(256), 0, c->stream, a);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
// ... and its counts: leaves cur / n on the new set
static int mvo_advance_collect(vo_mvo *s, MvoHdr *h) {
  vo_ctx *c = s->c;
  RC(mvo_wait_hdr(s));
  *h = *s->h_hdr;
  if (h->pad) return VO_OK;  // (inside the BA launch of a frame the host has to finish: nothing was built)
  if (h->overflow & 2)
    VO_FAIL(c, VO_ERR_CAPACITY, "a landmark has been tracked for %d frames or more: the pose of its first observation has left the frame-pose ring",
            MVO_FRAME_RING);
  if (h->overflow) VO_FAIL(c, VO_ERR_CAPACITY, "the next track set exceeds vo_config.max_points=%d", s->cap);
  c->next_landmark_id += h->n_new;
  s->cur ^= 1;
  s->n = h->n_next;
  return VO_OK;
}
static int mvo_advance(vo_mvo *s, const uint8_t *stage, const float *pts1, const float *cand1, const float *cand0, const uint8_t *mnew,
                       int m, const float T_obs[16], const float T_wc[16], MvoHdr *h) {
  RC(mvo_advance_launch(s, stage, pts1, cand1, cand0, mnew, m, T_obs, T_wc));
  return mvo_advance_collect(s, h);
}

// Keyframes::checkUpdateRule, keyframes.cpp:47-126
static bool mvo_keyframe_rule(const vo_mvo *s, int n_tracked, const float T_wc[16]) {
  const vo_svo &k = s->core;
  if (k.keyframes.empty()) return true;
  const float ratio = (float)n_tracked / (float)k.n_kf_lms;
  if (ratio <= s->prm.kf_overlap_ratio) return true;
  float T_kw[16], dT[16];
  svo_inv_se3(k.keyframes.back().T_wc, T_kw);
  svo_mul44(T_kw, T_wc, dT);
  float costheta = (((dT[0] + dT[5]) + dT[10]) - 1.0f) * 0.5f;
  if (costheta >= 0.999999f) costheta = 0.999999f;
  if (costheta <= -0.999999f) costheta = -0.999999f;
  const float rot = acosf(costheta);
  const float dtrans = sqrtf(dT[3] * dT[3] + (dT[7] * dT[7] + dT[11] * dT[11]));
  const float kf_rot = s->prm.kf_rotation_deg * (float)(3.14159265358979323846 / 180.0);
  return rot >= kf_rot || dtrans >= s->prm.kf_translation;
}

// the end of trackImage (mono_vo.cpp:1021-1163): keyframe rule, keyframe work, frame_prev_ <- frame_curr
static int mvo_finish_frame(vo_mvo *s, const MvoHdr &h, float T_wc[16], vo_mvo_frame_info *I) {
  vo_ctx *c = s->c;
  vo_svo &k = s->core;
  hipStream_t st = c->stream;
  const int n = s->n;
  I->n_kf_tracked = h.n_kf_tracked;
  if (mvo_keyframe_rule(s, h.n_kf_tracked, T_wc)) {
    I->is_keyframe = 1;
    MvoRecArgs a;
    memset(&a, 0, sizeof(a));
    a.s = s->ts[s->cur];
    a.n = n;
    a.f = s->f;
    memcpy(a.K, s->prm.frame.K, sizeof(a.K));
    svo_inv_se3(T_wc, a.T_cw);
    a.cos_thres = s->cos_thres;
    a.frameT = s->d_frameT;
    a.n_recon = nullptr;
    if (n > 0) {
      hipLaunchKernelGGL(mvo_keyframe_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a);
      VO_CHECK_HIP(c, hipGetLastError());
    }
    SvoKeyframe kf;
    kf.serial = k.n_keyframes++;
    kf.frame_id = s->frame_id;
    memcpy(kf.T_wc, T_wc, sizeof(kf.T_wc));
    if ((int)k.keyframes.size() == s->prm.kf_window) {
      k.keyframes.erase(k.keyframes.begin());
      s->kf_frame.erase(s->kf_frame.begin());
    }
    k.keyframes.push_back(kf);
    s->kf_frame.push_back(s->f);
    k.n_kf_lms = n;
    k.cur = s->cur;
    k.n = n;
    vo_svo_frame_info li;
    memset(&li, 0, sizeof(li));
    const int rc = vo_svo_local_ba(&k, &li, h.id_min);
    if (rc < 0) return rc;
    if (li.lba_ran) {
      I->lba_ran = 1;
      I->lba_err_first = li.lba_err_first;
      I->lba_err_last = li.lba_err_last;
      I->lba_landmarks = li.lba_landmarks;
      I->lba_observations = li.lba_observations;
      // kf->setPose(...) of the optimised keyframes: their entries of the frame table, and this frame's pose
      MvoPoseArgs p;
      memset(&p, 0, sizeof(p));
      for (size_t j = 2; j < k.keyframes.size() && p.n < 16; ++j) {
        p.f[p.n] = s->kf_frame[j];
        memcpy(p.T[p.n], k.keyframes[j].T_wc, sizeof(float) * 16);
        svo_inv_se3(k.keyframes[j].T_wc, p.T[p.n] + 16);
        ++p.n;
      }
      if (p.n > 0) {
        hipLaunchKernelGGL(mvo_pose_put_kernel, dim3(p.n), dim3(32), 0, st, p, s->d_frameT);
        VO_CHECK_HIP(c, hipGetLastError());
      }
      memcpy(T_wc, k.keyframes.back().T_wc, sizeof(float) * 16);
    }
  }
  memcpy(s->T_wp, T_wc, sizeof(float) * 16);
  memcpy(I->T_wc, T_wc, sizeof(float) * 16);
  I->n_tracks_out = n;
  return VO_OK;
}

// new points driven from the host (initialisation, 5-point fallback): updateWeightBin(final pixels), the table's best keypoint
// of every bin left empty, trackBidirection(I1, I0) — uploaded for the advance kernel. Returns the number of candidates.
static int mvo_host_new_points(vo_mvo *s, const std::vector<float> &final_px, int *m_out) {
  vo_ctx *c = s->c;
  const vo_bin_params &b = s->prm.bins;
  const int bins = b.n_bins_u * b.n_bins_v;
  std::vector<float> xy(2 * (size_t)bins), cand, p0;
  std::vector<uint8_t> has(bins), occ(bins, 0), m;
  RC(vo_new_point_candidates_get(c, s->tab_cur, xy.data(), has.data(), nullptr));
  for (size_t i = 0; i + 1 < final_px.size(); i += 2) {  // WeightBin::update, feature_extractor.h:116-135
    const int u = (int)floorf(final_px[i] / (float)b.u_step), v = (int)floorf(final_px[i + 1] / (float)b.v_step);
    const int bin = v * b.n_bins_u + u;
    if (bin >= 0 && bin < bins) occ[bin] = 1;
  }
  for (int j = 0; j < bins; ++j)
    if (has[j] && !occ[j]) {
      cand.push_back(xy[2 * j]);
      cand.push_back(xy[2 * j + 1]);
    }
  const int nc = (int)(cand.size() / 2);
  *m_out = nc;
  if (nc == 0) return VO_OK;
  p0.assign(2 * (size_t)nc, 0.f);
  m.assign((size_t)nc, 1);
  RC(vo_track_bidirection(c, s->slot[1], s->slot[0], cand.data(), nc, s->prm.frame.win, s->prm.frame.max_level, s->prm.frame.thres_err,
                          s->prm.frame.thres_bidirection, p0.data(), m.data()));
  hipStream_t st = c->stream;
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_cand1, cand.data(), sizeof(float) * 2 * nc, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_cand0, p0.data(), sizeof(float) * 2 * nc, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_mnew, m.data(), (size_t)nc, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipStreamSynchronize(st));  // (the vectors go out of scope)
  return VO_OK;
}

// mono_vo.cpp:528-561
static int mvo_first_image(vo_mvo *s, vo_mvo_frame_info *I) {
  vo_ctx *c = s->c;
  const int bins = s->prm.bins.n_bins_u * s->prm.bins.n_bins_v;
  std::vector<float> xy(2 * (size_t)bins), pts;
  std::vector<uint8_t> has(bins);
  RC(vo_new_point_candidates_get(c, s->tab_cur, xy.data(), has.data(), nullptr));  // resetWeightBin + extractORBwithBinning_fast
  for (int j = 0; j < bins; ++j)
    if (has[j]) {
      pts.push_back(xy[2 * j]);
      pts.push_back(xy[2 * j + 1]);
    }
  const int n = (int)(pts.size() / 2);
  if (n > s->cap) VO_FAIL(c, VO_ERR_CAPACITY, "%d initial landmarks exceed vo_config.max_points=%d", n, s->cap);
  MvoSet &t = s->ts[s->cur];
  hipStream_t st = c->stream;
  float I4[16];
  mvo_eye(I4);
  if (n > 0) {
    std::vector<int32_t> ids(n), zero(n, 0), age(n, 1), neg(n, -1);
    std::vector<float> cosn(n, MVO_COS_NONE);
    for (int i = 0; i < n; ++i) ids[i] = c->next_landmark_id + i;
    c->next_landmark_id += n;
    VO_CHECK_HIP(c, hipMemcpyAsync(t.t.pts_l, pts.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.p_first, pts.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.t.ids, ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemsetAsync(t.t.Xw, 0, sizeof(float) * 3 * n, st));
    VO_CHECK_HIP(c, hipMemsetAsync(t.t.flags, 0, (size_t)n, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.f_first, zero.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.age, age.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.cos_last, cosn.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemsetAsync(t.n_kf, 0, sizeof(int32_t) * n, st));
    VO_CHECK_HIP(c, hipMemsetAsync(t.p_kf_first, 0, sizeof(float) * 2 * n, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.kf_first, neg.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipStreamSynchronize(st));
  }
  {  // the frame table's entry 0: pose I
    MvoPoseArgs p;
    memset(&p, 0, sizeof(p));
    p.n = 1;
    p.f[0] = 0;
    memcpy(p.T[0], I4, sizeof(I4));
    memcpy(p.T[0] + 16, I4, sizeof(I4));
    hipLaunchKernelGGL(mvo_pose_put_kernel, dim3(1), dim3(32), 0, st, p, s->d_frameT);
    VO_CHECK_HIP(c, hipGetLastError());
  }
  s->n = n;
  // frame_curr->setPose(Identity); setPoseDiff10(T_init), T_init.t = (0, 0, -1) -> dT01_ = inverseSE3_f(T_init)
  float T_init[16];
  mvo_eye(T_init);
  T_init[11] = -1.0f;
  svo_inv_se3(T_init, s->dT01);
  s->got_first = true;
  I->is_first = 1;
  I->n_new = n;
  MvoHdr h;
  memset(&h, 0, sizeof(h));
  h.id_min = n > 0 ? c->next_landmark_id - n : c->next_landmark_id;
  return mvo_finish_frame(s, h, I4, I);
}

// mono_vo.cpp:562-696 — host-driven, once per stream
static int mvo_second_image(vo_mvo *s, vo_mvo_frame_info *I) {
  vo_ctx *c = s->c;
  const vo_mono_params &p = s->prm.frame;
  const int n = s->n;
  if (n <= 0) VO_FAIL(c, VO_ERR_GN_FAILED, "MonoVO initialisation: the first image gave no landmark");
  hipStream_t st = c->stream;
  std::vector<float> pts0(2 * (size_t)n), pts1(2 * (size_t)n, 0.f), p0k, p1k;
  std::vector<uint8_t> mask(n, 1), stage(n, 0), m5;
  VO_CHECK_HIP(c, hipMemcpyAsync(pts0.data(), s->ts[s->cur].t.pts_l, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, st));
  VO_CHECK_HIP(c, hipStreamSynchronize(st));
  RC(vo_track(c, s->slot[0], s->slot[1], pts0.data(), n, p.win, p.max_level, p.thres_err, pts1.data(), mask.data()));
  std::vector<int> idx;
  for (int i = 0; i < n; ++i)
    if (mask[i]) {
      idx.push_back(i);
      p0k.insert(p0k.end(), {pts0[2 * i], pts0[2 * i + 1]});
      p1k.insert(p1k.end(), {pts1[2 * i], pts1[2 * i + 1]});
    }
  const int nk = (int)idx.size();
  float R10[9], t10[3];
  m5.assign((size_t)std::max(nk, 1), 0);
  if (!s->prm.five_point(s->prm.five_point_user, p0k.data(), p1k.data(), nk, p.K, R10, t10, m5.data()))
    VO_FAIL(c, VO_ERR_GN_FAILED, "calcPose5PointsAlgorithm() is failed.");
  float F[9];
  mvo_fundamental(p.K, R10, t10, F);
  std::vector<float> dist((size_t)std::max(nk, 1)), fin;
  if (nk > 0) RC(vo_sampson_distance(c, p0k.data(), p1k.data(), nk, F, dist.data()));

  for (int q = 0; q < nk; ++q)
    if (m5[q] && dist[q] < p.thres_sampson) {
      stage[idx[q]] = 4;
      fin.insert(fin.end(), {p1k[2 * q], p1k[2 * q + 1]});

    }
  // dt10 = dt10 / dt10.norm() * 1.0f; dT10; dT01 = inverseSE3_f(dT10); pose = Twc_prev * dT01 (:606-612)
  const float nrm = sqrtf(t10[0] * t10[0] + (t10[1] * t10[1] + t10[2] * t10[2]));
  float dT10[16], dT01[16], T_wc[16], I4[16];
  mvo_eye(dT10);
  mvo_eye(I4);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) dT10[i * 4 + j] = R10[i * 3 + j];
    dT10[i * 4 + 3] = t10[i] / nrm * 1.0f;
  }
  svo_inv_se3(dT10, dT01);
  svo_mul44(s->T_wp, dT01, T_wc);
  svo_inv_se3(dT10, s->dT01);  // setPoseDiff10(dT10)
  int m = 0;
  RC(mvo_host_new_points(s, fin, &m));
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_stage, stage.data(), (size_t)n, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_pts1, pts1.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipStreamSynchronize(st));
  MvoHdr h;
  // (the observations of lmtrack_final are added before the frame's pose is set, :602-603 vs :611: they see the identity)
  RC(mvo_advance(s, s->d_stage, s->d_pts1, s->d_cand1, s->d_cand0, s->d_mnew, m, I4, T_wc, &h));
  {  // :660-687
    MvoRecArgs a;
    memset(&a, 0, sizeof(a));
    a.s = s->ts[s->cur];
    a.n = s->n;
    a.f = s->f;
    memcpy(a.K, p.K, sizeof(a.K));
    svo_inv_se3(T_wc, a.T_cw);
    a.cos_thres = s->cos_thres;
    a.frameT = s->d_frameT;
    a.n_recon = s->d_nrec;
    VO_CHECK_HIP(c, hipMemsetAsync(s->d_nrec, 0, sizeof(int), st));
    if (s->n > 0) hipLaunchKernelGGL(mvo_init_reconstruct_kernel, dim3((s->n + 255) / 256), dim3(256), 0, st, a);
    VO_CHECK_HIP(c, hipGetLastError());
    RC(vo_svo_lba_update_points(&s->core, s->ts[s->cur].t, s->n));  // (landmarks of keyframe 0 that just got their point)
    int nrec = 0;
    VO_CHECK_HIP(c, hipMemcpyAsync(&nrec, s->d_nrec, sizeof(int), hipMemcpyDeviceToHost, st));
    VO_CHECK_HIP(c, hipStreamSynchronize(st));
    I->n_reconstructed = nrec;
  }
  s->init_done = true;
  I->is_init = 1;
  I->used_five_point = 1;
  I->n_tracks_in = n;
  I->n_final = h.n_surv;
  I->n_new = h.n_new;
  memcpy(I->dT01, dT01, sizeof(dT01));
  return mvo_finish_frame(s, h, T_wc, I);
}

extern "C" int vo_mvo_enqueue(vo_mvo *s, const void *img, int stride, int on_device, double timestamp) {
  if (!s || !img) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "a frame is already in flight: call vo_mvo_result first");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  (void)timestamp;
  if (s->init_done && s->n <= 0) VO_FAIL(c, VO_ERR_GN_FAILED, "the track set is empty");
  // No image handed over early (trackImage as the reference's caller uses it), steady state: the image and its pyramid go in now
  // on the main stream; the keypoint detection runs on the side stream NEXT TO the features' tracking — from the caller's device
  // image it starts at once, before the pyramid is queued — and the candidates follow as a launch of their own (frame_mono.hip)
  int deferred = 0;
  if (!(s->prefetched && s->pre == img)) {
    const bool defer = s->init_done && s->n > 0 && c->ingest_side;
    if (defer) {
      deferred = 1;
      if (!s->prm.rectify && vo_orb_cand_table(c, s->tab_next)) {
        if (on_device) {
          const int rc = vo_new_point_candidates_enqueue_image(c, (const uint8_t *)img, stride, s->prm.frame.width, s->prm.frame.height,
                                                               &s->prm.bins, s->tab_next);
          if (rc < 0) return rc;
          if (rc == VO_OK) deferred = 2;
        } else {  // (a host image: behind its upload, vo_set_image_host_async)
          c->early_bins = &s->prm.bins;
          c->early_table = s->tab_next;
          c->early_issued = 0;
        }
      }
    }
    {
      const int rc_in = mvo_ingest(s, img, stride, on_device, !defer);
      if (c->early_bins) {
        if (c->early_issued) deferred = 2;
        c->early_bins = nullptr;
      }
      if (rc_in < 0) {
        if (deferred == 2) (void)hipStreamSynchronize(c->stream2);
        return rc_in;
      }
    }
  }
  // from here on the driver's state moves; every error return below puts ALL of it back (slots, tables, the advance step's
  // sequence number, the prefetch mark — as vo_svo_enqueue does), so that the caller can hand the image over again
  const int keep_slot[3] = {s->slot[0], s->slot[1], s->slot[2]}, keep_tc = s->tab_cur, keep_tn = s->tab_next;
  const uint32_t keep_seq = s->seq;
  const bool keep_pre = s->prefetched, keep_chained = s->chained;
  auto undo = [&](int rc) {
    memcpy(s->slot, keep_slot, sizeof(keep_slot));
    s->tab_cur = keep_tc;
    s->tab_next = keep_tn;
    s->seq = keep_seq;
    s->prefetched = keep_pre;
    s->chained = keep_chained;
    if (deferred == 2) (void)hipStreamSynchronize(c->stream2);  // (the detector reads the CALLER's image: finished before "refused")
    return rc;
  };
  s->prefetched = false;
  {  // previous <- current, current <- the image just ingested, the freed slot takes the next one
    const int p = s->slot[0];
    s->slot[0] = s->slot[1];
    s->slot[1] = s->slot[2];
    s->slot[2] = p;
    s->tab_cur = s->tab_next;
    s->tab_next ^= 1;
  }
  if (s->init_done) {
    // mono_vo.cpp:726-734: Twc_prev, Tcw_prev, dT01_prior = getPoseDiff01(), Tcw_prior = inverseSE3_f(Twc_prev * dT01_prior)
    float Tcw_prev[16], Twc_prior[16], Tcw_prior[16];
    svo_inv_se3(s->T_wp, Tcw_prev);
    svo_mul44(s->T_wp, s->dT01, Twc_prior);
    svo_inv_se3(Twc_prior, Tcw_prior);
    const MvoSet &t = s->ts[s->cur];
    s->chained = false;
    if (!c->dbg[VO_DBG_MVO_HOST_ADVANCE]) {
      // the next track set is the last thing the frame's BA launch does (mvo_advance_body): the pixels, stages and new points
      // are the launch's own (mono_gate.hpp fills them in), the pose is formed there from this one — vo_mvo_result waits for
      // the step's counts only, one host round trip per frame and no launch behind the BA launch
      MvoAdvArgs a;
      mvo_advance_args(s, &a, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s->f + 1);
      memcpy(a.T_wp, s->T_wp, sizeof(a.T_wp));
      s->chained = vo_mono_frame_set_advance(c, &a) >= 0;
    }
    // the operator's flag byte (bit 0 lm->isBundled(): prior and scale from the 3-D point; bit 1: the class the pose-only BA
    // takes — bundled with more than five window keyframes, triangulated otherwise, mono_vo.cpp:800-826; bit 2: dead,
    // landmark.cpp:251) is read off the track set's flags by the frame kernel itself
    int rc = vo_mono_frame_set_track_flags(c, s->core.keyframes.size() > 5 ? 2 : 1);
    if (rc >= 0 && deferred) rc = vo_frame_set_deferred_detection(c, deferred == 2 ? 1 : 0);
    if (rc < 0) return undo(rc);
    rc = vo_mono_frame_enqueue_closed(c, &s->prm.frame, s->slot[0], s->slot[1], t.t.pts_l, t.t.Xw, t.t.flags, s->n, Tcw_prev, Tcw_prior,
                                      s->dT01, &s->prm.bins, s->tab_cur, 1);
    if (rc < 0) return undo(rc);
  }
  s->frame_id = c->next_frame_id;  // Frame(cam, timestamp): id = frame_counter_++ (frame.cpp:22-41)
  c->next_frame_id += 1;
  ++s->f;
  s->pend_kind = !s->got_first ? 0 : (!s->init_done ? 1 : 2);
  if (!s->init_done) s->chained = false;
  s->pending = true;
  return VO_OK;
}

// mono_vo.cpp:909-949: the pose-only BA gave nothing — 5-point pose with the previous motion's length, the hook's mask as
// mask_motion, Sampson gate, new points: host-driven operator calls on the frame's device results
static int mvo_fallback(vo_mvo *s, float dT01[16], float dT10[16], int *m_new, vo_mvo_frame_info *I) {
  vo_ctx *c = s->c;
  // (a frame whose candidates were a launch of their own and whose BA gave nothing did not join them: they may still run)
  if (c->frame && c->frame->mono_split) VO_CHECK_HIP(c, hipStreamSynchronize(c->stream2));
  const vo_mono_params &p = s->prm.frame;
  vo_frame_state *f = c->frame;
  const int n = s->n;
  hipStream_t st = c->stream;
  std::vector<float> pts0(2 * (size_t)n), pts1(2 * (size_t)n), p0k, p1k;
  std::vector<uint8_t> stage(n);
  VO_CHECK_HIP(c, hipMemcpyAsync(pts0.data(), s->ts[s->cur].t.pts_l, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, st));
  VO_CHECK_HIP(c, hipStreamSynchronize(st));
  memcpy(pts1.data(), f->res_host + f->off_pl1, sizeof(float) * 2 * (size_t)n);
  memcpy(stage.data(), f->res_host + f->off_stage, (size_t)n);
  std::vector<int> idx;
  for (int i = 0; i < n; ++i)
    if (stage[i] >= 2) {
      idx.push_back(i);
      p0k.insert(p0k.end(), {pts0[2 * i], pts0[2 * i + 1]});
      p1k.insert(p1k.end(), {pts1[2 * i], pts1[2 * i + 1]});
    }
  const int nk = (int)idx.size();
  float R10[9], t10[3];
  std::vector<uint8_t> mm((size_t)std::max(nk, 1), 0);
  if (!s->prm.five_point(s->prm.five_point_user, p0k.data(), p1k.data(), nk, p.K, R10, t10, mm.data()))
    VO_FAIL(c, VO_ERR_GN_FAILED, "'calcPose5PointsAlgorithm()' is failed. Terminate the algorithm.");
  const float scale = sqrtf(s->dT01[3] * s->dT01[3] + (s->dT01[7] * s->dT01[7] + s->dT01[11] * s->dT01[11]));
  const float nrm = sqrtf(t10[0] * t10[0] + (t10[1] * t10[1] + t10[2] * t10[2]));
  mvo_eye(dT10);
  const float sc = scale / nrm;
  float ts[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) dT10[i * 4 + j] = R10[i * 3 + j];
    dT10[i * 4 + 3] = ts[i] = sc * t10[i];
  }
  svo_inv_se3(dT10, dT01);
  // lmtrack_motion = lmtrack_scaleok[mask_motion]; Sampson gate with dT10 (:951-963)
  std::vector<float> q0, q1, fin;
  std::vector<int> qi;
  for (int q = 0; q < nk; ++q)
    if (mm[q]) {
      qi.push_back(idx[q]);
      q0.insert(q0.end(), {p0k[2 * q], p0k[2 * q + 1]});
      q1.insert(q1.end(), {p1k[2 * q], p1k[2 * q + 1]});
    }
  float F[9];
  mvo_fundamental(p.K, R10, ts, F);
  std::vector<float> dist((size_t)std::max((int)qi.size(), 1));
  if (!qi.empty()) RC(vo_sampson_distance(c, q0.data(), q1.data(), (int)qi.size(), F, dist.data()));
  std::fill(stage.begin(), stage.end(), 0);
  for (size_t q = 0; q < qi.size(); ++q)
    if (dist[q] < p.thres_sampson) {
      stage[qi[q]] = 4;
      fin.insert(fin.end(), {q1[2 * q], q1[2 * q + 1]});
    }
  RC(mvo_host_new_points(s, fin, m_new));
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_stage, stage.data(), (size_t)n, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipMemcpyAsync(s->d_pts1, pts1.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  VO_CHECK_HIP(c, hipStreamSynchronize(st));
  I->used_five_point = 1;
  return VO_OK;
}

// An error return of vo_mvo_result ENDS THE STREAM, as the reference's throw ends the node (mono_vo.cpp:909-949 "Terminate the
// algorithm"; include/vo_hip.h says so): the frame is no longer in flight, track set and pose are where the failing step left
// them, and the next call reports "not in flight". A refused vo_mvo_enqueue, by contrast, leaves everything as it was.
extern "C" int vo_mvo_result(vo_mvo *s, vo_mvo_frame_info *info) {
  if (!s || !s->pending) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  s->pending = false;
  vo_mvo_frame_info I;
  memset(&I, 0, sizeof(I));
  I.frame_id = s->frame_id;
  int rc = VO_OK;
  if (s->pend_kind == 0) {
    rc = mvo_first_image(s, &I);
  } else if (s->pend_kind == 1) {
    rc = mvo_second_image(s, &I);
  } else {
    float dT01[16], dT10[16], T_wc[16];
    MvoHdr h;
    memset(&h, 0, sizeof(h));
    if (s->chained) {  // (the frame's own block is complete by then: the advance is the BA launch's last step)
      rc = mvo_advance_collect(s, &h);
      if (rc < 0) {
        (void)vo_mono_frame_result(c, nullptr, nullptr, nullptr, dT01, &I.counts, &I.gn);
        return rc;
      }
      c->frame->known_done = true;  // (the word just seen is the last store of the frame's last launch)
    }
    rc = vo_mono_frame_result(c, nullptr, nullptr, nullptr, dT01, &I.counts, &I.gn);
    if (rc < 0) return rc;
    vo_frame_state *f = c->frame;
    const vo_frame_hdr *fh = (const vo_frame_hdr *)f->res_host;
    I.n_tracks_in = s->n;
    const uint8_t *stage = f->res_dev + f->off_stage, *mnew = f->res_dev + f->off_mnew;
    const float *pts1 = (const float *)(f->res_dev + f->off_pl1);
    const float *cand1 = (const float *)(f->res_dev + f->off_newl), *cand0 = (const float *)(f->res_dev + f->off_newr);
    int m = fh->cnt[6];
    if (I.counts.need_five_point) {
      rc = mvo_fallback(s, dT01, dT10, &m, &I);
      if (rc < 0) return rc;
      stage = s->d_stage;
      pts1 = s->d_pts1;
      cand1 = s->d_cand1;
      cand0 = s->d_cand0;
      mnew = s->d_mnew;
    } else {
      svo_inv_se3(dT01, dT10);  // dT10 = inverseSE3_f(dT01) (:881)
    }
    svo_mul44(s->T_wp, dT01, T_wc);  // frame_curr->setPose(Twc_prev * dT01)
    svo_inv_se3(dT10, s->dT01);      // setPoseDiff10(dT10): dT01_ = inverseSE3_f(dT10)
    if (!s->chained || h.pad) {
      rc = mvo_advance(s, stage, pts1, cand1, cand0, mnew, m, T_wc, T_wc, &h);
      if (rc < 0) return rc;
    }
    I.n_final = h.n_surv;
    I.n_new = h.n_new;
    memcpy(I.dT01, dT01, sizeof(dT01));
    rc = mvo_finish_frame(s, h, T_wc, &I);
  }
  if (rc < 0) return rc;
  s->info = I;
  if (info) *info = I;
  return VO_OK;
}

extern "C" int vo_mvo_track(vo_mvo *s, const void *img, int stride, int on_device, double timestamp, vo_mvo_frame_info *info) {
  RC(vo_mvo_enqueue(s, img, stride, on_device, timestamp));
  return vo_mvo_result(s, info);
}

// A recorded sequence through the loop, driven from here (vo_svo_run's twin): collects frames k_begin .. k_end - 1 of the
// n_total images — vo_mvo_result(k), then at once vo_mvo_enqueue(k + 1) and vo_mvo_prefetch(k + 2). k_begin == 0 starts the
// sequence; otherwise frame k_begin is the one a previous call left in flight; frame k_end is in flight on return.
extern "C" int vo_mvo_run(vo_mvo *s, const void *const *img, int n_total, int stride, int on_device, int k_begin, int k_end,
                          vo_mvo_frame_info *infos, double *stamps) {
  if (!s || !img || n_total <= 0 || k_begin < 0 || k_end > n_total || k_begin >= k_end) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (k_begin == 0) {
    if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "a frame is already in flight: call vo_mvo_result first");
    RC(vo_mvo_enqueue(s, img[0], stride, on_device, 0.0));
    if (n_total > 1) RC(vo_mvo_prefetch(s, img[1], stride, on_device));
  } else if (!s->pending) {
    VO_FAIL(c, VO_ERR_INVALID, "vo_mvo_run: frame %d is not in flight", k_begin);
  }
  vo_mvo_frame_info info;
  for (int k = k_begin; k < k_end; ++k) {
    RC(vo_mvo_result(s, &info));
    if (stamps) {
      timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      stamps[k - k_begin] = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    if (k + 1 < n_total) {
      RC(vo_mvo_enqueue(s, img[k + 1], stride, on_device, 0.05 * (k + 1)));
      if (k + 2 < n_total) RC(vo_mvo_prefetch(s, img[k + 2], stride, on_device));
    }
    if (infos) infos[k - k_begin] = info;
  }
  return VO_OK;
}

extern "C" int vo_mvo_get_tracks(vo_mvo *s, int32_t *ids, float *pts, float *Xw, uint8_t *flags, int32_t *age, float *cos_parallax,
                                 int cap, int *n) {
  if (!s || !n) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_mvo_result first");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  *n = s->n;
  if (s->n > cap && (ids || pts || Xw || flags || age || cos_parallax)) VO_FAIL(c, VO_ERR_CAPACITY, "%d tracks, room for %d", s->n, cap);
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  const MvoSet &t = s->ts[s->cur];
  const size_t m = (size_t)s->n;
  if (m == 0) return VO_OK;
  if (ids) VO_CHECK_HIP(c, hipMemcpy(ids, t.t.ids, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
  if (pts) VO_CHECK_HIP(c, hipMemcpy(pts, t.t.pts_l, sizeof(float) * 2 * m, hipMemcpyDeviceToHost));
  if (Xw) VO_CHECK_HIP(c, hipMemcpy(Xw, t.t.Xw, sizeof(float) * 3 * m, hipMemcpyDeviceToHost));
  if (flags) VO_CHECK_HIP(c, hipMemcpy(flags, t.t.flags, m, hipMemcpyDeviceToHost));
  if (age) VO_CHECK_HIP(c, hipMemcpy(age, t.age, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
  if (cos_parallax) VO_CHECK_HIP(c, hipMemcpy(cos_parallax, t.cos_last, sizeof(float) * m, hipMemcpyDeviceToHost));
  return VO_OK;
}

extern "C" int vo_mvo_keyframe_count(vo_mvo *s, int *n_keyframes) {
  if (!s) return VO_ERR_INVALID;
  return vo_svo_keyframe_count(&s->core, n_keyframes);
}
extern "C" int vo_mvo_get_keyframes(vo_mvo *s, float *T_wc, int32_t *n_points, float *mappoints, size_t cap_points, size_t *total_points) {
  if (!s) return VO_ERR_INVALID;
  if (s->pending) VO_FAIL(s->c, VO_ERR_INVALID, "call vo_mvo_result first");
  return vo_svo_get_keyframes(&s->core, T_wc, n_points, mappoints, cap_points, total_points);
}
