// stereo_vo_lba.hip — the local bundle adjustment of a StereoVO keyframe: the landmark / keyframe bookkeeping that
// SparseBAParameters walks in the reference, around vo_sba_solve (sba.hip).
//
// Reference:
//   core/visual_odometry/stereo_vo/stereo_vo.cpp:802            localBundleAdjustmentSparseSolver_Stereo at every keyframe
//   core/visual_odometry/motion_estimator.cpp:1207-1340         window >= 3 keyframes, first two fixed, MAX_ITER 10, Huber 0.5
//   core/visual_odometry/keyframes.cpp:185-215                  addNewStereoKeyframe: addObservationAndRelatedKeyframe, left
//                                                               frame first, then the right one
//   core/visual_odometry/ba_solver/sparse_ba_parameters.h:292-466   setPosesAndPoints: landmarks of the window that are
//                                                               triangulated and alive, observations restricted to the window,
//                                                               at least 2 of them; reference frame = first keyframe of the
//                                                               window; translations and points scaled by 1 / 10
//   core/visual_odometry/ba_solver/sparse_bundle_adjustment.cpp:624-722  poses (inverseSE3_f of the float cast) and points back,
//                                                               setBundled / setDead (|X| > 3000), "large update!" (> 50 m)
// The reference iterates an std::unordered_set<LandmarkPtr> (sparse_ba_parameters.h:333): its landmark order depends on heap
// addresses. Here: ascending landmark id (as oracle/stereo_vo.py).
// The host part runs at keyframe rate on a few thousand landmarks; everything per observation is on the device (sba.hip).
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include <stdlib.h>
#include <time.h>

#include "stereo_vo.hpp"

static double lba_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}
#define LBA_T(k)                       do {                                   if (trace) {                           const double now_ = lba_now();       tt[k] += now_ - t_last;              t_last = now_;                     }                                  } while (0)

static void mul44d(const double A[16], const double B[16], double C[16]) {  // Matrix4d * Matrix4d, left to right over k
  double R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double r = A[i * 4 + 0] * B[0 * 4 + j];
      for (int k = 1; k < 4; ++k) r = A[i * 4 + k] * B[k * 4 + j] + r;
      R[i * 4 + j] = r;
    }
  for (int i = 0; i < 16; ++i) C[i] = R[i];
}
static void xformd(const double T[16], const double X[3], double Y[3]) {  // R * X + t, 3-term dot products e0 + (e1 + e2)
  double r[3];
  for (int i = 0; i < 3; ++i) r[i] = (T[i * 4 + 0] * X[0] + (T[i * 4 + 1] * X[1] + T[i * 4 + 2] * X[2])) + T[i * 4 + 3];
  Y[0] = r[0];
  Y[1] = r[1];
  Y[2] = r[2];
}
static void inv_se3d(const double T[16], double Ti[16]) {  // geometry::inverseSE3
  double Rt[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = T[j * 4 + i];
  const double t[3] = {T[3], T[7], T[11]};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Ti[i * 4 + j] = Rt[i * 3 + j];
    Ti[i * 4 + 3] = (-Rt[i * 3 + 0]) * t[0] + ((-Rt[i * 3 + 1]) * t[1] + (-Rt[i * 3 + 2]) * t[2]);
  }
  Ti[12] = Ti[13] = Ti[14] = 0.0;
  Ti[15] = 1.0;
}
static void to_d(const float T[16], double D[16]) {
  for (int i = 0; i < 12; ++i) D[i] = (double)T[i];
  D[12] = D[13] = D[14] = 0.0;
  D[15] = 1.0;
}

int vo_svo_local_ba(vo_svo *s, vo_svo_frame_info *info) {
  vo_ctx *c = s->c;
  const int n = s->n;
  SvoTrackSet &t = s->ts[s->cur];
  static const bool trace = getenv("VO_SVO_TRACE") != nullptr;
  static double tt[8];
  static int n_calls;
  double t_last = trace ? lba_now() : 0.0;
  hipStream_t st = c->stream;
  // ---- the new keyframe's related landmarks, after the reconstruction kernel (main stream), through pinned memory ----
  if (!s->h_ids) {
    const size_t cap = (size_t)s->cap;
    VO_CHECK_HIP(c, hipHostMalloc((void **)&s->h_ids, sizeof(int32_t) * cap, hipHostMallocDefault));
    VO_CHECK_HIP(c, hipHostMalloc((void **)&s->h_pl, sizeof(float) * 2 * cap, hipHostMallocDefault));
    VO_CHECK_HIP(c, hipHostMalloc((void **)&s->h_pr, sizeof(float) * 2 * cap, hipHostMallocDefault));
    VO_CHECK_HIP(c, hipHostMalloc((void **)&s->h_Xw, sizeof(float) * 3 * cap, hipHostMallocDefault));
    VO_CHECK_HIP(c, hipHostMalloc((void **)&s->h_fl, cap, hipHostMallocDefault));
  }
  int32_t *ids = s->h_ids;
  float *pl = s->h_pl, *pr = s->h_pr, *Xw = s->h_Xw;
  uint8_t *fl = s->h_fl;
  if (n > 0) {
    VO_CHECK_HIP(c, hipMemcpyAsync(ids, t.ids, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(pl, t.pts_l, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(pr, t.pts_r, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(Xw, t.Xw, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(fl, t.flags, (size_t)n, hipMemcpyDeviceToHost, st));
    VO_CHECK_HIP(c, hipStreamSynchronize(st));
  }
  LBA_T(0);
  SvoKeyframe &kf = s->keyframes.back();
  kf.ids.assign(ids, ids + n);
  kf.pl.assign(pl, pl + 2 * (size_t)n);
  kf.pr.assign(pr, pr + 2 * (size_t)n);
  if (n > 0 && (size_t)ids[n - 1] >= s->lmS.size()) {  // (ids ascend: the last one is the largest)
    const size_t want = (size_t)ids[n - 1] + 1 + 8192;
    s->lmX.resize(3 * want, 0.0f);
    s->lmS.resize(want, 0);
  }
  for (int k = 0; k < n; ++k) {  // the landmarks' state as of this keyframe
    const size_t id = (size_t)ids[k];
    s->lmX[3 * id] = Xw[3 * k];
    s->lmX[3 * id + 1] = Xw[3 * k + 1];
    s->lmX[3 * id + 2] = Xw[3 * k + 2];
    s->lmS[id] = (uint8_t)((s->lmS[id] & 2) | ((fl[k] & VO_LM_TRIANGULATED) ? 1 : 0));
  }
  LBA_T(1);
  std::vector<SvoKeyframe> &win = s->keyframes;
  const int nk = (int)win.size();
  if (nk < 3) return VO_OK;  // NUM_MINIMUM_REQUIRED_KEYFRAMES (motion_estimator.cpp:1245-1253)
  const double POSE_SCALE = 10.0, inv_scale = 1.0 / POSE_SCALE;
  // ---- SparseBAParameters::setPosesAndPoints: one merge over the window's (ascending) id lists ----
  double Twj_ref[16], Tjw_ref[16];
  to_d(win.front().T_wc, Twj_ref);
  inv_se3d(Twj_ref, Tjw_ref);
  std::vector<double> &X = s->ba_X, &px = s->ba_px, &T_jw = s->ba_T;
  std::vector<int32_t> &obs_ptr = s->ba_obs_ptr, &obs_frame = s->ba_obs_frame, &used = s->ba_used, &opt = s->ba_opt;
  std::vector<uint8_t> &obs_right = s->ba_obs_right;
  X.clear();
  px.clear();
  obs_ptr.assign(1, 0);
  obs_frame.clear();
  obs_right.clear();
  used.clear();
  size_t cur[32] = {0};
  if (nk > 32) VO_FAIL(c, VO_ERR_CAPACITY, "keyframe window of %d", nk);
  for (;;) {
    int32_t id = INT32_MAX;
    for (int j = 0; j < nk; ++j)
      if (cur[j] < win[j].ids.size() && win[j].ids[cur[j]] < id) id = win[j].ids[cur[j]];
    if (id == INT32_MAX) break;
    const bool take = (s->lmS[id] & 1) && !(s->lmS[id] & 2);  // isTriangulated() && isAlive()
    bool any = false;
    for (int j = 0; j < nk; ++j) {
      if (!(cur[j] < win[j].ids.size() && win[j].ids[cur[j]] == id)) continue;
      const size_t q = cur[j]++;
      if (!take) continue;
      any = true;  // (a stereo keyframe gives two observations: THRES_MINIMUM_SEEN = 2 always holds)
      obs_frame.push_back(j);
      obs_frame.push_back(j);
      obs_right.push_back(0);
      obs_right.push_back(1);
      px.push_back((double)win[j].pl[2 * q]);
      px.push_back((double)win[j].pl[2 * q + 1]);
      px.push_back((double)win[j].pr[2 * q]);
      px.push_back((double)win[j].pr[2 * q + 1]);
    }
    if (!any) continue;
    const double Xd[3] = {(double)s->lmX[3 * (size_t)id], (double)s->lmX[3 * (size_t)id + 1], (double)s->lmX[3 * (size_t)id + 2]};
    double Xr[3];
    xformd(Tjw_ref, Xd, Xr);  // warpToRef
    for (int k = 0; k < 3; ++k) X.push_back(Xr[k] * inv_scale);  // scalingPoint
    obs_ptr.push_back((int32_t)obs_frame.size());
    used.push_back(id);
  }
  if (used.empty()) return VO_OK;
  T_jw.resize(16 * (size_t)nk);
  opt.resize(nk);
  for (int j = 0; j < nk; ++j) {
    float Tjw_f[16];
    double Tjw[16];
    svo_inv_se3(win[j].T_wc, Tjw_f);  // getPoseInv()
    to_d(Tjw_f, Tjw);
    mul44d(Tjw, Twj_ref, &T_jw[16 * j]);  // changeInvPoseWorldToRef
    for (int r = 0; r < 3; ++r) T_jw[16 * j + r * 4 + 3] *= inv_scale;  // scalingPose
    opt[j] = j < 2 ? -1 : j - 2;  // NUM_FIX_KEYFRAMES_IN_WINDOW
  }
  vo_sba_problem p;
  memset(&p, 0, sizeof(p));
  p.n_frames = nk;
  p.n_opt = nk - 2;
  p.n_points = (int)used.size();
  p.n_obs = (int)obs_frame.size();
  p.stereo = 1;
  p.max_iter = 10;
  p.thres_huber = 0.5;
  for (int k = 0; k < 4; ++k) {
    p.Kl[k] = (double)s->prm.frame.Kl[k];
    p.Kr[k] = (double)s->prm.frame.Kr[k];
  }
  to_d(s->prm.frame.T_lr, p.T_lr);
  for (int r = 0; r < 3; ++r) p.T_lr[r * 4 + 3] *= inv_scale;  // scalingPose(T_stereo_)
  double err[16] = {0};
  LBA_T(2);
  int rc = vo_sba_solve(c, &p, T_jw.data(), opt.data(), X.data(), obs_ptr.data(), obs_frame.data(), obs_right.data(), px.data(), err);
  if (rc < 0) return rc;
  LBA_T(3);
  if (info) {
    info->lba_ran = 1;
    info->lba_err_first = err[0];
    info->lba_err_last = err[p.max_iter - 1];
    info->lba_landmarks = p.n_points;
    info->lba_observations = p.n_obs;
  }
  // ---- sparse_bundle_adjustment.cpp:624-722: poses and points back ----
  for (int j = 0; j < nk; ++j) {
    if (opt[j] < 0) continue;
    double T[16], Tjw[16], Twj_orig[16], dT[16];
    for (int k = 0; k < 16; ++k) T[k] = T_jw[16 * j + k];
    for (int r = 0; r < 3; ++r) T[r * 4 + 3] *= POSE_SCALE;  // recoverOriginalScalePose
    mul44d(T, Tjw_ref, Tjw);                                 // changeInvPoseRefToWorld
    to_d(win[j].T_wc, Twj_orig);
    mul44d(Twj_orig, Tjw, dT);
    const double tn = sqrt(dT[3] * dT[3] + (dT[7] * dT[7] + dT[11] * dT[11]));
    if (tn > 50) VO_FAIL(c, VO_ERR_LBA_NAN, "local BA: large update!");
    float Tf[16];
    for (int k = 0; k < 12; ++k) Tf[k] = (float)Tjw[k];
    Tf[12] = Tf[13] = Tf[14] = 0.0f;
    Tf[15] = 1.0f;
    svo_inv_se3(Tf, win[j].T_wc);  // kf->setPose(inverseSE3_f(Tjw_update_float))
  }
  for (size_t i = 0; i < used.size(); ++i) {
    double xs[3] = {X[3 * i] * POSE_SCALE, X[3 * i + 1] * POSE_SCALE, X[3 * i + 2] * POSE_SCALE}, xw[3];
    xformd(Twj_ref, xs, xw);  // warpToWorld
    const size_t id = (size_t)used[i];
    float *L = &s->lmX[3 * id];
    L[0] = (float)xw[0];
    L[1] = (float)xw[1];
    L[2] = (float)xw[2];
    s->lmS[id] |= 1;  // set3DPoint
    const float nrm = sqrtf(L[0] * L[0] + (L[1] * L[1] + L[2] * L[2]));
    if (!(nrm <= 3000)) s->lmS[id] |= 2;  // setDead
  }
  LBA_T(4);
  // ---- what the BA did to the landmarks the next frame tracks (the pinned staging arrays go back up) ----
  for (int k = 0; k < n; ++k) {
    const size_t id = (size_t)ids[k];
    Xw[3 * k] = s->lmX[3 * id];
    Xw[3 * k + 1] = s->lmX[3 * id + 1];
    Xw[3 * k + 2] = s->lmX[3 * id + 2];
    if (s->lmS[id] & 1) fl[k] |= VO_LM_TRIANGULATED;
    if (s->lmS[id] & 2) fl[k] |= VO_LM_DROPPED;
  }
  if (n > 0) {  // (stream-ordered in front of the next frame; the staging arrays are next written behind a synchronisation)
    VO_CHECK_HIP(c, hipMemcpyAsync(t.Xw, Xw, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
    VO_CHECK_HIP(c, hipMemcpyAsync(t.flags, fl, (size_t)n, hipMemcpyHostToDevice, st));
  }
  LBA_T(5);
  if (trace && (++n_calls % 10) == 0)
    fprintf(stderr, "[lba] per call (us): d2h %.0f  db %.0f  problem %.0f  solve %.0f  finish %.0f  h2d %.0f  (M=%d obs=%d)\n",
            tt[0] / n_calls, tt[1] / n_calls, tt[2] / n_calls, tt[3] / n_calls, tt[4] / n_calls, tt[5] / n_calls, p.n_points, p.n_obs);
  return VO_OK;
}
