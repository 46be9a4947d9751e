// frame_mono.hip — the steady-state MONO frame, chained on the device.
//
// Operator sequence of MonoVO::trackImage (core/visual_odometry/mono_vo/mono_vo.cpp)
//   prior pixels + patch scale            :739-761
//   trackBidirectionWithPrior I0 -> I1    :768-770, compaction :773
//   Sobel + trackWithScale                :779-786, compaction :788
//   index_ba selection (depth > 0.1)      :799-826
//   poseOnlyBundleAdjustment (core)       :856-867, mask_motion :872-879
//   Sampson distance gate                 :954-963
// What the landmark graph decides on the host comes in as flags: bit 0 = lm->isBundled() (prior and
// scale from the 3-D point), bit 1 = the landmark belongs to the class this frame uses for pose-only
// BA (:800-826). The 5-point fallback (:905-935, OpenCV calib3d) stays on the host: when the BA has
// too few points or fails, counts.need_five_point is set and the frame stops after the refinement.
//
// As in the stereo frame (frame_fused.hip) the per-feature steps — prior, forward KLT, backward KLT,
// bidirectional mask, IC refinement — are one wavefront of ONE launch; the strict-border replay, the
// selection of the BA set, the GN solve and the Sampson gate follow as launches on the same stream, and
// the host reads one packed block. stage[i] = number of gates feature i passed (1 tracked, 2 refined,
// 3 motion inlier or not part of the BA, 4 passed the Sampson gate).
#include "frame_state.hpp"
#include "ic_device.hpp"
#include "klt_device.hpp"
#include "vo_kernels.hpp"

struct MonoArgs {
  vo_level I0[VO_MAX_LEVELS], I1[VO_MAX_LEVELS];
  int max_level, n;
  const float *pts0, *Xw;
  const uint8_t *flags;
  float Tcw_prev[16], Tcw_prior[16], K[4];
  int W, H;
  float thres_err, thres_bidir;
  int strict;
  float *scale;     // out [n]
  float *k1;        // out [n][2] forward KLT result (= ic.pts_prior)
  float *Xp;        // out [n][3] point in the previous camera frame
  uint8_t *m1;      // out [n] trackBidirectionWithPrior mask
  uint8_t *ba_ok;   // out [n] BA class && depth > 0.1
  int32_t *orig;    // out [n] identity (compaction keeps it)
  IcArgs ic;        // pts0, scale, pts_prior = k1, pts_track = refined, mask = m2, records
};

template <int WIN>
struct MonoShared {
  uint32_t tt[KltCfg<WIN>::TT_H * KltCfg<WIN>::TT_WD];
  uint32_t tj[KltCfg<WIN>::TJ_H * KltCfg<WIN>::TJ_WD];
  IcShared ic;
};

template <int WIN>
__global__ __launch_bounds__(64) void mono_track_kernel(MonoArgs a) {
  __shared__ MonoShared<WIN> sh;
  const int i = blockIdx.x;
  if (i >= a.n) return;
  const int lane = threadIdx.x;
  const float p0x = a.pts0[2 * i], p0y = a.pts0[2 * i + 1];
  const int fl = a.flags[i];
  // ---- prior + patch scale (mono_vo.cpp:739-761) ----
  float Xp[3] = {0.f, 0.f, 0.f};
  float prx = p0x, pry = p0y, scale = 1.0f;
  const float *Xi = a.Xw + 3 * i;
  if (fl & 3) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      Xp[r] = ((a.Tcw_prev[r * 4 + 0] * Xi[0] + a.Tcw_prev[r * 4 + 1] * Xi[1]) + a.Tcw_prev[r * 4 + 2] * Xi[2]) +
              a.Tcw_prev[r * 4 + 3];
  }
  if (fl & 1) {
    float Xc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      Xc[r] = ((a.Tcw_prior[r * 4 + 0] * Xi[0] + a.Tcw_prior[r * 4 + 1] * Xi[1]) + a.Tcw_prior[r * 4 + 2] * Xi[2]) +
              a.Tcw_prior[r * 4 + 3];
    scale = Xp[2] / Xc[2];
    if (Xc[2] > 0) {  // Camera::projectToPixel, camera.cpp:208-213
      const float invz = 1.0f / Xc[2];
      prx = a.K[0] * Xc[0] * invz + a.K[2];
      pry = a.K[1] * Xc[1] * invz + a.K[3];
    }
  }
  // ---- trackBidirectionWithPrior (feature_tracker.cpp:88-169): forward with the prior as initial
  // flow, backward from the result with pts0 as initial flow, both at full maxLevel, {} criteria ----
  KltResult fwd, bwd;
  fwd.x = fwd.y = fwd.err = 0.f;
  fwd.status = 0;
  bwd = fwd;
  float q0x = p0x, q0y = p0y, ix = prx, iy = pry;
#pragma nounroll
  for (int pass = 0; pass < 2; ++pass) {
    const vo_level *I = pass == 0 ? a.I0 : a.I1;
    const vo_level *J = pass == 0 ? a.I1 : a.I0;
    const KltResult k = klt_point<WIN>(I, J, a.max_level, VO_KLT_USE_INITIAL_FLOW, 30, 0.01 * 0.01, 0.f, q0x, q0y, ix,
                                       iy, sh.tt, sh.tj, lane);
    if (pass == 0) {
      fwd = k;
      q0x = k.x;
      q0y = k.y;
      ix = p0x;
      iy = p0y;
    } else {
      bwd = k;
    }
  }
  // mask, feature_tracker.cpp:130-155
  bool m1;
  {
    const float dx = bwd.x - p0x, dy = bwd.y - p0y;
    const float dist2 = dx * dx + dy * dy;
    const float thres2 = a.thres_bidir * a.thres_bidir;
    const bool inimage = fwd.x > 0 && fwd.x < a.W && fwd.y > 0 && fwd.y < a.H;
    m1 = inimage && fwd.status && fwd.err <= a.thres_err && bwd.status && bwd.err <= a.thres_err && dist2 <= thres2 * 5;
  }
  // ---- trackWithScale, pass 1 (taps outside the image masked) ----
  const IcTaps tp = ic_make_taps(lane);
  IcState S;
  ic_state_clear(S);
  int cls = 0, touched = 0, n_iter = 0;
  float lpx = 0.f, lpy = 0.f;
  IcResult rf;
  rf.cls = 0;
  rf.ok = 0;
  rf.x = fwd.x;
  rf.y = fwd.y;
  rf.err_flag = 0;
  if (m1) {
    rf = ic_point<false>(a.I0[0], a.I1[0], tp, p0x, p0y, fwd.x, fwd.y, scale, lane, sh.ic, S, touched, lpx, lpy, n_iter);
    cls = rf.cls;
  }
  const int any_t = __any(touched);
  if (a.strict) ic_store_records(a.ic, i, lane, tp, S, cls);
  if (lane == 0) {
    a.scale[i] = scale;
    a.k1[2 * i] = fwd.x;
    a.k1[2 * i + 1] = fwd.y;
    a.Xp[3 * i] = Xp[0];
    a.Xp[3 * i + 1] = Xp[1];
    a.Xp[3 * i + 2] = Xp[2];
    a.m1[i] = m1 ? 1 : 0;
    a.ba_ok[i] = ((fl & 2) && Xp[2] > 0.1f) ? 1 : 0;  // mono_vo.cpp:808-809 / :821-822
    a.orig[i] = i;
    a.ic.pts_track[2 * i] = rf.x;
    a.ic.pts_track[2 * i + 1] = rf.y;
    a.ic.mask[i] = (uint8_t)rf.ok;
    if (rf.err_flag) atomicOr(a.ic.flags, rf.err_flag);
    if (a.strict) {
      a.ic.touched[i] = (uint8_t)(any_t ? 1 : 0);
      a.ic.cls[i] = (uint8_t)cls;
      a.ic.last_pu[2 * i] = lpx;
      a.ic.last_pu[2 * i + 1] = lpy;
      if (any_t) a.ic.tlist[atomicAdd(&a.ic.jac[IC_JAC_NT], 1)] = i;
    }
  }
}

// strict border: the touched features (ic_replay), then the sequential fallback if it was requested
__global__ __launch_bounds__(IC_T) void mono_replay_kernel(IcArgs a) {
  __shared__ IcReplayShared rs;
  (void)ic_replay(a, rs, threadIdx.x, [](int, const IcResult &) {});
}
__global__ __launch_bounds__(IC_T) void mono_fallback_kernel(IcArgs a) {
  __shared__ IcShared sh;
  if (a.jac[IC_JAC_OVF] == 0) return;
  const int pt = blockIdx.x;
  if (pt >= a.n) return;
  ic_strict_run(a, sh, pt, a.n, threadIdx.x, [](int, const IcResult &) {});
}

// ---- after the GN launch: mask_motion, Sampson gate, stages, counts (one workgroup) ----
struct MonoGateArgs {
  int n;
  const float *pts0, *k1, *ref;
  const uint8_t *m1, *m2, *ba_ok, *mG;
  const int32_t *C_orig;
  const int *n_ba;          // survivors handed to the GN solve
  const vo_gn_dev_info *gn;
  float *dT;                // in: GN result T01 ; out: the prior when the 5-point fallback is needed
  float dT_prior[16], K[4];
  float thres_sampson;
  uint8_t *motion;          // scratch [n]
  uint8_t *stage;           // out
  float *pts1;              // out
  int *cnt;                 // out [8]: n_klt, n_refine, n_ba, n_motion, n_final, need_five_point
  int *ctl;                 // frame control block: reported ([0] flags, replay count) and reset here
  int ctl_words, nt_word;
  int *hdr_flags;
  const uint32_t *res_dev;  // packed result block -> res_host (pinned, device-visible)
  uint32_t *res_host;
  int res_words;
};
__device__ __forceinline__ float mono_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);  // Eigen's unrolled 3-term redux
}
__global__ __launch_bounds__(1024) void mono_gate_kernel(MonoGateArgs a) {
  __shared__ float sF[9];
  __shared__ int s_cnt[8];
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  if (tid < 8) s_cnt[tid] = 0;
  const int n_ba = *a.n_ba;
  if (tid == 0) {
    const int ok = n_ba > 10 && !a.gn->is_nan;  // mono_vo.cpp:838, :866
    s_ok = ok;
    if (!ok) {
      for (int k = 0; k < 16; ++k) a.dT[k] = a.dT_prior[k];
    } else {
      // dT10 = inverseSE3_f(dT01) (geometry_library.cpp:554-560); F10 = Kinv^T [t10]x R10 Kinv (motion_estimator.cpp:551-552)
      float R10[9], t10[3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R10[i * 3 + j] = a.dT[j * 4 + i];
      const float t0 = a.dT[3], t1 = a.dT[7], t2 = a.dT[11];
      for (int i = 0; i < 3; ++i) t10[i] = ((-R10[i * 3 + 0]) * t0 + (-R10[i * 3 + 1]) * t1) + (-R10[i * 3 + 2]) * t2;
      const float fxi = 1.0f / a.K[0], fyi = 1.0f / a.K[1];
      const float Kinv[9] = {fxi, 0.0f, -a.K[2] * fxi, 0.0f, fyi, -a.K[3] * fyi, 0.0f, 0.0f, 1.0f};
      const float KinvT[9] = {Kinv[0], Kinv[3], Kinv[6], Kinv[1], Kinv[4], Kinv[7], Kinv[2], Kinv[5], Kinv[8]};
      const float Sx[9] = {0.0f, -t10[2], t10[1], t10[2], 0.0f, -t10[0], -t10[1], t10[0], 0.0f};
      float E[9], T[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          E[i * 3 + j] = mono_dot3(Sx[i * 3 + 0], R10[0 * 3 + j], Sx[i * 3 + 1], R10[1 * 3 + j], Sx[i * 3 + 2], R10[2 * 3 + j]);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          T[i * 3 + j] = mono_dot3(KinvT[i * 3 + 0], E[0 * 3 + j], KinvT[i * 3 + 1], E[1 * 3 + j], KinvT[i * 3 + 2], E[2 * 3 + j]);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          sF[i * 3 + j] = mono_dot3(T[i * 3 + 0], Kinv[0 * 3 + j], T[i * 3 + 1], Kinv[1 * 3 + j], T[i * 3 + 2], Kinv[2 * 3 + j]);
    }
    *a.hdr_flags = a.ctl[0];
    a.cnt[7] = a.ctl[16 + a.nt_word];
  }
  // mask_motion: true for every refined feature, the BA's inlier mask for the BA set (:845, :872-879)
  for (int i = tid; i < a.n; i += 1024) a.motion[i] = (a.m1[i] && a.m2[i]) ? 1 : 0;
  __syncthreads();
  const int ok = s_ok;
  if (ok)
    for (int c = tid; c < n_ba; c += 1024) a.motion[a.C_orig[c]] = a.mG[c];
  __syncthreads();
  for (int k = tid; k < a.ctl_words; k += 1024) a.ctl[k] = 0;
  int c_klt = 0, c_ref = 0, c_mot = 0, c_fin = 0;
  for (int i = tid; i < a.n; i += 1024) {
    const int m1 = a.m1[i], m2 = m1 && a.m2[i];
    int st = m1 ? (m2 ? 2 : 1) : 0;
    const float x1 = m1 ? a.ref[2 * i] : a.k1[2 * i], y1 = m1 ? a.ref[2 * i + 1] : a.k1[2 * i + 1];
    if (ok && m2 && a.motion[i]) {
      st = 3;
      // calcSampsonDistance, motion_estimator.cpp:553-569
      const float x0 = a.pts0[2 * i], y0 = a.pts0[2 * i + 1];
      float p[3], q[3];
      for (int r = 0; r < 3; ++r) p[r] = mono_dot3(sF[r * 3 + 0], x0, sF[r * 3 + 1], y0, sF[r * 3 + 2], 1.0f);
      for (int r = 0; r < 3; ++r) q[r] = mono_dot3(sF[0 * 3 + r], x1, sF[1 * 3 + r], y1, sF[2 * 3 + r], 1.0f);
      float num = mono_dot3(x1, p[0], y1, p[1], 1.0f, p[2]);
      num *= num;
      const float den = ((p[0] * p[0] + p[1] * p[1]) + q[0] * q[0]) + q[1] * q[1];
      if (num / den < a.thres_sampson) st = 4;
    }
    a.stage[i] = (uint8_t)st;
    a.pts1[2 * i] = x1;
    a.pts1[2 * i + 1] = y1;
    c_klt += st >= 1;
    c_ref += st >= 2;
    c_mot += st >= 3;
    c_fin += st >= 4;
  }
  atomicAdd(&s_cnt[0], c_klt);
  atomicAdd(&s_cnt[1], c_ref);
  atomicAdd(&s_cnt[3], c_mot);
  atomicAdd(&s_cnt[4], c_fin);
  __syncthreads();
  if (tid == 0) {
    a.cnt[0] = s_cnt[0];
    a.cnt[1] = s_cnt[1];
    a.cnt[2] = n_ba;
    a.cnt[3] = s_cnt[3];
    a.cnt[4] = s_cnt[4];
    a.cnt[5] = ok ? 0 : 1;
  }
  // every header word is written by a launch of this frame (cnt[6]: compaction, gn / dT: GN, the rest
  // above), so the block needs no clearing. It goes to pinned host memory from here (all final: earlier
  // launches' stores, and this workgroup's own above)
  __syncthreads();
  for (int k = tid; k < a.res_words; k += 1024) a.res_host[k] = a.res_dev[k];
}

// ---- host side ---------------------------------------------------------------------
static size_t m_align16(size_t v) { return (v + 15) & ~(size_t)15; }

template <int WIN>
static void mono_launch(vo_ctx *c, const MonoArgs &a) {
  vo_prof_begin(c, VO_K_KLT);
  hipLaunchKernelGGL(mono_track_kernel<WIN>, dim3(a.n), dim3(64), 0, c->stream, a);
  vo_prof_end(c);
}

extern "C" int vo_mono_frame_enqueue(vo_ctx *c, const vo_mono_params *prm, int slot0, int slot1, const float *pts0,
                                     const float *Xw, const uint8_t *flags, int n, const float Tcw_prev[16],
                                     const float Tcw_prior[16], const float dT01_prior[16], int inputs_on_device) {
  if (!c || !prm || !Tcw_prev || !Tcw_prior || !dT01_prior || n < 0) return VO_ERR_INVALID;
  if (n > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "n=%d exceeds vo_config.max_points=%d", n, c->cfg.max_points);
  if (n > 0 && (!pts0 || !Xw || !flags)) return VO_ERR_INVALID;
  if (prm->win != 13 && prm->win != 15 && prm->win != 21 && prm->win != 31)
    VO_FAIL(c, VO_ERR_INVALID, "mono frame kernel not instantiated for window %d (13, 15, 21, 31)", prm->win);
  if (prm->max_level < 0) VO_FAIL(c, VO_ERR_INVALID, "maxLevel >= 0 violated");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = vo_frame_init(c);
  if (rc < 0) return rc;
  vo_frame_state *f = c->frame;
  hipStream_t s = c->stream;
  const float *d_p0 = pts0, *d_X = Xw;
  const uint8_t *d_fl = flags;
  if (!inputs_on_device && n > 0) {
    VO_CHECK_HIP(c, hipMemcpyAsync(f->in_l0, pts0, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(f->in_X, Xw, sizeof(float) * 3 * n, hipMemcpyHostToDevice, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(f->st1, flags, (size_t)n, hipMemcpyHostToDevice, s));
    d_p0 = f->in_l0;
    d_X = f->in_X;
    d_fl = f->st1;
  }
  // packed result block: header | stage | pixels | scale
  f->n = n;
  f->n_new = 0;
  size_t off = m_align16(sizeof(vo_frame_hdr));
  f->off_stage = off;  off += m_align16((size_t)n);
  f->off_pl1 = off;    off += m_align16(sizeof(float) * 2 * (size_t)n);
  f->off_pr1 = off;    off += m_align16(sizeof(float) * (size_t)n);  // (scale)
  f->res_bytes = off;
  f->hdr = (vo_frame_hdr *)f->res_dev;
  f->stage = f->res_dev + f->off_stage;
  f->F_pl1 = (float *)(f->res_dev + f->off_pl1);
  if (n > 0) {
    if (slot0 < 0 || slot0 >= c->cfg.n_slots || slot1 < 0 || slot1 >= c->cfg.n_slots ||
        c->slots[slot0].n_levels <= 0 || c->slots[slot1].n_levels <= 0)
      VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
    const vo_pyramid &P0 = c->slots[slot0], &P1 = c->slots[slot1];
    if (P0.w != P1.w || P0.h != P1.h) VO_FAIL(c, VO_ERR_SIZE, "image size mismatch");
    MonoArgs a;
    memset(&a, 0, sizeof(a));
    int eff = vo_pyr_levels_host(P0.w, P0.h, prm->win, prm->max_level);
    if (eff > P0.n_levels - 1) eff = P0.n_levels - 1;
    if (eff > P1.n_levels - 1) eff = P1.n_levels - 1;
    for (int l = 0; l <= eff; ++l) {
      a.I0[l] = P0.lv[l];
      a.I1[l] = P1.lv[l];
    }
    a.max_level = eff;
    a.n = n;
    a.pts0 = d_p0;
    a.Xw = d_X;
    a.flags = d_fl;
    memcpy(a.Tcw_prev, Tcw_prev, sizeof(a.Tcw_prev));
    memcpy(a.Tcw_prior, Tcw_prior, sizeof(a.Tcw_prior));
    memcpy(a.K, prm->K, sizeof(a.K));
    a.W = prm->width;
    a.H = prm->height;
    a.thres_err = prm->thres_err;
    a.thres_bidir = prm->thres_bidirection;
    a.strict = c->frame_strict_ic;
    float *d_scale = (float *)(f->res_dev + f->off_pr1);  // written straight into the result block
    a.scale = d_scale;
    a.k1 = f->A_pl1;
    a.Xp = f->A_X;
    a.m1 = f->m1;
    a.ba_ok = f->m3;
    a.orig = f->F_orig;
    rc = vo_ic_frame_args(c, slot0, slot1, &a.ic, f->ctl, a.strict != 0);
    if (rc) return rc;
    a.ic.pts0 = d_p0;
    a.ic.scale = d_scale;
    a.ic.pts_prior = f->A_pl1;
    a.ic.pts_track = f->A_ref;
    a.ic.mask = f->m2;
    a.ic.touched = f->A_touched;
    a.ic.cls = f->A_cls;
    a.ic.last_pu = f->A_lastpu;
    a.ic.n = n;
    switch (prm->win) {
      case 13: mono_launch<13>(c, a); break;
      case 15: mono_launch<15>(c, a); break;
      case 21: mono_launch<21>(c, a); break;
      default: mono_launch<31>(c, a); break;
    }
    if (a.strict) {
      vo_prof_begin(c, VO_K_IC);
      if (a.strict == 2)
        (void)hipMemsetAsync(&a.ic.jac[IC_JAC_OVF], 1, sizeof(int), s);
      else
        hipLaunchKernelGGL(mono_replay_kernel, dim3(n < IC_JGRID ? n : IC_JGRID), dim3(IC_T), 0, s, a.ic);
      hipLaunchKernelGGL(mono_fallback_kernel, dim3(n), dim3(IC_T), 0, s, a.ic);
      vo_prof_end(c);
    }
    // BA set: refined && BA class && depth > 0.1, in index order (mono_vo.cpp:799-826, :846-860)
    CompactArgsHost h;
    h.mask = f->m1;
    h.alive = f->m2;
    h.tracked = f->m3;
    h.n = n;
    h.d_n_out = &f->hdr->cnt[6];
    h.in2[0] = f->A_ref;
    h.out2[0] = f->C_pl1;
    h.in3 = f->A_X;
    h.out3 = f->C_X;
    h.in_i = f->F_orig;
    h.out_i = f->C_orig;
    rc = vo_compact_enqueue(c, h);
    if (rc < 0) return rc;
    // poseOnlyBundleAdjustment (class-surface variant), T01 initialised with the motion prior (:856-867)
    rc = vo_gn_enqueue(c, false, true, f->C_X, f->C_pl1, nullptr, n, &f->hdr->cnt[6], prm->K, prm->K, nullptr,
                       (float)prm->thres_poseba, VO_GN_VARIANT_CORE, dT01_prior, f->hdr->dT, f->mG, &f->hdr->gn, true);
    if (rc < 0) return rc;
    MonoGateArgs g;
    memset(&g, 0, sizeof(g));
    g.n = n;
    g.pts0 = d_p0;
    g.k1 = f->A_pl1;
    g.ref = f->A_ref;
    g.m1 = f->m1;
    g.m2 = f->m2;
    g.ba_ok = f->m3;
    g.mG = f->mG;
    g.C_orig = f->C_orig;
    g.n_ba = &f->hdr->cnt[6];
    g.gn = &f->hdr->gn;
    g.dT = f->hdr->dT;
    memcpy(g.dT_prior, dT01_prior, sizeof(g.dT_prior));
    memcpy(g.K, prm->K, sizeof(g.K));
    g.thres_sampson = prm->thres_sampson;
    g.motion = f->st2;
    g.stage = f->stage;
    g.pts1 = f->F_pl1;
    g.cnt = f->hdr->cnt;
    g.ctl = f->ctl;
    g.ctl_words = (int)(vo_ic_ctl_bytes() / 4);
    g.nt_word = vo_ic_ctl_nt_word();
    g.hdr_flags = &f->hdr->flags;
    g.res_dev = (const uint32_t *)f->res_dev;
    g.res_host = (uint32_t *)f->res_host;
    g.res_words = (int)((f->res_bytes + 3) / 4);
    vo_prof_begin(c, VO_K_AUX);
    hipLaunchKernelGGL(mono_gate_kernel, dim3(1), dim3(1024), 0, s, g);
    vo_prof_end(c);
    VO_CHECK_HIP(c, hipGetLastError());
  } else {
    VO_CHECK_HIP(c, hipMemsetAsync(f->hdr, 0, sizeof(vo_frame_hdr), s));
    VO_CHECK_HIP(c, hipMemcpyAsync(f->hdr->dT, dT01_prior, sizeof(float) * 16, hipMemcpyHostToDevice, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(f->res_host, f->res_dev, f->res_bytes, hipMemcpyDeviceToHost, s));
  }
  VO_CHECK_HIP(c, hipEventRecord(f->ev_done, s));
  f->pending = true;
  return VO_OK;
}

extern "C" int vo_mono_frame_result(vo_ctx *c, float *pts1, float *scale, uint8_t *stage, float dT01[16],
                                    vo_mono_counts *counts, vo_gn_info *gn) {
  if (!c || !c->frame || !c->frame->pending) return VO_ERR_INVALID;
  vo_frame_state *f = c->frame;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_CHECK_HIP(c, hipEventSynchronize(f->ev_done));
  f->pending = false;
  const int n = f->n;
  const vo_frame_hdr *h = (const vo_frame_hdr *)f->res_host;
  if (pts1 && n) memcpy(pts1, f->res_host + f->off_pl1, sizeof(float) * 2 * (size_t)n);
  if (scale && n) memcpy(scale, f->res_host + f->off_pr1, sizeof(float) * (size_t)n);
  if (stage && n) memcpy(stage, f->res_host + f->off_stage, (size_t)n);
  if (dT01) memcpy(dT01, h->dT, sizeof(float) * 16);
  if (counts) {
    counts->n_klt = h->cnt[0];
    counts->n_refine = h->cnt[1];
    counts->n_ba = h->cnt[2];
    counts->n_motion = h->cnt[3];
    counts->n_final = h->cnt[4];
    counts->need_five_point = n > 0 ? h->cnt[5] : 1;
    counts->gn_iterations = (n > 0 && h->cnt[2] > 10) ? h->gn.iterations : 0;  // the BA is not called otherwise (:838)
    counts->n_replayed = h->cnt[7];
  }
  if (gn) {
    gn->iterations = h->gn.iterations;
    gn->err = h->gn.err;
    gn->delta_err = h->gn.delta_err;
    gn->delta_norm = h->gn.delta_norm;
    gn->cnt_invalid = h->gn.cnt_invalid;
    gn->is_nan = h->gn.is_nan;
  }
  if (h->flags) {
    if (h->flags & 1) VO_FAIL(c, VO_ERR_NAN_AXAY, "ax ay nan");
    if (h->flags & 2) VO_FAIL(c, VO_ERR_NAN_PATCH, "I0 I1 / du0 dv0 nan");
    VO_FAIL(c, VO_ERR_NAN_UPDATE, "dtu dtv nan");
  }
  return VO_OK;
}
