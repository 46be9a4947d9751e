// mono_gate.hpp — the tail of the mono frame (MonoVO::trackImage, mono_vo.cpp:838-879, :954-963) as a device
// function: it runs inside the GN launch, right after the iterations, so the frame needs no further launch.
#pragma once
#include "vo_internal.hpp"
#include "mvo_device.hpp"
#include "np_emit.hpp"

// ---- after the GN iterations of a mono frame: mask_motion, Sampson gate, stages, counts, result copy-out ----
struct MonoGateArgs {
  int n;
  const float *pts0, *k1, *ref;
  const uint8_t *m1, *m2, *ba_ok, *mG;
  const int32_t *C_orig;
  const vo_gn_dev_info *gn;
  float *dT;                // in: GN result T01 ; out: the prior when the 5-point fallback is needed
  float dT_prior[16], K[4];
  float thres_sampson;
  uint8_t *motion;          // scratch [n]
  uint8_t *stage;           // out
  float *pts1;              // out
  int *cnt;                 // [8]: in [0..2] n_klt, n_refine, n_ba and [3] replayed (GN prologue); out n_motion, n_final,
                            // need_five_point, [7] replayed
  const uint32_t *res_dev;  // packed result block -> res_host (pinned, device-visible)
  uint32_t *res_host;
  int res_words;
  VoNpArgs np;              // closed new-point step (mono_vo.cpp:977-1001): see np_emit.hpp; np.bins == 0: off
  // MonoVO (mono_vo.hip): the next track set is built here, behind everything else (mvo_advance_body); its stage / pixel /
  // new-point pointers are filled in from the fields above by the enqueue
  int adv_on;
  const int *hdr_flags;     // the frame's error flags (a frame that failed is the host's to finish)
  MvoAdvArgs adv;
};
__device__ __forceinline__ float mono_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);  // Eigen's unrolled 3-term redux
}
// Runs as the epilogue of gn_pose_kernel<false> (frame mode, mono): `nthr` lanes of one workgroup, n_ba = size of the
// BA set the solve just used. The caller has passed a __syncthreads() since the solve's last global stores.
// s_occ / s_wv: LDS scratch of the caller for the closed new-point step (np_emit.hpp).
template <int NTHR>
__device__ __forceinline__ void mono_gate_body(const MonoGateArgs &a, int tid, int n_ba, uint8_t *s_occ, int *s_wv) {
  constexpr int nthr = NTHR;
  __shared__ float sF[9];
  __shared__ int s_cnt[8];
  __shared__ int s_ok;
  if (tid < 8) s_cnt[tid] = 0;
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
    a.cnt[7] = a.cnt[3];  // features replayed by the strict-border pass (reported by the GN prologue)
  }
  // mask_motion: true for every refined feature, the BA's inlier mask for the BA set (:845, :872-879)
  for (int i = tid; i < a.n; i += nthr) a.motion[i] = (a.m1[i] && a.m2[i]) ? 1 : 0;
  __syncthreads();
  const int ok = s_ok;
  if (ok)
    for (int c = tid; c < n_ba; c += nthr) a.motion[a.C_orig[c]] = a.mG[c];
  __syncthreads();
  int c_klt = 0, c_ref = 0, c_mot = 0, c_fin = 0;
  for (int i = tid; i < a.n; i += nthr) {
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
    a.cnt[6] = 0;  // new points emitted (below)
  }
  if (a.np.bins > 0 && ok) {
    // ---- the new points of this frame: extractor_->updateWeightBin(lmtrack_final.pts1),
    // extractORBwithBinning_fast(I1) and trackBidirection(I1, I0, ...) (mono_vo.cpp:977-992) — the candidates were
    // found per bin before the frame and tracked by the frame kernel; here: which bins stayed empty, and their results.
    // (Without a pose from the BA the reference goes through the 5-point path first: the caller's, and so is this step.)
    if (a.np.cand_done) {  // the candidates' own launch (MonoVO's synchronous call) may still be running: bounded join
      if (tid == 0 && !vo_np_wait_candidates(a.np)) atomicOr(const_cast<int *>(a.hdr_flags), 8);
    }
    __syncthreads();  // the stages and pixels above are this workgroup's own stores
    const uint8_t *stg = a.stage;
    vo_np_emit(a.np, a.n, a.pts1, [&](int i) { return stg[i] == 4; }, tid, nthr, s_occ, s_wv, &a.cnt[6]);
  }
  // every header word is written by a launch of this frame (cnt[6]: compaction, gn / dT: GN, the rest
  // above), so the block needs no clearing. It goes to pinned host memory from here (all final: earlier
  // launches' stores, and this workgroup's own above)
  __syncthreads();
  for (int k = tid; k < a.res_words; k += nthr) a.res_host[k] = a.res_dev[k];
  if (a.adv_on)  // (cnt[6], the stages, pixels and new points: this workgroup's stores, behind the barrier above)
    mvo_advance_body<NTHR / 64>(a.adv, a.dT, !ok || *a.hdr_flags != 0, a.np.bins > 0 ? a.cnt[6] : 0, tid);
}

