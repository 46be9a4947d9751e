// frame_fused.hip — the per-point part of the steady-state stereo frame in ONE launch.
//
// Steps [3] .. [5] of StereoVO::trackStereoImages (core/visual_odometry/stereo_vo/stereo_vo.cpp)
//   [3]   prior pixels + patch scale                       :483-522
//   [4]   trackWithPrior  I0_L -> I1_L  (+ validity)       :531-538, feature_tracker.cpp:186-197
//   [4-1] trackWithScale refinement     (+ validity)       :549-558, feature_tracker.cpp:236-504
//   [5]   trackWithPrior  I1_L -> I1_R  (+ validity)       :564-571
// only ever combine data of ONE feature (the reference's std::vector compactions between the
// steps, landmark.cpp:291-332, just drop rejected features), so a feature is one gfx950
// wavefront that walks through all four steps; the compaction happens once, at the end, in index
// order (prologue of the GN launch, gn_pose.hip). Every step of the frame is latency-bound on a handful of
// stragglers (a KLT level that runs all 30 iterations, an IC point that does not converge);
// as separate launches the frame pays the slowest feature of EVERY step plus the launch gaps,
// fused it pays the slowest feature once.
//
// The one cross-feature dependency is the never-reset tap state of trackWithScale
// (ic_refine.hip): a feature whose IC window leaves the image ("touched") is deferred after its
// pass-1 refinement; frame_replay_kernel brings the touched features to the reference-exact
// state (ic_replay) and finishes step [5] for each right after its recomputation. If that replay
// asks for the sequential fallback, frame_fallback_kernel does the same work one run per wavefront.
#include "ic_device.hpp"
#include "klt_device.hpp"
#include "vo_kernels.hpp"

struct FrameArgs {
  vo_level L0[VO_MAX_LEVELS];  // previous left
  vo_level L1[VO_MAX_LEVELS];  // current left
  vo_level R1[VO_MAX_LEVELS];  // current right
  int max_level;               // effective OpenCV maxLevel
  int max_level_bwd;           // effective maxLevel of the candidates' backward track (maxLevel - 1 requested)
  int n;
  int wg_off;                  // workgroup b of the launch is feature / candidate b + wg_off (the candidates as a launch of their own)
  int *cand_done;              // non-null (candidates' own launch): results are written through and every candidate workgroup
                               // counts itself here when it is done — the BA launch, running meanwhile, joins on the count
  int n_new;                   // new-point candidates (step [10])
  const float *pts_new;        // [n_new][2]
  const uint8_t *cand_has;     // closed step [10]: candidate j = best keypoint of bin j, absent where cand_has[j] == 0
  float *new_r;                // out [n_new][2] forward result
  uint8_t *m_new;              // out [n_new]    trackBidirection mask
  float thres_bidir;
  const float *Xp, *pts_l0, *pts_r0;
  const uint8_t *lm_flags;     // [n] bit 0 = lm->isTriangulated(); null = every landmark is
  float T_cp[16], T_rl[16], Kl[4], Kr[4];
  int world;        // the reference's own data flow (stereo_vo.cpp:475-522): Xp holds WORLD points, T_cp = T_cw_prior =
  float T_pw2[4];   // inverseSE3_f(T_wp * dT_pc_prev), T_pw2 = row 2 of T_pw (X_l0(2) of the patch scale, :497-498)
  int W, H;
  float thres_err;
  int strict;
  float *scale;     // out [n]    patch scale (= ic.scale)
  float *k1;        // out [n][2] step [4] result (= ic.pts_prior)
  float *pr_prior;  // out [n][2] prior right pixels
  float *pl1;       // out [n][2] left pixels:  [4] result, replaced by the refined position when [4-1] accepts
  float *pr1;       // out [n][2] right pixels: the prior, replaced by the [5] result for features that reach [5]
  uint8_t *stage;   // out [n]    0 lost in [4], 1 in [4-1], 2 in [5], 3 survivor
  IcArgs ic;        // tap records / touched list / replay control (index space = input index)
  int *sync;        // hand-shakes with the concurrent replay (vo_frame_state::sync)
  int sync_signal;  // frame_fallback_kernel counts its workgroups in sync[1] (concurrent replay: the BA launch waits for it)
#ifdef FRAME_STAMP
  int *dbg;         // [n + n_new][8] diagnostic stamps (s_memrealtime, 100 MHz) and iteration counts
#endif
};

template <int WIN>
struct FrameShared {
  uint32_t tt[KltCfg<WIN>::TT_H * KltCfg<WIN>::TT_WD];
  uint32_t tj[KltCfg<WIN>::TJ_H * KltCfg<WIN>::TJ_WD];
  IcShared ic;
};

__device__ __forceinline__ bool frame_in_image(float x, float y, int W, int H) {
  const float offset = 3.0f;
  return !(x < offset || y < offset || x >= W - offset || y >= H - offset);
}

// trackWithPrior validity, feature_tracker.cpp:191-197
__device__ __forceinline__ bool frame_klt_valid(const KltResult &r, int W, int H, float thres_err) {
  return r.status > 0 && r.x > 0 && r.x < W && r.y > 0 && r.y < H && r.err <= thres_err;
}

// steps after [4-1] for one feature: `ok` / (rx, ry) are the refinement's mask and position
template <int WIN>
__device__ __forceinline__ void frame_tail(const FrameArgs &a, int i, int ok, float rx, float ry, float kx, float ky,
                                           float prx, float pry, uint32_t *s_tt, uint32_t *s_tj, int lane) {
  int stage = 1;
  float ox = prx, oy = pry;
  if (ok) {
    stage = 2;
    // [5] l1 -> r1 from the refined left pixel, initial flow = prior right pixel ({} criteria, {} minEig)
    const KltResult k2 = klt_point<WIN>(a.L1, a.R1, a.max_level, VO_KLT_USE_INITIAL_FLOW, 30, 0.01 * 0.01, 0.f, rx, ry,
                                        prx, pry, s_tt, s_tj, lane);
    ox = k2.x;  // reported for every feature that entered step [5], valid or not
    oy = k2.y;
    if (frame_klt_valid(k2, a.W, a.H, a.thres_err)) stage = 3;
  }
  if (lane == 0) {
    a.pl1[2 * i] = ok ? rx : kx;  // the step [4] result stays when the refinement is rejected
    a.pl1[2 * i + 1] = ok ? ry : ky;
    a.pr1[2 * i] = ox;
    a.pr1[2 * i + 1] = oy;
    a.stage[i] = (uint8_t)stage;
  }
}

// Workgroups 0..n-1 are the tracked features (steps [3],[4],[4-1],[5]); workgroups n..n+n_new-1 are
// the new-point candidates of step [10] (trackBidirection, feature_tracker.cpp:60-83: forward
// l1 -> r1, backward r1 -> l1 at maxLevel-1 with the candidates as initial flow, validity mask).
// Both roles are "KLT, something in between, KLT", so the kernel is a two-pass loop around ONE
// inlined copy of klt_point (its code is ~3000 instructions; one copy per call site would not fit
// the instruction cache).
#ifndef IC_TAIL_PRIO
#define IC_TAIL_PRIO 0
#endif
#ifndef IC_REPLAY_VGPRS
#define IC_REPLAY_VGPRS 288
#endif
#ifndef FRAME_WAVES_PER_EU
#define FRAME_WAVES_PER_EU 2
#endif
template <int WIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FRAME_WAVES_PER_EU, FRAME_WAVES_PER_EU))) void frame_track_kernel(FrameArgs a) {
  __shared__ FrameShared<WIN> sh;
  if ((int)blockIdx.x + a.wg_off >= a.n + a.n_new) return;
  // Issue priority of the two roles (measured, -DFRAME_PRIO_FEAT / _CAND / -DFRAME_WAVES_PER_EU sweeps, 400 frames each,
  // run-to-run noise ~1.5 %): candidates one step above the features 162-165 us per launch, equal 169-171 us, features
  // above 170 us — the candidates are dispatched last (3150 workgroups, 2048 resident at 211 VGPRs) and the launch ends
  // when the last of them does. Capping the kernel at 168 VGPRs (3 wavefronts per SIMD, everything resident at once)
  // costs 38 spilled VGPRs: 159-166 us with the features above, 177-182 us otherwise — no better than this.
#ifndef FRAME_PRIO_FEAT
#define FRAME_PRIO_FEAT 0
#define FRAME_PRIO_CAND 1
#endif
  if ((int)blockIdx.x + a.wg_off >= a.n)
    __builtin_amdgcn_s_setprio(FRAME_PRIO_CAND);
  else
    __builtin_amdgcn_s_setprio(FRAME_PRIO_FEAT);
  // Feature i is workgroup i: consecutive features (bucket order, i.e. image neighbours) go round-robin
  // over the 8 XCDs. The XCD-aware alternative — XCD x takes the x-th eighth of the list, a horizontal
  // band of the image that stays in its own L2 — was measured at 3840x2160 / 8000 features
  // (tools/tools_config5.py): 795 us instead of 603 us per launch. The kernel is bound by the slowest
  // wavefronts, slow features cluster in image regions (borders, low texture), and a band per XCD
  // concentrates them on one eighth of the chip; round-robin spreads them.
  const int i = blockIdx.x + a.wg_off;
  const int lane = threadIdx.x;
  const bool feat = i < a.n;
  const int j = i - a.n;  // candidate index (new-point role)
#ifdef FRAME_STAMP
#define FSTAMP(k) if (lane == 0) a.dbg[8 * i + (k)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#define FNOTE(k, v) if (lane == 0) a.dbg[8 * i + (k)] = (v);
#else
#define FSTAMP(k)
#define FNOTE(k, v)
#endif
  FSTAMP(0)
  float p0x, p0y, ix, iy;  // KLT operands of the current pass
  float prx = 0.f, pry = 0.f, scale = 1.f, l0x = 0.f, l0y = 0.f;
  if (feat) {
    // ---- [3] priors (stereo_vo.cpp:483-522); every lane computes the same values ----
    l0x = a.pts_l0[2 * i];
    l0y = a.pts_l0[2 * i + 1];
    float plx = l0x, ply = l0y;
    bool from_prev = true;  // prior = pts_l0 / pts_r0: untriangulated landmark (:515-519) or projection unusable (:505-510)
    if (!a.lm_flags || (a.lm_flags[i] & VO_LM_TRIANGULATED)) {
      const float *Xi = a.Xp + 3 * i;
      float Xl[3], Xr[3];
      if (a.world) {
        // `R * X + t` on fixed-size Eigen types: the product first, its 3-term dot products as e0 + (e1 + e2), then + t
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Xl[r] = (a.T_cp[r * 4 + 0] * Xi[0] + (a.T_cp[r * 4 + 1] * Xi[1] + a.T_cp[r * 4 + 2] * Xi[2])) + a.T_cp[r * 4 + 3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Xr[r] = (a.T_rl[r * 4 + 0] * Xl[0] + (a.T_rl[r * 4 + 1] * Xl[1] + a.T_rl[r * 4 + 2] * Xl[2])) + a.T_rl[r * 4 + 3];
        const float Xl0z = (a.T_pw2[0] * Xi[0] + (a.T_pw2[1] * Xi[1] + a.T_pw2[2] * Xi[2])) + a.T_pw2[3];
        scale = Xl0z / Xl[2];
      } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Xl[r] = ((a.T_cp[r * 4 + 0] * Xi[0] + a.T_cp[r * 4 + 1] * Xi[1]) + a.T_cp[r * 4 + 2] * Xi[2]) + a.T_cp[r * 4 + 3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Xr[r] = ((a.T_rl[r * 4 + 0] * Xl[0] + a.T_rl[r * 4 + 1] * Xl[1]) + a.T_rl[r * 4 + 2] * Xl[2]) + a.T_rl[r * 4 + 3];
        scale = Xi[2] / Xl[2];
      }
      const float izl = 1.0f / Xl[2], izr = 1.0f / Xr[2];
      plx = a.Kl[0] * Xl[0] * izl + a.Kl[2];
      ply = a.Kl[1] * Xl[1] * izl + a.Kl[3];
      prx = a.Kr[0] * Xr[0] * izr + a.Kr[2];
      pry = a.Kr[1] * Xr[1] * izr + a.Kr[3];
      from_prev = !frame_in_image(plx, ply, a.W, a.H) || !frame_in_image(prx, pry, a.W, a.H) || (double)Xl[2] < 0.1 ||
                  (double)Xr[2] < 0.1;
    }
    if (from_prev) {
      plx = l0x;
      ply = l0y;
      prx = a.pts_r0[2 * i];
      pry = a.pts_r0[2 * i + 1];
    }
    p0x = l0x;
    p0y = l0y;
    ix = plx;
    iy = ply;
  } else {
    if (a.cand_has && !a.cand_has[j]) {  // a bin without a keypoint
      if (lane == 0) {
        if (a.cand_done) {
          ic_st8(&a.m_new[j], 0);
          __builtin_amdgcn_s_waitcnt(0);
          __hip_atomic_fetch_add(a.cand_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          a.m_new[j] = 0;
        }
      }
      return;
    }
    p0x = ix = a.pts_new[2 * j];
    p0y = iy = a.pts_new[2 * j + 1];
  }
  KltResult first;  // pass 0 result: [4] of a feature / forward track of a candidate
  first.x = first.y = first.err = 0.f;
  first.status = 0;
  int rf_ok = 0;
  float rfx = 0.f, rfy = 0.f;
#pragma nounroll
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: feature [4] l0 -> l1 (initial flow = prior; {} criteria -> 30 / 0.01, {} minEig -> 0)
    //         candidate forward l1 -> r1 (no initial flow, 30 / 0.01, minEig 1e-4)
    // pass 1: feature [5] l1 -> r1 from the refined pixel (initial flow = prior right pixel)
    //         candidate backward r1 -> l1 at maxLevel - 1 (initial flow = the candidate)
    const vo_level *I = feat ? (pass == 0 ? a.L0 : a.L1) : (pass == 0 ? a.L1 : a.R1);
    const vo_level *J = feat ? (pass == 0 ? a.L1 : a.R1) : (pass == 0 ? a.R1 : a.L1);
    const int lvl = (!feat && pass == 1) ? a.max_level_bwd : a.max_level;
    const int flags = (!feat && pass == 0) ? 0 : VO_KLT_USE_INITIAL_FLOW;
    const float min_eig = (!feat && pass == 0) ? 1e-4f : 0.f;
    const KltResult k = klt_point<WIN>(I, J, lvl, flags, 30, 0.01 * 0.01, min_eig, p0x, p0y, ix, iy, sh.tt, sh.tj, lane);
#ifdef FRAME_STAMP
    FSTAMP(pass == 0 ? 1 : 3)
    FNOTE(pass == 0 ? 4 : 5, k.iters)
#endif
    if (!feat) {
      if (pass == 0) {
        first = k;
        p0x = k.x;  // backward: from the forward result, initial flow = the candidate (ix, iy unchanged)
        p0y = k.y;
        continue;
      }
      // trackBidirection validity, feature_tracker.cpp:74-83
      const float dx = k.x - ix, dy = k.y - iy;
      const float dist2 = dx * dx + dy * dy;
      const float thres2 = a.thres_bidir * a.thres_bidir;
      bool m = first.x > 3 && first.x < a.W - 3 && first.y > 3 && first.y < a.H - 3;
      m = m && first.status && k.status && first.err <= a.thres_err && k.err <= a.thres_err && dist2 <= thres2;
      if (lane == 0) {
        if (a.cand_done) {  // (read by a kernel that is running: write-through, then the count)
          ic_store<true>(&a.new_r[2 * j], first.x);
          ic_store<true>(&a.new_r[2 * j + 1], first.y);
          ic_st8(&a.m_new[j], (uint8_t)(m ? 1 : 0));
          __builtin_amdgcn_s_waitcnt(0);
          __hip_atomic_fetch_add(a.cand_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          a.new_r[2 * j] = first.x;
          a.new_r[2 * j + 1] = first.y;
          a.m_new[j] = m ? 1 : 0;
        }
      }
      return;
    }
    if (pass == 1) {
      // ---- end of [5] ----
      const int stage = frame_klt_valid(k, a.W, a.H, a.thres_err) ? 3 : 2;
      if (lane == 0) {
        a.pl1[2 * i] = rfx;
        a.pl1[2 * i + 1] = rfy;
        a.pr1[2 * i] = k.x;  // reported for every feature that entered step [5], valid or not
        a.pr1[2 * i + 1] = k.y;
        a.stage[i] = (uint8_t)stage;
      }
      return;
    }
    // ---- feature, after [4] ----
    first = k;
    // StereoLandmarkTracking(lmtrack_prev, mask_l0l1), landmark.cpp:305: mask && isAlive() && isTracked()
    const bool valid1 = frame_klt_valid(k, a.W, a.H, a.thres_err) && !(a.lm_flags && (a.lm_flags[i] & VO_LM_DROPPED));
    // Everything the strict-border replay reads or rewrites goes through to memory (ic_store<true>): that kernel runs
    // next to this one on other XCDs, and this kernel's dirty L2 lines would otherwise reach memory when IT ends.
    if (lane == 0) {
      ic_store<true>(&a.scale[i], scale);
      ic_store<true>(&a.k1[2 * i], k.x);
      ic_store<true>(&a.k1[2 * i + 1], k.y);
      ic_store<true>(&a.pr_prior[2 * i], prx);
      ic_store<true>(&a.pr_prior[2 * i + 1], pry);
    }
    const IcTaps tp = ic_make_taps(lane);
    IcState S;
    ic_state_clear(S);
    int cls = 0, touched = 0, n_iter = 0;
    float lpx = 0.f, lpy = 0.f;
    IcResult rf;
    rf.cls = 0;
    rf.ok = 0;
    rf.x = k.x;
    rf.y = k.y;
    rf.err_flag = 0;
    if (valid1) {
      // ---- [4-1] refinement of the left pixel (pass 1: taps outside the image are masked) ----
      rf = ic_point<false>(a.L0[0], a.L1[0], tp, l0x, l0y, k.x, k.y, scale, lane, sh.ic, S, touched, lpx, lpy, n_iter);
      cls = rf.cls;
    }
    FSTAMP(2)
    FNOTE(6, n_iter)
    const int any_t = __any(touched);
    if (lane == 0 && rf.err_flag) atomicOr(a.ic.flags, rf.err_flag);
    const bool deferred = a.strict && any_t;  // frame_replay_kernel (or frame_fallback_kernel) finishes this feature
    if (a.strict) {
      // records for the replay; pass-1 results of every point (the replay overwrites touched ones)
      ic_store_records<true>(a.ic, i, lane, tp, S, cls);
      if (lane == 0) {
        ic_store<true>(&a.ic.touched[i], (uint8_t)(any_t ? 1 : 0));
        ic_store<true>(&a.ic.cls[i], (uint8_t)cls);
        ic_store<true>(&a.ic.last_pu[2 * i], lpx);
        ic_store<true>(&a.ic.last_pu[2 * i + 1], lpy);
        ic_store<true>(&a.ic.pts_track[2 * i], rf.x);
        ic_store<true>(&a.ic.pts_track[2 * i + 1], rf.y);
        ic_store<true>(&a.ic.mask[i], (uint8_t)rf.ok);
        if (any_t) {
          const int slot = atomicAdd(&a.ic.jac[IC_JAC_NT], 1);
          ic_store<true>(&a.ic.tlist[slot], i);
          if (a.ic.tl2)  // concurrent replay: the entry says which frame it belongs to
            __hip_atomic_store(&a.ic.tl2[slot], ((unsigned long long)(unsigned)a.ic.epoch << 32) | (unsigned)i, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!deferred && (!valid1 || !rf.ok)) {
      if (lane == 0) {
        a.pl1[2 * i] = rf.x;  // the step [4] result unless the refinement accepted
        a.pl1[2 * i + 1] = rf.y;
        a.pr1[2 * i] = prx;
        a.pr1[2 * i + 1] = pry;
        a.stage[i] = valid1 ? 1 : 0;
      }
    }
    if (a.strict && a.ic.p1e) {
      // Concurrent replay: it may use this feature's pass-1 data from here on. All the stores above that it depends on
      // are write-through: wait for them, then stamp and count. (The stream-ordered replay needs neither.)
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) {
        ic_store<true>(&a.ic.p1e[i], a.ic.epoch);  // this feature's pass-1 data are in memory
        atomicAdd(&a.ic.p1_word[(i & (IC_P1_SHARDS - 1)) * IC_P1_STRIDE], 1);
      }
    }
    // (a deferred feature's outputs — pl1, pr1, stage — are written by the replay's tail, never here: two kernels
    // on two XCDs must not both own a byte)
    if (!valid1 || !rf.ok || deferred) return;
    rf_ok = rf.ok;
    rfx = rf.x;
    rfy = rf.y;
    p0x = rf.x;
    p0y = rf.y;
    ix = prx;
    iy = pry;
  }
  (void)rf_ok;
}

// Gate of the "gated" arrangement (strict mode 5, what mode 4 uses for a frame that is expected to replay nothing): the
// stream-ordered replay and its fallback sit on the replay stream behind this one wavefront, which waits until every
// feature of the frame kernel is past pass 1 — so the two (normally idle) launches run under the frame kernel's tail
// instead of between it and the BA launch, which joins on the device as in mode 3.
__global__ __launch_bounds__(64) void frame_gate_kernel(IcArgs a) {
  int polls = 0;
  while ((int)(ic_p1_count(a, threadIdx.x) - a.p1_target) < 0) {
    if (++polls > IC_SPIN_LIMIT) {  // the frame kernel is not running next to us: reported by the BA launch (flag 8)
      if (threadIdx.x == 0) atomicAdd(&a.p1_word[-IC_P1_STRIDE + 2], 1);
      return;
    }
    __builtin_amdgcn_s_sleep(32);  // ~1 us
  }
}

// strict border: replay of the touched features, then their step [5]
// 288 registers at most: 512 per SIMD minus one frame-kernel wavefront (224 with the allocation granule), so that a
// replay wavefront and a frame-kernel wavefront share a SIMD. With more (294 were used when unconstrained) a resident
// replay wavefront keeps the whole SIMD to itself — the concurrent pool then costs the frame kernel a quarter of the
// chip, or, started behind it, finds no room until SIMDs drain completely.
template <int WIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(IC_REPLAY_VGPRS))) void frame_replay_kernel(FrameArgs a) {
  __shared__ IcReplayShared rs;
  __shared__ uint32_t s_tt[KltCfg<WIN>::TT_H * KltCfg<WIN>::TT_WD];
  __shared__ uint32_t s_tj[KltCfg<WIN>::TJ_H * KltCfg<WIN>::TJ_WD];
  const int lane = threadIdx.x;
  __builtin_amdgcn_s_setprio(3);
  // step [5] of a feature follows its (re)computation at once: by then its record is published, so
  // nobody waits for this wavefront, and the feature is final unless an input changes later (rare;
  // the hook then runs again and overwrites the outputs)
  if (a.ic.tl2) {
    // next to the frame kernel (frame_launch, strict == 3): what that kernel wrote is read past the caches
    auto tail = [&](int i, const IcResult &r) {
      // step [5] of a replayed feature is off the dependency chains (its record is published): it must not starve the
      // frame-kernel wavefront it shares the SIMD with — that one would become the frame kernel's last
      __builtin_amdgcn_s_setprio(IC_TAIL_PRIO);
      frame_tail<WIN>(a, i, r.ok, r.x, r.y, ic_ldf(&a.k1[2 * i]), ic_ldf(&a.k1[2 * i + 1]), ic_ldf(&a.pr_prior[2 * i]),
                      ic_ldf(&a.pr_prior[2 * i + 1]), s_tt, s_tj, lane);
      __builtin_amdgcn_s_setprio(3);
    };
    (void)ic_replay<true>(a.ic, rs, lane, tail);
  } else {
    auto tail = [&](int i, const IcResult &r) {
      frame_tail<WIN>(a, i, r.ok, r.x, r.y, a.k1[2 * i], a.k1[2 * i + 1], a.pr_prior[2 * i], a.pr_prior[2 * i + 1], s_tt,
                      s_tj, lane);
    };
    (void)ic_replay(a.ic, rs, lane, tail);
  }
#ifdef IC_STAMP
  if (lane == 0) atomicMax(&a.ic.tlist[IC_DBG_OFF + 2], (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff));
#endif
}

// sequential fallback of the replay (returns at once unless it was requested): one wavefront walks a
// run of features, and step [5] follows each touched feature it recomputes
template <int WIN>
__global__ __launch_bounds__(64) void frame_fallback_kernel(FrameArgs a) {
  __shared__ IcShared sh;
  __shared__ uint32_t s_tt[KltCfg<WIN>::TT_H * KltCfg<WIN>::TT_WD];
  __shared__ uint32_t s_tj[KltCfg<WIN>::TJ_H * KltCfg<WIN>::TJ_WD];
  const int lane = threadIdx.x;
  // (a small grid that strides over the features: when nothing was asked for — the usual case — the launch is over
  // in the time it takes to start and drain it, which a workgroup per feature is not)
  bool work = a.ic.jac[IC_JAC_OVF] != 0;
  if (work && a.sync_signal && a.ic.p1e) {
    // Concurrent replay: this kernel is ordered behind the replay pool, not behind the frame kernel, whose pass-1
    // data it is about to read. The pool normally ends after the frame kernel's last pass 1; it ends early only when
    // it gave up waiting (kernels serialised across the queues by a tool): then wait here, bounded, and report.
    int polls = 0;
    while ((int)(ic_p1_count(a.ic, lane) - a.ic.p1_target) < 0) {
      if (++polls > IC_SPIN_LIMIT) {
        if (lane == 0) atomicOr(a.ic.flags, 8);  // reported as VO_ERR_HIP by vo_stereo_frame_result
        work = false;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (work) {
    auto tail = [&](int p, const IcResult &r) {
      frame_tail<WIN>(a, p, r.ok, r.x, r.y, a.k1[2 * p], a.k1[2 * p + 1], a.pr_prior[2 * p], a.pr_prior[2 * p + 1], s_tt,
                      s_tj, lane);
    };
    for (int i = blockIdx.x; i < a.n; i += gridDim.x) {
      __syncthreads();
      ic_strict_run(a.ic, sh, i, a.n, lane, tail);
    }
  }
  // This kernel is stream-ordered behind the replay: when all of its workgroups have counted, every touched feature
  // is final. The BA launch on the main stream waits for the count (no HIP event between the streams). The replay's
  // own stores reached memory when that kernel ended; only a workgroup that did fallback work has stores to release
  // (a release fence writes the XCD's whole L2 back: not something 1500 idle workgroups should each do).
  if (a.sync_signal) {
    if (work) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) atomicAdd(&a.sync[1], 1);
  }
}

#ifdef FRAME_STAMP
static int *vo_frame_dbg_ptr;
extern "C" int vo_debug_frame_stamps(vo_ctx *c, int *dst, int rows) {
  if (!vo_frame_dbg_ptr) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  VO_CHECK_HIP(c, hipMemcpy(dst, vo_frame_dbg_ptr, sizeof(int) * 8 * (size_t)rows, hipMemcpyDeviceToHost));
  return VO_OK;
}
#endif

// ---- host side ---------------------------------------------------------------------
static int vo_frame_fallback_grid(int n) { return n < 128 ? n : 128; }
// phase 0: the per-feature kernel; phase 1: the strict-border replay (nothing otherwise). Two phases so that
// the caller can feed other streams while the long first kernel is already running.
template <int WIN>
static void frame_launch(vo_ctx *c, const FrameArgs &a, int phase, int p1_target, int done_target, int conc_grid, int split_cands,
                         int *cand_done) {
  if (phase == 0) {
    vo_prof_begin(c, VO_K_KLT);
    hipLaunchKernelGGL(frame_track_kernel<WIN>, dim3(split_cands ? a.n : a.n + a.n_new), dim3(64), 0, c->stream, a);
    vo_prof_end(c);
    return;
  }
  if (phase == 2) {  // the candidates as a launch of their own (on c->stream: the caller's side stream, behind the detector)
    if (a.n_new > 0) {
      FrameArgs b = a;
      b.wg_off = a.n;
      b.cand_done = cand_done;
      hipLaunchKernelGGL(frame_track_kernel<WIN>, dim3(a.n_new), dim3(64), 0, c->stream, b);
    }
    return;
  }
  if (a.strict == 2) {  // validation mode: the sequential fallback does all the work, on the main stream
    (void)hipMemsetAsync(&a.ic.jac[IC_JAC_OVF], 1, sizeof(int), c->stream);
    hipLaunchKernelGGL(frame_fallback_kernel<WIN>, dim3(a.n < 1024 ? a.n : 1024), dim3(64), 0, c->stream, a);
  } else if (a.strict == 1) {  // the replay stream-ordered behind the frame kernel
    vo_prof_begin(c, VO_K_IC);
    hipLaunchKernelGGL(frame_replay_kernel<WIN>, dim3(a.n < IC_JGRID ? a.n : IC_JGRID), dim3(64), 0, c->stream, a);
    hipLaunchKernelGGL(frame_fallback_kernel<WIN>, dim3(vo_frame_fallback_grid(a.n)), dim3(64), 0, c->stream, a);
    vo_prof_end(c);
  } else if (a.strict == 3) {
    // The replay runs on its own stream NEXT TO the frame kernel, as a pool of IC_CONC_GRID resident workgroups that
    // pick the touched features up as the frame kernel lists them (ic_replay<true>): a dependency chain starts when
    // its members are through pass 1, not when the frame kernel's last wavefront is. No HIP event joins the two
    // streams (a cross-queue event wait costs ~30 us on this stack): the pool synchronises with the frame kernel
    // through the epoch stamps and the pass-1 count, and the BA launch (main stream, behind the frame kernel) polls
    // the count of finished workgroups of the fallback kernel, which is stream-ordered behind the pool. Both counts
    // are cumulative. (Earlier form: the whole replay behind a one-wavefront gate that waited for the LAST feature's
    // pass 1 — no faster than stream order, the pass-1 stragglers arrive 15-30 us before the kernel ends.)
    FrameArgs b = a;
    b.sync_signal = 1;
    hipStream_t main_stream = c->stream;
    c->stream = c->stream3;  // (the event brackets follow c->stream)
    vo_prof_begin(c, VO_K_IC);
    hipLaunchKernelGGL(frame_replay_kernel<WIN>, dim3(b.n < conc_grid ? b.n : conc_grid), dim3(64), 0, c->stream3, b);
    vo_prof_end(c);
    c->stream = main_stream;
    hipLaunchKernelGGL(frame_fallback_kernel<WIN>, dim3(vo_frame_fallback_grid(b.n)), dim3(64), 0, c->stream3, b);
  } else if (a.strict == 5) {
    FrameArgs b = a;
    b.sync_signal = 1;
    b.ic.tl2 = nullptr;  // (the replay itself is the stream-ordered one: everything it reads is complete behind the gate)
    hipLaunchKernelGGL(frame_gate_kernel, dim3(1), dim3(64), 0, c->stream3, a.ic);
    hipStream_t main_stream = c->stream;
    c->stream = c->stream3;
    vo_prof_begin(c, VO_K_IC);
    hipLaunchKernelGGL(frame_replay_kernel<WIN>, dim3(b.n < IC_JGRID ? b.n : IC_JGRID), dim3(64), 0, c->stream3, b);
    vo_prof_end(c);
    c->stream = main_stream;
    hipLaunchKernelGGL(frame_fallback_kernel<WIN>, dim3(vo_frame_fallback_grid(b.n)), dim3(64), 0, c->stream3, b);
  }
  (void)p1_target;
  (void)done_target;
  // (the frame's one compaction and the control-block reset are the prologue of the GN launch, gn_pose.hip)
}

// the window sizes of the reference's configurations (config/**.yaml: 13, 15, 21) and 31
int vo_frame_fused_supported(int win) { return win == 13 || win == 15 || win == 21 || win == 31; }

int vo_frame_fused_enqueue(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                           const float *d_l0, const float *d_r0, const float *d_X, const uint8_t *d_flags, int n,
                           const float T_cp[16], const float T_rl[16], const float *d_new, int n_new,
                           const vo_frame_fused_bufs &b, int phase, const float *T_pw) {
  if (n <= 0) return VO_OK;
  const int slots[3] = {slot_l0, slot_l1, slot_r1};
  for (int s : slots)
    if (s < 0 || s >= c->cfg.n_slots || c->slots[s].n_levels <= 0) VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
  const vo_pyramid &P0 = c->slots[slot_l0], &P1 = c->slots[slot_l1], &P2 = c->slots[slot_r1];
  if (P0.w != P1.w || P0.h != P1.h || P0.w != P2.w || P0.h != P2.h) VO_FAIL(c, VO_ERR_SIZE, "image size mismatch");
  if (prm->max_level < 0 || prm->win <= 2) VO_FAIL(c, VO_ERR_INVALID, "maxLevel >= 0 && winSize > 2 violated");
  FrameArgs a;
  memset(&a, 0, sizeof(a));
  const int eff = vo_pyr_levels_host(P0.w, P0.h, prm->win, prm->max_level);
  for (const vo_pyramid *P : {&P0, &P1, &P2}) VO_NEED_LEVELS(c, *P, eff);
  for (int s : slots)
    if (vo_slot_acquire(c, s) < 0) return VO_ERR_HIP;
  for (int l = 0; l <= eff; ++l) {
    a.L0[l] = P0.lv[l];
    a.L1[l] = P1.lv[l];
    a.R1[l] = P2.lv[l];
  }
  a.max_level = eff;
  if (n_new > 0) {
    if (prm->max_level - 1 < 0) VO_FAIL(c, VO_ERR_INVALID, "trackBidirection needs max_level >= 1");
    int effb = vo_pyr_levels_host(P0.w, P0.h, prm->win, prm->max_level - 1);
    a.max_level_bwd = effb < eff ? effb : eff;
  }
  a.n = n;
  a.n_new = n_new;
  a.pts_new = d_new;
  a.new_r = b.new_r;
  a.m_new = b.m_new;
  a.cand_has = b.cand_has;
  a.thres_bidir = prm->thres_bidirection;
#ifdef FRAME_STAMP
  {
    static int *dbg = nullptr;
    if (!dbg) (void)vo_dev_malloc(c, (void **)&dbg, sizeof(int) * 8 * (size_t)(2 * c->cfg.max_points));
    if (phase == 0) (void)hipMemsetAsync(dbg, 0, sizeof(int) * 8 * (size_t)(n + n_new), c->stream);
    a.dbg = dbg;
    vo_frame_dbg_ptr = dbg;
  }
#endif
  a.Xp = d_X;
  a.pts_l0 = d_l0;
  a.pts_r0 = d_r0;
  a.lm_flags = d_flags;
  memcpy(a.T_cp, T_cp, sizeof(a.T_cp));
  memcpy(a.T_rl, T_rl, sizeof(a.T_rl));
  a.world = T_pw ? 1 : 0;
  if (T_pw) memcpy(a.T_pw2, T_pw + 8, sizeof(a.T_pw2));
  memcpy(a.Kl, prm->Kl, sizeof(a.Kl));
  memcpy(a.Kr, prm->Kr, sizeof(a.Kr));
  a.W = prm->width;
  a.H = prm->height;
  a.thres_err = prm->thres_err;
  a.strict = c->frame_strict_now;  // 0 masked taps, 1 parallel replay (+ fallback), 2 sequential replay only, 3 concurrent
  a.scale = b.scale;
  a.k1 = b.k1;
  a.pr_prior = b.pr_prior;
  a.pl1 = b.pl1;
  a.pr1 = b.pr1;
  a.stage = b.stage;
  int rc = vo_ic_frame_args(c, slot_l0, slot_l1, &a.ic, b.ctl, a.strict != 0);
  if (rc) return rc;
  a.ic.pts0 = d_l0;
  a.ic.scale = b.scale;
  a.ic.pts_prior = b.k1;
  a.ic.pts_track = b.ref;
  a.ic.mask = b.m2;
  a.ic.touched = b.touched;
  a.ic.cls = b.cls;
  a.ic.last_pu = b.lastpu;
  a.ic.n = n;
  a.sync = b.sync;
  // running totals of the two hand-shake counters: what they will read when this frame's share has arrived
  if (phase == 0 && a.strict) {
    if (a.strict == 3 || a.strict == 5) {
      vo_wrap_add(*b.sync_p1_target, n);    // one count per feature past pass 1
      vo_wrap_add(*b.sync_done_target, vo_frame_fallback_grid(n));  // one count per workgroup of the fallback kernel
    }
  }
  const int p1_target = *b.sync_p1_target, done_target = *b.sync_done_target;
  if (a.strict == 3 || a.strict == 5) {
    // concurrent replay: list entries and per-feature stamps carry the frame's epoch — the pass-1 target, which is
    // different for every frame (and not 0)
    a.ic.epoch = p1_target != 0 ? p1_target : 1;
    a.ic.p1_word = b.sync + IC_P1_STRIDE;  // (the shards follow the block's first line)
    a.ic.p1_target = p1_target;
  } else {
    a.ic.tl2 = nullptr;
    a.ic.p1e = nullptr;
  }
  const int cg_dbg = c->dbg[VO_DBG_CONC_GRID];  // (tests/test_frame_gpu.py: a pool smaller than the list; experiments)
  const int cg = cg_dbg > 0 ? cg_dbg : (b.conc_grid > 0 ? b.conc_grid : IC_CONC_GRID);
  switch (prm->win) {
    case 13: frame_launch<13>(c, a, phase, p1_target, done_target, cg, b.split_cands, b.cand_done); break;
    case 15: frame_launch<15>(c, a, phase, p1_target, done_target, cg, b.split_cands, b.cand_done); break;
    case 21: frame_launch<21>(c, a, phase, p1_target, done_target, cg, b.split_cands, b.cand_done); break;
    case 31: frame_launch<31>(c, a, phase, p1_target, done_target, cg, b.split_cands, b.cand_done); break;
    default: VO_FAIL(c, VO_ERR_INVALID, "fused frame kernel not instantiated for window %d", prm->win);
  }
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
