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
// bidirectional mask, IC refinement — are one wavefront of ONE launch; the strict-border replay follows — stream-ordered
// (strict-border 1) or, as for stereo, as a pool of resident workgroups on a stream of its own NEXT TO the frame kernel
// (strict-border 3; 4 chooses per frame), joined by the BA launch on the device — then
// ONE more launch: the GN kernel in frame mode, whose prologue selects the BA set and whose epilogue is the
// Sampson gate (mono_gate.hpp); the host reads one packed block. stage[i] = number of gates feature i passed (1 tracked, 2 refined,
// 3 motion inlier or not part of the BA, 4 passed the Sampson gate).
#include "frame_state.hpp"
#include "ic_device.hpp"
#include "klt_device.hpp"
#include "mono_gate.hpp"
#include "vo_kernels.hpp"

struct MonoArgs {
  vo_level I0[VO_MAX_LEVELS], I1[VO_MAX_LEVELS];
  int max_level, n;
  const float *pts0, *Xw;
  const uint8_t *flags;
  int flag_mode;  // 0: operator flags (vo_hip.h); 1 / 2: a MonoVO track set's flags (VO_LM_*), the pose-only BA's class =
                  // triangulated / bundled landmarks (mono_vo.cpp:800-826)
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
  // closed new-point step: workgroups n .. n + n_new - 1 track the per-bin candidates (trackBidirection I1 -> I0)
  int n_new, max_level_bwd;
  const float *pts_new;     // [n_new][2] the table's pixels (image I1)
  const uint8_t *cand_has;  // [n_new] the bin holds a keypoint
  float *new_r;             // out [n_new][2] forward result (pixel in I0)
  uint8_t *m_new;           // out [n_new] trackBidirection mask
  int wg_off;               // workgroup b of the launch is feature / candidate b + wg_off (the candidates as a launch of their own)
  int *cand_done;           // non-null (candidates' own launch): results are written through and every candidate workgroup counts itself
};
// the replay's launches: the IC arguments and the hand-shake with the BA launch (vo_frame_state::sync)
struct MonoReplayArgs {
  IcArgs ic;
  int *sync;        // sync[1]: finished workgroups of mono_fallback_kernel, cumulative
  int sync_signal;  // concurrent replay: count there (the BA launch on the main stream waits for it)
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
  const int i = blockIdx.x + a.wg_off;
  if (i >= a.n + a.n_new) return;
  const int lane = threadIdx.x;
  // Workgroups n .. n + n_new - 1: the new-point candidate of bin j (mono_vo.cpp:989-991: trackBidirection(I1, I0,
  // pts1_new, ...), feature_tracker.cpp:39-86): forward I1 -> I0 (no initial flow, minEig 1e-4), backward I0 -> I1 at
  // maxLevel - 1 with the candidate as initial flow, validity mask. Same two-pass loop, same ONE copy of klt_point;
  // dispatched behind the features, one step up in issue priority (the launch ends when they do, as in the stereo kernel).
  const bool feat = i < a.n;
  const int j = i - a.n;
  if (!feat) {
    if (!a.cand_has[j]) {
      if (lane == 0) {
        if (a.cand_done) {  // (read by the BA launch, which may be running: write-through, then the count)
          ic_st8(&a.m_new[j], 0);
          __builtin_amdgcn_s_waitcnt(0);
          __hip_atomic_fetch_add(a.cand_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          a.m_new[j] = 0;
        }
      }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
  }
  const float p0x = feat ? a.pts0[2 * i] : a.pts_new[2 * j], p0y = feat ? a.pts0[2 * i + 1] : a.pts_new[2 * j + 1];
  int fl = feat ? a.flags[i] : 0;
  if (a.flag_mode && feat) {
    const int t = fl;
    fl = (t & VO_LM_BUNDLED) ? 1 : 0;
    if (t & (a.flag_mode == 2 ? VO_LM_BUNDLED : VO_LM_TRIANGULATED)) fl |= 2;
    if (t & VO_LM_DROPPED) fl |= VO_MONO_LM_DROPPED;
  }
  // ---- prior + patch scale (mono_vo.cpp:739-761) ----
  float Xp[3] = {0.f, 0.f, 0.f};
  float prx = p0x, pry = p0y, scale = 1.0f;
  const float *Xi = a.Xw + 3 * (feat ? i : 0);
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
    const bool from0 = (pass == 0) == feat;  // features: I0 -> I1 then back; candidates: I1 -> I0 then back
    const vo_level *I = from0 ? a.I0 : a.I1;
    const vo_level *J = from0 ? a.I1 : a.I0;
    const int lvl = (!feat && pass == 1) ? a.max_level_bwd : a.max_level;
    const int kflags = (!feat && pass == 0) ? 0 : VO_KLT_USE_INITIAL_FLOW;
    const float min_eig = (!feat && pass == 0) ? 1e-4f : 0.f;
    const KltResult k = klt_point<WIN>(I, J, lvl, kflags, 30, 0.01 * 0.01, min_eig, q0x, q0y, ix, iy, sh.tt, sh.tj, lane);
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
  if (!feat) {
    // trackBidirection validity, feature_tracker.cpp:74-83
    const float dx = bwd.x - p0x, dy = bwd.y - p0y;
    const float dist2 = dx * dx + dy * dy;
    const float thres2 = a.thres_bidir * a.thres_bidir;
    bool m = fwd.x > 3 && fwd.x < a.W - 3 && fwd.y > 3 && fwd.y < a.H - 3;
    m = m && fwd.status && bwd.status && fwd.err <= a.thres_err && bwd.err <= a.thres_err && dist2 <= thres2;
    if (lane == 0) {
      if (a.cand_done) {
        ic_store<true>(&a.new_r[2 * j], fwd.x);
        ic_store<true>(&a.new_r[2 * j + 1], fwd.y);
        ic_st8(&a.m_new[j], (uint8_t)(m ? 1 : 0));
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_fetch_add(a.cand_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        a.new_r[2 * j] = fwd.x;
        a.new_r[2 * j + 1] = fwd.y;
        a.m_new[j] = m ? 1 : 0;
      }
    }
    return;
  }
  // mask, feature_tracker.cpp:130-155
  bool m1;
  {
    const float dx = bwd.x - p0x, dy = bwd.y - p0y;
    const float dist2 = dx * dx + dy * dy;
    const float thres2 = a.thres_bidir * a.thres_bidir;
    const bool inimage = fwd.x > 0 && fwd.x < a.W && fwd.y > 0 && fwd.y < a.H;
    m1 = inimage && fwd.status && fwd.err <= a.thres_err && bwd.status && bwd.err <= a.thres_err && dist2 <= thres2 * 5;
    m1 = m1 && !(fl & VO_MONO_LM_DROPPED);  // LandmarkTracking(lmtrack_prev, mask_track), landmark.cpp:207
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
  // Everything the strict-border replay reads or rewrites goes through to memory (ic_store<true>): with the concurrent
  // arrangement that kernel runs next to this one on other XCDs (frame_fused.hip has the same stores for the same reason).
  if (a.strict) ic_store_records<true>(a.ic, i, lane, tp, S, cls);
  if (lane == 0) {
    a.Xp[3 * i] = Xp[0];
    a.Xp[3 * i + 1] = Xp[1];
    a.Xp[3 * i + 2] = Xp[2];
    a.m1[i] = m1 ? 1 : 0;
    a.ba_ok[i] = ((fl & 2) && Xp[2] > 0.1f) ? 1 : 0;  // mono_vo.cpp:808-809 / :821-822
    a.orig[i] = i;
    if (rf.err_flag) atomicOr(a.ic.flags, rf.err_flag);
    if (a.strict) {
      ic_store<true>(&a.scale[i], scale);
      ic_store<true>(&a.k1[2 * i], fwd.x);
      ic_store<true>(&a.k1[2 * i + 1], fwd.y);
      ic_store<true>(&a.ic.pts_track[2 * i], rf.x);
      ic_store<true>(&a.ic.pts_track[2 * i + 1], rf.y);
      ic_store<true>(&a.ic.mask[i], (uint8_t)rf.ok);
      ic_store<true>(&a.ic.touched[i], (uint8_t)(any_t ? 1 : 0));
      ic_store<true>(&a.ic.cls[i], (uint8_t)cls);
      ic_store<true>(&a.ic.last_pu[2 * i], lpx);
      ic_store<true>(&a.ic.last_pu[2 * i + 1], lpy);
      if (any_t) {
        const int slot = atomicAdd(&a.ic.jac[IC_JAC_NT], 1);
        ic_store<true>(&a.ic.tlist[slot], i);
        if (a.ic.tl2)  // concurrent replay: the entry says which frame it belongs to
          __hip_atomic_store(&a.ic.tl2[slot], ((unsigned long long)(unsigned)a.ic.epoch << 32) | (unsigned)i, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      a.scale[i] = scale;
      a.k1[2 * i] = fwd.x;
      a.k1[2 * i + 1] = fwd.y;
      a.ic.pts_track[2 * i] = rf.x;
      a.ic.pts_track[2 * i + 1] = rf.y;
      a.ic.mask[i] = (uint8_t)rf.ok;
    }
  }
  if (a.strict && a.ic.p1e) {
    // Concurrent replay: it may use this feature's pass-1 data from here on. The stores it depends on are write-through:
    // wait for them, then stamp and count.
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) {
      ic_store<true>(&a.ic.p1e[i], a.ic.epoch);
      atomicAdd(&a.ic.p1_word[(i & (IC_P1_SHARDS - 1)) * IC_P1_STRIDE], 1);
    }
  }
}

// strict border: the touched features (ic_replay), then the sequential fallback if it was requested.
// (Register cap as frame_replay_kernel's: a replay wavefront shares its SIMD with frame-kernel wavefronts.)
__global__ __launch_bounds__(IC_T) __attribute__((amdgpu_num_vgpr(288))) void mono_replay_kernel(MonoReplayArgs a) {
  __shared__ IcReplayShared rs;
  if (a.ic.tl2) {  // next to the frame kernel: what that kernel wrote is read past the caches
    __builtin_amdgcn_s_setprio(3);
    (void)ic_replay<true>(a.ic, rs, threadIdx.x, [](int, const IcResult &) {});
  } else {
    (void)ic_replay(a.ic, rs, threadIdx.x, [](int, const IcResult &) {});
  }
}
__global__ __launch_bounds__(IC_T) void mono_fallback_kernel(MonoReplayArgs a) {
  __shared__ IcShared sh;
  const int lane = threadIdx.x;
  bool work = a.ic.jac[IC_JAC_OVF] != 0;
  if (work && a.sync_signal && a.ic.p1e) {
    // Concurrent replay: ordered behind the replay pool, not behind the frame kernel whose pass-1 data this is about to
    // read. The pool normally ends after the frame kernel's last pass 1; when it gave up early: wait here, bounded, report.
    int polls = 0;
    while ((int)(ic_p1_count(a.ic, lane) - a.ic.p1_target) < 0) {
      if (++polls > IC_SPIN_LIMIT) {
        if (lane == 0) atomicOr(a.ic.flags, 8);  // the frame is issued again by vo_mono_frame_result
        work = false;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (work)
    for (int pt = blockIdx.x; pt < a.ic.n; pt += gridDim.x) {
      __syncthreads();
      ic_strict_run(a.ic, sh, pt, a.ic.n, lane, [](int, const IcResult &) {});
    }
  // stream-ordered behind the replay: when all of its workgroups have counted, every touched feature is final
  if (a.sync_signal) {
    if (work) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) atomicAdd(&a.sync[1], 1);
  }
}

// ---- host side ---------------------------------------------------------------------
static size_t m_align16(size_t v) { return (v + 15) & ~(size_t)15; }

// phase 0: features (+ candidates unless split); phase 2: the candidates as a launch of their own on c->stream (the side stream)
template <int WIN>
static void mono_launch(vo_ctx *c, const MonoArgs &a, int phase, bool split, int *cand_done) {
  if (phase == 0) {
    vo_prof_begin(c, VO_K_KLT);
    hipLaunchKernelGGL(mono_track_kernel<WIN>, dim3(split ? a.n : a.n + a.n_new), dim3(64), 0, c->stream, a);
    vo_prof_end(c);
  } else if (a.n_new > 0) {
    MonoArgs b = a;
    b.wg_off = a.n;
    b.cand_done = cand_done;
    hipLaunchKernelGGL(mono_track_kernel<WIN>, dim3(a.n_new), dim3(64), 0, c->stream, b);
  }
}
static void mono_launch_any(vo_ctx *c, int win, const MonoArgs &a, int phase, bool split, int *cand_done) {
  switch (win) {
    case 13: mono_launch<13>(c, a, phase, split, cand_done); break;
    case 15: mono_launch<15>(c, a, phase, split, cand_done); break;
    case 21: mono_launch<21>(c, a, phase, split, cand_done); break;
    default: mono_launch<31>(c, a, phase, split, cand_done); break;
  }
}

// MonoVO (mono_vo.hip): the next vo_mono_frame_enqueue* lets the BA launch build the next track set (mvo_advance_body)
int vo_mono_frame_set_advance(vo_ctx *c, const MvoAdvArgs *adv) {
  int rc = vo_frame_init(c);
  if (rc < 0) return rc;
  c->frame->mvo_adv = *adv;
  c->frame->mvo_adv_on = 1;
  return VO_OK;
}

int vo_mono_frame_set_track_flags(vo_ctx *c, int mode) {
  int rc = vo_frame_init(c);
  if (rc < 0) return rc;
  c->frame->mono_flag_mode = mode;
  return VO_OK;
}

// bp != null: the closed new-point step — the candidates are the per-bin best keypoints of table `table`
// (vo_new_point_candidates_enqueue on the image in slot1), all tracked speculatively, emitted by the BA launch's epilogue
static int mono_enqueue_impl(vo_ctx *c, const vo_mono_params *prm, int slot0, int slot1, const float *pts0,
                             const float *Xw, const uint8_t *flags, int n, const float Tcw_prev[16],
                             const float Tcw_prior[16], const float dT01_prior[16], int inputs_on_device,
                             const vo_bin_params *bp, int table) {
  if (!c || !prm || !Tcw_prev || !Tcw_prior || !dT01_prior || n < 0) return VO_ERR_INVALID;
  const vo_cand_table *tab = nullptr;
  int n_new = 0;
  if (bp) {
    if (n <= 0) VO_FAIL(c, VO_ERR_INVALID, "the closed new-point step needs a track set (the first frame is the caller's)");
    tab = vo_orb_cand_table(c, table);
    if (c->frame && c->frame->defer_detect && (!tab || tab->n_bins != bp->n_bins_u * bp->n_bins_v)) {
      c->frame->defer_detect = 0;  // deferred detection into a table that does not exist yet: detect now, in stream order
      {
        const int rcd = vo_new_point_candidates_enqueue(c, slot1, bp, table);
        if (rcd < 0) return rcd;
      }
      tab = vo_orb_cand_table(c, table);
    }
    if (!tab || tab->n_bins != bp->n_bins_u * bp->n_bins_v)
      VO_FAIL(c, VO_ERR_INVALID, "candidate table %d was not filled for %d x %d bins (vo_new_point_candidates_enqueue)", table,
              bp->n_bins_u, bp->n_bins_v);
    if (bp->u_step <= 0 || bp->v_step <= 0) VO_FAIL(c, VO_ERR_INVALID, "u_step / v_step must be positive");
    if (prm->max_level < 1) VO_FAIL(c, VO_ERR_INVALID, "trackBidirection needs max_level >= 1");
    n_new = tab->n_bins;
    if (n_new > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "%d bins exceed vo_config.max_points=%d", n_new, c->cfg.max_points);
  }
  if (n > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "n=%d exceeds vo_config.max_points=%d", n, c->cfg.max_points);
  if (n > 0 && (!pts0 || !Xw || !flags)) return VO_ERR_INVALID;
  if (prm->win != 13 && prm->win != 15 && prm->win != 21 && prm->win != 31)
    VO_FAIL(c, VO_ERR_INVALID, "mono frame kernel not instantiated for window %d (13, 15, 21, 31)", prm->win);
  if (prm->max_level < 0) VO_FAIL(c, VO_ERR_INVALID, "maxLevel >= 0 violated");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = vo_frame_init(c);
  if (rc < 0) return rc;
  vo_frame_state *f = c->frame;
  // one result block, one staging set and one completion event per context: a second frame would overwrite them
  // while the first one's kernels still use them
  if (f->pending) VO_FAIL(c, VO_ERR_INVALID, "a frame is already in flight: call vo_mono_frame_result first");
  // a track-set advance armed by vo_mono_frame_set_advance belongs to THIS enqueue, whether it gets as far as the BA launch or not
  const int adv_on = f->mvo_adv_on, flag_mode = f->mono_flag_mode, defer = f->defer_detect;
  f->mvo_adv_on = 0;
  f->mono_flag_mode = 0;
  f->defer_detect = 0;
  f->mono_split = 0;
  // MonoVO's synchronous call (vo_frame_set_deferred_detection): the features' part of the frame kernel goes out at once, the
  // keypoint detection of slot1 (defer == 1; 2: already on the side stream) and the candidates' launch follow on the side
  // stream, the BA launch joins them on the device (frame_pipeline.hip has the stereo form)
  const bool split = tab && defer && n > 0 && c->ingest_side && !c->frame_conc_off;
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
  if (tab && defer == 1 && !split) {  // (cannot overlap: now)
    const int rcd = vo_new_point_candidates_enqueue(c, slot1, bp, table);
    if (rcd < 0) return rcd;
  }
  if (tab && !split) VO_CHECK_HIP(c, hipStreamWaitEvent(s, tab->ready, 0));  // (filled on the side stream, long before)
  // packed result block: header | stage | pixels | scale [| closed: new-point masks | their I0 pixels | their I1 pixels]
  f->n = n;
  f->n_new = n_new;
  f->closed = tab ? 1 : 0;
  f->table = tab;
  size_t off = m_align16(sizeof(vo_frame_hdr));
  f->off_stage = off;  off += m_align16((size_t)n);
  f->off_pl1 = off;    off += m_align16(sizeof(float) * 2 * (size_t)n);
  f->off_pr1 = off;    off += m_align16(sizeof(float) * (size_t)n);  // (scale)
  const size_t bulk_end = off;  // what the BA launch's epilogue copies to the host as a block
  if (tab) {  // written entry by entry to the device AND the host block by the epilogue (np_emit.hpp)
    f->off_mnew = off;  off += m_align16((size_t)n_new);
    f->off_newr = off;  off += m_align16(sizeof(float) * 2 * (size_t)n_new);
    f->off_newl = off;  off += m_align16(sizeof(float) * 2 * (size_t)n_new);
  }
  f->res_bytes = off;
  if (f->res_bytes > f->res_cap) VO_FAIL(c, VO_ERR_CAPACITY, "result block of %zu bytes exceeds the context's (max_points too small)", f->res_bytes);
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
    const int eff = vo_pyr_levels_host(P0.w, P0.h, prm->win, prm->max_level);
    VO_NEED_LEVELS(c, P0, eff);
    VO_NEED_LEVELS(c, P1, eff);
    if (vo_slot_acquire(c, slot0) < 0 || vo_slot_acquire(c, slot1) < 0) return VO_ERR_HIP;
    for (int l = 0; l <= eff; ++l) {
      a.I0[l] = P0.lv[l];
      a.I1[l] = P1.lv[l];
    }
    a.max_level = eff;
    if (tab) {
      const int effb = vo_pyr_levels_host(P0.w, P0.h, prm->win, prm->max_level - 1);
      a.max_level_bwd = effb < eff ? effb : eff;
      a.n_new = n_new;
      a.pts_new = tab->xy;
      a.cand_has = tab->has;
      a.new_r = f->bin_r;
      a.m_new = f->bin_m;
    }
    a.n = n;
    a.pts0 = d_p0;
    a.Xw = d_X;
    a.flags = d_fl;
    a.flag_mode = flag_mode;
    memcpy(a.Tcw_prev, Tcw_prev, sizeof(a.Tcw_prev));
    memcpy(a.Tcw_prior, Tcw_prior, sizeof(a.Tcw_prior));
    memcpy(a.K, prm->K, sizeof(a.K));
    a.W = prm->width;
    a.H = prm->height;
    a.thres_err = prm->thres_err;
    a.thres_bidir = prm->thres_bidirection;
    // Strict-border arrangement of this frame (as frame_pipeline.hip): 4 takes the replay next to the frame kernel when the
    // previous frame replayed something and the launch fits the chip; the gated arrangement (5) is the stereo frame's only.
    int strict = c->frame_strict_ic;
    if (strict == 4) strict = (f->last_replayed >= VO_CONC_MIN_REPLAYED && n + n_new <= VO_CONC_MAX_WORKGROUPS) ? 3 : 1;
    if (strict == 5) strict = 1;
    if (c->frame_conc_off && strict >= 3) strict = 1;  // a join timed out before: stream order
    c->frame_strict_now = strict;
    a.strict = strict;
    {
      int g = ((f->last_replayed + 32 + 31) / 32) * 32;
      f->conc_grid = g < 64 ? 64 : (g > 256 ? 256 : g);
      if (c->frame_strict_ic == 3) f->conc_grid = 256;
      if (c->dbg[VO_DBG_CONC_GRID] > 0) f->conc_grid = c->dbg[VO_DBG_CONC_GRID];
    }
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
    const int fb_grid = n < 128 ? n : 128;
    const int p1_before = f->sync_p1_target, done_before = f->sync_done_target;
    if (strict == 3) {
      // running totals of the two hand-shake counters, and the frame's epoch (the pass-1 target: different for every frame)
      vo_wrap_add(f->sync_p1_target, n);
      vo_wrap_add(f->sync_done_target, fb_grid);
      a.ic.epoch = f->sync_p1_target != 0 ? f->sync_p1_target : 1;
      a.ic.p1_word = f->sync + IC_P1_STRIDE;
      a.ic.p1_target = f->sync_p1_target;
    } else {
      a.ic.tl2 = nullptr;
      a.ic.p1e = nullptr;
    }
    // what vo_mono_frame_result needs to issue this frame again (device pointers only)
    f->again_mono.prm = *prm;
    f->again_mono.slot0 = slot0;
    f->again_mono.slot1 = slot1;
    f->again_mono.n = n;
    f->again_mono.pts0 = d_p0;
    f->again_mono.Xw = d_X;
    f->again_mono.flags = d_fl;
    memcpy(f->again_mono.Tcw_prev, Tcw_prev, sizeof(f->again_mono.Tcw_prev));
    memcpy(f->again_mono.Tcw_prior, Tcw_prior, sizeof(f->again_mono.Tcw_prior));
    memcpy(f->again_mono.dT01_prior, dT01_prior, sizeof(f->again_mono.dT01_prior));
    f->again_mono.has_bins = bp ? 1 : 0;
    if (bp) f->again_mono.bins = *bp;
    f->again_mono.table = table;
    f->again_mono.flag_mode = flag_mode;
    mono_launch_any(c, prm->win, a, 0, split, nullptr);
    if (strict) {
      MonoReplayArgs r;
      memset(&r, 0, sizeof(r));
      r.ic = a.ic;
      r.sync = f->sync;
      if (strict == 3) {
        // The replay on its own stream next to the frame kernel, as a pool of resident workgroups that pick the touched
        // features up as the frame kernel lists them (ic_replay<true>); the fallback behind it counts its workgroups in
        // sync[1], the BA launch on the main stream waits for that count. No HIP event joins the streams (frame_fused.hip).
        r.sync_signal = 1;
        c->stream = c->stream3;
        vo_prof_begin(c, VO_K_IC);
        hipLaunchKernelGGL(mono_replay_kernel, dim3(n < f->conc_grid ? n : f->conc_grid), dim3(IC_T), 0, c->stream3, r);
        vo_prof_end(c);
        c->stream = s;
        hipLaunchKernelGGL(mono_fallback_kernel, dim3(fb_grid), dim3(IC_T), 0, c->stream3, r);
      } else {
        vo_prof_begin(c, VO_K_IC);
        if (strict == 2)
          (void)hipMemsetAsync(&a.ic.jac[IC_JAC_OVF], 1, sizeof(int), s);
        else
          hipLaunchKernelGGL(mono_replay_kernel, dim3(n < IC_JGRID ? n : IC_JGRID), dim3(IC_T), 0, s, r);
        hipLaunchKernelGGL(mono_fallback_kernel, dim3(strict == 2 ? (n < 1024 ? n : 1024) : fb_grid), dim3(IC_T), 0, s, r);
        vo_prof_end(c);
      }
    }
    if (split && hipGetLastError() == hipSuccess) {
      // the features (and the replay) are on their way: now the detector of the current image — unless it is there already —
      // and, behind it on the side stream, the candidates' launch; no event: the BA launch's epilogue waits for their count
      int rcd = defer == 1 ? vo_new_point_candidates_enqueue(c, slot1, bp, table) : VO_OK;
      if (rcd >= 0) {
        c->stream = c->stream2;
        if (vo_slot_acquire(c, slot0) < 0 || vo_slot_acquire(c, slot1) < 0) rcd = VO_ERR_HIP;
        if (rcd >= 0) mono_launch_any(c, prm->win, a, 2, true, f->cand_done);
        c->stream = s;
      }
      if (rcd < 0) {
        f->sync_p1_target = p1_before;
        f->sync_done_target = done_before;
        return rcd;
      }
      vo_wrap_add(f->cand_total, n_new);  // (cumulative, like the word the candidate workgroups count in)
      f->mono_split = 1;
    }
    if (hipGetLastError() != hipSuccess) {  // nothing of this frame will count: the cumulative targets go back
      f->sync_p1_target = p1_before;
      f->sync_done_target = done_before;
      VO_FAIL(c, VO_ERR_HIP, "launch of the mono frame failed");
    }
    // BA set (refined && BA class && depth > 0.1, in index order; mono_vo.cpp:799-826, :846-860), pose-only BA
    // (class-surface variant, T01 initialised with the motion prior, :856-867) and the tail of the frame: ONE launch.
    // The GN kernel's frame-mode prologue selects and compacts the set, its epilogue is mono_gate_body.
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
    g.gn = &f->hdr->gn;
    g.dT = f->hdr->dT;
    memcpy(g.dT_prior, dT01_prior, sizeof(g.dT_prior));
    memcpy(g.K, prm->K, sizeof(g.K));
    g.thres_sampson = prm->thres_sampson;
    g.motion = f->st2;
    g.stage = f->stage;
    g.pts1 = f->F_pl1;
    g.cnt = f->hdr->cnt;
    g.res_dev = (const uint32_t *)f->res_dev;
    g.res_host = (uint32_t *)f->res_host;
    g.res_words = (int)((bulk_end + 3) / 4);
    if (tab) {
      g.np.bins = tab->n_bins;
      g.np.bins_u = bp->n_bins_u;
      g.np.u_step = bp->u_step;
      g.np.v_step = bp->v_step;
      g.np.has = tab->has;
      g.np.xy = tab->xy;
      g.np.bin_r = f->bin_r;
      g.np.bin_m = f->bin_m;
      g.np.out_l = (float *)(f->res_dev + f->off_newl);
      g.np.out_r = (float *)(f->res_dev + f->off_newr);
      g.np.out_m = f->res_dev + f->off_mnew;
      g.np.host_l = (float *)(f->res_host + f->off_newl);
      g.np.host_r = (float *)(f->res_host + f->off_newr);
      g.np.host_m = f->res_host + f->off_mnew;
      if (f->mono_split) {
        g.np.cand_done = f->cand_done;
        g.np.cand_target = f->cand_total;
      }
    }
    g.hdr_flags = &f->hdr->flags;
    if (adv_on) {
      g.adv_on = 1;
      g.adv = f->mvo_adv;
      g.adv.stage = g.stage;
      g.adv.pts1 = g.pts1;
      g.adv.cand1 = g.np.out_l;
      g.adv.cand0 = g.np.out_r;
      g.adv.mnew = g.np.out_m;
    }
    vo_gn_frame gf;
    memset(&gf, 0, sizeof(gf));
    gf.n = n;
    gf.m1 = f->m1;
    gf.m2 = f->m2;
    gf.m3 = f->m3;
    gf.X = f->A_X;
    gf.pl1 = f->A_ref;
    gf.C_X = f->C_X;
    gf.C_pl1 = f->C_pl1;
    gf.C_orig = f->C_orig;
    gf.cnt = f->hdr->cnt;
    gf.ctl = f->ctl;
    gf.ctl_words = (int)(vo_ic_ctl_bytes() / 4);
    gf.nt_word = vo_ic_ctl_nt_word();
    gf.hdr_flags = &f->hdr->flags;
    if (strict == 3) {  // the replay runs on its own stream: join on the device
      gf.join_word = f->sync + 1;
      gf.join_target = f->sync_done_target;
      if (c->dbg[VO_DBG_FAIL_JOIN]) gf.join_target += 1 << 20;  // tests: a join that cannot be met
    }
    gf.mono_gate = &g;
    rc = vo_gn_enqueue(c, false, true, f->C_X, f->C_pl1, nullptr, n, nullptr, prm->K, prm->K, nullptr,
                       (float)prm->thres_poseba, VO_GN_VARIANT_CORE, dT01_prior, f->hdr->dT, f->mG, &f->hdr->gn, true,
                       nullptr, nullptr, 0, 0.f, &gf);
    if (rc < 0) return rc;
    VO_CHECK_HIP(c, hipGetLastError());
  } else {
    VO_CHECK_HIP(c, hipMemsetAsync(f->hdr, 0, sizeof(vo_frame_hdr), s));
    VO_CHECK_HIP(c, hipMemcpyAsync(f->hdr->dT, dT01_prior, sizeof(float) * 16, hipMemcpyHostToDevice, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(f->res_host, f->res_dev, f->res_bytes, hipMemcpyDeviceToHost, s));
  }
  VO_CHECK_HIP(c, hipEventRecord(f->ev_done, s));
  f->known_done = false;
  f->pending = true;
  c->frame_slots_busy = c->ingest_side;
  c->frame_slot[0] = slot0;
  c->frame_slot[1] = slot1;
  c->frame_slot[2] = -1;
  return VO_OK;
}

extern "C" int vo_mono_frame_enqueue(vo_ctx *c, const vo_mono_params *prm, int slot0, int slot1, const float *pts0,
                                     const float *Xw, const uint8_t *flags, int n, const float Tcw_prev[16],
                                     const float Tcw_prior[16], const float dT01_prior[16], int inputs_on_device) {
  const int rc = mono_enqueue_impl(c, prm, slot0, slot1, pts0, Xw, flags, n, Tcw_prev, Tcw_prior, dT01_prior, inputs_on_device, nullptr, 0);
  if (c && c->frame) c->frame->mvo_adv_on = c->frame->mono_flag_mode = 0;
  return rc;
}
extern "C" int vo_mono_frame_enqueue_closed(vo_ctx *c, const vo_mono_params *prm, int slot0, int slot1, const float *pts0,
                                            const float *Xw, const uint8_t *flags, int n, const float Tcw_prev[16],
                                            const float Tcw_prior[16], const float dT01_prior[16],
                                            const vo_bin_params *bins, int table, int inputs_on_device) {
  if (!bins) return VO_ERR_INVALID;
  const int rc = mono_enqueue_impl(c, prm, slot0, slot1, pts0, Xw, flags, n, Tcw_prev, Tcw_prior, dT01_prior, inputs_on_device, bins, table);
  if (c && c->frame) c->frame->mvo_adv_on = c->frame->mono_flag_mode = 0;
  return rc;
}

// the new points of the closed frame just received (vo_mono_frame_result first): pixels in I1 (the bucketed keypoints),
// their back-tracked pixels in I0 and the trackBidirection masks, bins ascending. Any pointer may be null.
extern "C" int vo_mono_frame_new_points(vo_ctx *c, float *pts1_new, float *pts0_new, uint8_t *mask_new, int *n_new) {
  if (!c || !c->frame) return VO_ERR_INVALID;
  vo_frame_state *f = c->frame;
  if (f->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_mono_frame_result first");
  const vo_frame_hdr *h = (const vo_frame_hdr *)f->res_host;
  const int nn = f->closed ? h->cnt[6] : 0;
  if (n_new) *n_new = nn;
  if (nn > 0) {
    if (pts1_new) memcpy(pts1_new, f->res_host + f->off_newl, sizeof(float) * 2 * (size_t)nn);
    if (pts0_new) memcpy(pts0_new, f->res_host + f->off_newr, sizeof(float) * 2 * (size_t)nn);
    if (mask_new) memcpy(mask_new, f->res_host + f->off_mnew, (size_t)nn);
  }
  return VO_OK;
}

extern "C" int vo_mono_frame_result(vo_ctx *c, float *pts1, float *scale, uint8_t *stage, float dT01[16],
                                    vo_mono_counts *counts, vo_gn_info *gn) {
  if (!c || !c->frame || !c->frame->pending) return VO_ERR_INVALID;
  vo_frame_state *f = c->frame;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (!f->known_done) VO_CHECK_HIP(c, hipEventSynchronize(f->ev_done));
  f->known_done = false;
  f->pending = false;
  c->frame_slots_busy = 0;
  f->recovered = 0;
  const int n = f->n;
  const vo_frame_hdr *h = (const vo_frame_hdr *)f->res_host;
  if ((h->flags & 8) && n > 0 && (c->frame_strict_now == 3 || f->mono_split)) {
    // The device-side join with the replay stream timed out (the two queues did not run concurrently: a serialising tool,
    // a busy GPU). As for the stereo frame: drain the streams, re-base the hand-shake words, switch the context to the
    // stream-ordered replay for good and issue the frame again from its (intact) device inputs.
    VO_CHECK_HIP(c, hipStreamSynchronize(c->stream3));
    VO_CHECK_HIP(c, hipStreamSynchronize(c->stream2));
    VO_CHECK_HIP(c, hipStreamSynchronize(c->stream_main));
    VO_CHECK_HIP(c, hipMemsetAsync(f->cand_done, 0, 64, c->stream_main));
    f->cand_total = 0;
    VO_CHECK_HIP(c, hipMemsetAsync(f->sync, 0, 128 + 64 * 128, c->stream_main));
    VO_CHECK_HIP(c, hipMemsetAsync(f->ctl, 0, vo_ic_ctl_bytes(), c->stream_main));
    f->sync_p1_target = f->sync_done_target = 0;
    c->frame_conc_off = 1;
    ++c->frame_recoveries;
    const auto g = f->again_mono;
    f->mono_flag_mode = g.flag_mode;
    // (MonoVO: the first attempt's BA launch reported "not built" for the next track set — the frame had failed — and the host
    // has that report by now: the advance of the re-issued frame is the host's, as launches of its own)
    const int rc2 = mono_enqueue_impl(c, &g.prm, g.slot0, g.slot1, g.pts0, g.Xw, g.flags, g.n, g.Tcw_prev, g.Tcw_prior, g.dT01_prior, 1,
                                      g.has_bins ? &g.bins : nullptr, g.table);
    if (rc2 < 0) return rc2;
    VO_CHECK_HIP(c, hipEventSynchronize(f->ev_done));
    f->pending = false;
    c->frame_slots_busy = 0;
    f->recovered = 1;
  }
  f->last_replayed = n > 0 ? h->cnt[7] : 0;
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
    if (h->flags & 8) VO_FAIL(c, VO_ERR_HIP, "the strict-border replay stream did not finish (device-side join timed out twice)");
    VO_FAIL(c, VO_ERR_NAN_UPDATE, "dtu dtv nan");
  }
  return VO_OK;
}
