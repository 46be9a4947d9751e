// frame_pipeline.hip — the steady-state stereo frame, chained on the device.
//
// Operator sequence of StereoVO::trackStereoImages
// (core/visual_odometry/stereo_vo/stereo_vo.cpp), steps
//   [3]  prior pixels + patch scale                    :483-522
//   [4]  trackWithPrior  I0_L -> I1_L  + compaction    :531-538
//   [4-1] Sobel + trackWithScale       + compaction    :549-558
//   [5]  trackWithPrior  I1_L -> I1_R  + compaction    :564-571
//   [6]  poseOnlyBundleAdjustment_Stereo               :595-646
//   [7]  the y > 660 gate                              :653-670
//   [10] trackBidirection for the new points           :706-711
// expressed in the previous left-camera frame. The reference returns to the
// host between every step (std::vector compaction in the StereoLandmarkTracking
// constructors, landmark.cpp:291-332); here every step is a kernel on one HIP
// stream, live counts stay in device memory (kernels take `const int* d_n` and
// surplus workgroups exit), and the host reads ONE packed result block per
// frame (a single D2H copy into pinned memory).
#include "frame_state.hpp"
#include "vo_kernels.hpp"

#include <sched.h>
#include <stddef.h>
#include <stdlib.h>
#include <time.h>
#ifdef VO_TRACE_HOST
#include <chrono>
static double vo_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define VO_TT(label) do { double _t = vo_now_us(); if (vo_tt_n < 40) fprintf(stderr, "  [host] %-14s +%.1f us\n", label, _t - vo_tt_last); vo_tt_last = _t; } while (0)
static double vo_tt_last; static int vo_tt_n;
#else
#define VO_TT(label)
#endif

template <typename T>
static hipError_t fs_alloc(vo_ctx *c, T **p, size_t n) {
  return vo_dev_malloc(c, (void **)p, (n ? n : 1) * sizeof(T));
}

static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

int vo_frame_init(vo_ctx *c) {
  if (c->frame) return VO_OK;
  vo_frame_state *f = (vo_frame_state *)calloc(1, sizeof(vo_frame_state));
  c->frame = f;
  const size_t N = (size_t)c->cfg.max_points;
  f->cap = (int)N;
  float **f2[] = {&f->in_l0, &f->in_r0, &f->in_new, &f->A_pl0, &f->A_pl1, &f->A_pr1, &f->B_pl1, &f->B_pr1,
                  &f->C_pl1, &f->C_pr1, &f->new_back, &f->A_ref, &f->A_lastpu, &f->bin_r};
  for (float **p : f2) VO_CHECK_HIP(c, fs_alloc(c, p, 2 * N));
  float **f3[] = {&f->in_X, &f->A_X, &f->B_X, &f->C_X};
  for (float **p : f3) VO_CHECK_HIP(c, fs_alloc(c, p, 3 * N));
  float **f1[] = {&f->F_scale, &f->A_scale, &f->e1, &f->e2, &f->e3};
  for (float **p : f1) VO_CHECK_HIP(c, fs_alloc(c, p, N));
  int32_t **i1[] = {&f->F_orig, &f->A_orig, &f->B_orig, &f->C_orig};
  for (int32_t **p : i1) VO_CHECK_HIP(c, fs_alloc(c, p, N));
  uint8_t **u1[] = {&f->m1, &f->m2, &f->m3, &f->mG, &f->st1, &f->st2, &f->st3, &f->A_touched, &f->A_cls, &f->in_flags,
                    &f->bin_m};
  for (uint8_t **p : u1) VO_CHECK_HIP(c, fs_alloc(c, p, N));
  f->res_cap = align16(sizeof(vo_frame_hdr)) + 2 * align16(N) + 4 * align16(sizeof(float) * 2 * N);
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&f->res_dev, f->res_cap));
  VO_CHECK_HIP(c, vo_host_malloc(c, (void **)&f->res_host, f->res_cap, hipHostMallocDefault));
  memset(f->res_host, 0, f->res_cap);  // (the polled sequence word must not match by accident)
  VO_CHECK_HIP(c, hipMemsetAsync(f->res_dev, 0, f->res_cap, c->stream));
  VO_CHECK_HIP(c, hipEventCreateWithFlags(&f->ev_done, hipEventDisableTiming));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&f->ctl, vo_ic_ctl_bytes()));
  VO_CHECK_HIP(c, hipMemsetAsync(f->ctl, 0, vo_ic_ctl_bytes(), c->stream));
  // [1] = finished workgroups of the replay's last kernel; from byte 128 on: 64 shards of the pass-1 count, 128 bytes apart
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&f->adv_done, 64));
  VO_CHECK_HIP(c, hipMemsetAsync(f->adv_done, 0, 64, c->stream));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&f->cand_done, 64));
  VO_CHECK_HIP(c, hipMemsetAsync(f->cand_done, 0, 64, c->stream));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&f->sync, 128 + 64 * 128));
  VO_CHECK_HIP(c, hipMemsetAsync(f->sync, 0, 128 + 64 * 128, c->stream));
  return VO_OK;
}

void vo_frame_free(vo_ctx *c) {
  vo_frame_state *f = c->frame;
  if (!f) return;
  void *bufs[] = {f->in_l0, f->in_r0, f->in_X, f->in_new, f->F_scale, f->F_orig, f->A_pl0, f->A_pl1, f->A_pr1,
                  f->A_X, f->A_scale, f->A_orig, f->B_pl1, f->B_pr1, f->B_X, f->B_orig, f->C_pl1, f->C_pr1,
                  f->C_X, f->C_orig, f->m1, f->m2, f->m3, f->mG, f->st1, f->st2, f->e1, f->e2, f->new_back,
                  f->A_ref, f->A_lastpu, f->A_touched, f->A_cls, f->res_dev, f->st3, f->e3, f->ctl, f->sync, f->in_flags, f->bin_r,
                  f->bin_m, f->adv_done, f->cand_done};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (f->res_host) (void)hipHostFree(f->res_host);
  if (f->ev_done) (void)hipEventDestroy(f->ev_done);
  free(f);
  c->frame = nullptr;
}

static void inv_se3(const float T[16], float Ti[16]) {
  float Rt[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = T[j * 4 + i];
  const float t[3] = {T[3], T[7], T[11]};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Ti[i * 4 + j] = Rt[i * 3 + j];
    Ti[i * 4 + 3] = ((-Rt[i * 3 + 0]) * t[0] + (-Rt[i * 3 + 1]) * t[1]) + (-Rt[i * 3 + 2]) * t[2];
  }
  Ti[12] = Ti[13] = Ti[14] = 0;
  Ti[15] = 1;
}

// (VO_CONC_MIN_REPLAYED / VO_CONC_MAX_WORKGROUPS: frame_state.hpp)
#define RC(x)                \
  do {                       \
    int _rc = (x);           \
    if (_rc < 0) return _rc; \
  } while (0)

// StereoVO (stereo_vo.hip): the next vo_frame_enqueue_impl lets the BA launch build the next track set
int vo_frame_set_advance(vo_ctx *c, const VoAdvArgs *adv) {
  int rc = vo_frame_init(c);
  if (rc < 0) return rc;
  c->frame->adv_next = *adv;
  c->frame->adv_next.on = 1;
  return VO_OK;
}

// StereoVO (stereo_vo.hip), trackStereoImages without look-ahead: the pair is in its slots but its keypoints are not
// detected yet. The next vo_frame_enqueue_impl (closed frame on the fused path) then
//   main stream:  frame kernel, features only -> replay -> ... BA launch
//   side stream:  detection + per-bin table of slot_l1 -> frame kernel, candidates only
// joined by one event in front of the BA launch — the 165 us detector chain runs under the features' tracking instead of
// in front of it. Same results: the same kernels on the same inputs, in two launches instead of one.
// issued != 0: the detection is already on the side stream (started from the image itself, in front of the pair's pyramids:
// vo_new_point_candidates_enqueue_image) — only the candidates' launch follows it there.
int vo_frame_set_deferred_detection(vo_ctx *c, int issued) {
  int rc = vo_frame_init(c);
  if (rc < 0) return rc;
  c->frame->defer_detect = issued ? 2 : 1;
  return VO_OK;
}

extern "C" int vo_stereo_frame_set_strict_border(vo_ctx *c, int strict) {
  if (!c) return VO_ERR_INVALID;
  c->frame_strict_ic = (strict >= 2 && strict <= 5) ? strict : (strict ? 1 : 0);
  c->frame_strict_now = c->frame_strict_ic == 4 ? 1 : c->frame_strict_ic;
  return VO_OK;
}

// bp != null: the closed step [10] — the candidates are the per-bin best keypoints of table `table`
// (vo_new_point_candidates_enqueue), all tracked speculatively, emitted by the BA launch's epilogue
// T_pw / T_cw_prior != null: the reference's own data flow — Xp holds WORLD points (vo_stereo_frame_enqueue_closed_world)
static int vo_frame_enqueue_body(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                                 const float *pts_l0, const float *pts_r0, const float *Xp, const uint8_t *flags, int n,
                                 const float dT_prior[16], const float *pts_new, int n_new, int inputs_on_device,
                                 const vo_bin_params *bp, int table, const float *T_pw, const float *T_cw_prior);
int vo_frame_enqueue_impl(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                          const float *pts_l0, const float *pts_r0, const float *Xp, const uint8_t *flags, int n,
                          const float dT_prior[16], const float *pts_new, int n_new, int inputs_on_device,
                          const vo_bin_params *bp, int table, const float *T_pw, const float *T_cw_prior) {
  const int rc = vo_frame_enqueue_body(c, prm, slot_l0, slot_l1, slot_r1, pts_l0, pts_r0, Xp, flags, n, dT_prior, pts_new, n_new,
                                       inputs_on_device, bp, table, T_pw, T_cw_prior);
  // a track-set advance armed by vo_frame_set_advance belongs to THIS enqueue, whether it got as far as the BA launch or
  // not: it never stays armed for a later frame of this context (its track sets may be gone by then)
  if (c && c->frame) {
    c->frame->adv_next.on = 0;
    c->frame->defer_detect = 0;
  }
  return rc;
}
static int vo_frame_enqueue_body(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                                 const float *pts_l0, const float *pts_r0, const float *Xp, const uint8_t *flags, int n,
                                 const float dT_prior[16], const float *pts_new, int n_new, int inputs_on_device,
                                 const vo_bin_params *bp, int table, const float *T_pw, const float *T_cw_prior) {
  if (!c || !prm || !dT_prior || n < 0 || n_new < 0) return VO_ERR_INVALID;
  if ((T_pw != nullptr) != (T_cw_prior != nullptr)) return VO_ERR_INVALID;
  if (T_pw && !vo_frame_fused_supported(prm->win))
    VO_FAIL(c, VO_ERR_INVALID, "world-frame landmarks need a window the fused frame kernel is built for (13, 15, 21, 31)");
  const vo_cand_table *tab = nullptr;
  if (bp) {
    if (n <= 0) VO_FAIL(c, VO_ERR_INVALID, "the closed step [10] needs a track set (the first frame is the caller's)");
    if (!vo_frame_fused_supported(prm->win))
      VO_FAIL(c, VO_ERR_INVALID, "the closed step [10] needs a window the fused frame kernel is built for (13, 15, 21, 31)");
    tab = vo_orb_cand_table(c, table);
    if (c->frame && c->frame->defer_detect && (!tab || tab->n_bins != bp->n_bins_u * bp->n_bins_v)) {
      // deferred detection into a table that does not exist yet (the stream's second frame): detect now, in stream order
      c->frame->defer_detect = 0;
      RC(vo_new_point_candidates_enqueue(c, slot_l1, bp, table));
      tab = vo_orb_cand_table(c, table);
    }
    if (!tab || tab->n_bins != bp->n_bins_u * bp->n_bins_v)
      VO_FAIL(c, VO_ERR_INVALID, "candidate table %d was not filled for %d x %d bins (vo_new_point_candidates_enqueue)", table,
              bp->n_bins_u, bp->n_bins_v);
    if (bp->u_step <= 0 || bp->v_step <= 0) VO_FAIL(c, VO_ERR_INVALID, "u_step / v_step must be positive");
    n_new = tab->n_bins;
    pts_new = tab->xy;  // device memory in either input mode
  }
#ifdef VO_TRACE_HOST
  vo_tt_last = vo_now_us();
#endif
  if (n > c->cfg.max_points || n_new > c->cfg.max_points)
    VO_FAIL(c, VO_ERR_CAPACITY, "n=%d / n_new=%d exceed vo_config.max_points=%d", n, n_new, c->cfg.max_points);
  if ((n > 0 && (!pts_l0 || !pts_r0 || !Xp)) || (n_new > 0 && !pts_new)) return VO_ERR_INVALID;
  if (n_new > 0 && prm->max_level - 1 < 0) VO_FAIL(c, VO_ERR_INVALID, "trackBidirection needs max_level >= 1");
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_TT("setdevice");
  RC(vo_frame_init(c));
#ifdef VO_TRACE_HOST
  ++vo_tt_n;
#endif
  vo_frame_state *f = c->frame;
  // one result block, one staging set and one completion event per context: a second frame would overwrite them
  // while the first one's kernels still use them
  if (f->pending) VO_FAIL(c, VO_ERR_INVALID, "a frame is already in flight: call vo_stereo_frame_result first");
  hipStream_t s = c->stream;
  const float *d_l0 = pts_l0, *d_r0 = pts_r0, *d_X = Xp, *d_new = pts_new;
  const uint8_t *d_fl = flags;
  if (!inputs_on_device) {
    if (n > 0 && flags) {
      VO_CHECK_HIP(c, hipMemcpyAsync(f->in_flags, flags, (size_t)n, hipMemcpyHostToDevice, s));
      d_fl = f->in_flags;
    }
    if (n > 0) {
      VO_CHECK_HIP(c, hipMemcpyAsync(f->in_l0, pts_l0, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
      VO_CHECK_HIP(c, hipMemcpyAsync(f->in_r0, pts_r0, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
      VO_CHECK_HIP(c, hipMemcpyAsync(f->in_X, Xp, sizeof(float) * 3 * n, hipMemcpyHostToDevice, s));
    }
    if (n_new > 0 && !tab) {
      VO_CHECK_HIP(c, hipMemcpyAsync(f->in_new, pts_new, sizeof(float) * 2 * n_new, hipMemcpyHostToDevice, s));
      d_new = f->in_new;
    }
    d_l0 = f->in_l0;
    d_r0 = f->in_r0;
    d_X = f->in_X;
  }
  // (the table is filled on the side stream, long before — unless its detection is deferred: then only the candidates'
  // launch, on the side stream itself, reads it)
  const bool split = tab && c->frame->defer_detect && vo_frame_fused_supported(prm->win) && n > 0 && c->ingest_side && !c->frame_conc_off;
  if (tab && c->frame->defer_detect == 1 && !split) RC(vo_new_point_candidates_enqueue(c, slot_l1, bp, table));  // (cannot overlap: now)
  if (tab && !split) VO_CHECK_HIP(c, hipStreamWaitEvent(s, tab->ready, 0));
  // ---- carve the packed result block for this frame ----
  f->n = n;
  f->n_new = n_new;
  f->closed = tab ? 1 : 0;
  f->table = tab;
  size_t off = align16(sizeof(vo_frame_hdr));
  f->off_stage = off;  off += align16((size_t)n);
  if (!tab) {          // open loop: the frame kernel writes the candidates' results, final before the BA launch
    f->off_mnew = off;  off += align16((size_t)n_new);
  }
  const size_t late_end = off;  // header + stage bytes: what the BA launch still changes, copied last
  f->off_pl1 = off;    off += align16(sizeof(float) * 2 * (size_t)n);
  f->off_pr1 = off;    off += align16(sizeof(float) * 2 * (size_t)n);
  if (!tab) {
    f->off_newr = off;  off += align16(sizeof(float) * 2 * (size_t)n_new);
    f->off_newl = 0;
  }
  // closed step [10]: mnew / newr / newl are written by the BA launch's epilogue, entry by entry, to the device AND
  // the host block; they are the tail of the block, outside both bulk copies
  const size_t bulk_end = off;
  if (tab) {
    f->off_mnew = off;  off += align16((size_t)n_new);
    f->off_newr = off;  off += align16(sizeof(float) * 2 * (size_t)n_new);
    f->off_newl = off;  off += align16(sizeof(float) * 2 * (size_t)n_new);
  }
  f->res_bytes = off;
  if (f->res_bytes > f->res_cap) VO_FAIL(c, VO_ERR_CAPACITY, "result block of %zu bytes exceeds the context's (max_points too small)", f->res_bytes);
  f->hdr = (vo_frame_hdr *)f->res_dev;
  f->stage = f->res_dev + f->off_stage;
  f->mNew = f->res_dev + f->off_mnew;
  f->F_pl1 = (float *)(f->res_dev + f->off_pl1);
  f->F_pr1 = (float *)(f->res_dev + f->off_pr1);
  f->new_r = (float *)(f->res_dev + f->off_newr);
  int *cnt = f->hdr->cnt;
  const bool fused = n > 0 && vo_frame_fused_supported(prm->win);
  // Strict-border mode 4: the replay next to the frame kernel pays (a pool of resident workgroups) when there is
  // something to replay, and a frame's border features are mostly the previous frame's — so the previous frame's count
  // decides; the results are the same either way. The pool is sized by that count too.
  c->frame_strict_now = c->frame_strict_ic;
  // (a frame expected to replay nothing: stream-ordered; the gated arrangement, mode 5, measures the same there)
  // (and a frame kernel of many times the chip's resident wavefronts is throughput-bound to its end: the pool then takes
  // from it more than the early start gives back — configs[4], 16000 workgroups: 788 against 892 frames/s)
  if (c->frame_strict_ic == 4)
    c->frame_strict_now = (fused && f->last_replayed >= VO_CONC_MIN_REPLAYED && n + n_new <= VO_CONC_MAX_WORKGROUPS) ? 3 : 1;
  if (c->frame_strict_ic == 5 && !fused) c->frame_strict_now = 1;
  if (c->frame_conc_off && c->frame_strict_now >= 3) c->frame_strict_now = 1;  // a join timed out before: stream order
  {
    int g = ((f->last_replayed + 32 + 31) / 32) * 32;
    f->conc_grid = g < 64 ? 64 : (g > 256 ? 256 : g);
    if (c->frame_strict_ic == 3) f->conc_grid = 256;  // (fixed when asked for explicitly)
  }
  // the fused path writes every header field itself and keeps its control block zero between frames
  if (!fused) VO_CHECK_HIP(c, hipMemsetAsync(f->hdr, 0, sizeof(vo_frame_hdr), s));  // counts, flags

  const int W = prm->width, H = prm->height;
  float T_rl[16], T_cp[16];
  inv_se3(prm->T_lr, T_rl);
  inv_se3(dT_prior, T_cp);
  if (T_cw_prior) memcpy(T_cp, T_cw_prior, sizeof(T_cp));  // world points: X_l1 = T_cw_prior * X (stereo_vo.cpp:493)

  if (n_new > 0 && !fused) VO_CHECK_HIP(c, hipEventRecord(c->ev_fork, s));  // the new pyramids are enqueued before this point
  bool cand_joined = false;  // the candidates went out as a launch of their own on the side stream: the BA launch joins them

  if (fused) {
    // steps [3] .. [5] of a feature are one wavefront of ONE launch (frame_fused.hip)
    vo_frame_fused_bufs b;
    b.scale = f->F_scale;
    b.k1 = f->A_pl1;
    b.pr_prior = f->A_pr1;
    b.pl1 = f->F_pl1;
    b.pr1 = f->F_pr1;
    b.ref = f->A_ref;
    b.lastpu = f->A_lastpu;
    b.stage = f->stage;
    b.m2 = f->m2;
    b.touched = f->A_touched;
    b.cls = f->A_cls;
    b.ctl = f->ctl;
    b.sync = f->sync;
    b.sync_p1_target = &f->sync_p1_target;
    b.sync_done_target = &f->sync_done_target;
    b.conc_grid = f->conc_grid;
    b.hdr_flags = &f->hdr->flags;
    b.C_X = f->C_X;
    b.C_pl1 = f->C_pl1;
    b.C_pr1 = f->C_pr1;
    b.C_orig = f->C_orig;
    b.cnt = cnt;
    b.new_r = tab ? f->bin_r : f->new_r;   // closed: per-bin scratch, compacted into the block by the BA launch
    b.m_new = tab ? f->bin_m : f->mNew;
    b.cand_has = tab ? tab->has : nullptr;
    b.split_cands = split ? 1 : 0;
    b.cand_done = split ? f->cand_done : nullptr;
    // what vo_stereo_frame_result needs to issue this frame again (device pointers only)
    f->again.prm = *prm;
    f->again.slot_l0 = slot_l0;
    f->again.slot_l1 = slot_l1;
    f->again.slot_r1 = slot_r1;
    f->again.l0 = d_l0;
    f->again.r0 = d_r0;
    f->again.X = d_X;
    f->again.fl = d_fl;
    f->again.pts_new = tab ? nullptr : d_new;
    f->again.n = n;
    f->again.n_new = tab ? 0 : n_new;
    memcpy(f->again.dT_prior, dT_prior, sizeof(f->again.dT_prior));
    f->again.has_bins = bp ? 1 : 0;
    if (bp) f->again.bins = *bp;
    f->again.table = table;
    f->again.adv = f->adv_next;  // (consumed below)
    f->again.has_world = T_pw ? 1 : 0;
    if (T_pw) {
      memcpy(f->again.T_pw, T_pw, sizeof(f->again.T_pw));
      memcpy(f->again.T_cw_prior, T_cw_prior, sizeof(f->again.T_cw_prior));
    }
    // [10] the new-point candidates are extra workgroups of the same launch
    VO_TT("setup");
    const int p1_before = f->sync_p1_target, done_before = f->sync_done_target;
    int rcf = vo_frame_fused_enqueue(c, prm, slot_l0, slot_l1, slot_r1, d_l0, d_r0, d_X, d_fl, n, T_cp, T_rl, d_new, n_new, b, 0, T_pw);
    VO_TT("track launch");
    if (rcf >= 0)
      rcf = vo_frame_fused_enqueue(c, prm, slot_l0, slot_l1, slot_r1, d_l0, d_r0, d_X, d_fl, n, T_cp, T_rl, d_new, n_new, b, 1, T_pw);
    if (rcf >= 0 && split) {
      // the features are on their way: now the detector chain of the current left image and, behind it on the same (side)
      // stream, the candidates' launch; the BA launch waits for both through ev_join
      if (c->frame->defer_detect == 1) rcf = vo_new_point_candidates_enqueue(c, slot_l1, bp, table);  // (2: already there)
      if (rcf >= 0) {
        c->stream = c->stream2;
        rcf = vo_frame_fused_enqueue(c, prm, slot_l0, slot_l1, slot_r1, d_l0, d_r0, d_X, d_fl, n, T_cp, T_rl, d_new, n_new, b, 2, T_pw);
        c->stream = s;
        // (no event: the BA launch joins the candidates on the device — its DLT workers and its epilogue wait for the
        // count of finished candidate workgroups, bounded; a cross-queue event pair in front of the BA launch kept its
        // iterations from running under the detection)
        if (rcf >= 0) {
          cand_joined = true;
          vo_wrap_add(f->cand_total, n_new);  // (cumulative, like the word the candidate workgroups count in)
        }
      }
    }
    if (rcf < 0) {  // nothing of this frame will count: the cumulative hand-shake targets go back
      f->sync_p1_target = p1_before;
      f->sync_done_target = done_before;
      return rcf;
    }
    VO_TT("phase1");
  } else if (n > 0) {
    // general window sizes: one launch per step, compaction in between
    // [3] priors
    RC(vo_stereo_prior_enqueue(c, d_X, d_l0, d_r0, d_fl, n, T_cp, T_rl, prm->Kl, prm->Kr, W, H, f->F_pl1, f->F_pr1,
                               f->F_scale, f->F_orig, f->stage));
    // [4] l0 -> l1 ({} criteria, {} minEig); validity mask fused into the compaction
    RC(vo_klt_enqueue(c, slot_l0, slot_l1, d_l0, nullptr, f->F_pl1, n, nullptr, prm->win, prm->max_level,
                      VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f, f->st1, f->e1));
    // (the main chain's first long kernel is queued; now feed the side stream)
    // [10] new points on the side stream: they depend only on the two new pyramids, not on the
    // main chain, and the chain's kernels leave most of the chip idle (one wave per point).
    if (n_new > 0) {
      VO_CHECK_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      c->stream = c->stream2;
      int rc2 = vo_klt_enqueue(c, slot_l1, slot_r1, d_new, nullptr, f->new_r, n_new, nullptr, prm->win,
                               prm->max_level, 0, 30, 0.01, 1e-4f, f->st3, f->e3);
      // backward: maxLevel-1, initial flow = pts_new, {} criteria / minEig (feature_tracker.cpp:69-71)
      if (rc2 >= 0)
        rc2 = vo_klt_enqueue(c, slot_r1, slot_l1, f->new_r, d_new, f->new_back, n_new, nullptr, prm->win,
                             prm->max_level - 1, VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f, f->st2, f->e2);
      if (rc2 >= 0)
        rc2 = vo_klt_mask_enqueue(c, 2, n_new, nullptr, W, H, prm->thres_err, prm->thres_bidirection, d_new,
                                  f->new_r, f->new_back, f->st3, f->st2, f->e3, f->e2, nullptr, f->mNew);
      c->stream = s;
      if (rc2 < 0) return rc2;
      VO_CHECK_HIP(c, hipEventRecord(c->ev_join, c->stream2));
    }
    {
      CompactArgsHost h;
      h.klt_status = f->st1;
      h.klt_err = f->e1;
      h.klt_pts = f->F_pl1;
      h.klt_thres_err = prm->thres_err;
      h.klt_W = W;
      h.klt_H = H;
      h.lm_flags = d_fl;  // landmark.cpp:305: && isAlive() && isTracked()
      h.lm_reject = VO_LM_DROPPED;
      h.n = n;
      h.d_n_out = &cnt[0];
      h.in2[0] = d_l0;      h.out2[0] = f->A_pl0;
      h.in2[1] = f->F_pl1;  h.out2[1] = f->A_pl1;
      h.in2[2] = f->F_pr1;  h.out2[2] = f->A_pr1;
      h.in3 = d_X;          h.out3 = f->A_X;
      h.in1 = f->F_scale;   h.out1 = f->A_scale;
      h.in_i = f->F_orig;   h.out_i = f->A_orig;
      h.stage = f->stage;
      h.stage_val = 1;
      RC(vo_compact_enqueue(c, h));
    }
    // [4-1] scale-compensated refinement on the compacted set (entry mask all true)
    RC(vo_ic_enqueue(c, slot_l0, slot_l1, f->A_pl0, f->A_scale, f->A_pl1, f->A_ref, nullptr, f->m2, f->A_touched,
                     f->A_cls, f->A_lastpu, n, &cnt[0], &f->hdr->flags, c->frame_strict_now != 0));
    if (c->frame_strict_now)
      RC(vo_ic_strict_enqueue(c, slot_l0, slot_l1, f->A_pl0, f->A_scale, f->A_pl1, f->A_ref, f->m2, f->A_touched,
                              f->A_cls, f->A_lastpu, n, &cnt[0], &f->hdr->flags));
    {
      CompactArgsHost h;
      h.mask = f->m2;
      h.n = n;
      h.d_n = &cnt[0];
      h.d_n_out = &cnt[1];
      h.in2[0] = f->A_ref;  h.out2[0] = f->B_pl1;
      h.in2[1] = f->A_pr1;  h.out2[1] = f->B_pr1;
      h.in3 = f->A_X;       h.out3 = f->B_X;
      h.in_i = f->A_orig;   h.out_i = f->B_orig;
      h.stage = f->stage;
      h.stage_val = 2;
      h.sc_src = f->A_ref;  // refined left pixels back to input index space
      h.sc_dst = f->F_pl1;
      RC(vo_compact_enqueue(c, h));
    }
    // [5] l1 -> r1
    RC(vo_klt_enqueue(c, slot_l1, slot_r1, f->B_pl1, nullptr, f->B_pr1, n, &cnt[1], prm->win, prm->max_level,
                      VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f, f->st1, f->e1));
    {
      CompactArgsHost h;
      h.klt_status = f->st1;
      h.klt_err = f->e1;
      h.klt_pts = f->B_pr1;
      h.klt_thres_err = prm->thres_err;
      h.klt_W = W;
      h.klt_H = H;
      h.n = n;
      h.d_n = &cnt[1];
      h.d_n_out = &cnt[2];
      h.in2[0] = f->B_pl1;  h.out2[0] = f->C_pl1;
      h.in2[1] = f->B_pr1;  h.out2[1] = f->C_pr1;
      h.in3 = f->B_X;       h.out3 = f->C_X;
      h.in_i = f->B_orig;   h.out_i = f->C_orig;
      h.stage = f->stage;
      h.stage_val = 3;
      h.sc_src = f->B_pr1;
      h.sc_dst = f->F_pr1;
      RC(vo_compact_enqueue(c, h));
    }
  }
  if (n == 0 && n_new > 0) VO_FAIL(c, VO_ERR_INVALID, "new-point candidates without a track set are not supported");
  // [6] stereo pose-only BA on the survivors (T01 init = constant-velocity prior; kept on NaN);
  // [7] its epilogue marks stage 4 for inliers that pass the y > 660 gate (thres_sampson = 60)
  // The GN launch's prologue selects the BA set — survivors of [5] whose landmark is triangulated (:599-613), in
  // index order — and counts the steps; its epilogue marks stage 4 (BA inlier, or survivor outside the BA set whose
  // mask_motion stays true, :582; both past the gate of [7]).
  vo_gn_frame gf;
  memset(&gf, 0, sizeof(gf));
  if (n > 0) {
    gf.n = n;
    gf.stage = f->stage;
    gf.lm_flags = d_fl;
    gf.X = d_X;
    gf.T_pw = T_pw;
    gf.pl1 = f->F_pl1;
    gf.pr1 = f->F_pr1;
    gf.C_X = f->C_X;
    gf.C_pl1 = f->C_pl1;
    gf.C_pr1 = f->C_pr1;
    gf.C_orig = f->C_orig;
    gf.cnt = cnt;
    gf.hdr_flags = &f->hdr->flags;
  }
  if (fused) {
    gf.ctl = f->ctl;
    if (c->frame_strict_now == 3 || c->frame_strict_now == 5) {  // the replay runs on its own stream: join on the device
      gf.join_word = f->sync + 1;
      gf.join_target = f->sync_done_target;
      if (c->dbg[VO_DBG_FAIL_JOIN]) gf.join_target += 1 << 20;  // tests: a join that cannot be met
    }
    gf.ctl_words = (int)(vo_ic_ctl_bytes() / 4);
    gf.nt_word = vo_ic_ctl_nt_word();
    gf.res_dev = f->res_dev;
    gf.res_host = f->res_host;  // pinned host memory is device-visible: the kernel copies the block out itself
    gf.res_bytes = bulk_end;
    f->seq = f->seq + 1 == 0 ? 1 : f->seq + 1;
    gf.seq = f->seq;
    gf.seq_word = (int)(offsetof(vo_frame_hdr, seq) / 4);
    gf.res_late_bytes = tab ? late_end : f->off_mnew;  // header + stage bytes
    if (tab) {
      gf.np_bins = tab->n_bins;
      gf.np_bins_u = bp->n_bins_u;
      gf.np_u_step = bp->u_step;
      gf.np_v_step = bp->v_step;
      gf.np_has = tab->has;
      gf.np_xy = tab->xy;
      gf.np_bin_r = f->bin_r;
      gf.np_bin_m = f->bin_m;
      gf.np_out_l = (float *)(f->res_dev + f->off_newl);
      gf.np_out_r = f->new_r;
      gf.np_out_m = f->mNew;
      gf.np_host_l = (float *)(f->res_host + f->off_newl);
      gf.np_host_r = (float *)(f->res_host + f->off_newr);
      gf.np_host_m = f->res_host + f->off_mnew;
      if (cand_joined) {
        gf.np_cand_done = f->cand_done;
        gf.np_cand_target = f->cand_total;
        if (c->dbg[VO_DBG_FAIL_JOIN]) gf.np_cand_target += 1 << 20;  // tests: a join that cannot be met
      }
    }
  }
  VoAdvArgs adv_now = f->adv_next;
  f->adv_next.on = 0;
  int adv_workers = 0;
  if (adv_now.on) {
    if (!(fused && tab)) VO_FAIL(c, VO_ERR_INVALID, "the track-set advance needs the closed frame on the fused path");
    adv_workers = (tab->n_bins + 63) / 64;  // one worker wavefront per 64 bins (gn_pose.hip)
    adv_now.dlt_done = f->adv_done;
    adv_now.dlt_target = (int)((unsigned)f->adv_total + (unsigned)adv_workers);
    gf.adv = &adv_now;
  }
  RC(vo_gn_enqueue(c, true, false, f->C_X, f->C_pl1, f->C_pr1, n, nullptr, prm->Kl, prm->Kr,
                   prm->T_lr, prm->thres_poseba, 0, dT_prior, f->hdr->dT, f->mG, &f->hdr->gn, true,
                   n > 0 ? f->stage : nullptr, f->C_orig, 4, prm->thres_sampson, n > 0 ? &gf : nullptr));
  vo_wrap_add(f->adv_total, adv_workers);  // (cumulative, like the word the workers count in: only a launch that went out counts)
  if (n_new > 0 && !fused) VO_CHECK_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
  VO_TT("gn launch");
  // one D2H of the packed block into pinned memory (general path; the fused path's GN launch did it)
  if (!fused) VO_CHECK_HIP(c, hipMemcpyAsync(f->res_host, f->res_dev, f->res_bytes, hipMemcpyDeviceToHost, s));
  VO_TT("memcpy");
  VO_CHECK_HIP(c, hipEventRecord(f->ev_done, s));
  VO_TT("event");
  f->seq_poll = fused;
  f->pending = true;
  c->frame_slots_busy = c->ingest_side;  // (one stream orders a rebuild behind the frame by itself)
  c->frame_slot[0] = slot_l0;
  c->frame_slot[1] = slot_l1;
  c->frame_slot[2] = slot_r1;
  return VO_OK;
}

extern "C" int vo_stereo_frame_enqueue(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1,
                                       int slot_r1, const float *pts_l0, const float *pts_r0, const float *Xp,
                                       const uint8_t *flags, int n, const float dT_prior[16], const float *pts_new,
                                       int n_new, int inputs_on_device) {
  return vo_frame_enqueue_impl(c, prm, slot_l0, slot_l1, slot_r1, pts_l0, pts_r0, Xp, flags, n, dT_prior, pts_new, n_new,
                               inputs_on_device, nullptr, 0, nullptr, nullptr);
}

extern "C" int vo_stereo_frame_enqueue_closed(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1,
                                              int slot_r1, const float *pts_l0, const float *pts_r0, const float *Xp,
                                              const uint8_t *flags, int n, const float dT_prior[16],
                                              const vo_bin_params *bins, int table, int inputs_on_device) {
  if (!bins) return VO_ERR_INVALID;
  return vo_frame_enqueue_impl(c, prm, slot_l0, slot_l1, slot_r1, pts_l0, pts_r0, Xp, flags, n, dT_prior, nullptr, 0,
                               inputs_on_device, bins, table, nullptr, nullptr);
}

extern "C" int vo_stereo_frame_enqueue_closed_world(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1,
                                                    int slot_r1, const float *pts_l0, const float *pts_r0, const float *Xw,
                                                    const uint8_t *flags, int n, const float dT_prior[16],
                                                    const float T_pw[16], const float T_cw_prior[16],
                                                    const vo_bin_params *bins, int table, int inputs_on_device) {
  if (!bins || !T_pw || !T_cw_prior) return VO_ERR_INVALID;
  return vo_frame_enqueue_impl(c, prm, slot_l0, slot_l1, slot_r1, pts_l0, pts_r0, Xw, flags, n, dT_prior, nullptr, 0,
                               inputs_on_device, bins, table, T_pw, T_cw_prior);
}

extern "C" int vo_stereo_frame_recoveries(const vo_ctx *c) { return c ? c->frame_recoveries : VO_ERR_INVALID; }

extern "C" int vo_debug_set(vo_ctx *c, int key, int value) {
  if (!c || key < 0 || key >= VO_DBG_COUNT) return VO_ERR_INVALID;
  c->dbg[key] = value;
  return VO_OK;
}
extern "C" int vo_debug_allocation_count(const vo_ctx *c, long long *count) {
  if (!c || !count) return VO_ERR_INVALID;
  *count = c->n_allocs;
  return VO_OK;
}

extern "C" int vo_stereo_frame_new_points(vo_ctx *c, float *pts_new, int *n_new) {
  if (!c || !c->frame || !n_new) return VO_ERR_INVALID;
  vo_frame_state *f = c->frame;
  if (f->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_stereo_frame_result first");
  if (!f->closed) VO_FAIL(c, VO_ERR_INVALID, "the last frame was not enqueued with vo_stereo_frame_enqueue_closed");
  const vo_frame_hdr *h = (const vo_frame_hdr *)f->res_host;
  *n_new = h->cnt[5];
  if (pts_new && h->cnt[5] > 0) memcpy(pts_new, f->res_host + f->off_newl, sizeof(float) * 2 * (size_t)h->cnt[5]);
  return VO_OK;
}

extern "C" int vo_stereo_frame_result(vo_ctx *c, float *pts_l1, float *pts_r1, uint8_t *stage, float dT[16],
                                      float *pts_new_r, uint8_t *mask_new, vo_frame_counts *counts,
                                      vo_gn_info *gn) {
  if (!c || !c->frame || !c->frame->pending) return VO_ERR_INVALID;
  vo_frame_state *f = c->frame;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  // wait for THIS frame's results only: work enqueued after it (e.g. the next frame's pyramids)
  // keeps running. Fused path: the BA launch writes the frame's sequence number into the pinned block after
  // everything else; polling that word costs a microsecond where an event wait costs tens (bounded: after ~2 ms
  // of polling, or on the general path, the event decides).
  bool seen = false;
  if (f->seq_poll) {
    volatile const int *seqp = &((volatile const vo_frame_hdr *)f->res_host)->seq;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int spin = 0;; ++spin) {
      if (*seqp == f->seq) {
        seen = true;
        break;
      }
      if ((spin & 255) == 255) {  // ~2 ms of polling at most, then the event decides
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec) > 2e6) break;
      }
      if (c->dbg[VO_OPT_POLL_YIELD]) sched_yield();
#if defined(__x86_64__)
      else __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  if (!seen) VO_CHECK_HIP(c, hipEventSynchronize(f->ev_done));
  f->pending = false;
  c->frame_slots_busy = 0;
  f->recovered = 0;
  const vo_frame_hdr *h = (const vo_frame_hdr *)f->res_host;
  if ((h->flags & 8) && f->seq_poll && f->again.n > 0) {
    // The device-side join with the replay stream timed out (the two queues did not run concurrently: a serialising tool,
    // a busy GPU, a stalled main stream). The frame is not lost: drain both streams, re-base the hand-shake words, switch
    // the context to the stream-ordered replay for good and issue the frame again from its (intact) device inputs.
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
    const auto g = f->again;  // (by value: the enqueue below rewrites f->again)
    f->adv_next = g.adv;      // (StereoVO: the re-issued frame builds the next track set again)
    int rc2 = vo_frame_enqueue_impl(c, &g.prm, g.slot_l0, g.slot_l1, g.slot_r1, g.l0, g.r0, g.X, g.fl, g.n, g.dT_prior, g.pts_new,
                                    g.n_new, 1, g.has_bins ? &g.bins : nullptr, g.table, g.has_world ? g.T_pw : nullptr,
                                    g.has_world ? g.T_cw_prior : nullptr);
    if (rc2 < 0) return rc2;
    VO_CHECK_HIP(c, hipEventSynchronize(f->ev_done));
    f->pending = false;
    c->frame_slots_busy = 0;
    f->recovered = 1;
  }
  f->last_replayed = f->n > 0 ? h->cnt[3] : 0;
  const int n = f->n, nn = f->closed ? h->cnt[5] : f->n_new;  // closed: what the BA launch's epilogue emitted
  if (f->closed && f->table) {  // the detector's capacity flags of the table this frame read
    int rcf = f->table->h_flags[0];
    if (rcf & 1) VO_FAIL(c, VO_ERR_CAPACITY, "more FAST corners on one pyramid level than the detector's lists hold");
    if (rcf & 2) VO_FAIL(c, VO_ERR_CAPACITY, "more keypoints than the detector's output buffer holds");
  }
  if (pts_l1 && n) memcpy(pts_l1, f->res_host + f->off_pl1, sizeof(float) * 2 * (size_t)n);
  if (pts_r1 && n) memcpy(pts_r1, f->res_host + f->off_pr1, sizeof(float) * 2 * (size_t)n);
  if (stage && n) memcpy(stage, f->res_host + f->off_stage, (size_t)n);
  if (dT) memcpy(dT, h->dT, sizeof(float) * 16);
  if (pts_new_r && nn) memcpy(pts_new_r, f->res_host + f->off_newr, sizeof(float) * 2 * (size_t)nn);
  if (mask_new && nn) memcpy(mask_new, f->res_host + f->off_mnew, (size_t)nn);
  if (counts) {
    counts->n_l0l1 = h->cnt[0];
    counts->n_refine = h->cnt[1];
    counts->n_l1r1 = h->cnt[2];
    int ninl = 0, nnew = 0;
    const uint8_t *sg = f->res_host + f->off_stage, *mn = f->res_host + f->off_mnew;
    for (int i = 0; i < n; ++i) ninl += sg[i] == 4;
    for (int i = 0; i < nn; ++i) nnew += mn[i] != 0;
    counts->n_inlier = ninl;
    counts->n_new_ok = nnew;
    counts->gn_iterations = h->gn.iterations;
    counts->n_replayed = h->cnt[3];
    counts->n_ba = h->cnt[4];
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
    if (h->flags & 16) VO_FAIL(c, VO_ERR_HIP, "the triangulation workers of the BA launch did not finish");
    VO_FAIL(c, VO_ERR_NAN_UPDATE, "dtu dtv nan");
  }
  if (h->gn.is_nan) VO_FAIL(c, VO_ERR_GN_FAILED, "PoseOnlyStereoBA is failed!");
  return VO_OK;
}
