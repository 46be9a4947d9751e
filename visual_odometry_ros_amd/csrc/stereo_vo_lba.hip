// stereo_vo_lba.hip — the local bundle adjustment of a StereoVO keyframe: the landmark / keyframe bookkeeping that
// SparseBAParameters walks in the reference, around the solver kernels of sba.hip — RESIDENT ON THE DEVICE.
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
//
// Layout. Landmark ids are dense integers handed out in order (landmark.cpp counter), a track set lists them ascending,
// and a landmark alive at some frame was alive at every keyframe since its creation — so everything the window can
// refer to lies in the id interval [first id of the oldest keyframe, next id to hand out). That makes every container of
// the reference a direct-address array in HBM:
//   landmark table   X (float3), state (bit 0 triangulated, bit 1 dead), tag (the id), slot = id mod 2^24
//   keyframe ring    per window keyframe its related landmarks' ids and both pixels (copied from the track set)
//   window scratch   per id of the interval: bit mask of the window keyframes that saw it, its entry index in each
// and the BA problem is built by six small launches — mark/copy, scatter, qualify (one lane per id, packed counts scanned
// per workgroup), scan (one workgroup over the workgroups' totals), fill (one lane per id), lists (one workgroup per
// gather list) — straight into the solver's arena; the solver's results go back into the
// table and the track set by two more. The host sees nine poses, ten error values and three counts per solve. (The
// first version merged the window's id lists on the host and uploaded 1.6 MB per keyframe: 0.41 ms of host time around
// 0.64 ms of kernels.)
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>

#include "sba_device.hpp"
#include "stereo_vo.hpp"

static double lba_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}

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

// ---- device side ------------------------------------------------------------------------------------------------------
#define LBA_KW 16            // window keyframes at most (bits of the mask that are used)
// landmark table: 2^k slots (17 B each), addressed by id mod 2^k with the id kept as a tag. k starts at LBA_TAB_FIRST_BITS
// and follows the ids the stream has handed out (every landmark ever created stays readable for stats_keyframe, like the
// reference's all_landmarks_) up to LBA_TAB_BITS; beyond that the oldest ids are overwritten (they read as the origin).
#define LBA_TAB_FIRST_BITS 18
#define LBA_TAB_BITS 24
// ids the window may span: from the first id of the oldest keyframe to the next id to hand out — set by the OLDEST landmark
// still tracked at the oldest keyframe (a track that never dies keeps the interval growing at ~200 ids per frame: 2^24 ids
// are ~80 000 frames). The window scratch (12 + 4 kf_window bytes per id of the interval) is sized at construction for
// LBA_SPAN_FIRST ids (or twice the window's capacity, if that is more) and doubles when an interval outgrows it.
#define LBA_SPAN_MAX (1 << LBA_TAB_BITS)
#define LBA_SPAN_FIRST (1 << 16)

struct LmTab {
  float *X;       // [slots][3] lm->get3DPoint()
  uint8_t *S;     // [slots] bit 0 isTriangulated(), bit 1 !isAlive()
  int32_t *tag;   // [slots] the id the slot holds (-1: none)
  int mask;       // slots - 1
};
struct LbaWin {
  int nk, No, W, base;  // window keyframes, optimised poses, ids spanned, first id
  int kw;               // entries per id in q_w (= kf_window of the stream)
  int opk;              // observations per keyframe entry: 2 (stereo: left, right) or 1 (MonoVO's window: one camera, and a
                        // landmark then needs two window keyframes, THRES_MINIMUM_SEEN of sparse_ba_parameters.h:313)
  int n[LBA_KW];        // related landmarks of window keyframe j
  const int32_t *ids[LBA_KW];
  const float *pl[LBA_KW], *pr[LBA_KW];
  int opt[LBA_KW];      // optimised-pose index of window keyframe j, -1: fixed
  int optf[LBA_KW];     // window keyframe of optimised pose k
  int optmask;          // bits of the optimised window keyframes
};
struct LbaProb {  // the problem inside the solver's arena (non-const views of SbaDev's lists) + the builder's own arrays
  double *T, *X, *px;
  int *opt_index, *opt_frame, *obs_ptr, *obs_frame, *slot_ptr, *slot_j, *slot_bobs, *slot_lm;
  uint8_t *obs_right;
  int *pose_obs_ptr, *pose_obs_end, *pose_obs, *pose_lm, *pose_slot_ptr, *pose_slot_end, *pose_slot;
  int *pair_ptr, *pair_end, *pair_a, *pair_b;
  int *flags, *dyn;
  double *avg_err;
  int32_t *used_id;
  int *lm_mask;
  int *mask_w, *q_w;                  // window scratch, per id of the interval
  unsigned long long *pre, *wg_tot;   // packed running counts: inside the workgroup / of the workgroups
  int pose_obs_stride, pose_slot_stride, pair_stride, max_iter;
};

// the new keyframe: its related landmarks' state as of now (Landmark::set3DPoint / isTriangulated at keyframe time; a
// dead landmark stays dead) and the keyframe's own copy of ids and pixels
__global__ void lba_keyframe_kernel(SvoTrackSet ts, int n, LmTab tab, int32_t *kf_ids, float *kf_pl, float *kf_pr, int32_t *all_ids) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int32_t id = ts.ids[k];
  const int t = id & tab.mask;
  const uint8_t dead = tab.tag[t] == id ? (uint8_t)(tab.S[t] & 2) : (uint8_t)0;
  tab.X[3 * (size_t)t] = ts.Xw[3 * k];
  tab.X[3 * (size_t)t + 1] = ts.Xw[3 * k + 1];
  tab.X[3 * (size_t)t + 2] = ts.Xw[3 * k + 2];
  tab.S[t] = (uint8_t)(dead | ((ts.flags[k] & VO_LM_TRIANGULATED) ? 1 : 0) | ((ts.flags[k] & VO_LM_BUNDLED) ? 4 : 0));
  tab.tag[t] = id;
  kf_ids[k] = id;
  all_ids[k] = id;  // (the keyframe's related landmarks, kept for good: stats_keyframe)
  kf_pl[2 * k] = ts.pts_l[2 * k];
  kf_pl[2 * k + 1] = ts.pts_l[2 * k + 1];
  kf_pr[2 * k] = ts.pts_r[2 * k];
  kf_pr[2 * k + 1] = ts.pts_r[2 * k + 1];
}

// Landmark::set3DPoint outside a keyframe (MonoVO's initialisation, mono_vo.cpp:660-687): landmarks that a keyframe already
// put into the table take their new point and state; the others enter the table at their first keyframe
__global__ void lba_table_update_kernel(SvoTrackSet ts, int n, LmTab tab) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int32_t id = ts.ids[k];
  const int t = id & tab.mask;
  if (tab.tag[t] != id) return;
  tab.X[3 * (size_t)t] = ts.Xw[3 * k];
  tab.X[3 * (size_t)t + 1] = ts.Xw[3 * k + 1];
  tab.X[3 * (size_t)t + 2] = ts.Xw[3 * k + 2];
  tab.S[t] = (uint8_t)((tab.S[t] & 2) | ((ts.flags[k] & VO_LM_TRIANGULATED) ? 1 : 0) | ((ts.flags[k] & VO_LM_BUNDLED) ? 4 : 0));
}

// getObservationsOnKeyframes restricted to the window: which keyframes saw id, and where
__global__ void lba_scatter_kernel(LbaWin w, int *mask_w, int *q_w) {
  const int j = blockIdx.y, q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= w.n[j]) return;
  const int idx = w.ids[j][q] - w.base;
  if ((unsigned)idx >= (unsigned)w.W) return;
  atomicOr(&mask_w[idx], 1 << j);
  q_w[(size_t)idx * w.kw + j] = q;
}

// exclusive scan of one int per thread over a workgroup of 1024 (16 wavefronts); every thread gets the total too
__device__ __forceinline__ int lba_block_scan(int v, int *s_w, int &total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  __syncthreads();  // (s_w may still be read from the previous scan)
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = s_w[k];
    before += k < wave ? c : 0;
    tot += c;
  }
  total = tot;
  return before + inc - v;
}

// the same for a 64-bit value and a workgroup of NW wavefronts (three counts packed into one word)
template <int NW>
__device__ __forceinline__ unsigned long long lba_block_scan64(unsigned long long v, unsigned long long *s_w,
                                                               unsigned long long &total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  unsigned long long before = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const unsigned long long c = s_w[k];
    before += k < wave ? c : 0;
    tot += c;
  }
  total = tot;
  return before + inc - v;
}
// three running counts place a landmark: its index, its observation pairs (one per keyframe that saw it), its slots —
// packed as 20 + 22 + 22 bits (the host checks the bounds)
#define LBA_PK_KF 20
#define LBA_PK_SL 42
#define LBA_QWG 256

// one lane per id of the interval: does it enter the problem (triangulated && alive, seen by the window: a stereo
// keyframe gives two observations, THRES_MINIMUM_SEEN = 2 always holds)? Its packed counts, scanned inside the workgroup.
__global__ __launch_bounds__(LBA_QWG) void lba_qualify_kernel(LbaWin w, LmTab tab, LbaProb p) {
  __shared__ unsigned long long s_w[LBA_QWG / 64];
  const int idx = blockIdx.x * LBA_QWG + threadIdx.x;
  int m = idx < w.W ? p.mask_w[idx] : 0;
  if (m) {
    const int32_t id = w.base + idx;
    const int t = id & tab.mask;
    const uint8_t st = tab.tag[t] == id ? tab.S[t] : (uint8_t)0;
    if (!((st & 1) && !(st & 2)) || w.opk * __popc(m) < 2) {  // isTriangulated() && isAlive(); kfs_seen.size() >= 2
      m = 0;
      p.mask_w[idx] = 0;
    }
  }
  const unsigned long long v =
      m ? 1ull | ((unsigned long long)__popc(m) << LBA_PK_KF) | ((unsigned long long)__popc(m & w.optmask) << LBA_PK_SL) : 0ull;
  unsigned long long total;
  const unsigned long long ex = lba_block_scan64<LBA_QWG / 64>(v, s_w, total);
  if (idx < w.W) p.pre[idx] = ex;
  if (threadIdx.x == 0) p.wg_tot[blockIdx.x] = total;
}

// ONE workgroup: the workgroups' totals become offsets, the totals of the interval the problem's counts. Also what the
// host would have uploaded: poses, index maps, zeroed flags — from the kernel arguments.
struct LbaHead {
  double T_jw[16 * LBA_KW];
};
__global__ __launch_bounds__(1024) void lba_scan_kernel(LbaWin w, LbaProb p, LbaHead h, int n_wg) {
  __shared__ unsigned long long s_w[16];
  const int tid = threadIdx.x;
  for (int k = tid; k < 16 * w.nk; k += 1024) p.T[k] = h.T_jw[k];
  if (tid < w.nk) p.opt_index[tid] = w.opt[tid];
  if (tid < w.No) p.opt_frame[tid] = w.optf[tid];
  if (tid < 16) p.flags[tid] = 0;
  if (tid <= p.max_iter) p.avg_err[tid] = 0.0;
  const int chunk = (n_wg + 1023) / 1024, i0 = tid * chunk, i1 = min(i0 + chunk, n_wg);
  unsigned long long c = 0;
  for (int g = i0; g < i1; ++g) c += p.wg_tot[g];
  unsigned long long total, off = lba_block_scan64<16>(c, s_w, total);
  for (int g = i0; g < i1; ++g) {
    const unsigned long long t = p.wg_tot[g];
    p.wg_tot[g] = off;
    off += t;
  }
  if (tid == 0) {
    const int t_lm = (int)(total & ((1ull << LBA_PK_KF) - 1)), t_kf = (int)((total >> LBA_PK_KF) & ((1ull << (LBA_PK_SL - LBA_PK_KF)) - 1));
    const int t_sl = (int)(total >> LBA_PK_SL);
    p.dyn[0] = t_lm;
    p.dyn[1] = w.opk * t_kf;
    p.dyn[2] = t_sl;
    p.obs_ptr[t_lm] = w.opk * t_kf;
    p.slot_ptr[t_lm] = t_sl;
  }
}

// one lane per id of the interval: the landmark's row of the problem. Points: warpToRef + scalingPoint in double
// (sparse_ba_parameters.h:377-398); observations in window order, left then right (keyframes.cpp:200-215)
struct LbaRef {
  double Tjw_ref[16], Twj_ref[16], inv_scale, pose_scale;
};
__global__ void lba_fill_kernel(LbaWin w, LmTab tab, LbaProb p, LbaRef r) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= w.W) return;
  const int m = p.mask_w[idx];
  if (!m) return;
  p.mask_w[idx] = 0;  // (last reader: the scratch is all zero again when the next solve scatters into it)
  const unsigned long long pk = p.pre[idx] + p.wg_tot[idx / LBA_QWG];
  const int i = (int)(pk & ((1ull << LBA_PK_KF) - 1)), kf0 = (int)((pk >> LBA_PK_KF) & ((1ull << (LBA_PK_SL - LBA_PK_KF)) - 1));
  const int s0 = (int)(pk >> LBA_PK_SL);
  const int32_t id = w.base + idx;
  const int t = id & tab.mask;
  p.used_id[i] = id;
  p.lm_mask[i] = m;
  const double Xd[3] = {(double)tab.X[3 * (size_t)t], (double)tab.X[3 * (size_t)t + 1], (double)tab.X[3 * (size_t)t + 2]};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double xr = (r.Tjw_ref[k * 4 + 0] * Xd[0] + (r.Tjw_ref[k * 4 + 1] * Xd[1] + r.Tjw_ref[k * 4 + 2] * Xd[2])) + r.Tjw_ref[k * 4 + 3];
    p.X[3 * (size_t)i + k] = xr * r.inv_scale;
  }
  p.obs_ptr[i] = w.opk * kf0;
  p.slot_ptr[i] = s0;
  int rr = 0, rs = 0;
  for (int mm = m; mm; mm &= mm - 1) {
    const int j = __ffs(mm) - 1;
    const int q = p.q_w[(size_t)idx * w.kw + j];
    const int o = w.opk * (kf0 + rr);
    p.obs_frame[o] = j;
    p.obs_right[o] = 0;
    p.px[2 * (size_t)o] = (double)w.pl[j][2 * q];
    p.px[2 * (size_t)o + 1] = (double)w.pl[j][2 * q + 1];
    if (w.opk == 2) {
      p.obs_frame[o + 1] = j;
      p.obs_right[o + 1] = 1;
      p.px[2 * (size_t)o + 2] = (double)w.pr[j][2 * q];
      p.px[2 * (size_t)o + 3] = (double)w.pr[j][2 * q + 1];
    }
    if (w.opt[j] >= 0) {  // slot: left observation in an optimised keyframe; its B block is the LAST observation's of the
      const int s = s0 + rs;  // landmark in that keyframe (:315 / :410 assign): the right one of a stereo keyframe
      p.slot_j[s] = w.opt[j];
      p.slot_bobs[s] = o + w.opk - 1;
      p.slot_lm[s] = i;
      ++rs;
    }
    ++rr;
  }
}

// the solver's gather lists, one workgroup each: workgroups 0 .. No - 1 the observation and slot lists of an optimised
// pose, the others the landmark pairs of a block (a, b), a <= b — compactions of the landmark sequence, in landmark
// order like the host builder's (sba.hip)
__global__ __launch_bounds__(1024) void lba_lists_kernel(LbaWin w, LbaProb p) {
  __shared__ int s_w[16];
  const int tid = threadIdx.x, M = p.dyn[0], No = w.No;
  const int chunk = (M + 1023) / 1024, i0 = tid * chunk, i1 = min(i0 + chunk, M);
  int u = blockIdx.x;
  if (u < No) {
    const int f = w.optf[u], bit = 1 << f;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += (p.lm_mask[i] & bit) ? 1 : 0;
    int total, off = lba_block_scan(cnt, s_w, total);
    const int Bo = u * p.pose_obs_stride, Bs = u * p.pose_slot_stride;
    for (int i = i0; i < i1; ++i) {
      const int m = p.lm_mask[i];
      if (!(m & bit)) continue;
      const int o = p.obs_ptr[i] + w.opk * __popc(m & (bit - 1));
      for (int e = 0; e < w.opk; ++e) {
        p.pose_obs[Bo + w.opk * off + e] = o + e;
        p.pose_lm[Bo + w.opk * off + e] = i;
      }
      p.pose_slot[Bs + off] = p.slot_ptr[i] + __popc(m & w.optmask & (bit - 1));
      ++off;
    }
    if (tid == 0) {
      p.pose_obs_ptr[u] = Bo;
      p.pose_obs_end[u] = Bo + w.opk * total;
      p.pose_slot_ptr[u] = Bs;
      p.pose_slot_end[u] = Bs + total;
    }
    return;
  }
  u -= No;
  int a = 0;
  while (u >= No - a) {
    u -= No - a;
    ++a;
  }
  const int b = a + u, jk = a * No + b;
  const int ba = 1 << w.optf[a], bb = 1 << w.optf[b], both = ba | bb;
  int cnt = 0;
  for (int i = i0; i < i1; ++i) cnt += ((p.lm_mask[i] & both) == both) ? 1 : 0;
  int total, off = lba_block_scan(cnt, s_w, total);
  const int Bp = jk * p.pair_stride;
  for (int i = i0; i < i1; ++i) {
    const int m = p.lm_mask[i];
    if ((m & both) != both) continue;
    const int sp = p.slot_ptr[i];
    p.pair_a[Bp + off] = sp + __popc(m & w.optmask & (ba - 1));
    p.pair_b[Bp + off] = sp + __popc(m & w.optmask & (bb - 1));
    ++off;
  }
  if (tid == 0) {
    p.pair_ptr[jk] = Bp;
    p.pair_end[jk] = Bp + total;
  }
}

// points back (sparse_bundle_adjustment.cpp:690-722): recoverOriginalScalePoint, warpToWorld, set3DPoint, setDead
__global__ void lba_writeback_kernel(LbaProb p, LmTab tab, LbaRef r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.dyn[0]) return;
  const double xs[3] = {p.X[3 * (size_t)i] * r.pose_scale, p.X[3 * (size_t)i + 1] * r.pose_scale, p.X[3 * (size_t)i + 2] * r.pose_scale};
  float L[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    L[k] = (float)((r.Twj_ref[k * 4 + 0] * xs[0] + (r.Twj_ref[k * 4 + 1] * xs[1] + r.Twj_ref[k * 4 + 2] * xs[2])) + r.Twj_ref[k * 4 + 3]);
  const int t = p.used_id[i] & tab.mask;
  tab.X[3 * (size_t)t] = L[0];
  tab.X[3 * (size_t)t + 1] = L[1];
  tab.X[3 * (size_t)t + 2] = L[2];
  uint8_t st = (uint8_t)(tab.S[t] | 1);
  const float nrm = sqrtf(L[0] * L[0] + (L[1] * L[1] + L[2] * L[2]));
  if (!(nrm <= 3000))
    st |= 2;  // setDead
  else
    st |= 4;  // setBundled (read by the mono driver only: isBundled() decides its priors and its pose-only BA set)
  tab.S[t] = st;
}
// what the BA did to the landmarks the next frame tracks
__global__ void lba_refresh_kernel(SvoTrackSet ts, int n, LmTab tab, int mono) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int t = ts.ids[k] & tab.mask;
  ts.Xw[3 * k] = tab.X[3 * (size_t)t];
  ts.Xw[3 * k + 1] = tab.X[3 * (size_t)t + 1];
  ts.Xw[3 * k + 2] = tab.X[3 * (size_t)t + 2];
  uint8_t fl = ts.flags[k];
  const uint8_t st = tab.S[t];
  if (st & 1) fl |= VO_LM_TRIANGULATED;
  if (st & 2) fl |= VO_LM_DROPPED;
  if (mono && (st & 4)) fl |= VO_LM_BUNDLED;
  ts.flags[k] = fl;
}

// stats_keyframe[j].mappoints: the CURRENT 3-D points of a keyframe's related landmarks (a slot taken over by a later id
// — 2^24 landmarks on — reads as the origin)
__global__ void lba_mappoints_kernel(const int32_t *ids, int n, LmTab tab, float *out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int32_t id = ids[k];
  const int t = id & tab.mask;
  const bool ok = tab.tag[t] == id;
  out[3 * k] = ok ? tab.X[3 * (size_t)t] : 0.0f;
  out[3 * k + 1] = ok ? tab.X[3 * (size_t)t + 1] : 0.0f;
  out[3 * k + 2] = ok ? tab.X[3 * (size_t)t + 2] : 0.0f;
}

// what the host needs of a solve (poses | errors | flags | counts, 2.3 KB) goes straight into pinned host memory, the solve's
// sequence number last: the host polls that word instead of sleeping in hipStreamSynchronize (whose wake-up costs tens to
// hundreds of microseconds after a wait of this length)
#define LBA_SEQ_WORD 1000  // of the 4096-byte pinned block
__global__ __launch_bounds__(256) void lba_result_kernel(const uint32_t *src, uint32_t *host, int words, uint32_t seq) {
  for (int k = threadIdx.x; k < words; k += 256) host[k] = src[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&host[LBA_SEQ_WORD], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the table doubles: every held id moves to id mod (new size) — a multiple of the old size, so no two collide
__global__ void lba_tab_rehash_kernel(LmTab o, LmTab n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > o.mask) return;
  const int32_t id = o.tag[t];
  if (id == -1) return;
  const int u = id & n.mask;
  n.X[3 * (size_t)u] = o.X[3 * (size_t)t];
  n.X[3 * (size_t)u + 1] = o.X[3 * (size_t)t + 1];
  n.X[3 * (size_t)u + 2] = o.X[3 * (size_t)t + 2];
  n.S[u] = o.S[t];
  n.tag[u] = id;
}

// ---- host side --------------------------------------------------------------------------------------------------------
#define LBA_POOL_CHUNK ((size_t)1 << 20)  // ids per chunk of the all-keyframes pool (4 MB: ~500 keyframes of 2000 landmarks)
namespace {
struct Arena {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};
// Where everything lies inside the solver's arena: fixed at construction from the window's capacity (kf_window keyframes of
// at most `cap` related landmarks each), so that a keyframe never allocates and the kernels' arguments repeat from solve to
// solve. The counts of a real problem are far below these bounds (landmarks <= keyframe entries).
struct LbaLayout {
  size_t res_bytes, oT, oAvg, oFl, oDyn, oOpt, oOfr, oX, oOp, oOf, oOr, oPx, oSp, oSj, oSb, oSl, oPop, oPo, oPl, oPsp, oPs, oPp,
      oPa, oPb, oUid, oLm, in_end, total;
  size_t E_max, Eopt_max;
  int No_max, maxn, n_err_max;
};
}  // namespace

struct vo_svo_lba {
  LmTab tab = {};
  int tab_bits = 0;
  long long id_lo = 0;  // smallest landmark id this stream has put into the table (valid once a keyframe exists)
  bool id_seen = false;
  int32_t *kf_ids[LBA_KW] = {};
  float *kf_pl[LBA_KW] = {}, *kf_pr[LBA_KW] = {};
  int *mask_w = nullptr, *q_w = nullptr;  // window scratch, for span_cap ids; mask_w is all zero between two solves
  unsigned long long *pre = nullptr;
  size_t span_cap = 0;
  uint8_t *arena = nullptr;
  LbaLayout lay = {};
  uint8_t *h_res = nullptr;  // pinned: poses, errors, flags, counts of a solve; word LBA_SEQ_WORD = the solve's sequence number
  uint32_t seq = 0;
  std::vector<int32_t *> pool;  // all keyframes' id lists, chunk by chunk
  size_t pool_used = 0;         // ids used in the last chunk
  float *d_map = nullptr;       // staging of one keyframe's map points
  float *d_map_all = nullptr;   // staging of all keyframes' map points (vo_svo_get_keyframes)
  size_t map_all_cap = 0;       // points
  std::vector<size_t> pool_fill;  // ids used in every chunk
  // VO_SVO_TRACE: host build + enqueue, device, write-back (us), per StereoVO
  double tt[4] = {0, 0, 0, 0};
  int n_calls = 0;
};

void vo_svo_lba_free(vo_svo *s) {
  vo_svo_lba *L = s->lba;
  if (!L) return;
  void *dev[] = {L->tab.X, L->tab.S, L->tab.tag, L->mask_w, L->q_w, L->pre, L->arena};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  for (int k = 0; k < LBA_KW; ++k) {
    if (L->kf_ids[k]) (void)hipFree(L->kf_ids[k]);
    if (L->kf_pl[k]) (void)hipFree(L->kf_pl[k]);
    if (L->kf_pr[k]) (void)hipFree(L->kf_pr[k]);
  }
  if (L->h_res) (void)hipHostFree(L->h_res);
  for (int32_t *p : L->pool)
    if (p) (void)hipFree(p);
  if (L->d_map) (void)hipFree(L->d_map);
  if (L->d_map_all) (void)hipFree(L->d_map_all);
  delete L;
  s->lba = nullptr;
}

static int lba_tab_alloc(vo_ctx *c, LmTab *t, int bits) {
  const size_t slots = (size_t)1 << bits;
  memset(t, 0, sizeof(*t));
  t->mask = (int)(slots - 1);
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->X, sizeof(float) * 3 * slots));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->S, slots));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&t->tag, sizeof(int32_t) * slots));
  VO_CHECK_HIP(c, hipMemsetAsync(t->tag, 0xFF, sizeof(int32_t) * slots, c->stream));
  VO_CHECK_HIP(c, hipMemsetAsync(t->S, 0, slots, c->stream));
  return VO_OK;
}
static void lba_tab_release(LmTab *t) {
  void *dev[] = {t->X, t->S, t->tag};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  memset(t, 0, sizeof(*t));
}
// the table follows the ids handed out so far (rarely: it doubles; never inside a steady stretch of less than 2^18 ids)
static int lba_tab_reserve(vo_svo *s, long long id_hi) {
  vo_ctx *c = s->c;
  vo_svo_lba *L = s->lba;
  int bits = L->tab_bits;
  while (bits < LBA_TAB_BITS && id_hi - L->id_lo > ((long long)1 << bits)) ++bits;
  if (bits == L->tab_bits) return VO_OK;
  LmTab nt;
  const int rc = lba_tab_alloc(c, &nt, bits);
  if (rc < 0) {
    lba_tab_release(&nt);
    return rc;
  }
  hipLaunchKernelGGL(lba_tab_rehash_kernel, dim3((unsigned)((L->tab.mask + 256) / 256)), dim3(256), 0, c->stream, L->tab, nt);
  VO_CHECK_HIP(c, hipGetLastError());
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  lba_tab_release(&L->tab);
  L->tab = nt;
  L->tab_bits = bits;
  return VO_OK;
}
static int lba_scratch_reserve(vo_svo *s, size_t ids) {
  vo_ctx *c = s->c;
  vo_svo_lba *L = s->lba;
  if (ids <= L->span_cap) return VO_OK;
  size_t want = L->span_cap ? L->span_cap : (size_t)LBA_SPAN_FIRST;
  while (want < ids) want *= 2;
  if (L->span_cap) VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  void *old[] = {L->mask_w, L->q_w, L->pre};
  for (void *q : old)
    if (q) (void)hipFree(q);
  L->mask_w = L->q_w = nullptr;
  L->pre = nullptr;
  L->span_cap = 0;
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->mask_w, sizeof(int) * want));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->q_w, sizeof(int) * want * (size_t)s->prm.kf_window));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->pre, sizeof(unsigned long long) * (want + want / LBA_QWG + 1)));
  VO_CHECK_HIP(c, hipMemsetAsync(L->mask_w, 0, sizeof(int) * want, c->stream));  // (lba_fill_kernel re-zeroes what a solve set)
  L->span_cap = want;
  return VO_OK;
}
static int lba_pool_reserve(vo_svo *s, size_t n) {
  vo_svo_lba *L = s->lba;
  if (!L->pool.empty() && L->pool_used + n <= LBA_POOL_CHUNK) return VO_OK;
  int32_t *chunk = nullptr;
  VO_CHECK_HIP(s->c, vo_dev_malloc(s->c, (void **)&chunk, sizeof(int32_t) * LBA_POOL_CHUNK));
  L->pool.push_back(chunk);
  L->pool_fill.push_back(0);
  L->pool_used = 0;
  return VO_OK;
}

static void lba_make_layout(LbaLayout *l, int kf_window, int cap) {
  Arena ar;
  const int No = std::max(kf_window - 2, 1);
  const size_t E = (size_t)kf_window * cap, E_opt = (size_t)No * cap, M_ub = E, nobs_ub = 2 * E, ns_ub = E_opt, maxn = (size_t)cap;
  l->E_max = E;
  l->Eopt_max = E_opt;
  l->No_max = No;
  l->maxn = cap;
  l->n_err_max = (int)((M_ub + 64 / SBA_LQ - 1) / (64 / SBA_LQ));
  // (what the host reads back is one piece: poses | errors | flags | counts)
  l->res_bytes = sizeof(double) * (16 * LBA_KW + 16) + sizeof(int) * (16 + 4);
  l->oT = ar.take(l->res_bytes);
  l->oAvg = l->oT + sizeof(double) * 16 * LBA_KW;
  l->oFl = l->oAvg + sizeof(double) * 16;
  l->oDyn = l->oFl + sizeof(int) * 16;
  l->oOpt = ar.take(sizeof(int) * LBA_KW);
  l->oOfr = ar.take(sizeof(int) * (No + 1));
  l->oX = ar.take(sizeof(double) * 3 * (M_ub + 1));
  l->oOp = ar.take(sizeof(int) * (M_ub + 1));
  l->oOf = ar.take(sizeof(int) * (nobs_ub + 2));
  l->oOr = ar.take(nobs_ub + 2);
  l->oPx = ar.take(sizeof(double) * 2 * (nobs_ub + 2));
  l->oSp = ar.take(sizeof(int) * (M_ub + 1));
  l->oSj = ar.take(sizeof(int) * (ns_ub + 1));
  l->oSb = ar.take(sizeof(int) * (ns_ub + 1));
  l->oSl = ar.take(sizeof(int) * (ns_ub + 1));
  l->oPop = ar.take(sizeof(int) * 2 * (No + 1));
  l->oPo = ar.take(sizeof(int) * 2 * maxn * No);
  l->oPl = ar.take(sizeof(int) * 2 * maxn * No);
  l->oPsp = ar.take(sizeof(int) * 2 * (No + 1));
  l->oPs = ar.take(sizeof(int) * maxn * No);
  l->oPp = ar.take(sizeof(int) * 2 * ((size_t)No * No + 1));
  l->oPa = ar.take(sizeof(int) * maxn * No * No);
  l->oPb = ar.take(sizeof(int) * maxn * No * No);
  l->oUid = ar.take(sizeof(int32_t) * (M_ub + 1));
  l->oLm = ar.take(sizeof(int) * (M_ub + 1));
  l->in_end = ar.off;
  SbaDev d;
  l->total = vo_sba_place_work(&d, nullptr, l->in_end, M_ub, ns_ub, No, 10, l->n_err_max);
}

// Build + solve + write-back of one window, enqueued on the main stream: the builder's five launches, the solver's
// iterations, the two write-back launches. Returns where the result piece (poses | errors | flags | counts) lies.
// seq != 0: the solver's last kernel sends the result piece to L->h_res itself, sequence word last, when it can
// (*delivered; otherwise the caller launches lba_result_kernel behind everything).
static int lba_enqueue(vo_svo *s, const LbaWin &w, int maxn, size_t M_ub, const LbaHead &head, const LbaRef &ref, const SvoTrackSet &t,
                       int n, int max_iter, const double **res, uint32_t seq = 0, bool *delivered = nullptr) {
  vo_ctx *c = s->c;
  vo_svo_lba *L = s->lba;
  const LbaLayout &lay = L->lay;
  hipStream_t st = c->stream;
  const int nk = w.nk, No = w.No;
  const double inv_scale = ref.inv_scale;
  const int n_err = std::max(1, (int)((M_ub + 64 / SBA_LQ - 1) / (64 / SBA_LQ)));
  // ---- the solver's arena: fixed offsets ----
  SbaDev d;
  memset(&d, 0, sizeof(d));
  uint8_t *base = L->arena;
  vo_sba_place_work(&d, base, lay.in_end, lay.E_max, lay.Eopt_max, lay.No_max, max_iter, lay.n_err_max);
  d.n_err = n_err;
  LbaProb p;
  memset(&p, 0, sizeof(p));
  p.T = (double *)(base + lay.oT);
  p.opt_index = (int *)(base + lay.oOpt);
  p.opt_frame = (int *)(base + lay.oOfr);
  p.X = (double *)(base + lay.oX);
  p.obs_ptr = (int *)(base + lay.oOp);
  p.obs_frame = (int *)(base + lay.oOf);
  p.obs_right = base + lay.oOr;
  p.px = (double *)(base + lay.oPx);
  p.slot_ptr = (int *)(base + lay.oSp);
  p.slot_j = (int *)(base + lay.oSj);
  p.slot_bobs = (int *)(base + lay.oSb);
  p.slot_lm = (int *)(base + lay.oSl);
  p.pose_obs_ptr = (int *)(base + lay.oPop);
  p.pose_obs_end = p.pose_obs_ptr + (lay.No_max + 1);
  p.pose_obs = (int *)(base + lay.oPo);
  p.pose_lm = (int *)(base + lay.oPl);
  p.pose_slot_ptr = (int *)(base + lay.oPsp);
  p.pose_slot_end = p.pose_slot_ptr + (lay.No_max + 1);
  p.pose_slot = (int *)(base + lay.oPs);
  p.pair_ptr = (int *)(base + lay.oPp);
  p.pair_end = p.pair_ptr + ((size_t)lay.No_max * lay.No_max + 1);
  p.pair_a = (int *)(base + lay.oPa);
  p.pair_b = (int *)(base + lay.oPb);
  p.flags = (int *)(base + lay.oFl);
  p.avg_err = (double *)(base + lay.oAvg);
  p.dyn = (int *)(base + lay.oDyn);
  p.used_id = (int32_t *)(base + lay.oUid);
  p.lm_mask = (int *)(base + lay.oLm);
  p.mask_w = L->mask_w;
  p.q_w = L->q_w;
  p.pre = L->pre;
  p.wg_tot = L->pre + L->span_cap;
  p.pose_obs_stride = 2 * lay.maxn;
  p.pose_slot_stride = lay.maxn;
  p.pair_stride = lay.maxn;
  p.max_iter = max_iter;
  // ---- build + solve, all on the main stream (mask_w is zero on entry: lba_fill_kernel clears what it has read) ----
  hipLaunchKernelGGL(lba_scatter_kernel, dim3((maxn + 255) / 256, nk), dim3(256), 0, st, w, L->mask_w, L->q_w);
  const int n_wg = (w.W + LBA_QWG - 1) / LBA_QWG;
  hipLaunchKernelGGL(lba_qualify_kernel, dim3(n_wg), dim3(LBA_QWG), 0, st, w, L->tab, p);
  hipLaunchKernelGGL(lba_scan_kernel, dim3(1), dim3(1024), 0, st, w, p, head, n_wg);
  hipLaunchKernelGGL(lba_fill_kernel, dim3((w.W + 255) / 256), dim3(256), 0, st, w, L->tab, p, ref);
  hipLaunchKernelGGL(lba_lists_kernel, dim3(No + No * (No + 1) / 2), dim3(1024), 0, st, w, p);
  VO_CHECK_HIP(c, hipGetLastError());
  d.n_frames = nk;
  d.n_opt = No;
  d.stereo = s->mono ? 0 : 1;
  d.max_iter = max_iter;
  d.dyn = p.dyn;
  for (int k = 0; k < 4; ++k) {
    d.Kl[k] = (double)s->prm.frame.Kl[k];
    d.Kr[k] = (double)s->prm.frame.Kr[k];
  }
  {  // geometry::inverseSE3(T_lr) of the scaled stereo pose (sba.hip: vo_sba_solve)
    double T_lr[16];
    to_d(s->prm.frame.T_lr, T_lr);
    for (int r = 0; r < 3; ++r) T_lr[r * 4 + 3] *= inv_scale;  // scalingPose(T_stereo_)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) d.R_rl[i * 3 + j] = T_lr[j * 4 + i];
    for (int i = 0; i < 3; ++i)
      d.t_rl[i] = (-d.R_rl[i * 3]) * T_lr[3] + ((-d.R_rl[i * 3 + 1]) * T_lr[7] + (-d.R_rl[i * 3 + 2]) * T_lr[11]);
  }
  d.thres_huber = 0.5;
  d.lambda = 0.00001;
  d.T = p.T;
  d.opt_index = p.opt_index;
  d.opt_frame = p.opt_frame;
  d.X = p.X;
  d.obs_ptr = p.obs_ptr;
  d.obs_frame = p.obs_frame;
  d.obs_right = p.obs_right;
  d.obs_px = p.px;
  d.slot_ptr = p.slot_ptr;
  d.slot_j = p.slot_j;
  d.slot_bobs = p.slot_bobs;
  d.slot_lm = p.slot_lm;
  d.pose_obs_ptr = p.pose_obs_ptr;
  d.pose_obs_end = p.pose_obs_end;
  d.pose_obs = p.pose_obs;
  d.pose_lm = p.pose_lm;
  d.pose_slot_ptr = p.pose_slot_ptr;
  d.pose_slot_end = p.pose_slot_end;
  d.pose_slot = p.pose_slot;
  d.pair_ptr = p.pair_ptr;
  d.pair_end = p.pair_end;
  d.pair_a = p.pair_a;
  d.pair_b = p.pair_b;
  d.avg_err = p.avg_err;
  d.flags = p.flags;
  if (seq) {
    d.res_host = (uint32_t *)L->h_res;
    d.res_src = (const uint32_t *)p.T;
    d.res_words = (int)(lay.res_bytes / 4);
    d.res_seq_word = LBA_SEQ_WORD;
    d.res_seq = seq;
    if (delivered) *delivered = vo_sba_delivers_result(c, d);
    if (!vo_sba_delivers_result(c, d)) d.res_host = nullptr;
  }
  vo_prof_begin(c, VO_K_AUX);
  {
    const int rc = vo_sba_enqueue_iterations(c, d, max_iter);
    if (rc < 0) return rc;
  }
  vo_prof_end(c);
  // ---- points back into the table, and into the track set the next frame starts from: enqueued right behind the
  // iterations, before the host looks at anything (a solve that ends in NaN or a "large update" ends the run as in the
  // reference; what it left in the table is then nobody's input; with no landmark in the problem both kernels do nothing) ----
  hipLaunchKernelGGL(lba_writeback_kernel, dim3((unsigned)((M_ub + 255) / 256)), dim3(256), 0, st, p, L->tab, ref);
  hipLaunchKernelGGL(lba_refresh_kernel, dim3((std::max(n, 1) + 255) / 256), dim3(256), 0, st, t, n, L->tab, s->mono);
  VO_CHECK_HIP(c, hipGetLastError());
  *res = p.T;
  return VO_OK;
}

// The first launch of a kernel costs the runtime 50-100 us (function lookup, kernel object, argument pool), and a local BA
// is ~12 different kernels — one register solve per window size among them: a stream's first solves used to cost 1-2 ms
// extra each. Every one of them is launched here once per window size on an EMPTY window (no keyframe entries, an id
// interval of one: every launch runs its code path and touches nothing but the zeroed arena).
static int lba_warm_up(vo_svo *s) {
  vo_ctx *c = s->c;
  vo_svo_lba *L = s->lba;
  VO_CHECK_HIP(c, hipMemsetAsync(L->arena, 0, L->lay.total, c->stream));
  LbaHead head;
  LbaRef ref;
  memset(&head, 0, sizeof(head));
  memset(&ref, 0, sizeof(ref));
  for (int k = 0; k < 4; ++k) ref.Tjw_ref[5 * k] = ref.Twj_ref[5 * k] = 1.0;
  ref.inv_scale = 0.1;
  ref.pose_scale = 10.0;
  for (int nk = 3; nk <= s->prm.kf_window; ++nk) {
    LbaWin w;
    memset(&w, 0, sizeof(w));
    w.nk = nk;
    w.No = nk - 2;
    w.kw = s->prm.kf_window;
    w.opk = s->mono ? 1 : 2;
    w.W = 1;
    for (int j = 0; j < nk; ++j) {
      w.ids[j] = L->kf_ids[0];
      w.pl[j] = L->kf_pl[0];
      w.pr[j] = L->kf_pr[0];
      w.opt[j] = j < 2 ? -1 : j - 2;
      if (j >= 2) {
        w.optf[j - 2] = j;
        w.optmask |= 1 << j;
      }
      for (int k = 0; k < 4; ++k) head.T_jw[16 * j + 5 * k] = 1.0;
    }
    const double *res = nullptr;
    const int rc = lba_enqueue(s, w, 1, 1, head, ref, s->ts[0], 0, 1, &res);
    if (rc < 0) return rc;
  }
  hipLaunchKernelGGL(lba_keyframe_kernel, dim3(1), dim3(256), 0, c->stream, s->ts[0], 0, L->tab, L->kf_ids[0], L->kf_pl[0], L->kf_pr[0],
                     L->pool.back());
  L->seq = 1;
  hipLaunchKernelGGL(lba_result_kernel, dim3(1), dim3(256), 0, c->stream, (const uint32_t *)L->arena, (uint32_t *)L->h_res, 16, L->seq);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// Everything the keyframes of a stream will need, allocated when the StereoVO is made (vo_svo_create): a keyframe — the
// first one, a growing window, the first solve — allocates nothing. (Grown later, each by doubling: the table past
// 2^18 landmark ids, the window scratch past its first interval, the keyframe pool every 2^20 related landmarks.)
int vo_svo_lba_init(vo_svo *s) {
  vo_ctx *c = s->c;
  if (s->prm.kf_window > LBA_KW) VO_FAIL(c, VO_ERR_CAPACITY, "keyframe window of %d (at most %d)", s->prm.kf_window, LBA_KW);
  vo_svo_lba *L = new vo_svo_lba();
  s->lba = L;
  L->tab_bits = LBA_TAB_FIRST_BITS;
  int rc = lba_tab_alloc(c, &L->tab, L->tab_bits);
  if (rc < 0) return rc;
  for (int k = 0; k < s->prm.kf_window; ++k) {
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->kf_ids[k], sizeof(int32_t) * (size_t)s->cap));
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->kf_pl[k], sizeof(float) * 2 * (size_t)s->cap));
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->kf_pr[k], sizeof(float) * 2 * (size_t)s->cap));
  }
  VO_CHECK_HIP(c, vo_host_malloc(c, (void **)&L->h_res, 4096, hipHostMallocDefault));
  memset(L->h_res, 0, 4096);
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->d_map, sizeof(float) * 3 * (size_t)s->cap));
  rc = lba_pool_reserve(s, (size_t)s->cap);
  if (rc < 0) return rc;
  if (s->prm.local_ba && s->prm.kf_window >= 3) {
    lba_make_layout(&L->lay, s->prm.kf_window, s->cap);
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->arena, L->lay.total));
    size_t first = (size_t)LBA_SPAN_FIRST;
    while (first < 2 * (size_t)s->prm.kf_window * (size_t)s->cap) first *= 2;
    rc = lba_scratch_reserve(s, first);
    if (rc < 0) return rc;
    rc = lba_warm_up(s);
    if (rc < 0) return rc;
  }
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

int vo_svo_lba_update_points(vo_svo *s, const SvoTrackSet &ts, int n) {
  vo_ctx *c = s->c;
  if (!s->lba || n <= 0) return VO_OK;
  hipLaunchKernelGGL(lba_table_update_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, ts, n, s->lba->tab);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

size_t vo_svo_lba_bytes(const vo_svo *s) {  // device memory held for the keyframes of this stream (test / report hook)
  const vo_svo_lba *L = s->lba;
  if (!L) return 0;
  size_t b = ((size_t)L->tab.mask + 1) * 17 + (size_t)s->prm.kf_window * (size_t)s->cap * 20 + sizeof(float) * 3 * (size_t)s->cap;
  b += L->pool.size() * LBA_POOL_CHUNK * sizeof(int32_t) + L->map_all_cap * 12;
  if (L->arena) b += L->lay.total;
  b += L->span_cap * (12 + 4 * (size_t)s->prm.kf_window) + (L->span_cap / LBA_QWG + 1) * 8;
  return b;
}

int vo_svo_local_ba(vo_svo *s, vo_svo_frame_info *info, int id_min) {
  vo_ctx *c = s->c;
  const int n = s->n;
  SvoTrackSet &t = s->ts[s->cur];
  static const bool trace = getenv("VO_SVO_TRACE") != nullptr;
  const double t_0 = trace ? lba_now() : 0.0;
  hipStream_t st = c->stream;
  vo_svo_lba *L = s->lba;
  if (!L) VO_FAIL(c, VO_ERR_INVALID, "StereoVO: the keyframe storage was not initialised");
  std::vector<SvoKeyframe> &win = s->keyframes;
  const int nk = (int)win.size();
  // ---- the new keyframe's related landmarks (behind the reconstruction kernel on the main stream) ----
  SvoKeyframe &kf = win.back();
  {
    unsigned used = 0;
    for (int j = 0; j + 1 < nk; ++j) used |= 1u << win[j].ring;
    int r = 0;
    while (used & (1u << r)) ++r;
    kf.ring = r;
    kf.n = n;
    kf.id_min = id_min;
    kf.global = (int)s->kf_all.size();
  }
  if (n > 0) {  // the table holds every id from the stream's first landmark on
    if (!L->id_seen || (long long)id_min < L->id_lo) L->id_lo = id_min;
    L->id_seen = true;
    const int rc = lba_tab_reserve(s, (long long)c->next_landmark_id);
    if (rc < 0) return rc;
  }
  {  // all_stkeyframes_: the keyframe's related landmarks go into the pool that only grows
    const int rc = lba_pool_reserve(s, (size_t)n);
    if (rc < 0) return rc;
    vo_svo::SvoKfAll rec;
    memcpy(rec.T_wc, kf.T_wc, sizeof(rec.T_wc));
    rec.n = n;
    rec.d_ids = L->pool.back() + L->pool_used;
    L->pool_used += (size_t)n;
    L->pool_fill.back() = L->pool_used;
    s->kf_all.push_back(rec);
  }
  if (n > 0) {
    hipLaunchKernelGGL(lba_keyframe_kernel, dim3((n + 255) / 256), dim3(256), 0, st, t, n, L->tab, L->kf_ids[kf.ring], L->kf_pl[kf.ring],
                       L->kf_pr[kf.ring], const_cast<int32_t *>(s->kf_all.back().d_ids));
    VO_CHECK_HIP(c, hipGetLastError());
  }
  if (!s->prm.local_ba) return VO_OK;
  if (nk < 3) return VO_OK;  // NUM_MINIMUM_REQUIRED_KEYFRAMES (motion_estimator.cpp:1245-1253)
  const double POSE_SCALE = 10.0, inv_scale = 1.0 / POSE_SCALE;
  const LbaLayout &lay = L->lay;
  // ---- the window ----
  LbaWin w;
  memset(&w, 0, sizeof(w));
  w.nk = nk;
  w.No = nk - 2;  // NUM_FIX_KEYFRAMES_IN_WINDOW
  w.kw = s->prm.kf_window;
  w.opk = s->mono ? 1 : 2;
  w.base = win.front().id_min;
  const long long span = (long long)c->next_landmark_id - (long long)w.base;
  if (span > LBA_SPAN_MAX) VO_FAIL(c, VO_ERR_CAPACITY, "local BA: the window spans %lld landmark ids (at most %d)", span, LBA_SPAN_MAX);
  w.W = (int)span;
  {  // the window scratch follows the interval (rarely: it doubles)
    const int rc = lba_scratch_reserve(s, (size_t)std::max(w.W, 0));
    if (rc < 0) return rc;
  }
  size_t E = 0, E_opt = 0;
  int maxn = 1;
  for (int j = 0; j < nk; ++j) {
    w.n[j] = win[j].n;
    w.ids[j] = L->kf_ids[win[j].ring];
    w.pl[j] = L->kf_pl[win[j].ring];
    w.pr[j] = L->kf_pr[win[j].ring];
    w.opt[j] = j < 2 ? -1 : j - 2;
    if (j >= 2) {
      w.optf[j - 2] = j;
      w.optmask |= 1 << j;
      E_opt += (size_t)win[j].n;
    }
    E += (size_t)win[j].n;
    maxn = std::max(maxn, win[j].n);
  }
  if (w.W <= 0 || E == 0) return VO_OK;
  if (E >= ((size_t)1 << LBA_PK_KF))  // (landmarks <= keyframe entries; pairs and slots <= entries: 20 / 22 / 22 bits)
    VO_FAIL(c, VO_ERR_CAPACITY, "local BA: %zu keyframe entries exceed the packed counters", E);
  if (E > lay.E_max || E_opt > lay.Eopt_max || maxn > lay.maxn || nk - 2 > lay.No_max)  // (cannot happen: n <= cap, nk <= kf_window)
    VO_FAIL(c, VO_ERR_CAPACITY, "local BA: the window (%zu entries) exceeds the arena laid out at construction", E);
  const int max_iter = 10;
  const size_t M_ub = std::min((size_t)w.W, E);
  // ---- poses: reference frame = first keyframe of the window (sparse_ba_parameters.h:340-375) ----
  LbaRef ref;
  LbaHead head;
  memset(&head, 0, sizeof(head));
  to_d(win.front().T_wc, ref.Twj_ref);
  inv_se3d(ref.Twj_ref, ref.Tjw_ref);
  ref.inv_scale = inv_scale;
  ref.pose_scale = POSE_SCALE;
  for (int j = 0; j < nk; ++j) {
    float Tjw_f[16];
    double Tjw[16];
    svo_inv_se3(win[j].T_wc, Tjw_f);  // getPoseInv()
    to_d(Tjw_f, Tjw);
    mul44d(Tjw, ref.Twj_ref, &head.T_jw[16 * j]);  // changeInvPoseWorldToRef
    for (int r = 0; r < 3; ++r) head.T_jw[16 * j + r * 4 + 3] *= inv_scale;  // scalingPose
  }
  const double *res_dev = nullptr;
  bool delivered = false;
  L->seq = L->seq + 1 == 0 ? 1 : L->seq + 1;
  {
    const int rc = lba_enqueue(s, w, maxn, M_ub, head, ref, t, n, max_iter, &res_dev, L->seq, &delivered);
    if (rc < 0) return rc;
  }
  // ---- what the host needs: poses, errors, flags, counts — sent by the last iteration's solve kernel itself (the host works
  // on them, and on the next frame's enqueue, while the final point update and the write-back launches run), or by a launch
  // behind everything when that kernel is not the one in use ----
  double *o_T = (double *)L->h_res, *o_e = o_T + 16 * LBA_KW;
  int *o_f = (int *)(o_e + 16), *o_d = o_f + 16;
  if (!delivered)
    hipLaunchKernelGGL(lba_result_kernel, dim3(1), dim3(256), 0, st, (const uint32_t *)res_dev, (uint32_t *)L->h_res, (int)(lay.res_bytes / 4),
                       L->seq);
  VO_CHECK_HIP(c, hipGetLastError());
  const double t_1 = trace ? lba_now() : 0.0;
  {
    volatile const uint32_t *seqp = (volatile const uint32_t *)L->h_res + LBA_SEQ_WORD;
    bool seen = false;
    timespec p0;
    clock_gettime(CLOCK_MONOTONIC, &p0);
    for (int spin = 0; !seen; ++spin) {
      if (*seqp == L->seq) {
        seen = true;
        break;
      }
      if ((spin & 255) == 255) {  // ~20 ms of polling at most, then the stream decides
        timespec p1;
        clock_gettime(CLOCK_MONOTONIC, &p1);
        if ((p1.tv_sec - p0.tv_sec) * 1e9 + (p1.tv_nsec - p0.tv_nsec) > 2e7) break;
      }
      if (c->dbg[VO_OPT_POLL_YIELD]) sched_yield();
#if defined(__x86_64__)
      else __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (!seen) VO_CHECK_HIP(c, hipStreamSynchronize(st));
  }
  const double t_2 = trace ? lba_now() : 0.0;
  if (o_d[0] <= 0) return VO_OK;  // no landmark qualifies: nothing to adjust
  if (o_f[0] & 1) VO_FAIL(c, VO_ERR_LBA_NAN, "In LBA, pose becomes nan!");
  if (o_f[0] & 2) VO_FAIL(c, VO_ERR_LBA_NAN, "Local BA NAN!");
  if (info) {
    info->lba_ran = 1;
    info->lba_err_first = o_e[0];
    info->lba_err_last = o_e[max_iter - 1];
    info->lba_landmarks = o_d[0];
    info->lba_observations = o_d[1];
  }
  // ---- sparse_bundle_adjustment.cpp:624-688: poses back ----
  for (int j = 0; j < nk; ++j) {
    if (w.opt[j] < 0) continue;
    double T[16], Tjw[16], Twj_orig[16], dT[16];
    for (int k = 0; k < 16; ++k) T[k] = o_T[16 * j + k];
    for (int r = 0; r < 3; ++r) T[r * 4 + 3] *= POSE_SCALE;  // recoverOriginalScalePose
    mul44d(T, ref.Tjw_ref, Tjw);                             // changeInvPoseRefToWorld
    to_d(win[j].T_wc, Twj_orig);
    mul44d(Twj_orig, Tjw, dT);
    const double tn = sqrt(dT[3] * dT[3] + (dT[7] * dT[7] + dT[11] * dT[11]));
    if (tn > 50) VO_FAIL(c, VO_ERR_LBA_NAN, "local BA: large update!");
    float Tf[16];
    for (int k = 0; k < 12; ++k) Tf[k] = (float)Tjw[k];
    Tf[12] = Tf[13] = Tf[14] = 0.0f;
    Tf[15] = 1.0f;
    svo_inv_se3(Tf, win[j].T_wc);  // kf->setPose(inverseSE3_f(Tjw_update_float))
    memcpy(s->kf_all[win[j].global].T_wc, win[j].T_wc, sizeof(win[j].T_wc));
  }
  if (trace) {
    const double t_3 = lba_now();
    L->tt[0] += t_1 - t_0;
    L->tt[1] += t_2 - t_1;
    L->tt[2] += t_3 - t_2;
    if ((++L->n_calls % 10) == 0)
      fprintf(stderr, "[lba] per call (us): host build + enqueue %.0f  device (build + solve) %.0f  write-back %.0f  (M=%d obs=%d slots=%d, span %d)\n",
              L->tt[0] / L->n_calls, L->tt[1] / L->n_calls, L->tt[2] / L->n_calls, o_d[0], o_d[1], o_d[2], w.W);
  }
  return VO_OK;
}

extern "C" int vo_svo_device_bytes(const vo_svo *s, size_t *bytes) {
  if (!s || !bytes) return VO_ERR_INVALID;
  *bytes = vo_svo_lba_bytes(s);
  return VO_OK;
}

extern "C" int vo_svo_keyframe_count(vo_svo *s, int *n_keyframes) {
  if (!s || !n_keyframes) return VO_ERR_INVALID;
  *n_keyframes = (int)s->kf_all.size();
  return VO_OK;
}

extern "C" int vo_svo_get_keyframe(vo_svo *s, int j, float T_wc[16], float *mappoints, int cap, int *n_points) {
  if (!s || j < 0 || j >= (int)s->kf_all.size()) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_svo_result first");
  const vo_svo::SvoKfAll &k = s->kf_all[j];
  if (T_wc) memcpy(T_wc, k.T_wc, sizeof(k.T_wc));
  if (n_points) *n_points = k.n;
  if (!mappoints || k.n == 0) return VO_OK;
  if (k.n > cap) VO_FAIL(c, VO_ERR_CAPACITY, "keyframe %d has %d related landmarks, room for %d", j, k.n, cap);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_svo_lba *L = s->lba;
  hipLaunchKernelGGL(lba_mappoints_kernel, dim3((k.n + 255) / 256), dim3(256), 0, c->stream, k.d_ids, k.n, L->tab, L->d_map);
  VO_CHECK_HIP(c, hipGetLastError());
  VO_CHECK_HIP(c, hipMemcpyAsync(mappoints, L->d_map, sizeof(float) * 3 * (size_t)k.n, hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

// the same for ALL keyframes at once — what the reference does at every keyframe (stereo_vo.cpp:813-821): one gather per
// pool chunk and one copy instead of a launch, a copy and a synchronisation per keyframe
extern "C" int vo_svo_get_keyframes(vo_svo *s, float *T_wc, int32_t *n_points, float *mappoints, size_t cap_points,
                                    size_t *total_points) {
  if (!s) return VO_ERR_INVALID;
  vo_ctx *c = s->c;
  if (s->pending) VO_FAIL(c, VO_ERR_INVALID, "call vo_svo_result first");
  size_t total = 0;
  for (size_t j = 0; j < s->kf_all.size(); ++j) {
    if (T_wc) memcpy(T_wc + 16 * j, s->kf_all[j].T_wc, sizeof(float) * 16);
    if (n_points) n_points[j] = s->kf_all[j].n;
    total += (size_t)s->kf_all[j].n;
  }
  if (total_points) *total_points = total;
  if (!mappoints || total == 0) return VO_OK;
  if (total > cap_points) VO_FAIL(c, VO_ERR_CAPACITY, "%zu map points, room for %zu", total, cap_points);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_svo_lba *L = s->lba;
  if (L->map_all_cap < total) {
    VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    if (L->d_map_all) (void)hipFree(L->d_map_all);
    L->d_map_all = nullptr;
    L->map_all_cap = 0;
    const size_t want = total + (total >> 1) + 4096;
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&L->d_map_all, sizeof(float) * 3 * want));
    L->map_all_cap = want;
  }
  size_t off = 0;  // (keyframes lie in the pool in their order, a keyframe never straddles two chunks)
  for (size_t q = 0; q < L->pool.size(); ++q) {
    const size_t m = L->pool_fill[q];
    if (!m) continue;
    hipLaunchKernelGGL(lba_mappoints_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, L->pool[q], (int)m, L->tab,
                       L->d_map_all + 3 * off);
    off += m;
  }
  VO_CHECK_HIP(c, hipGetLastError());
  VO_CHECK_HIP(c, hipMemcpyAsync(mappoints, L->d_map_all, sizeof(float) * 3 * total, hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}
