// klt_track.hip — pyramidal Lucas-Kanade, one gfx950 wavefront per feature point.
//
// Replaces cv::calcOpticalFlowPyrLK (OpenCV 4 modules/video/src/lkpyramid.cpp:
// SharrDerivInvoker + LKTrackerInvoker) as the reference calls it from
// FeatureTracker::track / trackBidirection / trackBidirectionWithPrior /
// trackWithPrior (core/visual_odometry/feature_tracker.cpp:29,60,69,108,117,186),
// and the validity masks those wrappers compute (:33-34, :74-83, :130-155, :191-197).
//
// Mapping (WIN x WIN window, W = 64 lanes):
//   lane -> (window row, run of RL consecutive columns); RL is the smallest run
//   for which rows x runs-per-row <= 64 (WIN 21: 21 rows x 3 runs of 7 = 63 lanes).
//   A lane keeps its RL template samples (I, Ix, Iy) in registers for the whole
//   level; neighbouring samples of a run share their bilinear neighbours, so one
//   iteration costs 2 x ceil((RL+4)/4) aligned LDS dword reads per lane.
//   All levels of a point run inside one launch (coarse -> fine), no host
//   round trip; both image pyramids are read with 4-byte-aligned, row-contiguous
//   dword loads into LDS tiles:
//     template tile : (WIN+3) rows incl. the Scharr halo; the derivative is
//                     computed on the fly from the u8 tile (never stored in HBM)
//     search tile   : (WIN+1+2M) rows around the current estimate, re-staged only
//                     when the window leaves the tile
//   Reductions: the products are integers, so lane partials are int32 and the
//   wave sum is an exact int64 (two DPP butterflies on 16-bit halves) rounded to
//   float once — order-independent and bit-identical to the oracle.
//
// Roofline: per point-level the kernel reads (WIN+3)^2-ish + (WIN+1+2M)^2-ish
// bytes of L2/HBM-resident pyramid; arithmetic is ~10 int ops per sample per
// iteration. At 1500 points it is latency-bound (dependent LDS->ALU->DPP chain
// per iteration), not HBM-bound; see DESIGN.md.
#include "klt_device.hpp"
#include "vo_kernels.hpp"

struct KltArgs {
  vo_level I[VO_MAX_LEVELS];
  vo_level J[VO_MAX_LEVELS];
  int max_level;  // effective OpenCV maxLevel: levels 0..max_level are used
  const float *pts0;
  float *pts1;             // out (and initial guess when pts1_init is null)
  const float *pts1_init;  // optional: initial nextPts read from here (USE_INITIAL_FLOW)
  int n;
  const int *d_n;
  int flags;
  int max_count;
  double epsilon;  // already squared
  float min_eig;
  uint8_t *status;
  float *err;
};

template <int WIN>
__global__ __launch_bounds__(64) void klt_track_kernel(KltArgs a) {
  using C = KltCfg<WIN>;
  __shared__ uint32_t s_tt[C::TT_H * C::TT_WD];
  __shared__ uint32_t s_tj[C::TJ_H * C::TJ_WD];
  const int n = a.d_n ? *a.d_n : a.n;
  const int pt = blockIdx.x;
  if (pt >= n) return;
  const int lane = threadIdx.x;
  const float *pin = a.pts1_init ? a.pts1_init : a.pts1;
  const KltResult r = klt_point<WIN>(a.I, a.J, a.max_level, a.flags, a.max_count, a.epsilon, a.min_eig, a.pts0[2 * pt],
                                     a.pts0[2 * pt + 1], pin[2 * pt], pin[2 * pt + 1], s_tt, s_tj, lane);
  if (lane == 0) {
    a.pts1[2 * pt] = r.x;
    a.pts1[2 * pt + 1] = r.y;
    a.status[pt] = (uint8_t)r.status;
    a.err[pt] = r.err;
  }
}

// ---- validity masks of the FeatureTracker wrappers ------------------------------
// mode 0: track()                      mask &= status>0 && err<=thr                         (:33-34)
// mode 1: trackWithPrior()             mask &= status>0 && 0<x<W && 0<y<H && err<=thr       (:191-197)
// mode 2: trackBidirection()           3-px border, both status/err, |bwd-pts0|^2<=thr^2    (:74-83)
// mode 3: trackBidirectionWithPrior()  in-image, both status/err, |bwd-pts0|^2<=5 thr^2     (:130-155)
struct MaskArgs {
  int mode;
  int n;
  const int *d_n;
  int n_cols, n_rows;
  float thres_err, thres_bidir;
  const float *pts0, *pts_track, *pts_back;
  const uint8_t *st_f, *st_b;
  const float *err_f, *err_b;
  const uint8_t *mask_in;  // optional (null = all true)
  uint8_t *mask;           // out (may alias mask_in)
};
__global__ __launch_bounds__(256) void klt_mask_kernel(MaskArgs a) {
  const int n = a.d_n ? *a.d_n : a.n;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a.pts_track[2 * i], y = a.pts_track[2 * i + 1];
  bool m = a.mask_in ? a.mask_in[i] != 0 : true;
  if (a.mode == 0) {
    m = m && a.st_f[i] > 0 && a.err_f[i] <= a.thres_err;
  } else if (a.mode == 1) {
    m = m && a.st_f[i] > 0 && x > 0 && x < a.n_cols && y > 0 && y < a.n_rows;
    m = m && a.err_f[i] <= a.thres_err;
  } else {
    const float dx = a.pts_back[2 * i] - a.pts0[2 * i], dy = a.pts_back[2 * i + 1] - a.pts0[2 * i + 1];
    const float dist2 = dx * dx + dy * dy;
    const float thres2 = a.thres_bidir * a.thres_bidir;
    if (a.mode == 2) {
      m = m && x > 3 && x < a.n_cols - 3 && y > 3 && y < a.n_rows - 3;
      m = m && a.st_f[i] && a.st_b[i] && a.err_f[i] <= a.thres_err && a.err_b[i] <= a.thres_err && dist2 <= thres2;
    } else {
      const bool inimage = x > 0 && x < a.n_cols && y > 0 && y < a.n_rows;
      m = m && inimage && a.st_f[i] && a.err_f[i] <= a.thres_err && a.st_b[i] && a.err_b[i] <= a.thres_err &&
          dist2 <= thres2 * 5;
    }
  }
  a.mask[i] = m ? 1 : 0;
}

// ---- host launchers ---------------------------------------------------------------
template <int WIN>
static void launch_klt(vo_ctx *c, const KltArgs &a, int grid) {
  hipLaunchKernelGGL(klt_track_kernel<WIN>, dim3(grid), dim3(64), 0, c->stream, a);
}

// Enqueue one calcOpticalFlowPyrLK on device-resident points. n_max bounds the grid
// when the live count is only known on the device (d_n).
int vo_klt_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_pts1_init,
                   float *d_pts1, int n_max,
                   const int *d_n, int win, int max_level, int flags, int max_iter, double eps,
                   float min_eig, uint8_t *d_status, float *d_err) {
  if (slot0 < 0 || slot0 >= c->cfg.n_slots || slot1 < 0 || slot1 >= c->cfg.n_slots)
    VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  const vo_pyramid &P0 = c->slots[slot0], &P1 = c->slots[slot1];
  if (P0.n_levels <= 0 || P1.n_levels <= 0) VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
  if (P0.w != P1.w || P0.h != P1.h) VO_FAIL(c, VO_ERR_SIZE, "prev/next image size mismatch");
  if (max_level < 0 || win <= 2) VO_FAIL(c, VO_ERR_INVALID, "maxLevel >= 0 && winSize > 2 violated");
  if (n_max <= 0) return VO_OK;
  KltArgs a;
  memset(&a, 0, sizeof(a));
  const int eff = vo_pyr_levels_host(P0.w, P0.h, win, max_level);
  VO_NEED_LEVELS(c, P0, eff);
  VO_NEED_LEVELS(c, P1, eff);
  if (vo_slot_acquire(c, slot0) < 0 || vo_slot_acquire(c, slot1) < 0) return VO_ERR_HIP;
  for (int l = 0; l <= eff; ++l) {
    a.I[l] = P0.lv[l];
    a.J[l] = P1.lv[l];
  }
  a.max_level = eff;
  a.pts0 = d_pts0;
  a.pts1 = d_pts1;
  a.pts1_init = d_pts1_init;
  a.n = n_max;
  a.d_n = d_n;
  a.flags = flags;
  a.max_count = max_iter <= 0 ? 30 : (max_iter > 100 ? 100 : max_iter);
  double e = eps <= 0 ? 0.01 : (eps > 10. ? 10. : eps);
  a.epsilon = e * e;
  a.min_eig = min_eig;
  a.status = d_status;
  a.err = d_err;
  vo_prof_begin(c, VO_K_KLT);
  switch (win) {
    case 7: launch_klt<7>(c, a, n_max); break;
    case 9: launch_klt<9>(c, a, n_max); break;
    case 11: launch_klt<11>(c, a, n_max); break;
    case 13: launch_klt<13>(c, a, n_max); break;
    case 15: launch_klt<15>(c, a, n_max); break;
    case 17: launch_klt<17>(c, a, n_max); break;
    case 19: launch_klt<19>(c, a, n_max); break;
    case 21: launch_klt<21>(c, a, n_max); break;
    case 23: launch_klt<23>(c, a, n_max); break;
    case 25: launch_klt<25>(c, a, n_max); break;
    case 31: launch_klt<31>(c, a, n_max); break;
    default:
      vo_prof_end(c);
      VO_FAIL(c, VO_ERR_INVALID, "window size %d not instantiated (odd 7..25, 31)", win);
  }
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return eff;
}

int vo_klt_mask_enqueue(vo_ctx *c, int mode, int n_max, const int *d_n, int n_cols, int n_rows,
                        float thres_err, float thres_bidir, const float *pts0, const float *pts_track,
                        const float *pts_back, const uint8_t *st_f, const uint8_t *st_b,
                        const float *err_f, const float *err_b, const uint8_t *mask_in, uint8_t *mask) {
  if (n_max <= 0) return VO_OK;
  MaskArgs a = {mode, n_max, d_n, n_cols, n_rows, thres_err, thres_bidir, pts0, pts_track, pts_back,
                st_f, st_b, err_f, err_b, mask_in, mask};
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(klt_mask_kernel, dim3((n_max + 255) / 256), dim3(256), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
