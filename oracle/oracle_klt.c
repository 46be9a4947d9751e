/*
 * oracle_klt.c — CPU restatement of the pyramidal Lucas-Kanade tracker that the
 * reference's FeatureTracker wraps, plus the FeatureTracker validity masks.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h). PARITY UNPINNED.
 *
 * Third-party algorithm, not in /root/reference: OpenCV 4.x
 *   modules/video/src/lkpyramid.cpp  (calcOpticalFlowPyrLK, buildOpticalFlowPyramid,
 *                                     SharrDerivInvoker, LKTrackerInvoker)
 *   modules/imgproc/src/pyramids.cpp (pyrDown, 5-tap [1 4 6 4 1]/16, BORDER_REFLECT_101)
 * pinned by the reference only as "OpenCV 4" (core/CMakeLists.txt:12).
 * UPSTREAM VERSION THIS WAS WRITTEN AGAINST: the OpenCV 4.5.x line, tag 4.5.4 (the libopencv-dev of Ubuntu 22.04 /
 * ROS 2 Humble, which is what the reference's ROS 2 nodes link). Nothing was read from disk or from the network —
 * the restatement is from the author's knowledge of that source — so a reader with OpenCV should diff, function by
 * function:   vo_ref_pyr_down            <-> pyramids.cpp  pyrDown_<FixPtCast<uchar,8>, ...>  (4.5.4)
 *             vo_ref_pyramid_levels      <-> lkpyramid.cpp buildOpticalFlowPyramid: the `level != 0 && (sz.width <=
 *                                            winSize.width || sz.height <= winSize.height)` stop
 *             vo_ref_scharr              <-> lkpyramid.cpp calcSharrDeriv / ScharrDerivInvoker (3/10/3, s16)
 *             klt_point (static)         <-> lkpyramid.cpp LKTrackerInvoker::operator(): W_BITS 14, descale 5 /
 *                                            W_BITS1-5, FLT_SCALE 2^-20, the minEig test, `j > 0 && |delta + prevDelta|
 *                                            < 0.01` back-off, the level-0-only status / err rules
 *             vo_ref_calc_optical_flow_pyr_lk <-> calcOpticalFlowPyrLK / SparsePyrLKOpticalFlowImpl::calc: criteria
 *                                            defaulting (maxCount clamp to [0,100], epsilon to [0,10], squared)
 * The 4.x line changed none of these between 4.2 and 4.8 to the author's knowledge; 3.x differs in the criteria
 * defaulting only.
 * Call sites that anchor the semantics (arguments, flags, defaulted criteria):
 *   core/visual_odometry/feature_tracker.cpp:29   track()                     defaults (30, 0.01), minEig 1e-4
 *   core/visual_odometry/feature_tracker.cpp:60,69   trackBidirection()       bwd: maxLevel-1, USE_INITIAL_FLOW, {} criteria, minEig {}=0
 *   core/visual_odometry/feature_tracker.cpp:108,117 trackBidirectionWithPrior()
 *   core/visual_odometry/feature_tracker.cpp:186  trackWithPrior()
 * Masks: feature_tracker.cpp:33-34, 74-83, 130-155, 191-197.
 *
 * One deliberate, documented choice: OpenCV accumulates the integer-valued
 * products ix*ix, ix*iy, iy*iy, diff*ix, diff*iy in float32, in an order that
 * depends on its SIMD build (they exceed 2^24, so that float sum rounds). Here
 * they are accumulated exactly in int64 and rounded to float once, which is
 * order-independent and within one float rounding per addend of every OpenCV
 * build. The HIP kernel does the same, so GPU-vs-oracle parity is bit-exact.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define W_BITS 14

/* diagnostic: LK iterations executed per point, summed over levels (reset by the reader) */
static int g_klt_iters[1 << 16];
void vo_ref_klt_iters(int *out, int n) {
  for (int i = 0; i < n && i < (1 << 16); ++i) {
    out[i] = g_klt_iters[i];
    g_klt_iters[i] = 0;
  }
}

static inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0)
      p = -p;
    else
      p = 2 * n - 2 - p;
  }
  return p;
}
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_floor(float v) { return (int)floorf(v); }
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

/* buildOpticalFlowPyramid level rule: after level L is produced the next size
 * is ((w+1)/2,(h+1)/2); the build stops when that is <= winSize in either
 * dimension. Returns the effective maxLevel (levels 0..ret exist). */
int vo_ref_pyramid_levels(int w, int h, int win, int max_level) {
  int level;
  for (level = 0; level <= max_level; ++level) {
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (w <= win || h <= win) return level;
  }
  return max_level;
}
void vo_ref_level_size(int w, int h, int level, int *lw, int *lh) {
  for (int l = 0; l < level; ++l) {
    w = (w + 1) / 2;
    h = (h + 1) / 2;
  }
  *lw = w;
  *lh = h;
}

/* cv::pyrDown for CV_8UC1, default border (REFLECT_101), dst = ((w+1)/2,(h+1)/2).
 * dst(x,y) = (sum_{i,j} k[i]k[j] src(2x+i-2, 2y+j-2) + 128) >> 8, k=[1 4 6 4 1]. */
void vo_ref_pyr_down(const uint8_t *src, int sw, int sh, int sstride,
                     uint8_t *dst, int dstride) {
  int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  static const int k[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      int s = 0;
      for (int j = 0; j < 5; ++j) {
        int sy = reflect101(2 * y + j - 2, sh);
        int r = 0;
        for (int i = 0; i < 5; ++i) {
          int sx = reflect101(2 * x + i - 2, sw);
          r += k[i] * src[sy * sstride + sx];
        }
        s += k[j] * r;
      }
      dst[y * dstride + x] = (uint8_t)((s + 128) >> 8);
    }
}

/* SharrDerivInvoker: 3x3 Scharr, [3 10 3] smoothing x [-1 0 1], REFLECT_101 at
 * the image edge, int16 output interleaved (dx,dy). */
void vo_ref_scharr(const uint8_t *src, int w, int h, int sstride, int16_t *dxy) {
  for (int y = 0; y < h; ++y) {
    const uint8_t *r0 = src + reflect101(y - 1, h) * sstride;
    const uint8_t *r1 = src + y * sstride;
    const uint8_t *r2 = src + reflect101(y + 1, h) * sstride;
    for (int x = 0; x < w; ++x) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10;
      int t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
      int t1m = r2[xm] - r0[xm];
      int t1c = r2[x] - r0[x];
      int t1p = r2[xp] - r0[xp];
      dxy[(y * w + x) * 2 + 0] = (int16_t)(t0p - t0m);
      dxy[(y * w + x) * 2 + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
    }
  }
}

/* cv::Sobel(src, CV_32F, 1,0 / 0,1, ksize 3, scale 1, delta 0, BORDER_DEFAULT)
 * as called at stereo_vo.cpp:551-552 and mono_vo.cpp:781-782. */
void vo_ref_sobel3(const uint8_t *src, int w, int h, int sstride, float *du,
                   float *dv) {
  for (int y = 0; y < h; ++y) {
    const uint8_t *r0 = src + reflect101(y - 1, h) * sstride;
    const uint8_t *r1 = src + y * sstride;
    const uint8_t *r2 = src + reflect101(y + 1, h) * sstride;
    for (int x = 0; x < w; ++x) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      int gx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
      int gy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
      du[y * w + x] = (float)gx;
      dv[y * w + x] = (float)gy;
    }
  }
}

/* A pyramid level as OpenCV holds it: the image with a winSize-wide
 * REFLECT_101 border, and (for the template side) the Scharr derivative with a
 * winSize-wide ZERO border (copyMakeBorder BORDER_CONSTANT in calc()). */
typedef struct {
  int w, h, pad, stride; /* stride of padded buffers, in elements */
  uint8_t *img;          /* (h+2pad) x stride */
  int16_t *deriv;        /* (h+2pad) x stride x 2, may be NULL */
} lk_level;

static void level_alloc(lk_level *L, int w, int h, int pad) {
  L->w = w;
  L->h = h;
  L->pad = pad;
  L->stride = w + 2 * pad;
  L->img = (uint8_t *)malloc((size_t)(h + 2 * pad) * (size_t)L->stride);
  L->deriv = NULL;
}
static void level_make_border(lk_level *L) {
  int pad = L->pad, w = L->w, h = L->h, st = L->stride;
  for (int y = -pad; y < h + pad; ++y) {
    int sy = reflect101(y, h);
    uint8_t *drow = L->img + (size_t)(y + pad) * st + pad;
    const uint8_t *srow = L->img + (size_t)(sy + pad) * st + pad;
    for (int x = -pad; x < w + pad; ++x) {
      if (x >= 0 && x < w && y >= 0 && y < h) continue;
      drow[x] = srow[reflect101(x, w)];
    }
  }
}
static void level_make_deriv(lk_level *L) {
  int pad = L->pad, w = L->w, h = L->h, st = L->stride;
  size_t total = (size_t)(h + 2 * pad) * (size_t)st * 2;
  L->deriv = (int16_t *)calloc(total, sizeof(int16_t));
  int16_t *tmp = (int16_t *)malloc((size_t)w * h * 2 * sizeof(int16_t));
  vo_ref_scharr(L->img + (size_t)pad * st + pad, w, h, st, tmp);
  for (int y = 0; y < h; ++y)
    memcpy(L->deriv + ((size_t)(y + pad) * st + pad) * 2, tmp + (size_t)y * w * 2,
           (size_t)w * 2 * sizeof(int16_t));
  free(tmp);
}
static void level_free(lk_level *L) {
  free(L->img);
  free(L->deriv);
}

static int build_pyramid(const uint8_t *img, int w, int h, int stride, int win,
                         int max_level, int with_deriv, lk_level *levels) {
  int nlev = vo_ref_pyramid_levels(w, h, win, max_level);
  level_alloc(&levels[0], w, h, win);
  for (int y = 0; y < h; ++y)
    memcpy(levels[0].img + (size_t)(y + win) * levels[0].stride + win,
           img + (size_t)y * stride, (size_t)w);
  level_make_border(&levels[0]);
  for (int l = 1; l <= nlev; ++l) {
    int lw = (levels[l - 1].w + 1) / 2, lh = (levels[l - 1].h + 1) / 2;
    level_alloc(&levels[l], lw, lh, win);
    const lk_level *P = &levels[l - 1];
    vo_ref_pyr_down(P->img + (size_t)P->pad * P->stride + P->pad, P->w, P->h, P->stride,
                    levels[l].img + (size_t)win * levels[l].stride + win, levels[l].stride);
    level_make_border(&levels[l]);
  }
  if (with_deriv)
    for (int l = 0; l <= nlev; ++l) level_make_deriv(&levels[l]);
  return nlev;
}

/* LKTrackerInvoker::operator() for one point at one level. */
static void lk_point_level(const lk_level *I, const lk_level *J, const float *prevPts,
                           float *nextPts, uint8_t *status, float *err, int ptidx,
                           int win, int level, int maxLevel, int flags,
                           int max_count, double epsilon, float minEigThreshold,
                           int16_t *IWinBuf, int16_t *dIWinBuf) {
  const float halfWin = (win - 1) * 0.5f;
  const float lscale = (float)(1. / (1 << level));
  float prevPt_x = prevPts[ptidx * 2] * lscale, prevPt_y = prevPts[ptidx * 2 + 1] * lscale;
  float nextPt_x, nextPt_y;
  if (level == maxLevel) {
    if (flags & VO_KLT_USE_INITIAL_FLOW) {
      nextPt_x = nextPts[ptidx * 2] * lscale;
      nextPt_y = nextPts[ptidx * 2 + 1] * lscale;
    } else {
      nextPt_x = prevPt_x;
      nextPt_y = prevPt_y;
    }
  } else {
    nextPt_x = nextPts[ptidx * 2] * 2.f;
    nextPt_y = nextPts[ptidx * 2 + 1] * 2.f;
  }
  nextPts[ptidx * 2] = nextPt_x;
  nextPts[ptidx * 2 + 1] = nextPt_y;

  prevPt_x -= halfWin;
  prevPt_y -= halfWin;
  int iprev_x = cv_floor(prevPt_x), iprev_y = cv_floor(prevPt_y);
  if (iprev_x < -win || iprev_x >= I->w || iprev_y < -win || iprev_y >= I->h) {
    if (level == 0) {
      status[ptidx] = 0;
      err[ptidx] = 0;
    }
    return;
  }
  float a = prevPt_x - iprev_x;
  float b = prevPt_y - iprev_y;
  const float FLT_SCALE = 1.f / (1 << 20);
  int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
  int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
  int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
  int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

  const int stI = I->stride, stJ = J->stride;
  const uint8_t *Ibase = I->img + (size_t)I->pad * stI + I->pad;
  const int16_t *Dbase = I->deriv + ((size_t)I->pad * stI + I->pad) * 2;
  const uint8_t *Jbase = J->img + (size_t)J->pad * stJ + J->pad;

  int64_t iA11 = 0, iA12 = 0, iA22 = 0;
  for (int y = 0; y < win; ++y) {
    const uint8_t *src = Ibase + (ptrdiff_t)(y + iprev_y) * stI + iprev_x;
    const int16_t *dsrc = Dbase + ((ptrdiff_t)(y + iprev_y) * stI + iprev_x) * 2;
    int dstep = stI * 2;
    for (int x = 0; x < win; ++x, dsrc += 2) {
      int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + stI] * iw10 +
                             src[x + stI + 1] * iw11,
                         W_BITS - 5);
      int ixval = descale(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 +
                              dsrc[dstep + 2] * iw11,
                          W_BITS);
      int iyval = descale(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 +
                              dsrc[dstep + 3] * iw11,
                          W_BITS);
      IWinBuf[y * win + x] = (int16_t)ival;
      dIWinBuf[(y * win + x) * 2] = (int16_t)ixval;
      dIWinBuf[(y * win + x) * 2 + 1] = (int16_t)iyval;
      iA11 += (int64_t)(ixval * ixval);
      iA12 += (int64_t)(ixval * iyval);
      iA22 += (int64_t)(iyval * iyval);
    }
  }
  float A11 = (float)iA11 * FLT_SCALE;
  float A12 = (float)iA12 * FLT_SCALE;
  float A22 = (float)iA22 * FLT_SCALE;
  float D = A11 * A22 - A12 * A12;
  float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                 (float)(2 * win * win);
  if (minEig < minEigThreshold || D < 1.19209290e-07f /* FLT_EPSILON */) {
    if (level == 0) status[ptidx] = 0;
    return;
  }
  D = 1.f / D;
  nextPt_x -= halfWin;
  nextPt_y -= halfWin;
  float prevDelta_x = 0, prevDelta_y = 0;
  for (int j = 0; j < max_count; ++j) {
    if (ptidx < (1 << 16)) g_klt_iters[ptidx]++;
    int inext_x = cv_floor(nextPt_x), inext_y = cv_floor(nextPt_y);
    if (inext_x < -win || inext_x >= J->w || inext_y < -win || inext_y >= J->h) {
      if (level == 0) status[ptidx] = 0;
      break;
    }
    a = nextPt_x - inext_x;
    b = nextPt_y - inext_y;
    iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    int64_t ib1 = 0, ib2 = 0;
    for (int y = 0; y < win; ++y) {
      const uint8_t *Jptr = Jbase + (ptrdiff_t)(y + inext_y) * stJ + inext_x;
      for (int x = 0; x < win; ++x) {
        int diff = descale(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stJ] * iw10 +
                               Jptr[x + stJ + 1] * iw11,
                           W_BITS - 5) -
                   IWinBuf[y * win + x];
        ib1 += (int64_t)(diff * dIWinBuf[(y * win + x) * 2]);
        ib2 += (int64_t)(diff * dIWinBuf[(y * win + x) * 2 + 1]);
      }
    }
    float b1 = (float)ib1 * FLT_SCALE;
    float b2 = (float)ib2 * FLT_SCALE;
    float delta_x = (float)((A12 * b2 - A22 * b1) * D);
    float delta_y = (float)((A12 * b1 - A11 * b2) * D);
    nextPt_x += delta_x;
    nextPt_y += delta_y;
    nextPts[ptidx * 2] = nextPt_x + halfWin;
    nextPts[ptidx * 2 + 1] = nextPt_y + halfWin;
    if ((double)delta_x * delta_x + (double)delta_y * delta_y <= epsilon) break;
    if (j > 0 && fabs((double)(delta_x + prevDelta_x)) < 0.01 &&
        fabs((double)(delta_y + prevDelta_y)) < 0.01) {
      nextPts[ptidx * 2] -= delta_x * 0.5f;
      nextPts[ptidx * 2 + 1] -= delta_y * 0.5f;
      break;
    }
    prevDelta_x = delta_x;
    prevDelta_y = delta_y;
  }
  if (status[ptidx] && level == 0) {
    float np_x = nextPts[ptidx * 2] - halfWin, np_y = nextPts[ptidx * 2 + 1] - halfWin;
    int ix = cv_floor(np_x), iy = cv_floor(np_y);
    if (ix < -win || ix >= J->w || iy < -win || iy >= J->h) {
      status[ptidx] = 0;
      return;
    }
    float aa = np_x - ix, bb = np_y - iy;
    iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
    iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
    iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    float errval = 0.f;
    for (int y = 0; y < win; ++y) {
      const uint8_t *Jptr = Jbase + (ptrdiff_t)(y + iy) * stJ + ix;
      for (int x = 0; x < win; ++x) {
        int diff = descale(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stJ] * iw10 +
                               Jptr[x + stJ + 1] * iw11,
                           W_BITS - 5) -
                   IWinBuf[y * win + x];
        errval += fabsf((float)diff);
      }
    }
    err[ptidx] = errval * 1.f / (32 * win * win);
  }
}

/* cv::calcOpticalFlowPyrLK(prevImg,nextImg,prevPts,nextPts,status,err,winSize,
 * maxLevel,criteria,flags,minEigThreshold).  max_iter<=0 / eps<=0 select the
 * defaults the reference gets from `{}` criteria (30 / 0.01). `eps` is the
 * un-squared epsilon. Returns the effective maxLevel, <0 on error. */
int vo_ref_calc_optical_flow_pyr_lk(const uint8_t *img0, const uint8_t *img1,
                                    int w, int h, int stride, const float *pts0,
                                    float *pts1, int n, int win, int max_level,
                                    int flags, int max_iter, double eps,
                                    float min_eig_thr, uint8_t *status,
                                    float *err, int n_threads) {
  if (max_level < 0 || win <= 2) return -1;
  if (n == 0) return 0;
  int max_count = max_iter <= 0 ? 30 : (max_iter > 100 ? 100 : max_iter);
  double epsilon = eps <= 0 ? 0.01 : (eps > 10. ? 10. : eps);
  epsilon *= epsilon;
  lk_level *P0 = (lk_level *)calloc((size_t)max_level + 1, sizeof(lk_level));
  lk_level *P1 = (lk_level *)calloc((size_t)max_level + 1, sizeof(lk_level));
  int nlev = build_pyramid(img0, w, h, stride, win, max_level, 1, P0);
  int nlev1 = build_pyramid(img1, w, h, stride, win, nlev, 0, P1);
  (void)nlev1;
  for (int i = 0; i < n; ++i) {
    status[i] = 1;
    err[i] = 0.f;
  }
  if (!(flags & VO_KLT_USE_INITIAL_FLOW))
    for (int i = 0; i < 2 * n; ++i) pts1[i] = 0.f;
  (void)n_threads;
  for (int level = nlev; level >= 0; --level) {
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
      int16_t *IWin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win);
      int16_t *dIWin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win * 2);
#pragma omp for schedule(static)
      for (int i = 0; i < n; ++i)
        lk_point_level(&P0[level], &P1[level], pts0, pts1, status, err, i, win, level,
                       nlev, flags, max_count, epsilon, min_eig_thr, IWin, dIWin);
      free(IWin);
      free(dIWin);
    }
  }
  for (int l = 0; l <= nlev; ++l) {
    level_free(&P0[l]);
    level_free(&P1[l]);
  }
  free(P0);
  free(P1);
  return nlev;
}

/* ---- FeatureTracker front-ends ---------------------------------------- */
/* feature_tracker.cpp:13-37 */
int vo_ref_track(const uint8_t *img0, const uint8_t *img1, int w, int h,
                 int stride, const float *pts0, int n, int win, int max_level,
                 float thres_err, float *pts_track, uint8_t *mask,
                 int n_threads) {
  uint8_t *status = (uint8_t *)malloc((size_t)n + 1);
  float *err = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  int rc = vo_ref_calc_optical_flow_pyr_lk(img0, img1, w, h, stride, pts0, pts_track, n, win,
                                           max_level, 0, 30, 0.01, 1e-4f, status, err,
                                           n_threads);
  for (int i = 0; i < n; ++i) mask[i] = (mask[i] && status[i] > 0 && err[i] <= thres_err);
  free(status);
  free(err);
  return rc;
}

/* feature_tracker.cpp:39-86 */
int vo_ref_track_bidirection(const uint8_t *img0, const uint8_t *img1, int w,
                             int h, int stride, const float *pts0, int n,
                             int win, int max_level, float thres_err,
                             float thres_bidirection, float *pts_track,
                             uint8_t *mask, int n_threads) {
  float thres2 = thres_bidirection * thres_bidirection;
  uint8_t *sf = (uint8_t *)malloc((size_t)n + 1), *sb = (uint8_t *)malloc((size_t)n + 1);
  float *ef = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  float *eb = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  float *back = (float *)malloc(sizeof(float) * 2 * ((size_t)n + 1));
  int rc = vo_ref_calc_optical_flow_pyr_lk(img0, img1, w, h, stride, pts0, pts_track, n, win,
                                           max_level, 0, 30, 0.01, 1e-4f, sf, ef, n_threads);
  memcpy(back, pts0, sizeof(float) * 2 * (size_t)n);
  /* :69-71  maxLevel-1, {} criteria, USE_INITIAL_FLOW, {} minEigThreshold (=0) */
  if (rc >= 0 && max_level - 1 >= 0)
    vo_ref_calc_optical_flow_pyr_lk(img1, img0, w, h, stride, pts_track, back, n, win,
                                    max_level - 1, VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f, sb,
                                    eb, n_threads);
  else
    rc = -1;
  for (int i = 0; i < n; ++i) {
    float dx = back[2 * i] - pts0[2 * i], dy = back[2 * i + 1] - pts0[2 * i + 1];
    float dist2 = dx * dx + dy * dy;
    float x = pts_track[2 * i], y = pts_track[2 * i + 1];
    uint8_t m = (mask[i] && x > 3 && x < w - 3 && y > 3 && y < h - 3);
    m = (m && sf[i] && sb[i] && ef[i] <= thres_err && eb[i] <= thres_err && dist2 <= thres2);
    mask[i] = m;
  }
  free(sf);
  free(sb);
  free(ef);
  free(eb);
  free(back);
  return rc;
}

/* feature_tracker.cpp:88-169 */
int vo_ref_track_bidirection_with_prior(const uint8_t *img0,
                                        const uint8_t *img1, int w, int h,
                                        int stride, const float *pts0, int n,
                                        int win, int max_level, float thres_err,
                                        float thres_bidirection,
                                        float *pts_track, uint8_t *mask,
                                        int n_threads) {
  float thres2 = thres_bidirection * thres_bidirection;
  uint8_t *sf = (uint8_t *)malloc((size_t)n + 1), *sb = (uint8_t *)malloc((size_t)n + 1);
  float *ef = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  float *eb = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  float *back = (float *)malloc(sizeof(float) * 2 * ((size_t)n + 1));
  int rc = vo_ref_calc_optical_flow_pyr_lk(img0, img1, w, h, stride, pts0, pts_track, n, win,
                                           max_level, VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f, sf,
                                           ef, n_threads);
  memcpy(back, pts0, sizeof(float) * 2 * (size_t)n);
  vo_ref_calc_optical_flow_pyr_lk(img1, img0, w, h, stride, pts_track, back, n, win, max_level,
                                  VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f, sb, eb, n_threads);
  for (int i = 0; i < n; ++i) {
    float dx = back[2 * i] - pts0[2 * i], dy = back[2 * i + 1] - pts0[2 * i + 1];
    float dist2 = dx * dx + dy * dy;
    float x = pts_track[2 * i], y = pts_track[2 * i + 1];
    int inimage = x > 0 && x < w && y > 0 && y < h;
    int bidir = dist2 <= thres2 * 5;
    mask[i] = (mask[i] && inimage && sf[i] && ef[i] <= thres_err && sb[i] &&
               eb[i] <= thres_err && bidir);
  }
  free(sf);
  free(sb);
  free(ef);
  free(eb);
  free(back);
  return rc;
}

/* feature_tracker.cpp:171-206 */
int vo_ref_track_with_prior(const uint8_t *img0, const uint8_t *img1, int w,
                            int h, int stride, const float *pts0, int n,
                            int win, int max_level, float thres_err,
                            float *pts_track, uint8_t *mask, int n_threads) {
  uint8_t *status = (uint8_t *)malloc((size_t)n + 1);
  float *err = (float *)malloc(sizeof(float) * ((size_t)n + 1));
  int rc = vo_ref_calc_optical_flow_pyr_lk(img0, img1, w, h, stride, pts0, pts_track, n, win,
                                           max_level, VO_KLT_USE_INITIAL_FLOW, 0, 0., 0.f,
                                           status, err, n_threads);
  for (int i = 0; i < n; ++i) {
    float x = pts_track[2 * i], y = pts_track[2 * i + 1];
    uint8_t m = (mask[i] && status[i] > 0 && x > 0 && x < w && y > 0 && y < h);
    mask[i] = (m && err[i] <= thres_err);
  }
  free(status);
  free(err);
  return rc;
}

/* feature_tracker.cpp:208-234 */
void vo_ref_calc_prior(const float *pts0, int n_pts0, const float *Xw, int n,
                       const float Tw1[16], const float K[9],
                       float *pts1_prior) {
  memcpy(pts1_prior, pts0, sizeof(float) * 2 * (size_t)n_pts0);
  float T1w[16];
  vo_ref_inverse4x4(Tw1, T1w);
  for (int i = 0; i < n; ++i) {
    const float *Xi = Xw + 3 * i;
    float X[3];
    for (int r = 0; r < 3; ++r)
      X[r] = ((T1w[r * 4 + 0] * Xi[0] + T1w[r * 4 + 1] * Xi[1]) + T1w[r * 4 + 2] * Xi[2]) +
             T1w[r * 4 + 3];
    float nrm = sqrtf((X[0] * X[0] + X[1] * X[1]) + X[2] * X[2]);
    if (nrm > 0) {
      pts1_prior[2 * i] = K[0] * X[0] / X[2] + K[2];
      pts1_prior[2 * i + 1] = K[4] * X[1] / X[2] + K[5];
    }
  }
}
