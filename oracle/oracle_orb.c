/* oracle_orb.c — CPU restatement (TEST INFRASTRUCTURE, parity unpinned — see vo_oracle.h) of the keypoint
 * detection behind FeatureExtractor::extractORBwithBinning_fast (core/visual_odometry/feature_extractor.cpp:
 * 211-318): `extractor_orb_->detect(img_in, fts)` with the parameters set in initParams (:30-57: 10000
 * features, scale 1.2, 8 levels, edge threshold 31, first level 0, HARRIS_SCORE, patch size 31,
 * FAST threshold THRES_FAST).
 *
 * cv::ORB is OpenCV 4 features2d (modules/features2d/src/orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp;
 * modules/imgproc/src/resize.cpp for INTER_LINEAR_EXACT) and is NOT in the reference tree nor in this
 * container. What follows restates its published algorithm from the upstream sources as the author knows
 * them; nothing here could be checked against an OpenCV build.
 * UPSTREAM VERSION THIS WAS WRITTEN AGAINST: the OpenCV 4.5.x line, tag 4.5.4 (Ubuntu 22.04 / ROS 2 Humble's
 * libopencv-dev). To diff with OpenCV at hand:
 *   vo_ref_orb_level_sizes          <-> orb.cpp ORB_Impl::detectAndCompute: layerInfo / getScale, computeKeyPoints:
 *                                       nfeaturesPerLevel (factor = (float)(1.0 / scaleFactor), cvRound, remainder)
 *   vo_ref_resize_linear_exact_u8   <-> resize.cpp resize_bitExact / interpolationLinear<uchar> (INTER_LINEAR_EXACT),
 *                                       fixedpoint.hpp ufixedpoint16 / ufixedpoint32 rounding
 *   vo_ref_fast_score_image         <-> fast.cpp FAST_t<16> (ring offsets, the threshold_tab test, N = 9 contiguous)
 *                                       and fast_score.cpp cornerScore<16>
 *   non-max suppression             <-> fast.cpp FAST_t: strict `score > prev/curr/pprev neighbours` over 3x3
 *   runByImageBorder / retainBest   <-> keypoint.cpp KeyPointsFilter (nth_element + keep the ties)
 *   harris_response (static)        <-> orb.cpp HarrisResponses (blockSize 7, scale = 1/((1 << 2) * blockSize * 255),
 *                                       scale_sq_sq, k = 0.04f)
 * The level loop in 4.5.4 detects on a per-level image with an edge-threshold border copied by copyMakeBorder
 * (BORDER_REFLECT_101) — detection results inside the border-filtered region do not depend on it.
 * Summary of what is restated:
 *   pyramid     level 0 = the image; level l = resize(level l-1, Size(cvRound(cols/s), cvRound(rows/s)),
 *               INTER_LINEAR_EXACT), s = (float)pow(1.2, l)
 *   per level   FAST-9/16 (threshold t, non-max suppression), runByImageBorder(31), retainBest(2 n_l) on the
 *               FAST score; HarrisResponses(blockSize 7, k 0.04); retainBest(n_l) on the Harris response;
 *               pt *= s
 *   n_l         nfeatures (1-f)/(1-f^8) f^l with f = (float)(1/1.2), cvRound, remainder to the last level
 * Orientation (IC_Angle) is not computed: the reference reads only pt and response (:253-275).
 * retainBest keeps everything that ties with the n-th best and leaves the order unspecified (nth_element):
 * the keypoint SET is reproduced; the order here is level, then raster order. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

/* ---- resize INTER_LINEAR_EXACT, 8-bit single channel (resize.cpp: resize_bitExact<uchar,
 * interpolationLinear<uchar>>). Coefficients are ufixedpoint16 (8 fractional bits) made from
 * softdouble arithmetic (= IEEE double, one rounding per operation); horizontal pass in 8.8, vertical pass
 * 8.8 x 8.8 -> 16.16, rounded to the byte with +0.5. */
static void linear_exact_coeffs(int src_size, int dst_size, int *ofs, int *c1, int *dst_min, int *dst_max) {
  /* interpolationLinear(inv_scale, ..): scale = softdouble::one() / softdouble(inv_scale), inv_scale = dst / src */
  const double scale = 1.0 / ((double)dst_size / (double)src_size);
  *dst_min = 0;
  *dst_max = dst_size;
  for (int v = 0; v < dst_size; ++v) {
    const double fval = scale * ((double)v + 0.5) - 0.5;
    const int ival = (int)floor(fval);
    ofs[v] = 0;
    c1[v] = 0;
    if (ival >= 0 && src_size > 1) {
      if (ival < src_size - 1) {
        ofs[v] = ival;
        c1[v] = (int)lrint((fval - (double)ival) * 256.0); /* ufixedpoint16(softdouble): cvRound(x * 2^8) */
      } else {
        ofs[v] = src_size - 1;
        if (v < *dst_max) *dst_max = v;
      }
    } else if (v + 1 > *dst_min)
      *dst_min = v + 1;
  }
}
void vo_ref_resize_linear_exact_u8(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dw, int dh) {
  int *ox = (int *)malloc(sizeof(int) * (size_t)dw), *cx = (int *)malloc(sizeof(int) * (size_t)dw);
  int *oy = (int *)malloc(sizeof(int) * (size_t)dh), *cy = (int *)malloc(sizeof(int) * (size_t)dh);
  int xmin, xmax, ymin, ymax;
  linear_exact_coeffs(w, dw, ox, cx, &xmin, &xmax);
  linear_exact_coeffs(h, dh, oy, cy, &ymin, &ymax);
  for (int y = 0; y < dh; ++y) {
    int y0, y1, wy1;
    if (y < ymin) { y0 = y1 = 0; wy1 = 0; }
    else if (y >= ymax) { y0 = y1 = h - 1; wy1 = 0; }
    else { y0 = oy[y]; y1 = y0 + 1; wy1 = cy[y]; }
    const uint8_t *r0 = src + (size_t)y0 * stride, *r1 = src + (size_t)y1 * stride;
    for (int x = 0; x < dw; ++x) {
      unsigned h0, h1; /* horizontal pass, 8.8 */
      if (x < xmin) { h0 = 256u * r0[0]; h1 = 256u * r1[0]; }
      else if (x >= xmax) { h0 = 256u * r0[w - 1]; h1 = 256u * r1[w - 1]; }
      else {
        const int o = ox[x], a1 = cx[x], a0 = 256 - a1;
        h0 = (unsigned)a0 * r0[o] + (unsigned)a1 * r0[o + 1];
        h1 = (unsigned)a0 * r1[o] + (unsigned)a1 * r1[o + 1];
      }
      const unsigned v = (unsigned)(256 - wy1) * h0 + (unsigned)wy1 * h1; /* 16.16 */
      const unsigned r = (v + 32768u) >> 16;
      dst[(size_t)y * dw + x] = (uint8_t)(r > 255 ? 255 : r);
    }
  }
  free(ox); free(cx); free(oy); free(cy);
}

/* ---- FAST-9/16 (fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) ---- */
static const int fast_off[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
/* 0 when the pixel is not a corner, else cornerScore: the largest threshold for which it still is, i.e.
 * max over the 16 arcs of 9 contiguous circle pixels of min(v - x) and of min(x - v), started at t, minus 1 */
static int fast_score_at(const uint8_t *p, int stride, int t) {
  int d[25];
  const int v = p[0];
  for (int k = 0; k < 25; ++k) d[k] = v - p[fast_off[k & 15][0] + fast_off[k & 15][1] * stride];
  /* corner test: 9 contiguous with d > t (darker ring) or d < -t (brighter ring) */
  int is_corner = 0, cd = 0, cb = 0;
  for (int k = 0; k < 25; ++k) {
    cd = d[k] > t ? cd + 1 : 0;
    cb = d[k] < -t ? cb + 1 : 0;
    if (cd > 8 || cb > 8) { is_corner = 1; break; }
  }
  if (!is_corner) return 0;
  int a0 = t;
  for (int k = 0; k < 16; k += 2) {
    int a = d[k + 1];
    for (int q = 2; q <= 8; ++q) a = a < d[k + q] ? a : d[k + q];
    int m = a < d[k] ? a : d[k];
    a0 = a0 > m ? a0 : m;
    m = a < d[k + 9] ? a : d[k + 9];
    a0 = a0 > m ? a0 : m;
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = d[k + 1];
    for (int q = 2; q <= 8; ++q) b = b > d[k + q] ? b : d[k + q];
    int m = b > d[k] ? b : d[k];
    b0 = b0 < m ? b0 : m;
    m = b > d[k + 9] ? b : d[k + 9];
    b0 = b0 < m ? b0 : m;
  }
  return -b0 - 1;
}
/* score image: 0 outside [3, w-3) x [3, h-3) and for non-corners */
void vo_ref_fast_score_image(const uint8_t *img, int w, int h, int stride, int threshold, uint8_t *score) {
  memset(score, 0, (size_t)w * h);
  /* rows are independent: spread over the host cores (cv::FAST is SIMD code; this is the restatement's way of not being
   * two orders of magnitude slower than the library it stands for when bench.py times it) */
#pragma omp parallel for schedule(static)
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) score[(size_t)y * w + x] = (uint8_t)fast_score_at(img + (size_t)y * stride + x, stride, threshold);
}
/* non-max suppression as in FAST_t: strictly greater than the 8 neighbours' scores */
static int fast_is_max(const uint8_t *s, int w, int x, int y) {
  const int c = s[(size_t)y * w + x];
  if (!c) return 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx)
      if ((dx || dy) && !(c > s[(size_t)(y + dy) * w + x + dx])) return 0;
  return 1;
}

/* ---- HarrisResponses (orb.cpp), blockSize 7 ---- */
static float harris_at(const uint8_t *img, int stride, int x0, int y0) {
  const int r = 3;
  int a = 0, b = 0, c = 0;
  for (int yy = -r; yy <= r; ++yy)
    for (int xx = -r; xx <= r; ++xx) {
      const uint8_t *p = img + (size_t)(y0 + yy) * stride + (x0 + xx);
      const int Ix = (p[1] - p[-1]) * 2 + (p[-stride + 1] - p[-stride - 1]) + (p[stride + 1] - p[stride - 1]);
      const int Iy = (p[stride] - p[-stride]) * 2 + (p[stride - 1] - p[-stride - 1]) + (p[stride + 1] - p[-stride + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

void vo_ref_orb_level_sizes(int w, int h, double scale_factor, int n_levels, int nfeatures, int *lw, int *lh,
                            float *lscale, int *nper) {
  for (int l = 0; l < n_levels; ++l) {
    const float s = (float)pow(scale_factor, (double)l);
    lscale[l] = s;
    lw[l] = (int)lrint((double)((float)w / s));
    lh[l] = (int)lrint((double)((float)h / s));
  }
  const float factor = (float)(1.0 / scale_factor);
  float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)n_levels));
  int sum = 0;
  for (int l = 0; l < n_levels - 1; ++l) {
    nper[l] = (int)lrint((double)nd);
    sum += nper[l];
    nd *= factor;
  }
  nper[n_levels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

static int cmp_int_desc(const void *a, const void *b) { return *(const int *)b - *(const int *)a; }
static int cmp_float_desc(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return x < y ? 1 : (x > y ? -1 : 0);
}

/* cv::ORB::detect. out: x, y (level-0 coordinates, pt * scale), response (Harris), octave. Returns the number
 * of keypoints (<= max_kp; -1 if max_kp is too small). levels_out (optional): n_levels images, tightly packed,
 * concatenated (test hook for the pyramid). */
int vo_ref_orb_detect(const uint8_t *img, int w, int h, int stride, int nfeatures, double scale_factor, int n_levels,
                      int edge_threshold, int fast_threshold, float *kp_xy, float *kp_response, int32_t *kp_octave,
                      int max_kp, uint8_t *levels_out) {
  int lw[32], lh[32], nper[32];
  float ls[32];
  if (n_levels > 32) return -1;
  vo_ref_orb_level_sizes(w, h, scale_factor, n_levels, nfeatures, lw, lh, ls, nper);
  uint8_t *prev = (uint8_t *)malloc((size_t)w * h), *cur = NULL;
  for (int y = 0; y < h; ++y) memcpy(prev + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  int n_out = 0;
  size_t lv_off = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int cw = lw[l], ch = lh[l];
    if (l > 0) {
      cur = (uint8_t *)malloc((size_t)cw * ch);
      vo_ref_resize_linear_exact_u8(prev, lw[l - 1], lh[l - 1], lw[l - 1], cur, cw, ch);
      free(prev);
      prev = cur;
    }
    if (levels_out) {
      memcpy(levels_out + lv_off, prev, (size_t)cw * ch);
      lv_off += (size_t)cw * ch;
    }
    uint8_t *score = (uint8_t *)malloc((size_t)cw * ch);
    vo_ref_fast_score_image(prev, cw, ch, cw, fast_threshold, score);
    /* FAST keypoints after NMS, inside the border (runByImageBorder: Rect(b, b, w-2b, h-2b).contains(pt)) */
    int cap = 1024, n = 0;
    int *kx = (int *)malloc(sizeof(int) * cap), *ky = (int *)malloc(sizeof(int) * cap), *ks = (int *)malloc(sizeof(int) * cap);
    if (cw > 2 * edge_threshold && ch > 2 * edge_threshold)
      for (int y = edge_threshold; y < ch - edge_threshold; ++y)
        for (int x = edge_threshold; x < cw - edge_threshold; ++x)
          if (fast_is_max(score, cw, x, y)) {
            if (n == cap) {
              cap *= 2;
              kx = (int *)realloc(kx, sizeof(int) * cap);
              ky = (int *)realloc(ky, sizeof(int) * cap);
              ks = (int *)realloc(ks, sizeof(int) * cap);
            }
            kx[n] = x; ky[n] = y; ks[n] = score[(size_t)y * cw + x];
            ++n;
          }
    /* retainBest(2 * n_l) on the FAST score: keep everything >= the (2 n_l)-th best */
    int keep_n = 2 * nper[l];
    int thr_s = 0;
    if (n > keep_n) {
      if (keep_n == 0) n = 0;
      else {
        int *tmp = (int *)malloc(sizeof(int) * n);
        memcpy(tmp, ks, sizeof(int) * n);
        qsort(tmp, n, sizeof(int), cmp_int_desc);
        thr_s = tmp[keep_n - 1];
        free(tmp);
      }
    }
    int m = 0;
    float *resp = (float *)malloc(sizeof(float) * (n + 1));
    for (int i = 0; i < n; ++i)
      if (ks[i] >= thr_s) {
        kx[m] = kx[i]; ky[m] = ky[i];
        resp[m] = harris_at(prev, cw, kx[m], ky[m]);
        ++m;
      }
    /* retainBest(n_l) on the Harris response */
    float thr_r = -INFINITY;
    int keep_all = 1;
    if (m > nper[l]) {
      if (nper[l] == 0) m = 0;
      else {
        float *tmp = (float *)malloc(sizeof(float) * m);
        memcpy(tmp, resp, sizeof(float) * m);
        qsort(tmp, m, sizeof(float), cmp_float_desc);
        thr_r = tmp[nper[l] - 1];
        keep_all = 0;
        free(tmp);
      }
    }
    for (int i = 0; i < m; ++i)
      if (keep_all || resp[i] >= thr_r) {
        if (n_out >= max_kp) { n_out = -1; break; }
        kp_xy[2 * n_out] = l ? (float)kx[i] * ls[l] : (float)kx[i];
        kp_xy[2 * n_out + 1] = l ? (float)ky[i] * ls[l] : (float)ky[i];
        kp_response[n_out] = resp[i];
        kp_octave[n_out] = l;
        ++n_out;
      }
    free(kx); free(ky); free(ks); free(resp); free(score);
    if (n_out < 0) break;
  }
  free(prev);
  return n_out;
}
