/* oracle_rectify.c — CPU restatement (TEST INFRASTRUCTURE, parity unpinned — see vo_oracle.h) of the
 * image ingestion in front of the trackers when flagDoUndistortion is set:
 *   Camera::generateImageUndistortMaps            core/visual_odometry/camera.cpp:56-90
 *   StereoCamera::generateStereoImagesUndistortAndRectifyMaps   camera.cpp:364-546
 *   Camera::undistortImage / StereoCamera::rectifyStereoImages  camera.cpp:166-183, :300-336
 *     = convertTo(CV_32FC1), cv::remap(float maps, INTER_LINEAR, BORDER_CONSTANT 0)
 *   followed by convertTo(CV_8UC1) in the drivers   stereo_vo.cpp:420-421, mono_vo.cpp:512
 * cv::remap is OpenCV 4 imgproc (modules/imgproc/src/imgwarp.cpp, not in the tree); its published
 * algorithm for CV_32FC1 maps + INTER_LINEAR is restated here: coordinates are quantised to 1/32 pixel
 * (INTER_BITS = 5: sx = cvRound(mapx * 32), integer part sx >> 5, fraction sx & 31), the four weights
 * come from the float table (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx with fx = k/32, the sample is
 * S00*w0 + S01*w1 + S10*w2 + S11*w3 in float, taps outside the source are the border value 0.
 * convertTo(CV_8UC1) is saturate_cast<uchar>(cvRound(v)), cvRound = round-half-to-even. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

/* camera.cpp:56-90. The literals 2.0 / 1.0 there are doubles: those sub-expressions are evaluated in
 * double and rounded on assignment to the float variables. */
void vo_ref_image_undistort_maps(int n_cols, int n_rows, const float K[4], const float D[5], float *map_u,
                                 float *map_v) {
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const float fxinv = 1.0f / fx, fyinv = 1.0f / fy;
  const float k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  for (int v = 0; v < n_rows; ++v) {
    const float y = ((float)v - cy) * fyinv;
    for (int u = 0; u < n_cols; ++u) {
      const float x = ((float)u - cx) * fxinv;
      const float xy2 = (float)((2.0 * (double)x) * (double)y);
      const float xx = x * x, yy = y * y;
      const float r2 = xx + yy;
      const float r4 = r2 * r2;
      const float r6 = r4 * r2;
      const float r_radial = (float)(((1.0 + (double)(k1 * r2)) + (double)(k2 * r4)) + (double)(k3 * r6));
      const float x_dist = (float)((double)(x * r_radial + p1 * xy2) + (double)p2 * ((double)r2 + 2.0 * (double)xx));
      const float y_dist = (float)(((double)(y * r_radial) + (double)p1 * ((double)r2 + 2.0 * (double)yy)) + (double)(p2 * xy2));
      map_u[(size_t)v * n_cols + u] = cx + x_dist * fx;
      map_v[(size_t)v * n_cols + u] = cy + y_dist * fy;
    }
  }
}

static float dot3e(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }
static void normalize3(float v[3]) {
  const float n = sqrtf(dot3e(v[0], v[0], v[1], v[1], v[2], v[2]));
  v[0] /= n; v[1] /= n; v[2] /= n;
}
static void cross3(const float a[3], const float b[3], float c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void mat3mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = dot3e(A[i * 3], B[j], A[i * 3 + 1], B[3 + j], A[i * 3 + 2], B[6 + j]);
}
/* Eigen's 3x3 inverse (cofactors of column 0 for the determinant), Inverse_impl size 3 */
static float cof3(const float m[9], int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void inv3(const float m[9], float r[9]) {
  const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const float det = dot3e(c0, m[0], c1, m[3], c2, m[6]);
  const float id = 1.0f / det;
  r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
  r[3] = cof3(m, 0, 1) * id; r[4] = cof3(m, 1, 1) * id; r[5] = cof3(m, 2, 1) * id;
  r[6] = cof3(m, 0, 2) * id; r[7] = cof3(m, 1, 2) * id; r[8] = cof3(m, 2, 2) * id;
}

/* The frame algebra of camera.cpp:364-432, :530-535: everything that does not depend on the pixel. */
void vo_ref_stereo_rectify_setup(int n_cols, int n_rows, const float Kl[4], const float Kr[4], const float T_lr[16],
                                 float M[9], float R_l0[9], float R_r0[9], float K_rect[4], float T_lr_rect[16]) {
  float R_0r[9], t_0r[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R_0r[i * 3 + j] = T_lr[i * 4 + j];
    t_0r[i] = T_lr[i * 4 + 3];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      R_l0[i * 3 + j] = i == j ? 1.0f : 0.0f;
      R_r0[i * 3 + j] = R_0r[j * 3 + i];
    }
  float k_n[3], i_n[3], j_n[3];
  for (int i = 0; i < 3; ++i) k_n[i] = ((i == 2 ? 1.0f : 0.0f) + R_0r[i * 3 + 2]) * 0.5f;
  normalize3(k_n);
  for (int i = 0; i < 3; ++i) i_n[i] = t_0r[i];
  normalize3(i_n);
  cross3(k_n, i_n, j_n);
  normalize3(j_n);
  cross3(i_n, j_n, k_n);
  normalize3(k_n);
  float R_0n[9];
  for (int i = 0; i < 3; ++i) {
    R_0n[i * 3 + 0] = i_n[i];
    R_0n[i * 3 + 1] = j_n[i];
    R_0n[i * 3 + 2] = k_n[i];
  }
  const float f_n = (Kl[0] + Kr[0]) * (1.0f / 2.0f);
  const float centu = (float)n_cols * 0.5f, centv = (float)n_rows * 0.5f;
  const float Kn[9] = {f_n, 0.0f, centu, 0.0f, f_n, centv, 0.0f, 0.0f, 1.0f};
  float Kn_inv[9];
  inv3(Kn, Kn_inv);
  mat3mul(R_0n, Kn_inv, M); /* P_0 = R_0n * K_rect_inv * p_n: the 3x3 product is evaluated first */
  K_rect[0] = f_n; K_rect[1] = f_n; K_rect[2] = centu; K_rect[3] = centv;
  /* T_lr_rect = [I, R_ln^T t_clcr], R_ln = R_l0 * R_0n (camera.cpp:530-535) */
  float R_ln[9];
  mat3mul(R_l0, R_0n, R_ln);
  memset(T_lr_rect, 0, sizeof(float) * 16);
  for (int i = 0; i < 3; ++i) {
    T_lr_rect[i * 4 + i] = 1.0f;
    T_lr_rect[i * 4 + 3] = dot3e(R_ln[0 * 3 + i], t_0r[0], R_ln[1 * 3 + i], t_0r[1], R_ln[2 * 3 + i], t_0r[2]);
  }
  T_lr_rect[15] = 1.0f;
}

static void distort_to_map(const float X[3], const float K[4], const float D[5], float *mu, float *mv) {
  const float k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  const float x = X[0] / X[2], y = X[1] / X[2];
  const float xx = x * x, yy = y * y, xy2 = x * y * 2.0f;
  const float r2 = xx + yy, r4 = r2 * r2, r6 = r4 * r2;
  const float r_radial = 1.0f + k1 * r2 + k2 * r4 + k3 * r6;
  const float x_dist = x * r_radial + p1 * xy2 + p2 * (r2 + 2.0f * xx);
  const float y_dist = y * r_radial + p2 * xy2 + p1 * (r2 + 2.0f * yy);
  *mu = x_dist * K[0] + K[2] - 1.0f;
  *mv = y_dist * K[1] + K[3] - 1.0f;
}

/* camera.cpp:434-527 */
void vo_ref_stereo_rectify_maps(int n_cols, int n_rows, const float Kl[4], const float Dl[5], const float Kr[4],
                                const float Dr[5], const float T_lr[16], float *map_lu, float *map_lv, float *map_ru,
                                float *map_rv, float K_rect[4], float T_lr_rect[16]) {
  float M[9], R_l0[9], R_r0[9];
  vo_ref_stereo_rectify_setup(n_cols, n_rows, Kl, Kr, T_lr, M, R_l0, R_r0, K_rect, T_lr_rect);
  for (int v = 0; v < n_rows; ++v)
    for (int u = 0; u < n_cols; ++u) {
      const float pn[3] = {(float)(u + 1), (float)(v + 1), 1.0f};
      float P0[3], xl[3], xr[3];
      for (int i = 0; i < 3; ++i) P0[i] = dot3e(M[i * 3], pn[0], M[i * 3 + 1], pn[1], M[i * 3 + 2], pn[2]);
      for (int i = 0; i < 3; ++i) {
        xl[i] = dot3e(R_l0[i * 3], P0[0], R_l0[i * 3 + 1], P0[1], R_l0[i * 3 + 2], P0[2]);
        xr[i] = dot3e(R_r0[i * 3], P0[0], R_r0[i * 3 + 1], P0[1], R_r0[i * 3 + 2], P0[2]);
      }
      const size_t o = (size_t)v * n_cols + u;
      distort_to_map(xl, Kl, Dl, &map_lu[o], &map_lv[o]);
      distort_to_map(xr, Kr, Dr, &map_ru[o], &map_rv[o]);
    }
}

/* convertTo(CV_32FC1) -> cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) -> convertTo(CV_8UC1) */
void vo_ref_remap_linear_u8(const uint8_t *src, int w, int h, int stride, const float *map_u, const float *map_v,
                            int dw, int dh, uint8_t *dst) {
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      const size_t o = (size_t)y * dw + x;
      const long fxq = lrintf(map_u[o] * 32.0f), fyq = lrintf(map_v[o] * 32.0f); /* cvRound: round half to even */
      long sx = fxq >> 5, sy = fyq >> 5;
      if (sx > 32767) sx = 32767; /* saturate_cast<short> */
      if (sx < -32768) sx = -32768;
      if (sy > 32767) sy = 32767;
      if (sy < -32768) sy = -32768;
      const float ax = (float)(fxq & 31) * (1.0f / 32.0f), ay = (float)(fyq & 31) * (1.0f / 32.0f);
      const float w0 = (1.0f - ay) * (1.0f - ax), w1 = (1.0f - ay) * ax, w2 = ay * (1.0f - ax), w3 = ay * ax;
      float val = 0.0f;
      if (!(sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0)) {
        const float s00 = (sx >= 0 && sx < w && sy >= 0 && sy < h) ? (float)src[(size_t)sy * stride + sx] : 0.0f;
        const float s01 = (sx + 1 >= 0 && sx + 1 < w && sy >= 0 && sy < h) ? (float)src[(size_t)sy * stride + sx + 1] : 0.0f;
        const float s10 = (sx >= 0 && sx < w && sy + 1 >= 0 && sy + 1 < h) ? (float)src[(size_t)(sy + 1) * stride + sx] : 0.0f;
        const float s11 = (sx + 1 >= 0 && sx + 1 < w && sy + 1 >= 0 && sy + 1 < h) ? (float)src[(size_t)(sy + 1) * stride + sx + 1] : 0.0f;
        val = s00 * w0 + s01 * w1 + s10 * w2 + s11 * w3;
      }
      long r = lrintf(val);
      dst[o] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}
