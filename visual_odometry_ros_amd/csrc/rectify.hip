// rectify.hip — undistortion / stereo-rectification maps and the image ingestion that uses them
// (SURVEY.md §8f #4). Reference: core/visual_odometry/camera.cpp
//   Camera::generateImageUndistortMaps                              :56-90   -> undistort_map_kernel
//   StereoCamera::generateStereoImagesUndistortAndRectifyMaps      :364-546  -> rectify_map_kernel (+ the
//       pixel-independent frame algebra on the host, rectify_setup_host)
//   Camera::undistortImage / StereoCamera::rectifyStereoImages     :166-183, :300-336 -> remap fused into
//       the pyramid's level-0 build (pyramid.hip: remap_level0_kernel)
// The maps are per-pixel closed forms (one thread per pixel, two or four coalesced float stores); they
// are built once per camera model and stay in HBM (8 B per pixel and camera).
#include "vo_internal.hpp"
#include "vo_kernels.hpp"

#define H2D(dst, src, bytes) VO_CHECK_HIP(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, c->stream))
#define D2H(dst, src, bytes) VO_CHECK_HIP(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream))
#define SYNC() VO_CHECK_HIP(c, hipStreamSynchronize(c->stream))

struct UndistMapArgs {
  int w, h;
  float fx, fy, cx, cy, k1, k2, p1, p2, k3;
  float *mu, *mv;
};
// camera.cpp:56-90. The literals 2.0 and 1.0 there are doubles, so those sub-expressions are double
// arithmetic rounded on assignment to the float variables; reproduced term by term.
__global__ __launch_bounds__(256) void undistort_map_kernel(UndistMapArgs a) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= a.w) return;
  const float fxinv = 1.0f / a.fx, fyinv = 1.0f / a.fy;
  const float y = ((float)v - a.cy) * fyinv;
  const float x = ((float)u - a.cx) * fxinv;
  const float xy2 = (float)((2.0 * (double)x) * (double)y);
  const float xx = x * x, yy = y * y;
  const float r2 = xx + yy;
  const float r4 = r2 * r2;
  const float r6 = r4 * r2;
  const float r_radial = (float)(((1.0 + (double)(a.k1 * r2)) + (double)(a.k2 * r4)) + (double)(a.k3 * r6));
  const float x_dist = (float)((double)(x * r_radial + a.p1 * xy2) + (double)a.p2 * ((double)r2 + 2.0 * (double)xx));
  const float y_dist =
      (float)(((double)(y * r_radial) + (double)a.p1 * ((double)r2 + 2.0 * (double)yy)) + (double)(a.p2 * xy2));
  const size_t o = (size_t)v * a.w + u;
  a.mu[o] = a.cx + x_dist * a.fx;
  a.mv[o] = a.cy + y_dist * a.fy;
}

struct RectMapArgs {
  int w, h;
  float M[9], R_l0[9], R_r0[9];  // row-major
  float Kl[4], Dl[5], Kr[4], Dr[5];
  float *lu, *lv, *ru, *rv;
};
__device__ __forceinline__ float rect_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);  // Eigen's unrolled 3-term redux
}
// camera.cpp:476-526 for one camera: normalise, distort, to (0-based) pixel
__device__ __forceinline__ void rect_distort(const float X[3], const float K[4], const float D[5], float &mu, float &mv) {
  const float k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  const float x = X[0] / X[2], y = X[1] / X[2];
  const float xx = x * x, yy = y * y, xy2 = x * y * 2.0f;
  const float r2 = xx + yy, r4 = r2 * r2, r6 = r4 * r2;
  const float r_radial = 1.0f + k1 * r2 + k2 * r4 + k3 * r6;
  const float x_dist = x * r_radial + p1 * xy2 + p2 * (r2 + 2.0f * xx);
  const float y_dist = y * r_radial + p2 * xy2 + p1 * (r2 + 2.0f * yy);
  mu = x_dist * K[0] + K[2] - 1.0f;
  mv = y_dist * K[1] + K[3] - 1.0f;
}
__global__ __launch_bounds__(256) void rectify_map_kernel(RectMapArgs a) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
  if (u >= a.w) return;
  const float pn[3] = {(float)(u + 1), (float)(v + 1), 1.0f};  // camera.cpp:459: 1-based pixel
  float P0[3], xl[3], xr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) P0[i] = rect_dot3(a.M[i * 3], pn[0], a.M[i * 3 + 1], pn[1], a.M[i * 3 + 2], pn[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xl[i] = rect_dot3(a.R_l0[i * 3], P0[0], a.R_l0[i * 3 + 1], P0[1], a.R_l0[i * 3 + 2], P0[2]);
    xr[i] = rect_dot3(a.R_r0[i * 3], P0[0], a.R_r0[i * 3 + 1], P0[1], a.R_r0[i * 3 + 2], P0[2]);
  }
  const size_t o = (size_t)v * a.w + u;
  float mu, mv;
  rect_distort(xl, a.Kl, a.Dl, mu, mv);
  a.lu[o] = mu;
  a.lv[o] = mv;
  rect_distort(xr, a.Kr, a.Dr, mu, mv);
  a.ru[o] = mu;
  a.rv[o] = mv;
}

// ---- host side ---------------------------------------------------------------------
// the pixel-independent algebra of camera.cpp:364-432 and :530-535, in Eigen's evaluation order
static float h_dot3(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }
static void h_normalize3(float v[3]) {
  const float n = sqrtf(h_dot3(v[0], v[0], v[1], v[1], v[2], v[2]));
  v[0] /= n;
  v[1] /= n;
  v[2] /= n;
}
static void h_cross3(const float a[3], const float b[3], float c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void h_mat3mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = h_dot3(A[i * 3], B[j], A[i * 3 + 1], B[3 + j], A[i * 3 + 2], B[6 + j]);
}
static float h_cof3(const float m[9], int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void h_inv3(const float m[9], float r[9]) {  // Eigen's fixed-size 3x3 inverse (cofactors, column-0 determinant)
  const float c0 = h_cof3(m, 0, 0), c1 = h_cof3(m, 1, 0), c2 = h_cof3(m, 2, 0);
  const float id = 1.0f / h_dot3(c0, m[0], c1, m[3], c2, m[6]);
  r[0] = c0 * id;
  r[1] = c1 * id;
  r[2] = c2 * id;
  r[3] = h_cof3(m, 0, 1) * id;
  r[4] = h_cof3(m, 1, 1) * id;
  r[5] = h_cof3(m, 2, 1) * id;
  r[6] = h_cof3(m, 0, 2) * id;
  r[7] = h_cof3(m, 1, 2) * id;
  r[8] = h_cof3(m, 2, 2) * id;
}
static void rectify_setup_host(int w, int h, const float Kl[4], const float Kr[4], const float T_lr[16], RectMapArgs &a,
                               float K_rect[4], float T_lr_rect[16]) {
  float R_0r[9], t_0r[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R_0r[i * 3 + j] = T_lr[i * 4 + j];
    t_0r[i] = T_lr[i * 4 + 3];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a.R_l0[i * 3 + j] = i == j ? 1.0f : 0.0f;
      a.R_r0[i * 3 + j] = R_0r[j * 3 + i];
    }
  float k_n[3], i_n[3], j_n[3];
  for (int i = 0; i < 3; ++i) k_n[i] = ((i == 2 ? 1.0f : 0.0f) + R_0r[i * 3 + 2]) * 0.5f;  // :385-388
  h_normalize3(k_n);
  for (int i = 0; i < 3; ++i) i_n[i] = t_0r[i];
  h_normalize3(i_n);
  h_cross3(k_n, i_n, j_n);
  h_normalize3(j_n);
  h_cross3(i_n, j_n, k_n);
  h_normalize3(k_n);
  float R_0n[9];
  for (int i = 0; i < 3; ++i) {
    R_0n[i * 3 + 0] = i_n[i];
    R_0n[i * 3 + 1] = j_n[i];
    R_0n[i * 3 + 2] = k_n[i];
  }
  const float f_n = (Kl[0] + Kr[0]) * (1.0f / 2.0f);  // :407-408
  const float centu = (float)w * 0.5f, centv = (float)h * 0.5f;
  const float Kn[9] = {f_n, 0.0f, centu, 0.0f, f_n, centv, 0.0f, 0.0f, 1.0f};
  float Kn_inv[9];
  h_inv3(Kn, Kn_inv);
  h_mat3mul(R_0n, Kn_inv, a.M);  // :460: R_0n * K_rect_inv is evaluated before the product with p_n
  K_rect[0] = f_n;
  K_rect[1] = f_n;
  K_rect[2] = centu;
  K_rect[3] = centv;
  float R_ln[9];
  h_mat3mul(a.R_l0, R_0n, R_ln);  // :530
  memset(T_lr_rect, 0, sizeof(float) * 16);
  for (int i = 0; i < 3; ++i) {
    T_lr_rect[i * 4 + i] = 1.0f;
    T_lr_rect[i * 4 + 3] = h_dot3(R_ln[0 * 3 + i], t_0r[0], R_ln[1 * 3 + i], t_0r[1], R_ln[2 * 3 + i], t_0r[2]);
  }
  T_lr_rect[15] = 1.0f;
}

static int rect_alloc(vo_ctx *c, int cam, int w, int h) {
  if (cam < 0 || cam > 1) VO_FAIL(c, VO_ERR_INVALID, "camera index must be 0 (left / mono) or 1 (right)");
  if (w <= 0 || h <= 0 || w > c->cfg.max_width || h > c->cfg.max_height)
    VO_FAIL(c, VO_ERR_CAPACITY, "map %dx%d exceeds vo_config %dx%d", w, h, c->cfg.max_width, c->cfg.max_height);
  if (!c->rect_u[cam]) {
    const size_t bytes = sizeof(float) * (size_t)c->cfg.max_width * c->cfg.max_height;
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&c->rect_u[cam], bytes));
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&c->rect_v[cam], bytes));
  }
  c->rect_w[cam] = w;
  c->rect_h[cam] = h;
  return VO_OK;
}

void vo_rectify_free(vo_ctx *c) {
  for (int k = 0; k < 2; ++k) {
    if (c->rect_u[k]) (void)hipFree(c->rect_u[k]);
    if (c->rect_v[k]) (void)hipFree(c->rect_v[k]);
    c->rect_u[k] = c->rect_v[k] = nullptr;
  }
}

extern "C" int vo_rectify_init_mono(vo_ctx *c, int cam, int width, int height, const float K[4], const float D[5]) {
  if (!c || !K || !D) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = rect_alloc(c, cam, width, height);
  if (rc) return rc;
  UndistMapArgs a;
  a.w = width;
  a.h = height;
  a.fx = K[0];
  a.fy = K[1];
  a.cx = K[2];
  a.cy = K[3];
  a.k1 = D[0];
  a.k2 = D[1];
  a.p1 = D[2];
  a.p2 = D[3];
  a.k3 = D[4];
  a.mu = c->rect_u[cam];
  a.mv = c->rect_v[cam];
  hipLaunchKernelGGL(undistort_map_kernel, dim3((width + 255) / 256, height), dim3(256), 0, c->stream, a);
  VO_CHECK_HIP(c, hipGetLastError());
  SYNC();
  return VO_OK;
}

extern "C" int vo_rectify_init_stereo(vo_ctx *c, int width, int height, const float Kl[4], const float Dl[5],
                                      const float Kr[4], const float Dr[5], const float T_lr[16], float K_rect[4],
                                      float T_lr_rect[16], float T_rl_rect[16]) {
  if (!c || !Kl || !Dl || !Kr || !Dr || !T_lr || !K_rect || !T_lr_rect) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = rect_alloc(c, 0, width, height);
  if (rc) return rc;
  rc = rect_alloc(c, 1, width, height);
  if (rc) return rc;
  RectMapArgs a;
  memset(&a, 0, sizeof(a));
  a.w = width;
  a.h = height;
  rectify_setup_host(width, height, Kl, Kr, T_lr, a, K_rect, T_lr_rect);
  if (T_rl_rect) {  // :535: [I, -R_ln^T t]
    memcpy(T_rl_rect, T_lr_rect, sizeof(float) * 16);
    for (int i = 0; i < 3; ++i) T_rl_rect[i * 4 + 3] = -T_lr_rect[i * 4 + 3];
  }
  memcpy(a.Kl, Kl, sizeof(a.Kl));
  memcpy(a.Dl, Dl, sizeof(a.Dl));
  memcpy(a.Kr, Kr, sizeof(a.Kr));
  memcpy(a.Dr, Dr, sizeof(a.Dr));
  a.lu = c->rect_u[0];
  a.lv = c->rect_v[0];
  a.ru = c->rect_u[1];
  a.rv = c->rect_v[1];
  hipLaunchKernelGGL(rectify_map_kernel, dim3((width + 255) / 256, height), dim3(256), 0, c->stream, a);
  VO_CHECK_HIP(c, hipGetLastError());
  SYNC();
  return VO_OK;
}

extern "C" int vo_rectify_set_maps(vo_ctx *c, int cam, const float *map_u, const float *map_v, int width, int height) {
  if (!c || !map_u || !map_v) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = rect_alloc(c, cam, width, height);
  if (rc) return rc;
  SYNC();
  VO_CHECK_HIP(c, hipMemcpy(c->rect_u[cam], map_u, sizeof(float) * (size_t)width * height, hipMemcpyHostToDevice));
  VO_CHECK_HIP(c, hipMemcpy(c->rect_v[cam], map_v, sizeof(float) * (size_t)width * height, hipMemcpyHostToDevice));
  return VO_OK;
}

extern "C" int vo_rectify_get_maps(vo_ctx *c, int cam, float *map_u, float *map_v, int *width, int *height) {
  if (!c || cam < 0 || cam > 1) return VO_ERR_INVALID;
  if (!c->rect_u[cam]) VO_FAIL(c, VO_ERR_INVALID, "no rectification map for camera %d", cam);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  SYNC();
  const size_t bytes = sizeof(float) * (size_t)c->rect_w[cam] * c->rect_h[cam];
  if (map_u) VO_CHECK_HIP(c, hipMemcpy(map_u, c->rect_u[cam], bytes, hipMemcpyDeviceToHost));
  if (map_v) VO_CHECK_HIP(c, hipMemcpy(map_v, c->rect_v[cam], bytes, hipMemcpyDeviceToHost));
  if (width) *width = c->rect_w[cam];
  if (height) *height = c->rect_h[cam];
  return VO_OK;
}

// ---- image ingestion through the maps ---------------------------------------------------
extern "C" int vo_set_image_rectified_device(vo_ctx *c, int slot, const void *dev, int width, int height, int stride,
                                             int cam) {
  if (!c || !dev) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  return vo_pyramid_build_rectified(c, slot, (const uint8_t *)dev, width, height, stride, cam);
}

extern "C" int vo_set_stereo_pair_rectified_device(vo_ctx *c, int slot_l, const void *dev_l, int slot_r,
                                                   const void *dev_r, int width, int height, int stride) {
  if (!c || !dev_l || !dev_r) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  return vo_pyramid_build_pair_rectified(c, slot_l, (const uint8_t *)dev_l, slot_r, (const uint8_t *)dev_r, width,
                                         height, stride);
}

extern "C" int vo_set_image_rectified(vo_ctx *c, int slot, const uint8_t *host, int width, int height, int stride,
                                      int cam) {
  if (!c || !host) return VO_ERR_INVALID;
  if (width <= 0 || height <= 0 || width > c->cfg.max_width || height > c->cfg.max_height)
    VO_FAIL(c, VO_ERR_CAPACITY, "image %dx%d exceeds vo_config %dx%d", width, height, c->cfg.max_width,
            c->cfg.max_height);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_ingest_scope ingest(c);
  SYNC();  // the previous use of the staging buffers must have drained
  for (int y = 0; y < height; ++y) memcpy(c->h_stage + (size_t)y * width, host + (size_t)y * stride, (size_t)width);
  H2D(c->d_img_stage, c->h_stage, (size_t)width * height);
  int rc = vo_pyramid_build_rectified(c, slot, c->d_img_stage, width, height, width, cam);
  if (rc) return rc;
  SYNC();
  return VO_OK;
}
