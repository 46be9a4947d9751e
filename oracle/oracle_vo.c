/*
 * oracle_vo.c — what closes the loop around the stereo frame: pose chaining, DLT triangulation of new
 * landmarks, keyframe reconstruction.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h). PARITY UNPINNED.
 * Follows:
 *   core/util/triangulate_3d.cpp:91-130                     (triangulateDLT, two-camera overload)
 *   core/visual_odometry/stereo_vo/stereo_vo.cpp:475-480    (T_wc_prior = T_wp * dT_pc_prev, inverseSE3_f)
 *   core/visual_odometry/stereo_vo/stereo_vo.cpp:714-739    (new landmarks: mask_new && Xl(2) > 0 && Xr(2) > 0)
 *   core/visual_odometry/stereo_vo/stereo_vo.cpp:763-797    (keyframe: reconstruction of lmtrack_final)
 *   core/visual_odometry/stereo_vo/stereo_vo.cpp:907-941    (first frame: reconstruction)
 *   core/visual_odometry/camera.cpp:208-213                 (projectToPixel)
 *
 * Third-party arithmetic restated here (Eigen 3, not in the reference tree, pinned only as `find_package(Eigen3)`):
 *   Eigen::JacobiSVD<Eigen::MatrixXf>(M, Eigen::ComputeFullV) for a 4x4 M, as Eigen 3.4.0 computes it
 *   (Eigen/src/SVD/JacobiSVD.h: scaling by the largest |coefficient|, no QR preconditioner for a square matrix,
 *   sweeps over the pairs (p, q < p) with the threshold max(FLT_MIN, 2 eps maxDiagEntry), real_2x2_jacobi_svd
 *   (Eigen/src/misc/RealSvd2x2.h), JacobiRotation::makeJacobi / operator* / transpose (Eigen/src/Jacobi/Jacobi.h),
 *   positive singular values, selection sort in descending order with the columns of V swapped along), written from
 *   the author's knowledge of that source: no Eigen on this machine to diff against.
 *   Fixed-size products: a 3-term dot product is Eigen's unrolled redux e0 + (e1 + e2) (DefaultTraversal,
 *   CompleteUnrolling); `A * x + b` evaluates the product first and adds b; a Matrix4f * Matrix4f product is the
 *   column-packet form res = a0 b0; res = a_k b_k + res (k = 1..3), i.e. left to right — without fused multiply-adds
 *   (the repository's one reproducible reading of `-O2 -march=native`).
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* Matrix4f * Matrix4f (stereo_vo.cpp:479 T_wp * dT_pc_prev, :640 T_wp * dT_pc_poBA), row-major in and out */
void vo_ref_mul44(const float A[16], const float B[16], float C[16]) {
  float R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float r = A[i * 4 + 0] * B[0 * 4 + j];
      for (int k = 1; k < 4; ++k) r = A[i * 4 + k] * B[k * 4 + j] + r;
      R[i * 4 + j] = r;
    }
  memcpy(C, R, sizeof(R));
}

static inline float dot3e(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2); /* Eigen's unrolled 3-term redux */
}

/* T.block<3,3>(0,0) * X + T.block<3,1>(0,3) (stereo_vo.cpp:493-497, :605, :793) */
void vo_ref_xform_eig(const float T[16], const float X[3], float Y[3]) {
  float r[3];
  for (int i = 0; i < 3; ++i) r[i] = dot3e(T[i * 4 + 0], X[0], T[i * 4 + 1], X[1], T[i * 4 + 2], X[2]) + T[i * 4 + 3];
  Y[0] = r[0];
  Y[1] = r[1];
  Y[2] = r[2];
}

typedef struct {
  float c, s;
} jrot;

/* JacobiRotation<float>::makeJacobi(x, y, z) */
static jrot make_jacobi(float x, float y, float z) {
  jrot j;
  const float deno = 2.0f * fabsf(y);
  if (deno < FLT_MIN) {
    j.c = 1.0f;
    j.s = 0.0f;
    return j;
  }
  const float tau = (x - z) / deno;
  const float w = sqrtf(tau * tau + 1.0f);
  float t;
  if (tau > 0.0f)
    t = 1.0f / (tau + w);
  else
    t = 1.0f / (tau - w);
  const float sign_t = t > 0.0f ? 1.0f : -1.0f;
  const float n = 1.0f / sqrtf(t * t + 1.0f);
  j.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
  j.c = n;
  return j;
}

/* apply_rotation_in_the_plane(x, y, j): x_i <- c x_i + s y_i ; y_i <- -s x_i + c y_i */
static void rot_apply(float *x, int sx, float *y, int sy, int n, jrot j) {
  if (j.c == 1.0f && j.s == 0.0f) return;
  for (int i = 0; i < n; ++i) {
    const float xi = x[i * sx], yi = y[i * sy];
    x[i * sx] = j.c * xi + j.s * yi;
    y[i * sy] = -j.s * xi + j.c * yi;
  }
}

/* JacobiSVD<MatrixXf>(M, ComputeFullV) of a row-major 4x4: V (row-major, columns sorted with the singular values,
 * descending) and the singular values. Returns the number of sweeps (0 when M is not finite: Eigen reports
 * InvalidInput and leaves V unset — here V = identity). */
int vo_ref_jacobi_svd4(const float M[16], float V[16], float sv[4]) {
  float W[16];
  float scale = 0.0f;
  int finite = 1;
  for (int i = 0; i < 16; ++i) {
    const float a = fabsf(M[i]);
    if (!(a <= FLT_MAX)) finite = 0;
    if (a > scale) scale = a;
  }
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  if (!finite) {
    for (int i = 0; i < 4; ++i) sv[i] = 0.0f;
    return 0;
  }
  if (scale == 0.0f) scale = 1.0f;
  for (int i = 0; i < 16; ++i) W[i] = M[i] / scale;
  const float precision = 2.0f * FLT_EPSILON;
  const float consider_as_zero = FLT_MIN;
  float max_diag = 0.0f;
  for (int i = 0; i < 4; ++i)
    if (fabsf(W[i * 5]) > max_diag) max_diag = fabsf(W[i * 5]);
  int sweeps = 0;
  for (int finished = 0; !finished;) {
    finished = 1;
    ++sweeps;
    for (int p = 1; p < 4; ++p)
      for (int q = 0; q < p; ++q) {
        const float pm = precision * max_diag;
        const float threshold = consider_as_zero > pm ? consider_as_zero : pm;
        if (fabsf(W[p * 4 + q]) > threshold || fabsf(W[q * 4 + p]) > threshold) {
          finished = 0;
          /* real_2x2_jacobi_svd(W, p, q, &j_left, &j_right) */
          float m00 = W[p * 4 + p], m01 = W[p * 4 + q], m10 = W[q * 4 + p], m11 = W[q * 4 + q];
          jrot rot1;
          const float t = m00 + m11;
          const float d = m10 - m01;
          if (fabsf(d) < FLT_MIN) {
            rot1.s = 0.0f;
            rot1.c = 1.0f;
          } else {
            const float u = t / d;
            const float tmp = sqrtf(1.0f + u * u);
            rot1.s = 1.0f / tmp;
            rot1.c = u / tmp;
          }
          { /* m.applyOnTheLeft(0, 1, rot1) */
            float r0[2] = {m00, m01}, r1[2] = {m10, m11};
            rot_apply(r0, 1, r1, 1, 2, rot1);
            m00 = r0[0];
            m01 = r0[1];
            m10 = r1[0];
            m11 = r1[1];
          }
          const jrot jr = make_jacobi(m00, m01, m11);
          /* j_left = rot1 * j_right.transpose() */
          const jrot jrt = {jr.c, -jr.s};
          jrot jl;
          jl.c = rot1.c * jrt.c - rot1.s * jrt.s;
          jl.s = rot1.c * jrt.s + rot1.s * jrt.c;
          /* m_workMatrix.applyOnTheLeft(p, q, j_left): rows p and q */
          rot_apply(&W[p * 4], 1, &W[q * 4], 1, 4, jl);
          /* applyOnTheRight(p, q, j_right): columns p and q with j_right.transpose() */
          rot_apply(&W[p], 4, &W[q], 4, 4, jrt);
          rot_apply(&V[p], 4, &V[q], 4, 4, jrt);
          const float a = fabsf(W[p * 4 + p]), b = fabsf(W[q * 4 + q]);
          const float mx = a > b ? a : b;
          if (mx > max_diag) max_diag = mx;
        }
      }
    if (sweeps > 1000) break; /* (Eigen has no bound; NaN cannot loop because the comparisons above are false for it) */
  }
  for (int i = 0; i < 4; ++i) sv[i] = fabsf(W[i * 5]); /* (the sign goes into U, which is not computed) */
  for (int i = 0; i < 4; ++i) sv[i] *= scale;
  for (int i = 0; i < 4; ++i) {
    int pos = i;
    float best = sv[i];
    for (int k = i + 1; k < 4; ++k)
      if (sv[k] > best) { /* maxCoeff(&pos): the first of equal maxima */
        best = sv[k];
        pos = k;
      }
    if (best == 0.0f) break;
    if (pos != i) {
      const float ts = sv[i];
      sv[i] = sv[pos];
      sv[pos] = ts;
      for (int r = 0; r < 4; ++r) {
        const float tv = V[r * 4 + i];
        V[r * 4 + i] = V[r * 4 + pos];
        V[r * 4 + pos] = tv;
      }
    }
  }
  return sweeps;
}

/* P10 = [K1 * R10, K1 * t10] (triangulate_3d.cpp:104), K1 = cam1->K() as a Matrix3f, row-major 3x4 out */
void vo_ref_dlt_projection(const float K1[4], const float R10[9], const float t10[3], float P10[12]) {
  const float Km[9] = {K1[0], 0.0f, K1[2], 0.0f, K1[1], K1[3], 0.0f, 0.0f, 1.0f};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      P10[i * 4 + j] = dot3e(Km[i * 3 + 0], R10[0 * 3 + j], Km[i * 3 + 1], R10[1 * 3 + j], Km[i * 3 + 2], R10[2 * 3 + j]);
    P10[i * 4 + 3] = dot3e(Km[i * 3 + 0], t10[0], Km[i * 3 + 1], t10[1], Km[i * 3 + 2], t10[2]);
  }
}

/* mapping::triangulateDLT(pt0, pt1, R10, t10, cam0, cam1, X0, X1), triangulate_3d.cpp:91-130.
 * K0 / K1 = (fx, fy, cx, cy). Returns the number of Jacobi sweeps. */
int vo_ref_triangulate_dlt(const float pt0[2], const float pt1[2], const float R10[9], const float t10[3],
                           const float K0[4], const float K1[4], float X0[3], float X1[3]) {
  float P10[12], M[16];
  vo_ref_dlt_projection(K1, R10, t10, P10);
  memset(M, 0, sizeof(M));
  M[0] = -K0[0];
  M[5] = -K0[1];
  M[2] = pt0[0] - K0[2];
  M[6] = pt0[1] - K0[3];
  for (int c = 0; c < 4; ++c) {
    M[8 + c] = pt1[0] * P10[8 + c] - P10[0 + c];
    M[12 + c] = pt1[1] * P10[8 + c] - P10[4 + c];
  }
  float V[16], sv[4];
  const int sweeps = vo_ref_jacobi_svd4(M, V, sv);
  const float w = V[3 * 4 + 3];
  X0[0] = V[0 * 4 + 3] / w;
  X0[1] = V[1 * 4 + 3] / w;
  X0[2] = V[2 * 4 + 3] / w;
  for (int i = 0; i < 3; ++i)
    X1[i] = dot3e(R10[i * 3 + 0], X0[0], R10[i * 3 + 1], X0[1], R10[i * 3 + 2], X0[2]) + t10[i];
  return sweeps;
}

static void project_px(const float K[4], const float X[3], float *px, float *py) { /* camera.cpp:208-213 */
  const float invz = 1.0f / X[2];
  *px = K[0] * X[0] * invz + K[2];
  *py = K[1] * X[1] * invz + K[3];
}

/* Step [10]'s landmark test for n candidates (stereo_vo.cpp:714-739): accept[i] = mask_new[i] && Xl(2) > 0 &&
 * Xr(2) > 0 with (Xl, Xr) = triangulateDLT(pt_l, pt_r, R_rl, t_rl). Xl (n x 3, may be NULL) receives the left-camera
 * points of the candidates that were triangulated (mask_new set). Returns the number accepted. */
int vo_ref_new_landmark_accept(const float *pts_l, const float *pts_r, const uint8_t *mask_new, int n,
                               const float T_rl[16], const float Kl[4], const float Kr[4], uint8_t *accept, float *Xl_out) {
  const float R[9] = {T_rl[0], T_rl[1], T_rl[2], T_rl[4], T_rl[5], T_rl[6], T_rl[8], T_rl[9], T_rl[10]};
  const float t[3] = {T_rl[3], T_rl[7], T_rl[11]};
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    accept[i] = 0;
    if (Xl_out) Xl_out[3 * i] = Xl_out[3 * i + 1] = Xl_out[3 * i + 2] = 0.0f;
    if (!mask_new[i]) continue;
    float Xl[3], Xr[3];
    vo_ref_triangulate_dlt(pts_l + 2 * i, pts_r + 2 * i, R, t, Kl, Kr, Xl, Xr);
    if (Xl_out) memcpy(Xl_out + 3 * i, Xl, sizeof(Xl));
    if (Xl[2] > 0 && Xr[2] > 0) {
      accept[i] = 1;
      ++cnt;
    }
  }
  return cnt;
}

/* Reconstruction at a keyframe (stereo_vo.cpp:763-797) and at the first frame (:907-941, T_wc = identity there and
 * Xworld = Xl is assigned without the transform: pass T_wc = NULL): for every feature triangulateDLT, reprojection
 * error of Xl in the left and of Xr in the right image each at most 1 px (squared, compared in double like the
 * reference's `> 1.0`), both depths positive -> set[i] = 1 and Xw[i] = T_wc.block<3,3> * Xl + T_wc.block<3,1>.
 * Entries with set[i] = 0 keep their Xw. Returns the number reconstructed. */
int vo_ref_keyframe_reconstruct(const float *pts_l, const float *pts_r, int n, const float T_rl[16], const float Kl[4],
                                const float Kr[4], const float *T_wc, float *Xw, uint8_t *set) {
  const float R[9] = {T_rl[0], T_rl[1], T_rl[2], T_rl[4], T_rl[5], T_rl[6], T_rl[8], T_rl[9], T_rl[10]};
  const float t[3] = {T_rl[3], T_rl[7], T_rl[11]};
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    set[i] = 0;
    float Xl[3], Xr[3], px, py;
    vo_ref_triangulate_dlt(pts_l + 2 * i, pts_r + 2 * i, R, t, Kl, Kr, Xl, Xr);
    project_px(Kl, Xl, &px, &py);
    float dx = pts_l[2 * i] - px, dy = pts_l[2 * i + 1] - py;
    float d2 = dx * dx + dy * dy;
    if ((double)d2 > 1.0) continue;
    project_px(Kr, Xr, &px, &py);
    dx = pts_r[2 * i] - px;
    dy = pts_r[2 * i + 1] - py;
    d2 = dx * dx + dy * dy;
    if ((double)d2 > 1.0) continue;
    if (Xl[2] > 0 && Xr[2] > 0) {
      if (T_wc)
        vo_ref_xform_eig(T_wc, Xl, Xw + 3 * i);
      else
        memcpy(Xw + 3 * i, Xl, sizeof(Xl));
      set[i] = 1;
      ++cnt;
    }
  }
  return cnt;
}

/* ---- MonoVO (mono_vo.cpp) -------------------------------------------------------------------------------------------- */

/* Landmark::addObservationAndRelatedFrame, landmark.cpp:100-121: the parallax of the newest observation p1 (seen from a
 * frame of pose T_wc_last) with respect to the oldest one p0 (frame of inverse pose T_cw_first): T01 =
 * front()->getPoseInv() * back()->getPose(); bearings through fxinv / fyinv (camera.cpp:28-29); x1 = R01 * x1 (Eigen's
 * 3-term redux); costheta = x0.dot(x1) / (|x0| |x1|), pushed inside (-1, 1) by the two tests of :113-116; acosf. */
float vo_ref_parallax(const float p0[2], const float p1[2], const float K[4], const float T_cw_first[16],
                      const float T_wc_last[16], float *cos_out) {
  float T01[16];
  vo_ref_mul44(T_cw_first, T_wc_last, T01);
  const float fxinv = 1.0f / K[0], fyinv = 1.0f / K[1];
  const float x0[3] = {(p0[0] - K[2]) * fxinv, (p0[1] - K[3]) * fyinv, 1.0f};
  const float x1[3] = {(p1[0] - K[2]) * fxinv, (p1[1] - K[3]) * fyinv, 1.0f};
  float r[3];
  for (int i = 0; i < 3; ++i) r[i] = dot3e(T01[i * 4 + 0], x1[0], T01[i * 4 + 1], x1[1], T01[i * 4 + 2], x1[2]);
  const float dot = dot3e(x0[0], r[0], x0[1], r[1], x0[2], r[2]);
  const float n0 = sqrtf(dot3e(x0[0], x0[0], x0[1], x0[1], x0[2], x0[2]));
  const float n1 = sqrtf(dot3e(r[0], r[0], r[1], r[1], r[2], r[2]));
  float c = dot / (n0 * n1);
  if (c >= 1.0f) c = 0.99999f;
  if (c <= -1.0f) c = -0.99999f;
  if (cos_out) *cos_out = c;
  return acosf(c);
}

/* MonoVO's two reconstructions of a landmark from its first and its last observation (mapping::triangulateDLT, one camera):
 *   keyframe_rule = 0  mono_vo.cpp:669-686 (initialisation): T10 = T1w * Tw0, X0(2) > 0 -> Xworld = Tw0 * X0
 *   keyframe_rule = 1  mono_vo.cpp:1041-1073 (new keyframe): additionally both reprojection errors at most 1 px (squared,
 *                      compared with the double 1.0) and X1(2) > 0
 * Returns 1 and Xw when the landmark is reconstructed. */
int vo_ref_mono_reconstruct(const float pt0[2], const float pt1[2], const float T_w0[16], const float T_1w[16],
                            const float K[4], int keyframe_rule, float Xw[3]) {
  float T10[16];
  vo_ref_mul44(T_1w, T_w0, T10);
  const float R[9] = {T10[0], T10[1], T10[2], T10[4], T10[5], T10[6], T10[8], T10[9], T10[10]};
  const float t[3] = {T10[3], T10[7], T10[11]};
  float X0[3], X1[3], px, py;
  vo_ref_triangulate_dlt(pt0, pt1, R, t, K, K, X0, X1);
  if (keyframe_rule) {
    project_px(K, X0, &px, &py);
    float dx = pt0[0] - px, dy = pt0[1] - py;
    float d2 = dx * dx + dy * dy;
    if ((double)d2 > 1.0) return 0;
    project_px(K, X1, &px, &py);
    dx = pt1[0] - px;
    dy = pt1[1] - py;
    d2 = dx * dx + dy * dy;
    if ((double)d2 > 1.0) return 0;
    if (!(X0[2] > 0 && X1[2] > 0)) return 0;
  } else if (!(X0[2] > 0)) {
    return 0;
  }
  vo_ref_xform_eig(T_w0, X0, Xw);
  return 1;
}
