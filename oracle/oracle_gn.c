/*
 * oracle_gn.c — CPU restatement of the pose-only Gauss-Newton motion estimator.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h). PARITY UNPINNED (no reference
 * fixtures exist); follows, line by line:
 *   core/visual_odometry/motion_estimator.cpp:665-861   (mono)
 *   core/visual_odometry/motion_estimator.cpp:863-1088  (stereo)
 *   core/visual_odometry/motion_estimator.cpp:1342-1576 (calcJtJ_x/_y, calcJtWJ_x/_y)
 *   core/util/geometry_library.cpp:386-440 (se3Exp_f), :554-560 (inverseSE3_f)
 *   standalone/motion_estimator/motion_estimator.cpp:4-411 (same math, scalar K)
 * Eigen's LDLT (6x6, robust Cholesky with symmetric pivoting on the largest
 * remaining |diagonal|, lower-triangular storage) is restated from its
 * published algorithm (Eigen 3.x src/Cholesky/LDLT.h); Eigen is not in the
 * reference tree.
 *
 * Build with -ffp-contract=off: every float operation rounds once, in the
 * order written here.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* geometry_library.cpp:386-440. `sin`/`cos` are called unqualified on a float:
 * with only <cmath> visible that resolves to ::sin(double), so the coefficient
 * is formed in double and rounded once when it scales the float matrix. */
void vo_ref_se3_exp(const float xi[6], float T[16]) {
  float v[3] = {xi[0], xi[1], xi[2]};
  float w[3] = {xi[3], xi[4], xi[5]};
  float theta = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  float wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  float wx2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0.0f;
      for (int k = 0; k < 3; ++k) s += wx[i * 3 + k] * wx[k * 3 + j];
      wx2[i * 3 + j] = s;
    }
  float R[9], V[9];
  float a, b, c; /* R = I + a wx + b wx^2 ; V = I + b' wx + c wx^2 */
  float bV;
  if (theta < 1e-7) {
    a = 1.0f;
    b = 0.5f;
    bV = 0.5f;
    c = 0.33333333333333333333333333f;
  } else {
    double th = (double)theta;
    a = (float)(sin(th) / th);
    b = (float)((1 - cos(th)) / (double)(theta * theta));
    bV = b;
    c = (float)((th - sin(th)) / (double)(theta * theta * theta));
  }
  for (int i = 0; i < 9; ++i) {
    float I = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
    R[i] = (I + a * wx[i]) + b * wx2[i];
    V[i] = (I + bV * wx[i]) + c * wx2[i];
  }
  float t[3];
  for (int i = 0; i < 3; ++i)
    t[i] = (V[i * 3 + 0] * v[0] + V[i * 3 + 1] * v[1]) + V[i * 3 + 2] * v[2];
  memset(T, 0, 16 * sizeof(float));
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
    T[i * 4 + 3] = t[i];
  }
  T[15] = 1.0f;
}

/* geometry_library.cpp:554-560: Tinv = [R^T, -R^T t; 0 0 0 1]. */
void vo_ref_inverse_se3(const float T[16], float Tinv[16]) {
  float Rt[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = T[j * 4 + i];
  float t[3] = {T[3], T[7], T[11]};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Tinv[i * 4 + j] = Rt[i * 3 + j];
    Tinv[i * 4 + 3] =
        ((-Rt[i * 3 + 0]) * t[0] + (-Rt[i * 3 + 1]) * t[1]) + (-Rt[i * 3 + 2]) * t[2];
  }
  Tinv[12] = 0;
  Tinv[13] = 0;
  Tinv[14] = 0;
  Tinv[15] = 1;
}

/* General 4x4 inverse by cofactors (motion_estimator.cpp:700 and
 * feature_tracker.cpp:215 call Eigen's Matrix4f::inverse()). */
void vo_ref_inverse4x4(const float m[16], float inv[16]) {
  float s0 = m[0] * m[5] - m[4] * m[1];
  float s1 = m[0] * m[6] - m[4] * m[2];
  float s2 = m[0] * m[7] - m[4] * m[3];
  float s3 = m[1] * m[6] - m[5] * m[2];
  float s4 = m[1] * m[7] - m[5] * m[3];
  float s5 = m[2] * m[7] - m[6] * m[3];
  float c5 = m[10] * m[15] - m[14] * m[11];
  float c4 = m[9] * m[15] - m[13] * m[11];
  float c3 = m[9] * m[14] - m[13] * m[10];
  float c2 = m[8] * m[15] - m[12] * m[11];
  float c1 = m[8] * m[14] - m[12] * m[10];
  float c0 = m[8] * m[13] - m[12] * m[9];
  float det = ((s0 * c5 - s1 * c4) + s2 * c3 + s3 * c2 - s4 * c1) + s5 * c0;
  float id = 1.0f / det;
  inv[0] = ((m[5] * c5 - m[6] * c4) + m[7] * c3) * id;
  inv[1] = ((-m[1] * c5 + m[2] * c4) - m[3] * c3) * id;
  inv[2] = ((m[13] * s5 - m[14] * s4) + m[15] * s3) * id;
  inv[3] = ((-m[9] * s5 + m[10] * s4) - m[11] * s3) * id;
  inv[4] = ((-m[4] * c5 + m[6] * c2) - m[7] * c1) * id;
  inv[5] = ((m[0] * c5 - m[2] * c2) + m[3] * c1) * id;
  inv[6] = ((-m[12] * s5 + m[14] * s2) - m[15] * s1) * id;
  inv[7] = ((m[8] * s5 - m[10] * s2) + m[11] * s1) * id;
  inv[8] = ((m[4] * c4 - m[5] * c2) + m[7] * c0) * id;
  inv[9] = ((-m[0] * c4 + m[1] * c2) - m[3] * c0) * id;
  inv[10] = ((m[12] * s4 - m[13] * s2) + m[15] * s0) * id;
  inv[11] = ((-m[8] * s4 + m[9] * s2) - m[11] * s0) * id;
  inv[12] = ((-m[4] * c3 + m[5] * c1) - m[6] * c0) * id;
  inv[13] = ((m[0] * c3 - m[1] * c1) + m[2] * c0) * id;
  inv[14] = ((-m[12] * s3 + m[13] * s1) - m[14] * s0) * id;
  inv[15] = ((m[8] * s3 - m[9] * s1) + m[10] * s0) * id;
}

/* Eigen LDLT<Matrix<float,6,6>,Lower>::compute + solve, unblocked, in place on
 * the lower triangle (motion_estimator.cpp:823,1054: JtWJ.ldlt().solve(mJtWr)).
 * Returns 0; x receives the solution. */
int vo_ref_ldlt6_solve(const float A[36], const float b[6], float x[6]) {
  enum { N = 6 };
  float m[N][N];
  int tr[N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) m[i][j] = A[i * N + j];
  float temp[N];
  for (int k = 0; k < N; ++k) {
    /* pivot: largest |diagonal| among k..N-1 (first one on ties) */
    int piv = k;
    float best = fabsf(m[k][k]);
    for (int i = k + 1; i < N; ++i) {
      float a = fabsf(m[i][i]);
      if (a > best) {
        best = a;
        piv = i;
      }
    }
    tr[k] = piv;
    if (piv != k) {
      /* symmetric row/column swap on the lower triangle */
      for (int j = 0; j < k; ++j) {
        float t = m[k][j];
        m[k][j] = m[piv][j];
        m[piv][j] = t;
      }
      for (int i = piv + 1; i < N; ++i) {
        float t = m[i][k];
        m[i][k] = m[i][piv];
        m[i][piv] = t;
      }
      {
        float t = m[k][k];
        m[k][k] = m[piv][piv];
        m[piv][piv] = t;
      }
      for (int i = k + 1; i < piv; ++i) {
        float t = m[i][k];
        m[i][k] = m[piv][i];
        m[piv][i] = t;
      }
    }
    int rs = N - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
      float s = 0.0f;
      for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
      m[k][k] -= s;
      for (int i = 0; i < rs; ++i) {
        float d = 0.0f;
        for (int j = 0; j < k; ++j) d += m[k + 1 + i][j] * temp[j];
        m[k + 1 + i][k] -= d;
      }
    }
    float akk = m[k][k];
    if (fabsf(akk) > 0.0f) {
      for (int i = 0; i < rs; ++i) m[k + 1 + i][k] /= akk;
    }
  }
  /* solve: x = P^T L^-T D^+ L^-1 P b */
  float y[N];
  for (int i = 0; i < N; ++i) y[i] = b[i];
  for (int k = 0; k < N; ++k)
    if (tr[k] != k) {
      float t = y[k];
      y[k] = y[tr[k]];
      y[tr[k]] = t;
    }
  for (int i = 0; i < N; ++i) {
    float s = y[i];
    for (int j = 0; j < i; ++j) s -= m[i][j] * y[j];
    y[i] = s;
  }
  const float tol = 1.17549435e-38f; /* std::numeric_limits<float>::min() */
  for (int i = 0; i < N; ++i) {
    if (fabsf(m[i][i]) > tol)
      y[i] /= m[i][i];
    else
      y[i] = 0.0f;
  }
  for (int i = N - 1; i >= 0; --i) {
    float s = y[i];
    for (int j = i + 1; j < N; ++j) s -= m[j][i] * y[j];
    y[i] = s;
  }
  for (int k = N - 1; k >= 0; --k)
    if (tr[k] != k) {
      float t = y[k];
      y[k] = y[tr[k]];
      y[tr[k]] = t;
    }
  for (int i = 0; i < N; ++i) x[i] = y[i];
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Accumulator: 21 upper-triangular JtWJ entries (row-major upper order
 * (0,0),(0,1)...(0,5),(1,1)...(5,5)), 6 mJtWr entries, err, cnt_invalid. */
typedef struct {
  float H[21];
  float g[6];
  float err;
  int cnt_invalid;
} gn_acc;

static const int UT[6][6] = {{0, 1, 2, 3, 4, 5},      {1, 6, 7, 8, 9, 10},
                             {2, 7, 11, 12, 13, 14},  {3, 8, 12, 15, 16, 17},
                             {4, 9, 13, 16, 18, 19},  {5, 10, 14, 17, 19, 20}};

/* calcJtWJ_x (motion_estimator.cpp:1398-1458): Jt(1)==0; products (w*Jt(a))*Jt(b). */
static void acc_row_x_w(gn_acc *A, float w, const float Jt[6]) {
  float wJ[6];
  for (int i = 0; i < 6; ++i) wJ[i] = w * Jt[i];
  static const int idx[5] = {0, 2, 3, 4, 5};
  for (int a = 0; a < 5; ++a)
    for (int b = a; b < 5; ++b) A->H[UT[idx[a]][idx[b]]] += wJ[idx[a]] * Jt[idx[b]];
}
/* calcJtJ_x (:1342-1396) */
static void acc_row_x(gn_acc *A, const float Jt[6]) {
  static const int idx[5] = {0, 2, 3, 4, 5};
  for (int a = 0; a < 5; ++a)
    for (int b = a; b < 5; ++b) A->H[UT[idx[a]][idx[b]]] += Jt[idx[a]] * Jt[idx[b]];
}
/* calcJtWJ_y (:1516-1576): Jt(0)==0 */
static void acc_row_y_w(gn_acc *A, float w, const float Jt[6]) {
  float wJ[6];
  for (int i = 0; i < 6; ++i) wJ[i] = w * Jt[i];
  for (int a = 1; a < 6; ++a)
    for (int b = a; b < 6; ++b) A->H[UT[a][b]] += wJ[a] * Jt[b];
}
/* calcJtJ_y (:1460-1514) */
static void acc_row_y(gn_acc *A, const float Jt[6]) {
  for (int a = 1; a < 6; ++a)
    for (int b = a; b < 6; ++b) A->H[UT[a][b]] += Jt[a] * Jt[b];
}
static void acc_g(gn_acc *A, float s, const float Jt[6]) {
  for (int i = 0; i < 6; ++i) A->g[i] -= s * Jt[i];
}

/* one stereo point, motion_estimator.cpp:920-1040 */
static void stereo_point(gn_acc *A, const float R10[9], const float t10[3],
                         const float Rrl[9], const float trl[3], const float *X,
                         const float *pl, const float *pr, const float Kl[4],
                         const float Kr[4], float thres, uint8_t *inlier) {
  const float THRES_HUBER = 0.5f;
  float Xl[3], Xr[3];
  for (int i = 0; i < 3; ++i)
    Xl[i] = ((R10[i * 3 + 0] * X[0] + R10[i * 3 + 1] * X[1]) + R10[i * 3 + 2] * X[2]) + t10[i];
  for (int i = 0; i < 3; ++i)
    Xr[i] = ((Rrl[i * 3 + 0] * Xl[0] + Rrl[i * 3 + 1] * Xl[1]) + Rrl[i * 3 + 2] * Xl[2]) + trl[i];
  const float fx_l = Kl[0], fy_l = Kl[1], cx_l = Kl[2], cy_l = Kl[3];
  const float fx_r = Kr[0], fy_r = Kr[1], cx_r = Kr[2], cy_r = Kr[3];

  const float iz_l = 1.0f / Xl[2];
  const float xiz_l = Xl[0] * iz_l;
  const float yiz_l = Xl[1] * iz_l;
  const float fxxiz_l = fx_l * xiz_l;
  const float fyyiz_l = fy_l * yiz_l;
  const float rx_l = (fxxiz_l + cx_l) - pl[0];
  const float ry_l = (fyyiz_l + cy_l) - pl[1];

  const float iz_r = 1.0f / Xr[2];
  const float xiz_r = Xr[0] * iz_r;
  const float yiz_r = Xr[1] * iz_r;
  const float fxxiz_r = fx_r * xiz_r;
  const float fyyiz_r = fy_r * yiz_r;
  const float rx_r = (fxxiz_r + cx_r) - pr[0];
  const float ry_r = (fyyiz_r + cy_r) - pr[1];

  float weight = 1.0f;
  float absrxry = ((fabsf(rx_l) + fabsf(ry_l)) + fabsf(rx_r)) + fabsf(ry_r);
  absrxry *= 0.5f;
  if (absrxry >= THRES_HUBER) weight = THRES_HUBER / absrxry;
  if (absrxry >= thres) {
    *inlier = 0;
    ++A->cnt_invalid;
  } else
    *inlier = 1;

  float Jt[6];
  /* Left x */
  Jt[0] = fx_l * iz_l;
  Jt[1] = 0.0f;
  Jt[2] = -fxxiz_l * iz_l;
  Jt[3] = -fxxiz_l * yiz_l;
  Jt[4] = fx_l * (1.0f + xiz_l * xiz_l);
  Jt[5] = -fx_l * yiz_l;
  acc_row_x_w(A, weight, Jt);
  acc_g(A, weight * rx_l, Jt);
  A->err += rx_l * rx_l;
  /* Left y */
  Jt[0] = 0.0f;
  Jt[1] = fy_l * iz_l;
  Jt[2] = -fyyiz_l * iz_l;
  Jt[3] = -fy_l * (1.0f + yiz_l * yiz_l);
  Jt[4] = fyyiz_l * xiz_l;
  Jt[5] = fy_l * xiz_l;
  acc_row_y_w(A, weight, Jt);
  acc_g(A, weight * ry_l, Jt);
  A->err += ry_l * ry_l;
  /* Right x — the left-camera Jacobian form evaluated at Xr (:1009-1014) */
  Jt[0] = fx_r * iz_r;
  Jt[1] = 0.0f;
  Jt[2] = -fxxiz_r * iz_r;
  Jt[3] = -fxxiz_r * yiz_r;
  Jt[4] = fx_r * (1.0f + xiz_r * xiz_r);
  Jt[5] = -fx_r * yiz_r;
  acc_row_x_w(A, weight, Jt);
  acc_g(A, weight * rx_r, Jt);
  A->err += rx_r * rx_r;
  /* Right y */
  Jt[0] = 0.0f;
  Jt[1] = fy_r * iz_r;
  Jt[2] = -fyyiz_r * iz_r;
  Jt[3] = -fy_r * (1.0f + yiz_r * yiz_r);
  Jt[4] = fyyiz_r * xiz_r;
  Jt[5] = fy_r * xiz_r;
  acc_row_y_w(A, weight, Jt);
  acc_g(A, weight * ry_r, Jt);
  A->err += ry_r * ry_r;
}

/* one mono point, motion_estimator.cpp:713-810 */
static void mono_point(gn_acc *A, const float R10[9], const float t10[3],
                       const float *X, const float *pt, const float K[4],
                       float thres, int variant, uint8_t *inlier) {
  const float THRES_HUBER = 0.5f;
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  float Xw[3];
  for (int i = 0; i < 3; ++i)
    Xw[i] = ((R10[i * 3 + 0] * X[0] + R10[i * 3 + 1] * X[1]) + R10[i * 3 + 2] * X[2]) + t10[i];
  float iz = 1.0f / Xw[2];
  float xiz = Xw[0] * iz;
  float yiz = Xw[1] * iz;
  float fxxiz = fx * xiz;
  float fyyiz = fy * yiz;
  float rx = (fxxiz + cx) - pt[0];
  float ry = (fyyiz + cy) - pt[1];

  float weight = 1.0f;
  int flag_weight = 0;
  float absrxry = fabsf(rx) + fabsf(ry);
  if (absrxry >= THRES_HUBER) {
    weight = THRES_HUBER / absrxry;
    flag_weight = 1;
  }
  if (absrxry >= thres) {
    *inlier = 0;
    ++A->cnt_invalid;
  } else
    *inlier = 1;

  float Jt[6];
  Jt[0] = fx * iz;
  Jt[1] = 0.0f;
  Jt[2] = -fxxiz * iz;
  Jt[3] = -fxxiz * yiz;
  Jt[4] = fx * (1.0f + xiz * xiz);
  Jt[5] = -fx * yiz;
  if (flag_weight) {
    float w_rx = weight * rx;
    acc_row_x_w(A, weight, Jt);
    acc_g(A, w_rx, Jt);
    A->err += rx * rx; /* :771 adds the UNweighted square */
  } else {
    acc_row_x(A, Jt);
    acc_g(A, rx, Jt);
    A->err += rx * rx;
  }
  Jt[0] = 0.0f;
  Jt[1] = fy * iz;
  Jt[2] = -fyyiz * iz;
  Jt[3] = -fy * (1.0f + yiz * yiz);
  Jt[4] = fyyiz * xiz;
  Jt[5] = fy * xiz;
  if (flag_weight) {
    float w_ry = weight * ry;
    acc_row_y_w(A, weight, Jt);
    acc_g(A, w_ry, Jt);
    /* core :793-799 adds w*ry^2; standalone :135 adds ry^2 */
    if (variant == VO_GN_VARIANT_CORE)
      A->err += w_ry * ry;
    else
      A->err += ry * ry;
  } else {
    acc_row_y(A, Jt);
    acc_g(A, ry, Jt);
    A->err += ry * ry;
  }
}

/* balanced binary tree over `n` partial floats, natural order, adjacent first */
static float tree_sum(const float *v, int lo, int hi) {
  if (hi - lo == 1) return v[lo];
  int mid = lo + (hi - lo) / 2;
  return tree_sum(v, lo, mid) + tree_sum(v, mid, hi);
}

static void reduce_tree(const gn_acc *parts, int T, gn_acc *out) {
  float *col = (float *)malloc(sizeof(float) * (size_t)T);
  for (int k = 0; k < 21; ++k) {
    for (int t = 0; t < T; ++t) col[t] = parts[t].H[k];
    out->H[k] = tree_sum(col, 0, T);
  }
  for (int k = 0; k < 6; ++k) {
    for (int t = 0; t < T; ++t) col[t] = parts[t].g[k];
    out->g[k] = tree_sum(col, 0, T);
  }
  for (int t = 0; t < T; ++t) col[t] = parts[t].err;
  out->err = tree_sum(col, 0, T);
  out->cnt_invalid = 0;
  for (int t = 0; t < T; ++t) out->cnt_invalid += parts[t].cnt_invalid;
  free(col);
}

static void expand_H(const gn_acc *A, float JtWJ[36]) {
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) JtWJ[i * 6 + j] = A->H[UT[i][j]];
}

static void matmul4(const float A[16], const float B[16], float C[16]) {
  float R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = A[i * 4 + 0] * B[0 * 4 + j];
      s += A[i * 4 + 1] * B[1 * 4 + j];
      s += A[i * 4 + 2] * B[2 * 4 + j];
      s += A[i * 4 + 3] * B[3 * 4 + j];
      R[i * 4 + j] = s;
    }
  memcpy(C, R, sizeof(R));
}

static float norm16(const float T[16]) {
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += T[i] * T[i];
  return sqrtf(s);
}

static int is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

/* shared GN driver; stereo != 0 selects :863-1088, else :665-861 */
static int gn_run(int stereo, const float *X, const float *p1, const float *p2,
                  int n, const float Kl[4], const float Kr[4],
                  const float T_rl[16], float thres, float T10[16],
                  uint8_t *mask, int variant, int sum_mode, int T,
                  vo_ref_gn_info *info) {
  const int MAX_ITER = 100;
  const float THRES_DELTA_XI = 1e-6;
  const float THRES_DELTA_ERROR = 1e-7;
  const float lambda = 0.00001f;
  float err_prev = 1e10f;
  gn_acc *parts = NULL;
  if (sum_mode == VO_SUM_TREE) {
    if (!is_pow2(T)) return -2;
    parts = (gn_acc *)malloc(sizeof(gn_acc) * (size_t)T);
  }
  float Rrl[9], trl[3];
  if (stereo) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Rrl[i * 3 + j] = T_rl[i * 4 + j];
      trl[i] = T_rl[i * 4 + 3];
    }
  }
  int iter = 0;
  float err_curr = 0, delta_err = 0, dnorm = 0;
  int cnt_invalid = 0;
  for (iter = 0; iter < MAX_ITER; ++iter) {
    float R10[9], t10[3];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R10[i * 3 + j] = T10[i * 4 + j];
      t10[i] = T10[i * 4 + 3];
    }
    gn_acc acc;
    memset(&acc, 0, sizeof(acc));
    float inv_npts = 1.0f / (float)n;
    if (sum_mode == VO_SUM_SEQ) {
      for (int i = 0; i < n; ++i) {
        if (stereo)
          stereo_point(&acc, R10, t10, Rrl, trl, X + 3 * i, p1 + 2 * i, p2 + 2 * i, Kl, Kr, thres, mask + i);
        else
          mono_point(&acc, R10, t10, X + 3 * i, p1 + 2 * i, Kl, thres, variant, mask + i);
      }
    } else {
      memset(parts, 0, sizeof(gn_acc) * (size_t)T);
      for (int t = 0; t < T; ++t)
        for (int i = t; i < n; i += T) {
          if (stereo)
            stereo_point(&parts[t], R10, t10, Rrl, trl, X + 3 * i, p1 + 2 * i, p2 + 2 * i, Kl, Kr, thres, mask + i);
          else
            mono_point(&parts[t], R10, t10, X + 3 * i, p1 + 2 * i, Kl, thres, variant, mask + i);
        }
      reduce_tree(parts, T, &acc);
    }
    cnt_invalid = acc.cnt_invalid;
    err_curr = acc.err;
    err_curr *= (inv_npts * 0.5f);
    if (stereo) err_curr = sqrtf(err_curr); /* :1043 ; mono :812 has no sqrt */
    delta_err = fabsf(err_curr - err_prev);

    float JtWJ[36];
    expand_H(&acc, JtWJ);
    for (int k = 0; k < 6; ++k) JtWJ[k * 6 + k] *= (1.0f + lambda);
    float dxi[6];
    vo_ref_ldlt6_solve(JtWJ, acc.g, dxi);
    float dT[16];
    vo_ref_se3_exp(dxi, dT);
    matmul4(dT, T10, T10);
    err_prev = err_curr;
    float s = 0.0f;
    for (int k = 0; k < 6; ++k) s += dxi[k] * dxi[k];
    dnorm = sqrtf(s);
    if (dnorm < THRES_DELTA_XI || delta_err < THRES_DELTA_ERROR) {
      ++iter;
      break;
    }
  }
  if (info) {
    info->iterations = iter;
    info->err = err_curr;
    info->delta_err = delta_err;
    info->delta_norm = dnorm;
    info->cnt_invalid = cnt_invalid;
    info->is_nan = isnan(norm16(T10)) ? 1 : 0;
  }
  free(parts);
  return 0;
}

/* motion_estimator.cpp:665-861. Returns 1 on success (reference `true`), 0 if
 * the pose went NaN (pose left untouched), <0 on precondition violation. */
int vo_ref_gn_pose_mono(const float *X, const float *pts1, int n,
                        const float K[4], int thres_reproj_outlier,
                        float R01[9], float t01[3], uint8_t *mask_inlier,
                        int variant, int sum_mode, int tree_width,
                        vo_ref_gn_info *info) {
  float T01[16], T10[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T01[i * 4 + j] = R01[i * 3 + j];
    T01[i * 4 + 3] = t01[i];
  }
  T01[12] = T01[13] = T01[14] = 0;
  T01[15] = 1;
  vo_ref_inverse4x4(T01, T10); /* :700 T01_init.inverse() */
  vo_ref_gn_info li;
  int rc = gn_run(0, X, pts1, NULL, n, K, K, NULL, (float)thres_reproj_outlier,
                  T10, mask_inlier, variant, sum_mode, tree_width, &li);
  if (rc < 0) return rc;
  if (info) *info = li;
  if (!li.is_nan) {
    float Tu[16];
    vo_ref_inverse_se3(T10, Tu);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R01[i * 3 + j] = Tu[i * 4 + j];
      t01[i] = Tu[i * 4 + 3];
    }
    return 1;
  }
  return 0;
}

/* motion_estimator.cpp:863-1088 */
int vo_ref_gn_pose_stereo(const float *X, const float *pts_l1,
                          const float *pts_r1, int n, const float Kl[4],
                          const float Kr[4], const float T_lr[16],
                          float thres_reproj_outlier, float T01[16],
                          uint8_t *mask_inlier, int sum_mode, int tree_width,
                          vo_ref_gn_info *info) {
  float T_rl[16], T10[16];
  vo_ref_inverse_se3(T_lr, T_rl); /* :869 */
  for (int i = 0; i < n; ++i) mask_inlier[i] = 1; /* :878 assign(n,true) */
  vo_ref_inverse_se3(T01, T10); /* :904, same closed form */
  vo_ref_gn_info li;
  int rc = gn_run(1, X, pts_l1, pts_r1, n, Kl, Kr, T_rl, thres_reproj_outlier,
                  T10, mask_inlier, 0, sum_mode, tree_width, &li);
  if (rc < 0) return rc;
  if (info) *info = li;
  if (!li.is_nan) {
    vo_ref_inverse_se3(T10, T01);
    return 1;
  }
  return 0;
}
