/* oracle_sba.c — CPU restatement (TEST INFRASTRUCTURE, parity unpinned — see vo_oracle.h) of
 * SparseBundleAdjustmentSolver::solveForFiniteIterations
 * (core/visual_odometry/ba_solver/sparse_bundle_adjustment.cpp:150-643) on flat arrays: what
 * SparseBAParameters hands the solver (poses T_jw in the reference frame and scaled, which of them are
 * optimised, the landmarks and their observation lists) comes in as CSR lists. double throughout
 * (_BA_Numeric, define_ba_type.h:9), the reference's loop and summation order.
 * Behaviour that is easy to miss and is reproduced:
 *  - lambda is the constant 1e-5 and the iteration count is fixed (:186, :188) — no LM adaptation;
 *  - B_[j][i] is ASSIGNED, not accumulated (:315, :410): with a left and a right observation of
 *    landmark i in stereo frame j the later one in the list wins, while A_, a_, C_, b_ take both;
 *  - the Schur loops visit left observations only (:462, :482): a landmark seen only in the right image
 *    of frame j contributes to A_j / C_i but not to B C^-1 B^T;
 *  - BCinvBt_[j][k] is accumulated for list positions kk >= jj and afterwards the lower block triangle
 *    is overwritten with the transposed upper one (:495-497), diagonal blocks are transposed in place;
 *  - calc_Qij_t_Qij_weight (:986-1041) assumes Q(0,1) = Q(1,0) = 0, which holds for the left-image
 *    Jacobian only: entry (0,1)/(1,0) of Q^T Q is left zero and the (0,x)/(1,x) rows take one term;
 *  - the pose update is exp(log(exp(x) * exp(log(T)))) (:563-575, geometry_library.cpp:546-552).
 * Third-party arithmetic: Eigen's pivoted LDLT (C_[i].ldlt(), the dense reduced system :460, :531),
 * restated as in oracle_gn.c. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

/* Eigen::LDLT (lower, pivoting on the largest |diagonal|): factor in place, then solve nrhs columns.
 * m: n x n row-major (lower triangle used), B: n x nrhs row-major in/out. */
int vo_ref_ldlt_solve_f64(int n, double *m, int nrhs, double *B) {
  int *tr = (int *)malloc(sizeof(int) * (size_t)n);
  double *temp = (double *)malloc(sizeof(double) * (size_t)n);
#define M(i, j) m[(size_t)(i) * n + (j)]
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(M(k, k));
    for (int i = k + 1; i < n; ++i) {
      const double a = fabs(M(i, i));
      if (a > best) {
        best = a;
        piv = i;
      }
    }
    tr[k] = piv;
    if (piv != k) {
      for (int j = 0; j < k; ++j) { double t = M(k, j); M(k, j) = M(piv, j); M(piv, j) = t; }
      for (int i = piv + 1; i < n; ++i) { double t = M(i, k); M(i, k) = M(i, piv); M(i, piv) = t; }
      { double t = M(k, k); M(k, k) = M(piv, piv); M(piv, piv) = t; }
      for (int i = k + 1; i < piv; ++i) { double t = M(i, k); M(i, k) = M(piv, i); M(piv, i) = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = M(j, j) * M(k, j);
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += M(k, j) * temp[j];
      M(k, k) -= s;
      for (int i = 0; i < rs; ++i) {
        double d = 0.0;
        for (int j = 0; j < k; ++j) d += M(k + 1 + i, j) * temp[j];
        M(k + 1 + i, k) -= d;
      }
    }
    const double akk = M(k, k);
    if (fabs(akk) > 0.0)
      for (int i = 0; i < rs; ++i) M(k + 1 + i, k) /= akk;
  }
  const double tol = 2.2250738585072014e-308; /* std::numeric_limits<double>::min() */
  for (int c = 0; c < nrhs; ++c) {
#define Y(i) B[(size_t)(i) * nrhs + c]
    for (int k = 0; k < n; ++k)
      if (tr[k] != k) { double t = Y(k); Y(k) = Y(tr[k]); Y(tr[k]) = t; }
    for (int i = 0; i < n; ++i) {
      double s = Y(i);
      for (int j = 0; j < i; ++j) s -= M(i, j) * Y(j);
      Y(i) = s;
    }
    for (int i = 0; i < n; ++i) Y(i) = fabs(M(i, i)) > tol ? Y(i) / M(i, i) : 0.0;
    for (int i = n - 1; i >= 0; --i) {
      double s = Y(i);
      for (int j = i + 1; j < n; ++j) s -= M(j, i) * Y(j);
      Y(i) = s;
    }
    for (int k = n - 1; k >= 0; --k)
      if (tr[k] != k) { double t = Y(k); Y(k) = Y(tr[k]); Y(tr[k]) = t; }
#undef Y
  }
#undef M
  free(tr);
  free(temp);
  return 0;
}

static void mat3mul_d(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + (A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j]);
}

/* geometry::se3Exp, geometry_library.cpp:336-384 (T row-major 4x4) */
void vo_ref_se3_exp_f64(const double xi[6], double T[16]) {
  const double v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
  const double theta = sqrt(w[0] * w[0] + (w[1] * w[1] + w[2] * w[2]));
  const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double wxwx[9], R[9], V[9];
  mat3mul_d(wx, wx, wxwx);
  double a, b, c, d;
  if (theta < 1e-9) {
    a = 1.0; b = 0.5; c = 0.5; d = 0.33333333333333333333333333;
  } else {
    const double invtheta2 = 1.0 / (theta * theta);
    a = sin(theta) / theta;
    b = (1 - cos(theta)) * invtheta2;
    c = (1 - cos(theta)) * invtheta2;
    d = (theta - sin(theta)) / (theta * theta * theta);
  }
  for (int k = 0; k < 9; ++k) {
    const double I = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
    R[k] = (I + a * wx[k]) + b * wxwx[k];
    V[k] = (I + c * wx[k]) + d * wxwx[k];
  }
  memset(T, 0, sizeof(double) * 16);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
    T[i * 4 + 3] = V[i * 3] * v[0] + (V[i * 3 + 1] * v[1] + V[i * 3 + 2] * v[2]);
  }
  T[15] = 1.0;
}

/* geometry::SE3Log, geometry_library.cpp:442-495 */
void vo_ref_se3_log_f64(const double T[16], double xi[6]) {
  double R[9], t[3], Vin[9], w[3] = {0, 0, 0};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j];
    t[i] = T[i * 4 + 3];
  }
  const double inCos = (((R[0] + R[4]) + R[8]) - 1.0) * 0.5;
  for (int k = 0; k < 9; ++k) Vin[k] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
  if (!(inCos >= 0.999999999)) {
    const double theta = acos(inCos);
    const double invTheta = 1.0 / theta, invTheta2 = invTheta * invTheta;
    const double f = theta / (2.0 * sin(theta));
    double lnR[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) lnR[i * 3 + j] = f * (R[i * 3 + j] - R[j * 3 + i]);
    w[0] = -lnR[1 * 3 + 2];
    w[1] = lnR[0 * 3 + 2];
    w[2] = -lnR[0 * 3 + 1];
    const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double wxwx[9];
    mat3mul_d(wx, wx, wxwx);
    const double A = sin(theta) * invTheta;
    const double B = (1.0 - cos(theta)) * invTheta2;
    const double g = invTheta2 * (1.0 - A / (2.0 * B));
    for (int k = 0; k < 9; ++k) Vin[k] = (Vin[k] - 0.5 * wx[k]) + g * wxwx[k];
  }
  for (int i = 0; i < 3; ++i) xi[i] = Vin[i * 3] * t[0] + (Vin[i * 3 + 1] * t[1] + Vin[i * 3 + 2] * t[2]);
  xi[3] = w[0];
  xi[4] = w[1];
  xi[5] = w[2];
}

static void mat4mul_d(const double A[16], const double B[16], double C[16]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
      C[i * 4 + j] = s;
    }
}

/* the pose update of sparse_bundle_adjustment.cpp:563-575 */
void vo_ref_sba_pose_update(double T[16], const double x[6]) {
  double xi[6], Tjw[16], dT[16], P[16];
  vo_ref_se3_log_f64(T, xi);
  vo_ref_se3_exp_f64(xi, Tjw); /* addFrontse3 */
  vo_ref_se3_exp_f64(x, dT);
  mat4mul_d(dT, Tjw, P); /* Tjw.noalias() = dT*Tjw: the intended product (the in-place form is only safe with
                            whole-column packets, which -march=native gives on AVX hosts) */
  vo_ref_se3_log_f64(P, xi);
  vo_ref_se3_exp_f64(xi, T);
}

/* one observation: residual, Huber weight, Rij (2x3), Qij (2x6) */
typedef struct {
  double r[2], w, R[6], Q[12];
} sba_obs;
static void sba_linearize(const vo_ref_sba_dims *d, const double *Tjw, const double X[3], const double px[2], int right,
                          sba_obs *o) {
  double Rjw[9], tjw[3], Xij[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Rjw[i * 3 + j] = Tjw[i * 4 + j];
    tjw[i] = Tjw[i * 4 + 3];
  }
  for (int i = 0; i < 3; ++i) Xij[i] = (Rjw[i * 3] * X[0] + (Rjw[i * 3 + 1] * X[1] + Rjw[i * 3 + 2] * X[2])) + tjw[i];
  if (right) {
    double T_rl[16], R_rl[9], t_rl[3], RR[9], Xr[3];
    /* geometry::inverseSE3(T_lr) */
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R_rl[i * 3 + j] = d->T_lr[j * 4 + i];
    }
    for (int i = 0; i < 3; ++i)
      t_rl[i] = (-R_rl[i * 3]) * d->T_lr[3] + ((-R_rl[i * 3 + 1]) * d->T_lr[7] + (-R_rl[i * 3 + 2]) * d->T_lr[11]);
    (void)T_rl;
    mat3mul_d(R_rl, Rjw, RR);
    for (int i = 0; i < 3; ++i) Xr[i] = (R_rl[i * 3] * Xij[0] + (R_rl[i * 3 + 1] * Xij[1] + R_rl[i * 3 + 2] * Xij[2])) + t_rl[i];
    const double fx = d->Kr[0], fy = d->Kr[1], cx = d->Kr[2], cy = d->Kr[3];
    const double invz = 1.0 / Xr[2];
    const double fxinvz = fx * invz, fyinvz = fy * invz, xinvz = Xr[0] * invz, yinvz = Xr[1] * invz;
    const double fx_xinvz2 = fxinvz * xinvz, fy_yinvz2 = fyinvz * yinvz;
    o->r[0] = (fx * xinvz + cx) - px[0];
    o->r[1] = (fy * yinvz + cy) - px[1];
    for (int c = 0; c < 3; ++c) {
      o->R[c] = fxinvz * RR[c] - fx_xinvz2 * RR[6 + c];
      o->R[3 + c] = fyinvz * RR[3 + c] - fy_yinvz2 * RR[6 + c];
    }
    /* Qij = [dp_dX*R_rl, -dp_dX*R_rl*skew(Xij)] (:292-300) */
    const double dp[6] = {fxinvz, 0, -fx_xinvz2, 0, fyinvz, -fy_yinvz2};
    const double sk[9] = {0, -Xij[2], Xij[1], Xij[2], 0, -Xij[0], -Xij[1], Xij[0], 0};
    double DR[6], nDR[6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {
        DR[i * 3 + j] = dp[i * 3] * R_rl[j] + (dp[i * 3 + 1] * R_rl[3 + j] + dp[i * 3 + 2] * R_rl[6 + j]);
        nDR[i * 3 + j] = (-dp[i * 3]) * R_rl[j] + ((-dp[i * 3 + 1]) * R_rl[3 + j] + (-dp[i * 3 + 2]) * R_rl[6 + j]);
      }
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {
        o->Q[i * 6 + j] = DR[i * 3 + j];
        o->Q[i * 6 + 3 + j] = nDR[i * 3] * sk[j] + (nDR[i * 3 + 1] * sk[3 + j] + nDR[i * 3 + 2] * sk[6 + j]);
      }
  } else {
    const double fx = d->Kl[0], fy = d->Kl[1], cx = d->Kl[2], cy = d->Kl[3];
    const double invz = 1.0 / Xij[2];
    const double fxinvz = fx * invz, fyinvz = fy * invz, xinvz = Xij[0] * invz, yinvz = Xij[1] * invz;
    const double fx_xinvz2 = fxinvz * xinvz, fy_yinvz2 = fyinvz * yinvz, xinvz_yinvz = xinvz * yinvz;
    o->r[0] = (fx * xinvz + cx) - px[0];
    o->r[1] = (fy * yinvz + cy) - px[1];
    for (int c = 0; c < 3; ++c) {
      o->R[c] = fxinvz * Rjw[c] - fx_xinvz2 * Rjw[6 + c];
      o->R[3 + c] = fyinvz * Rjw[3 + c] - fy_yinvz2 * Rjw[6 + c];
    }
    const double Q[12] = {fxinvz, 0, -fx_xinvz2, -fx * xinvz_yinvz, fx * (1.0 + xinvz * xinvz), -fx * yinvz,
                          0, fyinvz, -fy_yinvz2, -fy * (1.0 + yinvz * yinvz), fy * xinvz_yinvz, fy * xinvz};
    memcpy(o->Q, Q, sizeof(Q));
  }
  const double absr = fabs(o->r[0]) + fabs(o->r[1]);
  o->w = absr > d->thres_huber ? d->thres_huber / absr : 1.0;
}

/* test hook: residual, Huber weight, Rij (2x3 row-major) and Qij (2x6) of one observation */
void vo_ref_sba_linearize(const vo_ref_sba_dims *d, const double T_jw[16], const double X[3], const double px[2],
                          int right, double r[2], double *w, double R[6], double Q[12]) {
  sba_obs o;
  sba_linearize(d, T_jw, X, px, right, &o);
  memcpy(r, o.r, sizeof(o.r));
  *w = o.w;
  memcpy(R, o.R, sizeof(o.R));
  memcpy(Q, o.Q, sizeof(o.Q));
}

/* calc_Qij_t_Qij_weight, :986-1041 (upper triangle as written there, mirrored; entry (0,1) stays zero) */
static void qtq_weight(double w, const double a[12], double out[36]) {
  double wa[12];
  for (int k = 0; k < 12; ++k) wa[k] = w * a[k];
  memset(out, 0, sizeof(double) * 36);
#define A_(r, c) a[(r) * 6 + (c)]
#define W_(r, c) wa[(r) * 6 + (c)]
#define O_(r, c) out[(r) * 6 + (c)]
  O_(0, 0) = W_(0, 0) * A_(0, 0);
  for (int c = 2; c < 6; ++c) O_(0, c) = W_(0, 0) * A_(0, c);
  for (int c = 1; c < 6; ++c) O_(1, c) = W_(1, 1) * A_(1, c);
  for (int r = 2; r < 6; ++r)
    for (int c = r; c < 6; ++c) O_(r, c) = W_(0, r) * A_(0, c) + W_(1, r) * A_(1, c);
  for (int r = 0; r < 6; ++r)
    for (int c = r + 1; c < 6; ++c)
      if (!(r == 0 && c == 1)) O_(c, r) = O_(r, c);
#undef A_
#undef W_
#undef O_
}

int vo_ref_sba_solve(const vo_ref_sba_dims *d, double *T_jw, const int *opt_index, double *X, const int *obs_ptr,
                     const int *obs_frame, const uint8_t *obs_right, const double *obs_px, double *avg_err) {
  const int No = d->n_opt, M = d->n_points, n = 6 * No;
  const double lambda = 0.00001;
  double *A = (double *)malloc(sizeof(double) * 36 * (size_t)(No + 1)), *a = (double *)malloc(sizeof(double) * 6 * (size_t)(No + 1));
  double *C = (double *)malloc(sizeof(double) * 9 * (size_t)(M + 1)), *b = (double *)malloc(sizeof(double) * 3 * (size_t)(M + 1));
  double *Cinv = (double *)malloc(sizeof(double) * 9 * (size_t)(M + 1)), *Cinvb = (double *)malloc(sizeof(double) * 3 * (size_t)(M + 1));
  /* B_[j][i] dense as in the reference (6x3 blocks) */
  double *B = (double *)calloc((size_t)(No + 1) * (size_t)(M + 1) * 18, sizeof(double));
  double *S = (double *)malloc(sizeof(double) * 36 * (size_t)(No * No + 1)), *BCb = (double *)malloc(sizeof(double) * 6 * (size_t)(No + 1));
  double *Sm = (double *)malloc(sizeof(double) * (size_t)(n * n + 1)), *rhs = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  int rc = 1;
  for (int iter = 0; iter < d->max_iter; ++iter) {
    memset(A, 0, sizeof(double) * 36 * (size_t)No);
    memset(a, 0, sizeof(double) * 6 * (size_t)No);
    memset(C, 0, sizeof(double) * 9 * (size_t)M);
    memset(b, 0, sizeof(double) * 3 * (size_t)M);
    memset(B, 0, sizeof(double) * (size_t)No * (size_t)M * 18);
    memset(S, 0, sizeof(double) * 36 * (size_t)(No * No));
    memset(BCb, 0, sizeof(double) * 6 * (size_t)No);
    double err = 0.0;
    for (int i = 0; i < M; ++i)
      for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
        const int f = obs_frame[o], j = opt_index[f];
        sba_obs L;
        sba_linearize(d, T_jw + 16 * (size_t)f, X + 3 * (size_t)i, obs_px + 2 * (size_t)o, obs_right[o], &L);
        /* calc_Rij_t_Rij_weight (:911-930), Rij_t_rij = weight*(Rij^T rij) */
        double RtR[9];
        for (int r = 0; r < 3; ++r)
          for (int c = r; c < 3; ++c) RtR[r * 3 + c] = RtR[c * 3 + r] = L.w * (L.R[r] * L.R[c] + L.R[3 + r] * L.R[3 + c]);
        for (int k = 0; k < 9; ++k) C[9 * (size_t)i + k] += RtR[k];
        for (int r = 0; r < 3; ++r) b[3 * (size_t)i + r] += -(L.w * (L.R[r] * L.r[0] + L.R[3 + r] * L.r[1]));
        if (j >= 0) {
          double QtQ[36];
          qtq_weight(L.w, L.Q, QtQ);
          for (int k = 0; k < 36; ++k) A[36 * (size_t)j + k] += QtQ[k];
          double *Bji = B + ((size_t)j * M + i) * 18;
          for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 3; ++c) Bji[r * 3 + c] = L.w * (L.Q[r] * L.R[c] + L.Q[6 + r] * L.R[3 + c]);
            a[6 * (size_t)j + r] += -(L.w * (L.Q[r] * L.r[0] + L.Q[6 + r] * L.r[1]));
          }
          for (int k = 0; k < 36; ++k)
            if (isnan(QtQ[k])) rc = -1; /* :318: throw "In LBA, pose becomes nan!" */
        }
        err += L.r[0] * L.r[0] + L.r[1] * L.r[1];
      }
    if (rc < 0) break;
    for (int j = 0; j < No; ++j)
      for (int k = 0; k < 6; ++k) A[36 * (size_t)j + k * 7] += lambda * A[36 * (size_t)j + k * 7];
    for (int i = 0; i < M; ++i) {
      double *Ci = C + 9 * (size_t)i;
      for (int k = 0; k < 3; ++k) Ci[k * 4] += lambda * Ci[k * 4];
      double m[9], I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(m, Ci, sizeof(m));
      vo_ref_ldlt_solve_f64(3, m, 3, I);
      memcpy(Cinv + 9 * (size_t)i, I, sizeof(I));
      for (int r = 0; r < 3; ++r)
        Cinvb[3 * (size_t)i + r] = I[r * 3] * b[3 * (size_t)i] + (I[r * 3 + 1] * b[3 * (size_t)i + 1] + I[r * 3 + 2] * b[3 * (size_t)i + 2]);
    }
    /* 3) BCinv, BCinv_b, BCinvBt over LEFT observations (:458-493) */
    for (int i = 0; i < M; ++i)
      for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
        if (obs_right[o]) continue;
        const int j = opt_index[obs_frame[o]];
        if (j < 0) continue;
        const double *Bji = B + ((size_t)j * M + i) * 18, *Ci = Cinv + 9 * (size_t)i;
        double BC[18];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 3; ++c) BC[r * 3 + c] = Bji[r * 3] * Ci[c] + (Bji[r * 3 + 1] * Ci[3 + c] + Bji[r * 3 + 2] * Ci[6 + c]);
        for (int r = 0; r < 6; ++r)
          BCb[6 * (size_t)j + r] += BC[r * 3] * b[3 * (size_t)i] + (BC[r * 3 + 1] * b[3 * (size_t)i + 1] + BC[r * 3 + 2] * b[3 * (size_t)i + 2]);
        for (int o2 = o; o2 < obs_ptr[i + 1]; ++o2) {
          if (obs_right[o2]) continue;
          const int k = opt_index[obs_frame[o2]];
          if (k < 0) continue;
          const double *Bki = B + ((size_t)k * M + i) * 18;
          double *Sjk = S + 36 * ((size_t)j * No + k);
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) Sjk[r * 6 + c] += BC[r * 3] * Bki[c * 3] + (BC[r * 3 + 1] * Bki[c * 3 + 1] + BC[r * 3 + 2] * Bki[c * 3 + 2]);
        }
      }
    /* :495-497 lower <- upper^T (diagonal blocks transposed in place) */
    for (int j = 0; j < No; ++j)
      for (int u = j; u < No; ++u) {
        double t[36];
        const double *up = S + 36 * ((size_t)j * No + u);
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) t[r * 6 + c] = up[c * 6 + r];
        memcpy(S + 36 * ((size_t)u * No + j), t, sizeof(t));
      }
    /* reduced system (:499-529) and its solve (:531) */
    for (int j = 0; j < No; ++j) {
      for (int u = 0; u < No; ++u)
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            const double s = S[36 * ((size_t)j * No + u) + r * 6 + c];
            Sm[(size_t)(6 * j + r) * n + 6 * u + c] = j == u ? A[36 * (size_t)j + r * 6 + c] - s : -s;
          }
      for (int r = 0; r < 6; ++r) rhs[6 * j + r] = a[6 * (size_t)j + r] - BCb[6 * (size_t)j + r];
    }
    if (n > 0) vo_ref_ldlt_solve_f64(n, Sm, 1, rhs);
    /* 2) y (:537-556), updates (:583-590) */
    for (int i = 0; i < M; ++i) {
      double cbx[3] = {0, 0, 0};
      const double *Ci = Cinv + 9 * (size_t)i;
      for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
        if (obs_right[o]) continue;
        const int j = opt_index[obs_frame[o]];
        if (j < 0) continue;
        const double *Bji = B + ((size_t)j * M + i) * 18;
        double BC[18];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 3; ++c) BC[r * 3 + c] = Bji[r * 3] * Ci[c] + (Bji[r * 3 + 1] * Ci[3 + c] + Bji[r * 3 + 2] * Ci[6 + c]);
        for (int c = 0; c < 3; ++c) { /* CinvBt_[i][j] = BCinv^T */
          double s = 0.0;
          for (int r = 0; r < 6; ++r) s += BC[r * 3 + c] * rhs[6 * j + r];
          cbx[c] += s;
        }
      }
      for (int c = 0; c < 3; ++c) X[3 * (size_t)i + c] += Cinvb[3 * (size_t)i + c] - cbx[c];
    }
    for (int f = 0; f < d->n_frames; ++f)
      if (opt_index[f] >= 0) vo_ref_sba_pose_update(T_jw + 16 * (size_t)f, rhs + 6 * opt_index[f]);
    const double average_error = sqrt(err / (double)d->n_obs);
    if (avg_err) avg_err[iter] = average_error;
    if (isnan(err)) { /* :604-613: throw */
      rc = -1;
      break;
    }
    rc = average_error <= 1.0 ? 1 : 0; /* THRES_SUCCESS_AVG_ERROR (:158, :599-601) */
  }
  free(A); free(a); free(C); free(b); free(Cinv); free(Cinvb); free(B); free(S); free(BCb); free(Sm); free(rhs);
  return rc;
}
