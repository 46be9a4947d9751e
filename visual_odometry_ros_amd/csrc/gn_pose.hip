// gn_pose.hip — pose-only Gauss-Newton motion estimator on one gfx950 workgroup.
//
// Replaces MotionEstimator::poseOnlyBundleAdjustment      (core/visual_odometry/motion_estimator.cpp:665-861)
//      and MotionEstimator::poseOnlyBundleAdjustment_Stereo (core/visual_odometry/motion_estimator.cpp:863-1088)
// including calcJtJ_x/_y, calcJtWJ_x/_y (:1342-1576), geometry::se3Exp_f and
// inverseSE3_f (core/util/geometry_library.cpp:386-440, :554-560) and the 6x6
// Eigen LDLT solve (:823, :1054).
//
// Shape of the work: N (~1500) points x 4 residuals, Jacobian rows generated on
// the fly and contracted into 21 + 6 + 2 scalars. That is a reduction, not a
// GEMM (a 6x6 output would waste >85 % of any MFMA tile), so the kernel is a
// wavefront reduction: one persistent workgroup of GN_T threads runs ALL
// iterations without returning to the host:
//   thread t   : serial partial over points t, t+GN_T, ...      (29 registers)
//   wavefront  : DPP butterfly (quad_perm, row_half_mirror, row_mirror, readlane)
//   workgroup  : GN_T/64 wave totals through LDS, balanced tree
//   lane 0     : damped 6x6 LDLT (symmetric pivoting), se3 exp, update, stop test
// The summation tree is the balanced binary tree over the GN_T thread partials
// in natural order — the oracle's VO_SUM_TREE with tree_width = GN_T.
#include "vo_internal.hpp"
#include "vo_kernels.hpp"
#include "mono_gate.hpp"
#include "svo_device.hpp"

#define GN_T 512
#define GN_NW (GN_T / 64)
#define GN_NACC 29  // 21 H + 6 g + err + cnt

struct GnArgs {
  const float *X;      // n x 3
  const float *p1;     // n x 2 (left / mono pixels)
  const float *p2;     // n x 2 (right pixels, stereo only)
  int n;
  const int *d_n;      // optional device-side count (frame pipeline)
  float Kl[4], Kr[4];
  float Rrl[9], trl[3];
  float thres;
  int variant;
  float T10[16];       // initial T10 (row-major)
  const float *d_T10;  // optional device-side initial T10 (overrides T10)
  float *T_out;        // 16 floats: T01 on success; untouched if NaN (or T01_init when nan_writes_init)
  float T01_init[16];
  int nan_writes_init;
  uint8_t *mask;
  vo_gn_dev_info *info;
  // optional epilogue of the frame pipeline: stage[orig[i]] = stage_val for every inlier that also
  // passes the y > 660 gate of stereo_vo.cpp:653-668
  uint8_t *stage;
  const int32_t *orig;
  int stage_val;
  float gate_thres;
  // optional frame mode (frame_fused.hip): the kernel first does the one compaction of the frame
  // (survivors = stage 3, in index order, + the three step counts), reports / resets the frame's
  // control block, and at the end copies the packed result block to pinned host memory itself:
  // two launches (compaction, D2H blit) fewer on the critical path of every frame
  const int *f_join_word;   // frame mode: wait for the concurrent replay (see vo_gn_frame)
  int f_join_target;
  int f_n;                  // > 0: frame mode, features in input index space
  const uint8_t *f_stage;   // [f_n]
  const uint8_t *f_lmflags; // [f_n] stereo: bit 0 = landmark triangulated (null: all); the BA set is stage 3 && triangulated
  const float *f_X, *f_pl1, *f_pr1;
  int f_world;              // f_X holds world points: the BA set takes T_pw * X (stereo_vo.cpp:605), f_Tpw = rows 0..2 of T_pw
  float f_Tpw[12];
  float *f_CX, *f_Cpl1, *f_Cpr1;
  int32_t *f_Corig;
  int *f_cnt;               // [6]: three step counts, replayed features, size of the BA set, new-point candidates emitted
  int *f_ctl;               // control block; [0] = error flags, [16 + f_nt_word] = replayed features
  int f_ctl_words, f_nt_word;
  int *f_hdr_flags;
  const uint32_t *f_res_dev;  // packed result block (device) -> f_res_host (pinned, device-visible)
  uint32_t *f_res_host;
  int f_res_words;
  int f_res_late_words;     // leading words (header, stage bytes) that this kernel still changes: copied last
  int f_seq, f_seq_word;    // the host block's "result complete" word
  VoNpArgs np;              // closed step [10] (vo_gn_frame::np_*)
  VoAdvArgs adv;            // StereoVO: the next track set (svo_device.hpp); workgroups 1.. of the launch are its DLT workers
  const uint8_t *f_m1, *f_m2, *f_m3;  // mono frame: selection masks in place of f_stage
  int f_mono;               // epilogue = mono_gate_body(f_gate)
  MonoGateArgs f_gate;
};
static_assert(sizeof(GnArgs) <= 3072, "kernel arguments of the BA launch: keep well below the 4 KB segment");

// upper-triangular index of (i,j), i<=j, row-major: matches oracle UT[][]
__device__ __forceinline__ constexpr int ut(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// The 29 per-thread sums of a GN iteration. Every accumulator receives its terms in the same order and with the same
// two roundings (product, then sum) as the reference's calcJtWJ_x/_y; entries (.,2),(.,3) and (.,4),(.,5) of a row of
// JtWJ are neighbours in a register pair so that one v_pk_mul_f32 + one v_pk_add_f32 serve two of them (the compiler's
// own vectoriser found about half of these pairs and paid for them in register moves).
typedef float gn_f2 __attribute__((ext_vector_type(2)));
struct GnAcc {
  gn_f2 h02, h04, h12, h14, h22, h24, h34, h44;  // (a,b),(a,b+1)
  float h00, h11, h33, h55;
  gn_f2 g23, g45;
  float g0, g1;
  float err;
  float cnt;
};
__device__ __forceinline__ void gn_acc_clear(GnAcc &A) {
  const gn_f2 z = {0.0f, 0.0f};
  A.h02 = A.h04 = A.h12 = A.h14 = A.h22 = A.h24 = A.h34 = A.h44 = A.g23 = A.g45 = z;
  A.h00 = A.h11 = A.h33 = A.h55 = A.g0 = A.g1 = A.err = A.cnt = 0.0f;
}
// entry k of the upper triangle (oracle UT[][] order) / of g
__device__ __forceinline__ void gn_acc_unpack(const GnAcc &A, float (&H)[21], float (&g)[6]) {
  H[ut(0, 0)] = A.h00;  H[ut(0, 1)] = 0.0f;  // (x rows have Jt(1) == 0, y rows Jt(0) == 0: never touched)
  H[ut(0, 2)] = A.h02.x;  H[ut(0, 3)] = A.h02.y;  H[ut(0, 4)] = A.h04.x;  H[ut(0, 5)] = A.h04.y;
  H[ut(1, 1)] = A.h11;  H[ut(1, 2)] = A.h12.x;  H[ut(1, 3)] = A.h12.y;  H[ut(1, 4)] = A.h14.x;  H[ut(1, 5)] = A.h14.y;
  H[ut(2, 2)] = A.h22.x;  H[ut(2, 3)] = A.h22.y;  H[ut(2, 4)] = A.h24.x;  H[ut(2, 5)] = A.h24.y;
  H[ut(3, 3)] = A.h33;  H[ut(3, 4)] = A.h34.x;  H[ut(3, 5)] = A.h34.y;
  H[ut(4, 4)] = A.h44.x;  H[ut(4, 5)] = A.h44.y;  H[ut(5, 5)] = A.h55;
  g[0] = A.g0;  g[1] = A.g1;  g[2] = A.g23.x;  g[3] = A.g23.y;  g[4] = A.g45.x;  g[5] = A.g45.y;
}
// rows 2..5 of a residual row's contribution (common to x and y rows): l_a = (w *) Jt(a)
template <bool W>
__device__ __forceinline__ void acc_row_tail(GnAcc &A, float w, gn_f2 P23, gn_f2 P45) {
  const gn_f2 L23 = W ? w * P23 : P23, L45 = W ? w * P45 : P45;
  A.h22 += L23.x * P23;
  A.h24 += L23.x * P45;
  A.h33 += L23.y * P23.y;
  A.h34 += L23.y * P45;
  A.h44 += L45.x * P45;
  A.h55 += L45.y * P45.y;
}
// x rows: Jt(1) == 0 (calcJtWJ_x / calcJtJ_x)
template <bool W>
__device__ __forceinline__ void acc_row_x(GnAcc &A, float w, const float (&Jt)[6]) {
  const gn_f2 P23 = {Jt[2], Jt[3]}, P45 = {Jt[4], Jt[5]};
  const float l0 = W ? w * Jt[0] : Jt[0];
  A.h00 += l0 * Jt[0];
  A.h02 += l0 * P23;
  A.h04 += l0 * P45;
  acc_row_tail<W>(A, w, P23, P45);
}
// y rows: Jt(0) == 0 (calcJtWJ_y / calcJtJ_y)
template <bool W>
__device__ __forceinline__ void acc_row_y(GnAcc &A, float w, const float (&Jt)[6]) {
  const gn_f2 P23 = {Jt[2], Jt[3]}, P45 = {Jt[4], Jt[5]};
  const float l1 = W ? w * Jt[1] : Jt[1];
  A.h11 += l1 * Jt[1];
  A.h12 += l1 * P23;
  A.h14 += l1 * P45;
  acc_row_tail<W>(A, w, P23, P45);
}
__device__ __forceinline__ void acc_g_x(GnAcc &A, float s, const float (&Jt)[6]) {
  const gn_f2 P23 = {Jt[2], Jt[3]}, P45 = {Jt[4], Jt[5]};
  A.g0 -= s * Jt[0];
  A.g23 -= s * P23;
  A.g45 -= s * P45;
}
__device__ __forceinline__ void acc_g_y(GnAcc &A, float s, const float (&Jt)[6]) {
  const gn_f2 P23 = {Jt[2], Jt[3]}, P45 = {Jt[4], Jt[5]};
  A.g1 -= s * Jt[1];
  A.g23 -= s * P23;
  A.g45 -= s * P45;
}

__device__ __forceinline__ void jac_x(float (&Jt)[6], float f, float iz, float fxxiz, float xiz, float yiz) {
  Jt[0] = f * iz;
  Jt[1] = 0.0f;
  Jt[2] = -fxxiz * iz;
  Jt[3] = -fxxiz * yiz;
  Jt[4] = f * (1.0f + xiz * xiz);
  Jt[5] = -f * yiz;
}
__device__ __forceinline__ void jac_y(float (&Jt)[6], float f, float iz, float fyyiz, float xiz, float yiz) {
  Jt[0] = 0.0f;
  Jt[1] = f * iz;
  Jt[2] = -fyyiz * iz;
  Jt[3] = -f * (1.0f + yiz * yiz);
  Jt[4] = fyyiz * xiz;
  Jt[5] = f * xiz;
}

// ---- lane-0 small dense algebra (same operation order as the oracle) ---------
// Eigen LDLT<Matrix<float,6,6>,Lower>: unblocked, symmetric pivoting on the largest
// remaining |diagonal|. Everything is fully unrolled with compile-time indices; the
// data-dependent pivot becomes predicated swaps, so the 6x6 lives in registers
// (no scratch memory on the critical path of every GN iteration).
__device__ __forceinline__ void swapf(float &a, float &b) {
  const float t = a;
  a = b;
  b = t;
}
// Eigen's LDLT picks the pivot of step k among the diagonal entries of rows k.., and in its left-looking in-place form
// those still hold their ORIGINAL values at that point (step k only ever writes column k): the whole transposition
// sequence depends on the original diagonal alone. So: the sequence first (6 magnitudes, predicated swaps of 2 x 6
// values instead of whole rows and columns), then the matrix is GATHERED already permuted — entry (i, j) of P A P^T is
// A(perm[i], perm[j]), read from the upper-triangle sums in LDS — and the factorisation runs without a single swap: the
// same operations on the same values as the swapping form (and as sba_solve_kernel does it for the local BA).
// tot: the 21 upper-triangle sums in oracle UT[][] order followed by g[6]; lambda damps the diagonal (:820-822, :1051-1053).
__device__ __forceinline__ void ldlt6_solve(const float *__restrict__ tot, float lambda, float (&x)[6]) {
  int perm[6];
  float mag[6], dg[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    perm[k] = k;
    dg[k] = tot[ut(k, k)] * (1.0f + lambda);
    mag[k] = fabsf(dg[k]);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int piv = k;
    float best = mag[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (mag[i] > best) {  // the first of equal maxima, as Eigen's maxCoeff
        best = mag[i];
        piv = i;
      }
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (piv == p) {
        swapf(mag[k], mag[p]);
        swapf(dg[k], dg[p]);
        const int t = perm[k];
        perm[k] = perm[p];
        perm[p] = t;
      }
  }
  float m[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    m[i][i] = dg[i];
#pragma unroll
    for (int j = 0; j < i; ++j) {
      const int a = perm[i], b = perm[j];
      const int lo = a < b ? a : b, hi = a < b ? b : a;
      m[i][j] = tot[lo * 6 - ((lo * (lo - 1)) >> 1) + (hi - lo)];  // ut(lo, hi)
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k > 0) {
      float temp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
      m[k][k] -= s;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        float d = 0.0f;
#pragma unroll
        for (int j = 0; j < k; ++j) d += m[i][j] * temp[j];
        m[i][k] -= d;
      }
    }
    const float akk = m[k][k];
    if (fabsf(akk) > 0.0f) {
#pragma unroll
      for (int i = k + 1; i < 6; ++i) m[i][k] /= akk;
    }
  }
  // y = P b: the transposition sequence applied to a vector is the gather by perm
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = tot[21 + perm[i]];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = y[i];
#pragma unroll
    for (int j = 0; j < i; ++j) s -= m[i][j] * y[j];
    y[i] = s;
  }
  const float tol = 1.17549435e-38f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (fabsf(m[i][i]) > tol)
      y[i] /= m[i][i];
    else
      y[i] = 0.0f;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) s -= m[j][i] * y[j];
    y[i] = s;
  }
  // x = P^T y: scatter back through perm
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float v = y[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) v = perm[q] == i ? y[q] : v;
    x[i] = perm[0] == i ? y[0] : v;
  }
}

__device__ void se3_exp_dev(const float (&xi)[6], float (&T)[16]) {
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
  float a, b, bV, c;
  if ((double)theta < 1e-7) {
    a = 1.0f;
    b = 0.5f;
    bV = 0.5f;
    c = 0.33333333333333333333333333f;
  } else {
    // The reference evaluates sin/cos in double and rounds each coefficient to float once.
    // GN steps are small: below 0.25 rad the double Taylor series (terms to th^13, truncation
    // < 1e-19) gives the same doubles as libm to within an ulp, at a fraction of the cost.
    const double th = (double)theta;
    double sn, cs1;  // sin(th), 1 - cos(th)
    if (th < 0.25) {
      const double t2 = th * th;
      sn = th * (1.0 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 + t2 * (-1.0 / 39916800 + t2 * (1.0 / 6227020800.0)))))));
      cs1 = t2 * (0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600.0))))));
    } else {
      sn = sin(th);
      cs1 = 1 - cos(th);
    }
    a = (float)(sn / th);
    b = (float)(cs1 / (double)(theta * theta));
    bV = b;
    c = (float)((th - sn) / (double)(theta * theta * theta));
  }
  float R[9], V[9];
  for (int i = 0; i < 9; ++i) {
    float I = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
    R[i] = (I + a * wx[i]) + b * wx2[i];
    V[i] = (I + bV * wx[i]) + c * wx2[i];
  }
  float t[3];
  for (int i = 0; i < 3; ++i) t[i] = (V[i * 3 + 0] * v[0] + V[i * 3 + 1] * v[1]) + V[i * 3 + 2] * v[2];
  for (int i = 0; i < 16; ++i) T[i] = 0.0f;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
    T[i * 4 + 3] = t[i];
  }
  T[15] = 1.0f;
}

// residual rows of one point (motion_estimator.cpp:733-800 mono, :935-1020 stereo)
template <bool STEREO>
__device__ __forceinline__ void gn_point(GnAcc &A, const GnArgs &a, const float (&R10)[9], const float (&t10)[3], float X0,
                                         float X1, float X2, float plx, float ply, float prx, float pry, int i) {
  const float THRES_HUBER = 0.5f;
  const float fx_l = a.Kl[0], fy_l = a.Kl[1], cx_l = a.Kl[2], cy_l = a.Kl[3];
  const float fx_r = a.Kr[0], fy_r = a.Kr[1], cx_r = a.Kr[2], cy_r = a.Kr[3];
  const float thres = a.thres;
  float Xl[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) Xl[r] = ((R10[r * 3 + 0] * X0 + R10[r * 3 + 1] * X1) + R10[r * 3 + 2] * X2) + t10[r];
  float Jt[6];
  if (STEREO) {
    float Xr[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      Xr[r] = ((a.Rrl[r * 3 + 0] * Xl[0] + a.Rrl[r * 3 + 1] * Xl[1]) + a.Rrl[r * 3 + 2] * Xl[2]) + a.trl[r];
    const float iz_l = 1.0f / Xl[2];
    const float xiz_l = Xl[0] * iz_l, yiz_l = Xl[1] * iz_l;
    const float fxxiz_l = fx_l * xiz_l, fyyiz_l = fy_l * yiz_l;
    const float rx_l = (fxxiz_l + cx_l) - plx, ry_l = (fyyiz_l + cy_l) - ply;
    const float iz_r = 1.0f / Xr[2];
    const float xiz_r = Xr[0] * iz_r, yiz_r = Xr[1] * iz_r;
    const float fxxiz_r = fx_r * xiz_r, fyyiz_r = fy_r * yiz_r;
    const float rx_r = (fxxiz_r + cx_r) - prx, ry_r = (fyyiz_r + cy_r) - pry;
    float weight = 1.0f;
    float absrxry = ((fabsf(rx_l) + fabsf(ry_l)) + fabsf(rx_r)) + fabsf(ry_r);
    absrxry *= 0.5f;
    if (absrxry >= THRES_HUBER) weight = THRES_HUBER / absrxry;
    const bool outl = absrxry >= thres;
    a.mask[i] = outl ? 0 : 1;
    if (outl) A.cnt += 1.0f;
    jac_x(Jt, fx_l, iz_l, fxxiz_l, xiz_l, yiz_l);
    acc_row_x<true>(A, weight, Jt);
    acc_g_x(A, weight * rx_l, Jt);
    A.err += rx_l * rx_l;
    jac_y(Jt, fy_l, iz_l, fyyiz_l, xiz_l, yiz_l);
    acc_row_y<true>(A, weight, Jt);
    acc_g_y(A, weight * ry_l, Jt);
    A.err += ry_l * ry_l;
    jac_x(Jt, fx_r, iz_r, fxxiz_r, xiz_r, yiz_r);
    acc_row_x<true>(A, weight, Jt);
    acc_g_x(A, weight * rx_r, Jt);
    A.err += rx_r * rx_r;
    jac_y(Jt, fy_r, iz_r, fyyiz_r, xiz_r, yiz_r);
    acc_row_y<true>(A, weight, Jt);
    acc_g_y(A, weight * ry_r, Jt);
    A.err += ry_r * ry_r;
  } else {
    const float iz = 1.0f / Xl[2];
    const float xiz = Xl[0] * iz, yiz = Xl[1] * iz;
    const float fxxiz = fx_l * xiz, fyyiz = fy_l * yiz;
    const float rx = (fxxiz + cx_l) - plx, ry = (fyyiz + cy_l) - ply;
    float weight = 1.0f;
    bool flag_weight = false;
    const float absrxry = fabsf(rx) + fabsf(ry);
    if (absrxry >= THRES_HUBER) {
      weight = THRES_HUBER / absrxry;
      flag_weight = true;
    }
    const bool outl = absrxry >= thres;
    a.mask[i] = outl ? 0 : 1;
    if (outl) A.cnt += 1.0f;
    jac_x(Jt, fx_l, iz, fxxiz, xiz, yiz);
    if (flag_weight) {
      acc_row_x<true>(A, weight, Jt);
      acc_g_x(A, weight * rx, Jt);
    } else {
      acc_row_x<false>(A, 1.0f, Jt);
      acc_g_x(A, rx, Jt);
    }
    A.err += rx * rx;
    jac_y(Jt, fy_l, iz, fyyiz, xiz, yiz);
    if (flag_weight) {
      const float w_ry = weight * ry;
      acc_row_y<true>(A, weight, Jt);
      acc_g_y(A, w_ry, Jt);
      if (a.variant == VO_GN_VARIANT_CORE)
        A.err += w_ry * ry;
      else
        A.err += ry * ry;
    } else {
      acc_row_y<false>(A, 1.0f, Jt);
      acc_g_y(A, ry, Jt);
      A.err += ry * ry;
    }
  }
}

#ifdef GN_STAMP
__device__ long long vo_gn_stamps[12];
extern "C" int vo_debug_gn_stamps(vo_ctx *c, long long out[12]) {
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  VO_CHECK_HIP(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(vo_gn_stamps), sizeof(long long) * 12));
  return VO_OK;
}
#endif
#define GN_PC 4  // points per thread kept in registers across the iterations (n <= GN_PC * GN_T: no reloads)

template <bool STEREO>
__global__ __launch_bounds__(GN_T) void gn_pose_kernel(GnArgs a) {
  // thread partials, transposed: s_red[k][t]; wavefront w then reduces sums 4w .. 4w+3
  __shared__ float s_red[GN_NACC * GN_T];
  __shared__ float s_tot[32];
  __shared__ float s_T10[16];
  __shared__ int s_stop;
  __shared__ int s_pcnt[4 * GN_NW * 4];  // frame prologue: per chunk and wavefront the four counts of a super-chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x > 0) {
    // ---- StereoVO, workgroups 1..: stereo_vo.cpp:721-725 for EVERY bin's candidate (speculatively, next to the
    // iterations of workgroup 0): mapping::triangulateDLT of the candidate and its tracked right pixel, both depths
    // positive. One wavefront per 64 bins; the verdicts are written through, then the wavefront counts itself done.
    if (STEREO && wave == 0 && a.adv.on) {
      const int b = ((int)blockIdx.x - 1) * 64 + lane;
      if (a.np.cand_done) {  // the candidates' own launch may still be running (synchronous call): bounded join
        int ok = 1;
        if (lane == 0) ok = vo_np_wait_candidates(a.np) ? 1 : 0;
        ok = __builtin_amdgcn_readfirstlane(ok);
        if (!ok && lane == 0) atomicOr(a.f_hdr_flags, 8);  // (0.1 s later than the prologue's report: straight into the header;
                                                           //  the frame is then issued again in stream order)
      }
      if (b < a.np.bins) {
        int acc = 0;
        if (vo_np_ld8(a.np, &a.np.has[b]) && vo_np_ld8(a.np, &a.np.bin_m[b])) {
          float Xl[3], Xr[3];
          svo_triangulate(a.adv.cam, vo_np_ldf(a.np, &a.np.xy[2 * b]), vo_np_ldf(a.np, &a.np.xy[2 * b + 1]), vo_np_ldf(a.np, &a.np.bin_r[2 * b]),
                          vo_np_ldf(a.np, &a.np.bin_r[2 * b + 1]), Xl, Xr);
          acc = (Xl[2] > 0 && Xr[2] > 0) ? 1 : 0;
        }
        __hip_atomic_store(&a.adv.acc_bin[b], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) atomicAdd(a.adv.dlt_done, 1);
    }
    return;
  }
#ifdef GN_STAMP
#define GSTAMP(k) if (tid == 0) vo_gn_stamps[k] = (long long)__builtin_amdgcn_s_memrealtime();
#else
#define GSTAMP(k)
#endif
  GSTAMP(0)
  __builtin_amdgcn_s_setprio(3);  // one workgroup on the frame's critical path, next to the side stream's kernels
  int n = a.d_n ? *a.d_n : a.n;
  if (a.f_n > 0 && a.f_join_word) {
    // the strict-border replay runs on a stream of its own: its last kernel counts its finished workgroups
    if (tid == 0) {
      int polls = 0, ok = 1;
      while ((int)(__hip_atomic_load(a.f_join_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.f_join_target) < 0) {
        if (++polls > (1 << 17)) {  // ~0.1 s: the replay stream is stuck; report instead of hanging
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(16);
      }
      s_stop = ok;
    }
    __syncthreads();
    GSTAMP(5)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!s_stop && tid == 0) atomicOr(a.f_ctl, 8);  // reported through the frame's error flags
    // (word [2] of the block: replay workgroups that gave up waiting for the frame kernel; consumed here)
    if (tid == 0 && __hip_atomic_load(a.f_join_word + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      __hip_atomic_store(const_cast<int *>(a.f_join_word) + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(a.f_ctl, 8);
    }
    __syncthreads();
  }
  if (a.f_n > 0) {
    // ---- frame mode prologue: survivors in index order + step counts (the one compaction of the frame) ----
    // Super-chunks of GN_SC * GN_T features: every load of a super-chunk — the stage bytes, the landmark flags and the
    // points themselves, selected or not — is issued before anything waits, so the memory latency is exposed once per
    // super-chunk (one for 1500 features) instead of twice per 512 features, and two barriers replace nine.
    constexpr int GN_SC = 4;  // chunks of GN_T features per super-chunk
    int *s_run = (int *)s_tot;  // [4] running totals: st >= 1, >= 2, >= 3, BA set
    if (tid < 4) s_run[tid] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < a.f_n; c0 += GN_SC * GN_T) {
      int st[GN_SC], fl[GN_SC];
      float gx[GN_SC][3], gl[GN_SC][2], gr[GN_SC][2];
#pragma unroll
      for (int q = 0; q < GN_SC; ++q) {
        const int i = c0 + q * GN_T + tid;
        const bool in = i < a.f_n;
        const int ii = in ? i : 0;
        if (a.f_m1)  // mono: tracked / refined / member of the BA set (mono_vo.cpp:773, :788, :799-826)
          st[q] = in ? (a.f_m1[ii] ? (a.f_m2[ii] ? (a.f_m3[ii] ? 3 : 2) : 1) : 0) : 0;
        else
          st[q] = in ? a.f_stage[ii] : 0;
        fl[q] = a.f_lmflags ? a.f_lmflags[ii] : VO_LM_TRIANGULATED;
        gx[q][0] = a.f_X[3 * ii];
        gx[q][1] = a.f_X[3 * ii + 1];
        gx[q][2] = a.f_X[3 * ii + 2];
        if (a.f_world) {  // Xp = T_pw.block<3,3>(0,0) * X + T_pw.block<3,1>(0,3): 3-term dot products as e0 + (e1 + e2)
          const float x0 = gx[q][0], x1 = gx[q][1], x2 = gx[q][2];
#pragma unroll
          for (int r = 0; r < 3; ++r)
            gx[q][r] = (a.f_Tpw[r * 4 + 0] * x0 + (a.f_Tpw[r * 4 + 1] * x1 + a.f_Tpw[r * 4 + 2] * x2)) + a.f_Tpw[r * 4 + 3];
        }
        gl[q][0] = a.f_pl1[2 * ii];
        gl[q][1] = a.f_pl1[2 * ii + 1];
        gr[q][0] = a.f_pr1 ? a.f_pr1[2 * ii] : 0.f;
        gr[q][1] = a.f_pr1 ? a.f_pr1[2 * ii + 1] : 0.f;
      }
      bool in_ba[GN_SC];
      int below[GN_SC];
#pragma unroll
      for (int q = 0; q < GN_SC; ++q) {
        // stereo_vo.cpp:599: only triangulated landmarks enter the pose-only BA
        in_ba[q] = st[q] >= 3 && (fl[q] & VO_LM_TRIANGULATED);
        const unsigned long long b1 = __ballot(st[q] >= 1), b2 = __ballot(st[q] >= 2), b3 = __ballot(st[q] >= 3),
                                 b4 = __ballot(in_ba[q]);
        below[q] = __popcll(b4 & ((1ull << lane) - 1ull));
        if (lane == 0) {
          int *d = s_pcnt + (q * GN_NW + wave) * 4;
          d[0] = __popcll(b1);
          d[1] = __popcll(b2);
          d[2] = __popcll(b3);
          d[3] = __popcll(b4);
        }
      }
      __syncthreads();
      int base = s_run[3];
#pragma unroll
      for (int q = 0; q < GN_SC; ++q) {
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_pcnt[(q * GN_NW + w) * 4 + 3];
        if (in_ba[q]) {
          const int i = c0 + q * GN_T + tid;
          const int o = base + woff + below[q];
          a.f_CX[3 * o] = gx[q][0];
          a.f_CX[3 * o + 1] = gx[q][1];
          a.f_CX[3 * o + 2] = gx[q][2];
          a.f_Cpl1[2 * o] = gl[q][0];
          a.f_Cpl1[2 * o + 1] = gl[q][1];
          if (a.f_pr1) {
            a.f_Cpr1[2 * o] = gr[q][0];
            a.f_Cpr1[2 * o + 1] = gr[q][1];
          }
          a.f_Corig[o] = i;
        }
        for (int w = 0; w < GN_NW; ++w) base += s_pcnt[(q * GN_NW + w) * 4 + 3];
      }
      __syncthreads();
      if (tid < 4) {
        int tot = 0;
        for (int k = 0; k < GN_SC * GN_NW; ++k) tot += s_pcnt[k * 4 + tid];
        s_run[tid] += tot;
      }
      __syncthreads();
    }
    n = s_run[3];
    if (tid < 3) a.f_cnt[tid] = s_run[tid];
    if (tid == 3) a.f_cnt[4] = n;
    // every producer / consumer of the control block ran before this kernel: report, then reset
    if (tid == 0 && a.f_ctl) {
      *a.f_hdr_flags = a.f_ctl[0];
      a.f_cnt[3] = a.f_ctl[16 + a.f_nt_word];
    }
    __syncthreads();  // (also: the compacted arrays written above are read below by other threads)
    if (a.f_ctl)
      for (int k = tid; k < a.f_ctl_words; k += GN_T) a.f_ctl[k] = 0;
    // the pixel arrays and new-point results are final: their copy to the host runs under the iterations
    if (a.f_res_host)
      for (int k = a.f_res_late_words + tid; k < a.f_res_words; k += GN_T) a.f_res_host[k] = a.f_res_dev[k];
  }
  GSTAMP(1)
  float T10[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T10[k] = a.d_T10 ? a.d_T10[k] : a.T10[k];

  // the first GN_PC points of this thread stay in registers
  float cX[GN_PC][3], cP1[GN_PC][2], cP2[GN_PC][2];
#pragma unroll
  for (int q = 0; q < GN_PC; ++q) {
    const int i = tid + q * GN_T;
    const int ii = i < n ? i : 0;
    const bool ok = n > 0;
    cX[q][0] = ok ? a.X[3 * ii] : 0.f;
    cX[q][1] = ok ? a.X[3 * ii + 1] : 0.f;
    cX[q][2] = ok ? a.X[3 * ii + 2] : 0.f;
    cP1[q][0] = ok ? a.p1[2 * ii] : 0.f;
    cP1[q][1] = ok ? a.p1[2 * ii + 1] : 0.f;
    cP2[q][0] = (STEREO && ok) ? a.p2[2 * ii] : 0.f;
    cP2[q][1] = (STEREO && ok) ? a.p2[2 * ii + 1] : 0.f;
  }

  GSTAMP(2)
  float err_prev = 1e10f;
  int iter = 0;
  float last_err = 0, last_derr = 0, last_dnorm = 0;
  int last_cnt = 0;

  for (iter = 0; iter < 100; ++iter) {
    float R10[9], t10[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) R10[i * 3 + j] = T10[i * 4 + j];
      t10[i] = T10[i * 4 + 3];
    }
    GnAcc A;
    gn_acc_clear(A);

    // thread t: points t, t + GN_T, ... in ascending order
#pragma unroll
    for (int q = 0; q < GN_PC; ++q) {
      const int i = tid + q * GN_T;
      if (i < n)
        gn_point<STEREO>(A, a, R10, t10, cX[q][0], cX[q][1], cX[q][2], cP1[q][0], cP1[q][1], cP2[q][0], cP2[q][1], i);
    }
    for (int i = tid + GN_PC * GN_T; i < n; i += GN_T)
      gn_point<STEREO>(A, a, R10, t10, a.X[3 * i], a.X[3 * i + 1], a.X[3 * i + 2], a.p1[2 * i], a.p1[2 * i + 1],
                       STEREO ? a.p2[2 * i] : 0.f, STEREO ? a.p2[2 * i + 1] : 0.f, i);

    // ---- reduction: balanced binary tree over the GN_T thread partials in natural order ----
    // transposed through LDS so that a lane adds 8 neighbouring partials (three tree levels) and
    // ONE 4-way DPP butterfly per wavefront finishes the other six, instead of eight butterflies
    {
      float AH[21], Ag[6];
      gn_acc_unpack(A, AH, Ag);
#pragma unroll
      for (int k = 0; k < 21; ++k) s_red[k * GN_T + tid] = AH[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) s_red[(21 + k) * GN_T + tid] = Ag[k];
      s_red[27 * GN_T + tid] = A.err;
      s_red[28 * GN_T + tid] = A.cnt;
    }
    __syncthreads();
    {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 4 * wave + c;
        if (k < GN_NACC) {
          const float4 lo = *(const float4 *)&s_red[k * GN_T + 8 * lane];
          const float4 hi = *(const float4 *)&s_red[k * GN_T + 8 * lane + 4];
          v[c] = ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
        } else {
          v[c] = 0.0f;
        }
      }
      wave_sum4_f32(v[0], v[1], v[2], v[3]);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s_tot[4 * wave + c] = v[c];
      }
    }
    __syncthreads();
    // ---- wavefront 0 (all lanes alike): damped normal equations, LDLT, pose update, stop test.
    // The ~1600-instruction solve is issue-bound; run by every wavefront it would compete with
    // itself for the four SIMDs of the CU.
    if (wave == 0) {
      float dxi[6];
      float err_curr = s_tot[27];
      const float inv_npts = 1.0f / (float)n;
      err_curr *= (inv_npts * 0.5f);
      if (STEREO) err_curr = sqrtf(err_curr);
      const float delta_err = fabsf(err_curr - err_prev);
      const float lambda = 0.00001f;
      ldlt6_solve(s_tot, lambda, dxi);
      float dT[16], Tn[16];
      se3_exp_dev(dxi, dT);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = dT[i * 4 + 0] * T10[0 * 4 + j];
          s += dT[i * 4 + 1] * T10[1 * 4 + j];
          s += dT[i * 4 + 2] * T10[2 * 4 + j];
          s += dT[i * 4 + 3] * T10[3 * 4 + j];
          Tn[i * 4 + j] = s;
        }
      err_prev = err_curr;
      float s2 = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) s2 += dxi[k] * dxi[k];
      const float dnorm = sqrtf(s2);
      last_err = err_curr;
      last_derr = delta_err;
      last_dnorm = dnorm;
      last_cnt = (int)s_tot[28];
      const bool stop = dnorm < (float)1e-6 || delta_err < (float)1e-7;
      if (lane < 16) {
        float tv = Tn[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) tv = lane == k ? Tn[k] : tv;
        s_T10[lane] = tv;
      }
      if (lane == 0) s_stop = stop ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) T10[k] = s_T10[k];
    if (s_stop) {
      ++iter;
      break;
    }
  }
  GSTAMP(3)
  if (a.stage) {
    for (int i = tid; i < n; i += GN_T) {
      const float gate = a.p1[2 * i + 1] > 660 ? 100.f : 0.f;
      if (a.mask[i] && gate < a.gate_thres) a.stage[a.orig[i]] = (uint8_t)a.stage_val;
    }
    // survivors of [5] outside the BA set keep mask_motion = true (stereo_vo.cpp:582) and meet the same gate
    if (a.f_n > 0 && a.f_lmflags && !a.f_mono)
      for (int i = tid; i < a.f_n; i += GN_T)
        if (a.f_stage[i] == 3 && !(a.f_lmflags[i] & VO_LM_TRIANGULATED)) {
          const float gate = a.f_pl1[2 * i + 1] > 660 ? 100.f : 0.f;
          if (gate < a.gate_thres) a.stage[i] = (uint8_t)a.stage_val;
        }
  }
  if (tid == 0) {
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += s_T10[i] * s_T10[i];
    const float nrm = sqrtf(s);
    const int is_nan = isnan(nrm) ? 1 : 0;
    if (!is_nan) {
      // inverseSE3_f
      float Rt[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = s_T10[j * 4 + i];
      const float t0 = s_T10[3], t1 = s_T10[7], t2 = s_T10[11];
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) a.T_out[i * 4 + j] = Rt[i * 3 + j];
        a.T_out[i * 4 + 3] = ((-Rt[i * 3 + 0]) * t0 + (-Rt[i * 3 + 1]) * t1) + (-Rt[i * 3 + 2]) * t2;
      }
      a.T_out[12] = 0;
      a.T_out[13] = 0;
      a.T_out[14] = 0;
      a.T_out[15] = 1;
    } else if (a.nan_writes_init) {
      for (int i = 0; i < 16; ++i) a.T_out[i] = a.T01_init[i];
    }
    if (a.info) {
      a.info->iterations = iter;
      a.info->err = last_err;
      a.info->delta_err = last_derr;
      a.info->delta_norm = last_dnorm;
      a.info->cnt_invalid = last_cnt;
      a.info->is_nan = is_nan;
    }
  }
  bool gate_done = false;
  if constexpr (!STEREO) {
    if (a.f_n > 0 && a.f_mono) {
      // mono frame: mask_motion, Sampson gate, stages, counts, the copy of the result block and MonoVO's next track set
      // (mono_gate.hpp). (Compiled into the mono kernel only: with the gate's arguments referenced from the stereo kernel
      // too, that one kept a 1.9 KB copy of its argument block in scratch memory — 45 us per frame.)
      __syncthreads();  // inlier mask, pose and info above are this workgroup's own stores
      mono_gate_body<GN_T>(a.f_gate, tid, n, (uint8_t *)s_red, (int *)s_tot);
      gate_done = true;
    }
  }
  if (gate_done) {
  } else if (a.f_n > 0 && a.f_res_host) {
    // frame mode epilogue: the packed result block goes to pinned host memory from here
    __syncthreads();  // stage marks, pose and info above are this workgroup's own stores
    if (STEREO && a.adv.on) {
      // the DLT workers (workgroups 1..) finished long ago — they take ~15 us, the iterations above 40+; bounded anyway
      if (tid == 0) {
        int polls = 0;
        while ((int)(__hip_atomic_load(a.adv.dlt_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.adv.dlt_target) < 0) {
          if (++polls > (1 << 17)) {
            atomicOr(a.f_hdr_flags, 16);  // reported by vo_stereo_frame_result
            break;
          }
          __builtin_amdgcn_s_sleep(16);
        }
      }
      __syncthreads();
    }
    GSTAMP(6)
    if (a.np.bins > 0 && a.np.cand_done) {  // (the candidates' own launch: see VoNpArgs::cand_done)
      if (tid == 0 && !vo_np_wait_candidates(a.np)) atomicOr(a.f_hdr_flags, 8);
      __syncthreads();
    }
    if (a.np.bins > 0) {
      // ---- closed step [10]: updateWeightBin(lmtrack_final.pts_l1) + emission (np_emit.hpp), with the
      // trackBidirection results (stereo_vo.cpp:706-711) the frame kernel computed for every bin's candidate.
      // LDS: the partial sums are dead.
      const uint8_t *stg = a.stage;
      const uint8_t sv = (uint8_t)a.stage_val;
      vo_np_emit(a.np, a.f_n, a.f_pl1, [&](int i) { return stg[i] == sv; }, tid, GN_T, (uint8_t *)s_red, (int *)s_tot,
                 &a.f_cnt[5]);
    }
    GSTAMP(7)
    if (STEREO && a.adv.on) {
      // ---- StereoVO: the next frame's track set. lmtrack_final (stereo_vo.cpp:670: the stage-4 features in index order,
      // with their landmarks), then the new landmarks of step [10] (:729-734: candidate order, ids from the landmark
      // counter, NOT triangulated: set3DPoint is commented out at :736). setStereoPtsSeenAndRelatedLandmarks (:752).
      // Four chunks of GN_T entries per round (one round up to 2048 tracks): the keep flags of a round are gathered
      // first, ONE barrier pair serves them all, and a thread's loads of the round are issued before its stores.
      constexpr int NCH = 4;
      int *s_wv = s_pcnt;  // [NCH * GN_NW] per-wavefront counts of a round + [GN_NW] members of the last keyframe among the survivors
      const VoAdvArgs &v = a.adv;
      const uint8_t sv4 = (uint8_t)a.stage_val;
      int base = 0, kf = 0;
      for (int c0 = 0; c0 < a.f_n; c0 += NCH * GN_T) {
        bool keep[NCH];
        int below[NCH];
        uint8_t fl[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
          const int i = c0 + q * GN_T + tid;
          keep[q] = i < a.f_n && a.stage[i] == sv4;
          fl[q] = keep[q] ? v.cur.flags[i] : (uint8_t)0;
          const unsigned long long bal = __ballot(keep[q]);
          below[q] = __popcll(bal & ((1ull << lane) - 1ull));
          kf += __popcll(__ballot(keep[q] && (fl[q] & VO_LM_KF_MEMBER)));
          if (lane == 0) s_wv[q * GN_NW + wave] = __popcll(bal);
        }
        __syncthreads();
        int off = base, o[NCH];
        float pl[NCH][2], pr[NCH][2], xw[NCH][3];
        int32_t id[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
          int woff = 0, tot = 0;
#pragma unroll
          for (int w = 0; w < GN_NW; ++w) {
            const int cw = s_wv[q * GN_NW + w];
            woff += w < wave ? cw : 0;
            tot += cw;
          }
          o[q] = off + woff + below[q];
          off += tot;
          if (keep[q]) {
            const int i = c0 + q * GN_T + tid;
            pl[q][0] = a.f_pl1[2 * i];
            pl[q][1] = a.f_pl1[2 * i + 1];
            pr[q][0] = a.f_pr1[2 * i];
            pr[q][1] = a.f_pr1[2 * i + 1];
            xw[q][0] = v.cur.Xw[3 * i];
            xw[q][1] = v.cur.Xw[3 * i + 1];
            xw[q][2] = v.cur.Xw[3 * i + 2];
            id[q] = v.cur.ids[i];
          }
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q)
          if (keep[q] && o[q] < v.cap) {
            const int oo = o[q];
            v.nxt.pts_l[2 * oo] = pl[q][0];
            v.nxt.pts_l[2 * oo + 1] = pl[q][1];
            v.nxt.pts_r[2 * oo] = pr[q][0];
            v.nxt.pts_r[2 * oo + 1] = pr[q][1];
            v.nxt.Xw[3 * oo] = xw[q][0];
            v.nxt.Xw[3 * oo + 1] = xw[q][1];
            v.nxt.Xw[3 * oo + 2] = xw[q][2];
            v.nxt.flags[oo] = fl[q];
            v.nxt.ids[oo] = id[q];
          }
        base = off;
        __syncthreads();
      }
      if (lane == 0) s_wv[NCH * GN_NW + wave] = kf;
      const int n_surv = base;
      const int n_emit = a.np.bins > 0 ? a.f_cnt[5] : 0;  // (thread 0's store, behind vo_np_emit's last barrier)
      for (int c0 = 0; c0 < n_emit; c0 += GN_T) {
        const int j = c0 + tid;
        const bool keep = j < n_emit && a.np.out_acc[j];
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wv[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < GN_NW; ++w) {
          const int cw = s_wv[w];
          woff += w < wave ? cw : 0;
          tot += cw;
        }
        if (keep) {
          const int o = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
          if (o < v.cap) {
            v.nxt.pts_l[2 * o] = a.np.out_l[2 * j];
            v.nxt.pts_l[2 * o + 1] = a.np.out_l[2 * j + 1];
            v.nxt.pts_r[2 * o] = a.np.out_r[2 * j];
            v.nxt.pts_r[2 * o + 1] = a.np.out_r[2 * j + 1];
            v.nxt.Xw[3 * o] = 0.0f;
            v.nxt.Xw[3 * o + 1] = 0.0f;
            v.nxt.Xw[3 * o + 2] = 0.0f;
            v.nxt.flags[o] = 0;
            v.nxt.ids[o] = v.id_base + (o - n_surv);
          }
        }
        base += tot;
        __syncthreads();
      }
      __syncthreads();
      if (tid == 0) {
        int kft = 0;
        for (int w = 0; w < GN_NW; ++w) kft += s_wv[NCH * GN_NW + w];
        SvoHdr h;
        h.n_surv = n_surv;
        h.n_kf_tracked = kft;
        h.n_new = base - n_surv;
        h.n_next = base;
        h.n_emit = n_emit;
        h.overflow = base > v.cap ? 1 : 0;
        h.seq = 0;
        h.id_min = base > 0 ? v.nxt.ids[0] : v.id_base;  // (another thread's store, behind the barrier above)
        *v.hdr_dev = h;
        *v.hdr_host = h;  // (pinned; made visible by the system-scope fence in front of the frame's sequence word below)
      }
    }
    GSTAMP(8)
    for (int k = tid; k < a.f_res_late_words; k += GN_T)
      if (k != a.f_seq_word || !a.f_seq) a.f_res_host[k] = a.f_res_dev[k];
    if (a.f_seq) {
      // the host polls this word instead of waiting for a HIP event (tens of microseconds of wake-up latency per
      // frame): every store of the block above must be visible in host memory before it
      __threadfence_system();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&a.f_res_host[a.f_seq_word], (uint32_t)a.f_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  GSTAMP(4)
}

// geometry::se3Exp_f / inverseSE3_f on their own (one lane), for the device unit test of the small-angle branch
__global__ void se3_exp_kernel(const float *xi, float *T, float *Tinv) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float x[6], M[16];
  for (int i = 0; i < 6; ++i) x[i] = xi[i];
  se3_exp_dev(x, M);
  for (int i = 0; i < 16; ++i) T[i] = M[i];
  float Rt[9];  // inverseSE3_f, the form of the GN kernel's last step
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = M[j * 4 + i];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Tinv[i * 4 + j] = Rt[i * 3 + j];
    Tinv[i * 4 + 3] = ((-Rt[i * 3 + 0]) * M[3] + (-Rt[i * 3 + 1]) * M[7]) + (-Rt[i * 3 + 2]) * M[11];
  }
  Tinv[12] = Tinv[13] = Tinv[14] = 0.f;
  Tinv[15] = 1.f;
}

extern "C" int vo_se3_exp(vo_ctx *c, const float xi[6], float T[16], float Tinv[16]) {
  if (!c || !xi || !T) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_CHECK_HIP(c, hipMemcpyAsync(c->d_mat, xi, sizeof(float) * 6, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(se3_exp_kernel, dim3(1), dim3(64), 0, c->stream, c->d_mat, c->d_mat + 8, c->d_mat + 24);
  VO_CHECK_HIP(c, hipGetLastError());
  float out[32];
  VO_CHECK_HIP(c, hipMemcpyAsync(out, c->d_mat + 8, sizeof(out), hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(T, out, sizeof(float) * 16);
  if (Tinv) memcpy(Tinv, out + 16, sizeof(float) * 16);
  return VO_OK;
}

// ---- host side ---------------------------------------------------------------
static void inverse_se3_host(const float T[16], float Ti[16]) {
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

// general 4x4 inverse by cofactors (Matrix4f::inverse() at motion_estimator.cpp:700)
static void inverse4x4_host(const float m[16], float inv[16]) {
  float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
  float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
  float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
  float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
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

// Enqueue a GN solve on device-resident inputs. T01_init is the reference's
// in/out pose; the kernel writes T01 to d_Tout unless the pose went NaN.
int vo_gn_enqueue(vo_ctx *c, bool stereo, bool mono_general_inverse, const float *dX, const float *dP1,
                  const float *dP2, int n, const int *d_n, const float Kl[4], const float Kr[4],
                  const float T_lr[16], float thres, int variant, const float T01_init[16],
                  float *d_Tout, uint8_t *d_mask, vo_gn_dev_info *d_info, bool write_init_on_nan, uint8_t *d_stage,
                  const int32_t *d_orig, int stage_val, float gate_thres, const vo_gn_frame *frame) {
  GnArgs a;
  memset(&a, 0, sizeof(a));
  if (frame && frame->n > 0) {
    a.f_n = frame->n;
    a.f_stage = frame->stage;
    a.f_lmflags = frame->lm_flags;
    a.f_X = frame->X;
    if (frame->T_pw) {
      a.f_world = 1;
      memcpy(a.f_Tpw, frame->T_pw, sizeof(a.f_Tpw));
    }
    a.f_pl1 = frame->pl1;
    a.f_pr1 = frame->pr1;
    a.f_CX = frame->C_X;
    a.f_Cpl1 = frame->C_pl1;
    a.f_Cpr1 = frame->C_pr1;
    a.f_Corig = frame->C_orig;
    a.f_cnt = frame->cnt;
    a.f_ctl = frame->ctl;
    a.f_ctl_words = frame->ctl_words;
    a.f_nt_word = frame->nt_word;
    a.f_hdr_flags = frame->hdr_flags;
    a.f_join_word = frame->join_word;
    a.f_join_target = frame->join_target;
    a.f_res_dev = (const uint32_t *)frame->res_dev;
    a.f_res_host = (uint32_t *)frame->res_host;
    a.f_res_words = (int)((frame->res_bytes + 3) / 4);
    a.f_res_late_words = (int)(frame->res_late_bytes / 4);
    a.f_seq = frame->seq;
    a.f_seq_word = frame->seq_word;
    a.np.bins = frame->np_bins;
    a.np.bins_u = frame->np_bins_u;
    a.np.u_step = frame->np_u_step;
    a.np.v_step = frame->np_v_step;
    a.np.has = frame->np_has;
    a.np.xy = frame->np_xy;
    a.np.bin_r = frame->np_bin_r;
    a.np.bin_m = frame->np_bin_m;
    a.np.out_l = frame->np_out_l;
    a.np.out_r = frame->np_out_r;
    a.np.out_m = frame->np_out_m;
    a.np.host_l = frame->np_host_l;
    a.np.host_r = frame->np_host_r;
    a.np.host_m = frame->np_host_m;
    a.np.cand_done = frame->np_cand_done;
    a.np.cand_target = frame->np_cand_target;
    if (frame->adv && frame->adv->on) {
      a.adv = *frame->adv;
      a.np.acc_bin = a.adv.acc_bin;
      a.np.out_acc = a.adv.accept;
    }
    a.f_m1 = frame->m1;
    a.f_m2 = frame->m2;
    a.f_m3 = frame->m3;
    if (frame->mono_gate) {
      a.f_mono = 1;
      a.f_gate = *frame->mono_gate;
    }
  }
  a.X = dX;
  a.p1 = dP1;
  a.p2 = dP2;
  a.n = n;
  a.d_n = d_n;
  for (int i = 0; i < 4; ++i) {
    a.Kl[i] = Kl[i];
    a.Kr[i] = Kr ? Kr[i] : Kl[i];
  }
  if (stereo) {
    float Trl[16];
    inverse_se3_host(T_lr, Trl);  // motion_estimator.cpp:869
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) a.Rrl[i * 3 + j] = Trl[i * 4 + j];
      a.trl[i] = Trl[i * 4 + 3];
    }
  }
  a.thres = thres;
  a.variant = variant;
  if (mono_general_inverse)
    inverse4x4_host(T01_init, a.T10);  // :700
  else
    inverse_se3_host(T01_init, a.T10);  // :904
  a.d_T10 = nullptr;
  memcpy(a.T01_init, T01_init, sizeof(a.T01_init));
  a.nan_writes_init = write_init_on_nan ? 1 : 0;
  a.T_out = d_Tout;
  a.mask = d_mask;
  a.info = d_info;
  a.stage = d_stage;
  a.orig = d_orig;
  a.stage_val = stage_val;
  a.gate_thres = gate_thres;
  vo_prof_begin(c, VO_K_GN);
  const int workers = (stereo && a.adv.on) ? (a.np.bins + 63) / 64 : 0;  // StereoVO: DLT of every bin's candidate
  if (stereo)
    hipLaunchKernelGGL(gn_pose_kernel<true>, dim3(1 + workers), dim3(GN_T), 0, c->stream, a);
  else
    hipLaunchKernelGGL(gn_pose_kernel<false>, dim3(1), dim3(GN_T), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
