// sba.hip — sparse local bundle adjustment on the device.
// Reference: SparseBundleAdjustmentSolver::solveForFiniteIterations
// (core/visual_odometry/ba_solver/sparse_bundle_adjustment.cpp:150-643), double precision
// (_BA_Numeric, define_ba_type.h:9), fixed iteration count, constant lambda = 1e-5, Huber weights.
//
// The reference walks landmark -> observation and scatters 6x6 / 6x3 / 3x3 blocks into dense block
// arrays (B_ alone is N_opt x M blocks). Here the sparsity pattern is turned into index lists once per
// problem (it does not change between iterations) — on the host by vo_sba_solve below, on the device by the
// StereoVO driver (stereo_vo_lba.hip) — and every reduction becomes a gather with a fixed order, so there
// are no atomics and results are run-to-run identical:
//   sba_update_point_kernel  four lanes per landmark: y_i, X_i += y_i of the previous iteration; then C_i, b_i,
//                      damping, 3x3 pivoted LDLT inverse, C^-1 b, and per "slot" (left observation in an optimised
//                      keyframe) the blocks B_ji, B_ji C_i^-1 and (B_ji C_i^-1) b_i
//   sba_pose_schur_kernel  SBA_PG workgroups of 256 per optimised pose: A_j, a_j over the pose's observation list,
//                      (B C^-1 b)_j over its slot list; SBA_SG workgroups per block (j,k), j <= k: sum of
//                      (B_ji C_i^-1) B_ki^T over the landmark pairs; lane-strided partial sums, DPP + LDS reduction
//   sba_solve_reg_kernel<42>  the steady-state window: reduced system assembled into LDS by 512 lanes, then one
//                      wavefront: pivot order from the original diagonal, rows of the permuted matrix in registers,
//                      Eigen-order LDLT, x; pose updates exp(log(exp(x) exp(log T))); average error (second wavefront)
//   sba_assemble_kernel + sba_solve_kernel  the same for any other size (one lane per entry; matrix in LDS)
// Three launches per iteration in the steady-state window (four otherwise), no host round trip inside the solve.
// The quirks listed in oracle/oracle_sba.c (B assigned not accumulated, left-only Schur loops, symmetrisation
// overwrite, calc_Qij_t_Qij_weight's zero entries) are reproduced.
#include <stdlib.h>
#include <time.h>

#include <vector>

#include "vo_internal.hpp"
#include "vo_kernels.hpp"

#ifdef SBA_STAMP  // measurement build: per-phase maxima over the workgroups of the last iteration, 10 ns ticks
#define SBA_TICK() ((long long)__builtin_amdgcn_s_memrealtime())
#define SBA_STAMP_MAX(slot, t1, t0) \
  if ((threadIdx.x & 63) == 0) atomicMax(d.flags + (slot), (int)((t1) - (t0)))
#else
#define SBA_TICK() 0LL
#define SBA_STAMP_MAX(slot, t1, t0) ((void)(t1), (void)(t0))
#endif
#include "sba_device.hpp"

struct SbaObs {
  double r[2], w, R[6], Q[12];
};

// one observation: residual, Huber weight, Rij (2x3), Qij (2x6); sparse_bundle_adjustment.cpp:207-300 (right
// image) and :331-398 (left image), expression by expression
__device__ __forceinline__ void sba_linearize(const SbaDev &d, const double *__restrict__ Tjw, const double X[3],
                                              const double px[2], int right, SbaObs &o) {
  double Rjw[9], tjw[3], Xij[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rjw[i * 3 + j] = Tjw[i * 4 + j];
    tjw[i] = Tjw[i * 4 + 3];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) Xij[i] = (Rjw[i * 3] * X[0] + (Rjw[i * 3 + 1] * X[1] + Rjw[i * 3 + 2] * X[2])) + tjw[i];
  if (right) {
    double RR[9], Xr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        RR[i * 3 + j] = d.R_rl[i * 3] * Rjw[j] + (d.R_rl[i * 3 + 1] * Rjw[3 + j] + d.R_rl[i * 3 + 2] * Rjw[6 + j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Xr[i] = (d.R_rl[i * 3] * Xij[0] + (d.R_rl[i * 3 + 1] * Xij[1] + d.R_rl[i * 3 + 2] * Xij[2])) + d.t_rl[i];
    const double fx = d.Kr[0], fy = d.Kr[1], cx = d.Kr[2], cy = d.Kr[3];
    const double invz = 1.0 / Xr[2];
    const double fxinvz = fx * invz, fyinvz = fy * invz, xinvz = Xr[0] * invz, yinvz = Xr[1] * invz;
    const double fx_xinvz2 = fxinvz * xinvz, fy_yinvz2 = fyinvz * yinvz;
    o.r[0] = (fx * xinvz + cx) - px[0];
    o.r[1] = (fy * yinvz + cy) - px[1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o.R[c] = fxinvz * RR[c] - fx_xinvz2 * RR[6 + c];
      o.R[3 + c] = fyinvz * RR[3 + c] - fy_yinvz2 * RR[6 + c];
    }
    const double dp[6] = {fxinvz, 0, -fx_xinvz2, 0, fyinvz, -fy_yinvz2};
    const double sk[9] = {0, -Xij[2], Xij[1], Xij[2], 0, -Xij[0], -Xij[1], Xij[0], 0};
    double nDR[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        o.Q[i * 6 + j] = dp[i * 3] * d.R_rl[j] + (dp[i * 3 + 1] * d.R_rl[3 + j] + dp[i * 3 + 2] * d.R_rl[6 + j]);
        nDR[i * 3 + j] = (-dp[i * 3]) * d.R_rl[j] + ((-dp[i * 3 + 1]) * d.R_rl[3 + j] + (-dp[i * 3 + 2]) * d.R_rl[6 + j]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        o.Q[i * 6 + 3 + j] = nDR[i * 3] * sk[j] + (nDR[i * 3 + 1] * sk[3 + j] + nDR[i * 3 + 2] * sk[6 + j]);
  } else {
    const double fx = d.Kl[0], fy = d.Kl[1], cx = d.Kl[2], cy = d.Kl[3];
    const double invz = 1.0 / Xij[2];
    const double fxinvz = fx * invz, fyinvz = fy * invz, xinvz = Xij[0] * invz, yinvz = Xij[1] * invz;
    const double fx_xinvz2 = fxinvz * xinvz, fy_yinvz2 = fyinvz * yinvz, xinvz_yinvz = xinvz * yinvz;
    o.r[0] = (fx * xinvz + cx) - px[0];
    o.r[1] = (fy * yinvz + cy) - px[1];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o.R[c] = fxinvz * Rjw[c] - fx_xinvz2 * Rjw[6 + c];
      o.R[3 + c] = fyinvz * Rjw[3 + c] - fy_yinvz2 * Rjw[6 + c];
    }
    o.Q[0] = fxinvz;
    o.Q[1] = 0;
    o.Q[2] = -fx_xinvz2;
    o.Q[3] = -fx * xinvz_yinvz;
    o.Q[4] = fx * (1.0 + xinvz * xinvz);
    o.Q[5] = -fx * yinvz;
    o.Q[6] = 0;
    o.Q[7] = fyinvz;
    o.Q[8] = -fy_yinvz2;
    o.Q[9] = -fy * (1.0 + yinvz * yinvz);
    o.Q[10] = fy * xinvz_yinvz;
    o.Q[11] = fy * xinvz;
  }
  const double absr = fabs(o.r[0]) + fabs(o.r[1]);
  o.w = absr > d.thres_huber ? d.thres_huber / absr : 1.0;
}

// Eigen::LDLT of a 3x3 (lower, pivoting on the largest |diagonal|), solve for the identity: C^-1 as
// C_[i].ldlt().solve(I) gives it (:460)
__device__ __forceinline__ void sba_inv3_ldlt(const double Cin[9], double out[9]) {
  double m[3][3];
  int tr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m[i][j] = Cin[i * 3 + j];
  double temp[3];
  for (int k = 0; k < 3; ++k) {
    int piv = k;
    double best = fabs(m[k][k]);
    for (int i = k + 1; i < 3; ++i) {
      const double a = fabs(m[i][i]);
      if (a > best) {
        best = a;
        piv = i;
      }
    }
    tr[k] = piv;
    if (piv != k) {
      for (int j = 0; j < k; ++j) { const double t = m[k][j]; m[k][j] = m[piv][j]; m[piv][j] = t; }
      for (int i = piv + 1; i < 3; ++i) { const double t = m[i][k]; m[i][k] = m[i][piv]; m[i][piv] = t; }
      { const double t = m[k][k]; m[k][k] = m[piv][piv]; m[piv][piv] = t; }
      for (int i = k + 1; i < piv; ++i) { const double t = m[i][k]; m[i][k] = m[piv][i]; m[piv][i] = t; }
    }
    const int rs = 3 - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
      m[k][k] -= s;
      for (int i = 0; i < rs; ++i) {
        double dd = 0.0;
        for (int j = 0; j < k; ++j) dd += m[k + 1 + i][j] * temp[j];
        m[k + 1 + i][k] -= dd;
      }
    }
    const double akk = m[k][k];
    if (fabs(akk) > 0.0)
      for (int i = 0; i < rs; ++i) m[k + 1 + i][k] /= akk;
  }
  const double tol = 2.2250738585072014e-308;
  for (int c = 0; c < 3; ++c) {
    double y[3] = {c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0};
    for (int k = 0; k < 3; ++k)
      if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < 3; ++i) {
      double s = y[i];
      for (int j = 0; j < i; ++j) s -= m[i][j] * y[j];
      y[i] = s;
    }
    for (int i = 0; i < 3; ++i) y[i] = fabs(m[i][i]) > tol ? y[i] / m[i][i] : 0.0;
    for (int i = 2; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < 3; ++j) s -= m[j][i] * y[j];
      y[i] = s;
    }
    for (int k = 2; k >= 0; --k)
      if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < 3; ++i) out[i * 3 + c] = y[i];
  }
}

// ---- per landmark -------------------------------------------------------------------
// SBA_LQ lanes share a landmark: lane `sub` takes observations (and slots) sub, sub + 4, ... and the partial sums meet
// in a quad butterfly — a landmark seen by all nine stereo keyframes costs 5 dependent rounds instead of 18, and a
// wavefront's duration is that of its longest landmark. Every lane of the quad ends with the same bits (a + b == b + a).
#define SBA_LDS_FRAMES 32  // poses of up to this many frames are staged in LDS (a lane's pose depends on its observation)
template <int CTRL>
__device__ __forceinline__ double sba_dpp_f64(double v) {
  int2 p = __builtin_bit_cast(int2, v);
  p.x = dpp_i32<CTRL>(p.x);
  p.y = dpp_i32<CTRL>(p.y);
  return __builtin_bit_cast(double, p);
}
__device__ __forceinline__ double sba_quad_sum(double v) {  // sum over the SBA_LQ lanes of a landmark; all must be active
  v += sba_dpp_f64<0xB1>(v);
  v += sba_dpp_f64<0x4E>(v);
#if SBA_LQ == 8
  v += sba_dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the eight
#endif
  return v;
}

__device__ __forceinline__ double sba_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;  // lane 0 holds the total
}

// ---- per optimised pose: A_j, a_j, (B C^-1 b)_j -----------------------------------------
// Workgroups of SBA_WG lanes: SBA_PG (SBA_SG) of them share a pose (a block), lane-strided partial sums, a butterfly
// per wavefront and the wavefronts' sums added in LDS — one partial per workgroup goes to HBM. With 8 x 256 lanes a
// pose of the bench window (3 800 observations) is two dependent rounds of loads per lane.
#define SBA_WG 256
// sum over the workgroup, value by value: DPP butterfly inside rows of 16 lanes (two instructions move a double), the
// SBA_WG / 16 row sums through LDS, lane k < NV adds them in row order (a 64-lane __shfl_down tree costs two
// ds_bpermute per step and was three quarters of the kernel)
__device__ __forceinline__ double sba_row16_sum(double v) {  // every lane must be active
  v += sba_dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += sba_dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += sba_dpp_f64<0x141>(v);  // row_half_mirror
  v += sba_dpp_f64<0x140>(v);  // row_mirror
  return v;
}
template <int NV>
__device__ __forceinline__ void sba_group_reduce(double (&acc)[NV], double *__restrict__ out, int tid) {
  __shared__ double s_part[SBA_WG / 16][NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double t = sba_row16_sum(acc[k]);
    if ((tid & 15) == 0) s_part[tid >> 4][k] = t;
  }
  __syncthreads();
  if (tid < NV) {
    double t = s_part[0][tid];
#pragma unroll
    for (int w = 1; w < SBA_WG / 16; ++w) t += s_part[w][tid];
    out[tid] = t;
  }
}

__device__ __forceinline__ void sba_pose_body(const SbaDev &d, int j, int g, int tid) {
  double acc[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) acc[k] = 0.0;
  int nan_seen = 0;
  const long long st0 = SBA_TICK();
  double Tj[16];  // every observation of the list is one in the pose's own frame (wave-uniform)
  {
    const double *Tp = d.T + 16 * (size_t)d.opt_frame[j];
#pragma unroll
    for (int k = 0; k < 16; ++k) Tj[k] = Tp[k];
  }
  const long long st1 = SBA_TICK();
  const int po_end = d.pose_obs_end ? d.pose_obs_end[j] : d.pose_obs_ptr[j + 1];
  for (int q = d.pose_obs_ptr[j] + g * SBA_WG + tid; q < po_end; q += SBA_WG * SBA_PG) {
    const int o = d.pose_obs[q], i = d.pose_lm[q];
    const double X[3] = {d.X[3 * (size_t)i], d.X[3 * (size_t)i + 1], d.X[3 * (size_t)i + 2]};
    const double px[2] = {d.obs_px[2 * o], d.obs_px[2 * o + 1]};
    SbaObs L;
    sba_linearize(d, Tj, X, px, d.obs_right[o], L);
    // calc_Qij_t_Qij_weight (:986-1041): written for Q(0,1) = Q(1,0) = 0; entry (0,1) stays zero
    double wa[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) wa[k] = L.w * L.Q[k];
    double q6[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) q6[k] = 0.0;
    q6[0] = wa[0] * L.Q[0];
#pragma unroll
    for (int c = 2; c < 6; ++c) q6[c] = wa[0] * L.Q[c];
#pragma unroll
    for (int c = 1; c < 6; ++c) q6[6 + c] = wa[7] * L.Q[6 + c];
#pragma unroll
    for (int r = 2; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 6; ++c) q6[r * 6 + c] = wa[r] * L.Q[c] + wa[6 + r] * L.Q[6 + c];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r + 1; c < 6; ++c)
        if (!(r == 0 && c == 1)) q6[c * 6 + r] = q6[r * 6 + c];
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      acc[k] += q6[k];
      nan_seen |= (q6[k] != q6[k]);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[36 + r] += -(L.w * (L.Q[r] * L.r[0] + L.Q[6 + r] * L.r[1]));
  }
  const int ps_end = d.pose_slot_end ? d.pose_slot_end[j] : d.pose_slot_ptr[j + 1];
  for (int q = d.pose_slot_ptr[j] + g * SBA_WG + tid; q < ps_end; q += SBA_WG * SBA_PG) {
    const double *BCb = d.BCb + 6 * (size_t)d.pose_slot[q];  // (B_ji C_i^-1) b_i, formed by the point kernel (:473)
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[42 + r] += BCb[r];
  }
  const long long st2 = SBA_TICK();
  sba_group_reduce<48>(acc, d.Apart + 48 * ((size_t)j * SBA_PG + g), tid);
  const long long st3 = SBA_TICK();
  SBA_STAMP_MAX(4, st1, st0);
  SBA_STAMP_MAX(5, st2, st1);
  SBA_STAMP_MAX(6, st3, st2);
  if (__any(nan_seen) && (tid & 63) == 0) atomicOr(d.flags, 1);  // :318 "In LBA, pose becomes nan!"
}

// ---- per block (j,k) of B C^-1 B^T -------------------------------------------------------
// (blocks below the diagonal are overwritten by the transposed upper ones (:495-497) before anything reads them:
// whatever an observation list in reverse keyframe order accumulates there is discarded — they are not computed)
__device__ __forceinline__ void sba_schur_body(const SbaDev &d, int jk, int g, int tid) {
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  const long long st1 = SBA_TICK();
  const int pr_end = d.pair_end ? d.pair_end[jk] : d.pair_ptr[jk + 1];
  for (int q = d.pair_ptr[jk] + g * SBA_WG + tid; q < pr_end; q += SBA_WG * SBA_SG) {
    const double *BC = d.BCs + 18 * (size_t)d.pair_a[q], *Bk = d.Bs + 18 * (size_t)d.pair_b[q];
    double bc[18], bk[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      bc[k] = BC[k];
      bk[k] = Bk[k];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        acc[r * 6 + c] += bc[r * 3] * bk[c * 3] + (bc[r * 3 + 1] * bk[c * 3 + 1] + bc[r * 3 + 2] * bk[c * 3 + 2]);  // :489
  }
  const long long st2 = SBA_TICK();
  sba_group_reduce<36>(acc, d.S + 36 * ((size_t)jk * SBA_SG + g), tid);
  const long long st3 = SBA_TICK();
  SBA_STAMP_MAX(8, st2, st1);
  SBA_STAMP_MAX(9, st3, st2);
}

// A_j / a_j / (B C^-1 b)_j and the blocks of B C^-1 B^T do not depend on one another: ONE launch, workgroups
// 0 .. n_opt * SBA_PG - 1 take the pose sums, the others the blocks on and above the diagonal, row by row
__global__ __launch_bounds__(SBA_WG) void sba_pose_schur_kernel(SbaDev d) {
  const int t = blockIdx.x, tid = threadIdx.x, No = d.n_opt;
  const int n_pose = No * SBA_PG;
  if (t < n_pose) {
    sba_pose_body(d, t / SBA_PG, t % SBA_PG, tid);
    return;
  }
  int u = (t - n_pose) / SBA_SG, j = 0;
  while (u >= No - j) {
    u -= No - j;
    ++j;
  }
  sba_schur_body(d, j * No + j + u, (t - n_pose) % SBA_SG, tid);
}

// ---- se3 exp / log in double (geometry_library.cpp:336-384, :442-495) ---------------------
__device__ void sba_mat3mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + (A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j]);
}
__device__ void sba_se3_exp(const double xi[6], double T[16]) {
  const double v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
  const double theta = sqrt(w[0] * w[0] + (w[1] * w[1] + w[2] * w[2]));
  const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double wxwx[9], V[9];
  sba_mat3mul(wx, wx, wxwx);
  double a, b, c, dd;
  if (theta < 1e-9) {
    a = 1.0;
    b = 0.5;
    c = 0.5;
    dd = 0.33333333333333333333333333;
  } else {
    const double invtheta2 = 1.0 / (theta * theta);
    a = sin(theta) / theta;
    b = (1 - cos(theta)) * invtheta2;
    c = (1 - cos(theta)) * invtheta2;
    dd = (theta - sin(theta)) / (theta * theta * theta);
  }
  for (int k = 0; k < 16; ++k) T[k] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int k = i * 3 + j;
      const double I = i == j ? 1.0 : 0.0;
      T[i * 4 + j] = (I + a * wx[k]) + b * wxwx[k];
      V[k] = (I + c * wx[k]) + dd * wxwx[k];
    }
  for (int i = 0; i < 3; ++i) T[i * 4 + 3] = V[i * 3] * v[0] + (V[i * 3 + 1] * v[1] + V[i * 3 + 2] * v[2]);
  T[15] = 1.0;
}
__device__ void sba_se3_log(const double T[16], double xi[6]) {
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
    w[0] = -lnR[5];
    w[1] = lnR[2];
    w[2] = -lnR[1];
    const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double wxwx[9];
    sba_mat3mul(wx, wx, wxwx);
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
// :563-575: xi = log(T); addFrontse3(xi, x) = log(exp(x) exp(xi)); T = exp(xi) — in two halves: the first does not depend
// on the solve's x (the register solve lets an idle wavefront compute it while the factorisation runs)
__device__ void sba_pose_pre(const double *T, double Tjw[16]) {
  double Tin[16], xi[6];
  for (int k = 0; k < 16; ++k) Tin[k] = T[k];
  sba_se3_log(Tin, xi);
  sba_se3_exp(xi, Tjw);
}
__device__ void sba_pose_post(double *T, const double Tjw[16], const double x[6]) {
  double xi[6], dT[16], P[16], To[16];
  sba_se3_exp(x, dT);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += dT[i * 4 + k] * Tjw[k * 4 + j];
      P[i * 4 + j] = s;
    }
  sba_se3_log(P, xi);
  sba_se3_exp(xi, To);
  for (int k = 0; k < 16; ++k) T[k] = To[k];
}
__device__ void sba_pose_update(double *T, const double x[6]) {
  double Tjw[16];
  sba_pose_pre(T, Tjw);
  sba_pose_post(T, Tjw, x);
}

// ---- reduced system: assemble, LDLT, solve, pose update, average error (one wavefront) ------------------
// The matrix lives in LDS with M(i,j) at m[j * n + i]: the row sweeps below touch M(row, j) for consecutive
// rows in consecutive lanes, i.e. consecutive addresses.
#define SBA_CH 8
// max over the 64 lanes (DPP butterfly, same pattern as wave_sum_i32 in vo_internal.hpp); all lanes get it
__device__ __forceinline__ unsigned sba_wave_max_u32(unsigned v) {
  int x = (int)v, t;
  // every DPP is evaluated unconditionally (inside a select it would run under a partial EXEC mask and read
  // zeros from the disabled lanes)
#define SBA_UMAX_STEP(expr) \
  t = (expr);               \
  x = (unsigned)x > (unsigned)t ? x : t;
  SBA_UMAX_STEP(dpp_i32<0xB1>(x))
  SBA_UMAX_STEP(dpp_i32<0x4E>(x))
  SBA_UMAX_STEP(dpp_i32<0x141>(x))
  SBA_UMAX_STEP(dpp_i32<0x140>(x))
  SBA_UMAX_STEP(__builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false))
  SBA_UMAX_STEP(__builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false))
#undef SBA_UMAX_STEP
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ double sba_row_dot(const double *__restrict__ m, const double *__restrict__ temp, int n,
                                              int row, int k) {
  // sum_{j<k} M(row,j) temp[j] in increasing j (the CPU order). Software-pipelined: the LDS loads of chunk c + 1 are
  // issued before the multiply-adds of chunk c, so a step waits for the LDS once instead of once per chunk.
  double dd = 0.0;
  int j = 0;
  double a[SBA_CH], t[SBA_CH];
  if (k >= SBA_CH) {
#pragma unroll
    for (int q = 0; q < SBA_CH; ++q) {
      a[q] = m[(size_t)q * n + row];
      t[q] = temp[q];
    }
    for (j = SBA_CH; j + SBA_CH <= k; j += SBA_CH) {
      double a2[SBA_CH], t2[SBA_CH];
#pragma unroll
      for (int q = 0; q < SBA_CH; ++q) {
        a2[q] = m[(size_t)(j + q) * n + row];
        t2[q] = temp[j + q];
      }
#pragma unroll
      for (int q = 0; q < SBA_CH; ++q) dd += a[q] * t[q];
#pragma unroll
      for (int q = 0; q < SBA_CH; ++q) {
        a[q] = a2[q];
        t[q] = t2[q];
      }
    }
#pragma unroll
    for (int q = 0; q < SBA_CH; ++q) dd += a[q] * t[q];
  }
  for (; j < k; ++j) dd += m[(size_t)j * n + row] * temp[j];
  return dd;
}
// entry (row, col), row >= col, of the reduced matrix Am_BCinvBt (:499-506) from the partial sums in HBM:
// block (j,u) = [j == u] A_j - BCinvBt_[j][u], where the blocks on and below the diagonal are the transposed
// blocks on and above it (:495-497; diagonal blocks are transposed in place)
__device__ __forceinline__ double sba_reduced_entry(const SbaDev &d, int row, int col) {
  const int No = d.n_opt;
  const int j = row / 6, r = row - 6 * j, u = col / 6, c = col - 6 * u;  // u <= j
  const double *blk = d.S + 36 * (((size_t)u * No + j) * SBA_SG);      // block (u,j), read transposed
  const int idx = c * 6 + r;
  double s = 0.0;
#pragma unroll
  for (int g = 0; g < SBA_SG; ++g) s += blk[36 * g + idx];
  if (j != u) return -s;
  double A = 0.0;
#pragma unroll
  for (int g = 0; g < SBA_PG; ++g) A += d.Apart[48 * ((size_t)j * SBA_PG + g) + r * 6 + c];
  if (r == c) A += d.lambda * A;  // :433-441
  return A - s;
}

// the reduced system out of the partial sums: one lane per entry of the lower triangle (and of the right-hand
// side), so that the one-wavefront solve kernel below starts from n^2 coalesced loads instead of 16 n^2 serial ones
__global__ __launch_bounds__(64) void sba_assemble_kernel(SbaDev d) {
  const int n = 6 * d.n_opt;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n * n) {
    const int row = e / n, col = e - row * n;
    if (row >= col) d.G[e] = sba_reduced_entry(d, row, col);
  } else if (e < n * n + n) {
    const int q = e - n * n, j = q / 6, r = q - 6 * j;
    double a = 0.0, bcb = 0.0;
    for (int g = 0; g < SBA_PG; ++g) {
      a += d.Apart[48 * ((size_t)j * SBA_PG + g) + 36 + r];
      bcb += d.Apart[48 * ((size_t)j * SBA_PG + g) + 42 + r];
    }
    d.G[e] = a - bcb;  // :508-509
  }
}

__global__ __launch_bounds__(64) void sba_solve_kernel(SbaDev d, int iter) {
  extern __shared__ double sm[];
  const int No = d.n_opt, n = 6 * No, lane = threadIdx.x;
  double *m = sm;                  // n x n, M(i,j) = m[j * n + i], lower triangle
  double *y = sm + (size_t)n * n;  // n
  double *temp = y + n;            // n
  double *dg = temp + n;           // n: |diagonal| during the pivot pre-pass
  int *tr = (int *)(dg + n);       // n: transpositions
  int *sig = tr + n;               // n: position -> original index
  const long long t_0 = (long long)__builtin_amdgcn_s_memrealtime();
#define MM(i, j) m[(size_t)(j) * n + (i)]
  // ---- pivot order. Eigen's LDLT looks for the largest |diagonal| among rows k.. at step k, and in its
  // left-looking form the diagonal entries of rows > k still hold their ORIGINAL values at that point: the whole
  // transposition sequence depends on the original diagonal only. It is computed first, the matrix is then
  // assembled already permuted, and the factorisation runs without swaps — same operations on the same values.
  for (int e = lane; e < n; e += 64) {
    dg[e] = fabs(d.G[(size_t)e * n + e]);
    sig[e] = e;
  }
  __syncthreads();
  // Without exact ties among the |diagonal| values the selection order is simply their descending order: position k of
  // the permutation holds the k-th largest — every lane counts how many values beat its own (n <= 64: one value per
  // lane, broadcast lane by lane). n sequential steps with two reductions and a barrier each (below: the general form,
  // kept for ties and for n > 64) were a quarter of this kernel's time.
  bool ranked = false;
  if (n <= 64) {
    const double v = lane < n ? dg[lane] : -1.0;
    int rank = 0, tie = 0;
    for (int j = 0; j < n; ++j) {
      const double vj = __shfl(v, j);
      rank += (vj > v) ? 1 : 0;
      tie |= (vj == v && j != lane) ? 1 : 0;
    }
    if (!__any(tie && lane < n)) {
      if (lane < n) sig[rank] = lane;
      ranked = true;
    }
    __syncthreads();
  }
  for (int k = 0; !ranked && k < n; ++k) {
    // largest, first one on ties. |d| >= 0: the IEEE bit pattern orders like an unsigned integer, so two 32-bit
    // DPP max reductions and a ballot replace a 64-bit shuffle tree
    const int i0 = k + lane, i1 = k + lane + 64;
    const bool v0 = i0 < n, v1 = i1 < n;
    const unsigned long long k0 = v0 ? (unsigned long long)__double_as_longlong(dg[i0]) : 0ull;
    const unsigned long long k1 = v1 ? (unsigned long long)__double_as_longlong(dg[i1]) : 0ull;
    const bool second = v1 && k1 > k0;  // within a lane the first slot wins ties (smaller index)
    const unsigned long long key = second ? k1 : k0;
    const unsigned hi = sba_wave_max_u32((unsigned)(key >> 32));
    const bool c_hi = v0 && (unsigned)(key >> 32) == hi;
    const unsigned lo = sba_wave_max_u32(c_hi ? (unsigned)key : 0u);
    const bool cand = c_hi && (unsigned)key == lo;
    const unsigned long long b_first = __ballot(cand && !second), b_second = __ballot(cand && second);
    const int piv = b_first ? k + (int)__builtin_ctzll(b_first) : k + 64 + (int)__builtin_ctzll(b_second);
    if (lane == 0) {
      tr[k] = piv;
      if (piv != k) {
        const double t = dg[k];
        dg[k] = dg[piv];
        dg[piv] = t;
        const int ti = sig[k];
        sig[k] = sig[piv];
        sig[piv] = ti;
      }
    }
    __syncthreads();
  }
  // ---- assemble the permuted lower triangle: entry (i,j), i >= j, is the original lower-triangle entry between
  // sig[i] and sig[j] (the symmetric swaps of the reference only ever move lower-triangle storage)
#pragma unroll 8
  for (int e = lane; e < n * n; e += 64) {  // (independent iterations: the loads of several are in flight together)
    const int col = e / n, row = e - col * n;
    if (row >= col) {
      const int sr = sig[row], sc = sig[col];
      MM(row, col) = sr >= sc ? d.G[(size_t)sr * n + sc] : d.G[(size_t)sc * n + sr];
    }
  }
  for (int e = lane; e < n; e += 64) temp[e] = d.G[(size_t)n * n + e];
  __syncthreads();
  // P rhs: the transposition sequence applied to a vector is the gather by sig (sig went through the same swaps)
  for (int e = lane; e < n; e += 64) y[e] = temp[sig[e]];
  __syncthreads();
  const long long t_1 = (long long)__builtin_amdgcn_s_memrealtime();
  // ---- Eigen::LDLT (lower, in place) on the permuted matrix; every inner sum in the sequential order of the
  // CPU restatement
  for (int k = 0; k < n; ++k) {
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = lane; j < k; j += 64) temp[j] = MM(j, j) * MM(k, j);
      __syncthreads();
      // rows k (the diagonal entry) and k+1.. in one sweep: none of them reads what another writes
      for (int t = lane; t <= rs; t += 64) MM(k + t, k) -= sba_row_dot(m, temp, n, k + t, k);
      __syncthreads();
    }
    const double akk = MM(k, k);
    if (fabs(akk) > 0.0)
      for (int i = lane; i < rs; i += 64) MM(k + 1 + i, k) /= akk;
    __syncthreads();
  }
  // solve: x = P^T L^-T D^+ L^-1 (P rhs)
  // L sweep, column-oriented: entry i subtracts M(i,j) y_j in increasing j, as the row form does
  const double tol = 2.2250738585072014e-308;
  if (n <= 64) {
    // one entry per lane, in a register: the pivot entry of a step reaches the others by a lane broadcast — no LDS
    // round trip for y and no barrier per step (same subtractions in the same order as the loops below)
    double yi = lane < n ? y[lane] : 0.0;
    for (int j = 0; j < n; ++j) {
      const double yj = __shfl(yi, j);
      if (lane > j && lane < n) yi -= MM(lane, j) * yj;
    }
    if (lane < n) yi = fabs(MM(lane, lane)) > tol ? yi / MM(lane, lane) : 0.0;
    for (int q = n - 1; q > 0; --q) {
      const double yq = __shfl(yi, q);
      if (lane < q) yi -= MM(q, lane) * yq;
    }
    if (lane < n) y[lane] = yi;
    __syncthreads();
  } else {
  for (int j = 0; j < n; ++j) {
    const double yj = y[j];
    for (int i = j + 1 + lane; i < n; i += 64) y[i] -= MM(i, j) * yj;
    __syncthreads();
  }
  for (int i = lane; i < n; i += 64) y[i] = fabs(MM(i, i)) > tol ? y[i] / MM(i, i) : 0.0;
  __syncthreads();
  // L^T sweep, column-oriented as well: entry i subtracts M(q,i) y_q in DECREASING q (the CPU restatement's row
  // form goes up; the difference is the rounding of an n-term sum, ~1e-16 relative)
  for (int q = n - 1; q > 0; --q) {
    const double yq = y[q];
    for (int i = lane; i < q; i += 64) y[i] -= MM(q, i) * yq;
    __syncthreads();
  }
  }
  // P^T: scatter back through sig
  for (int e = lane; e < n; e += 64) temp[sig[e]] = y[e];
  __syncthreads();
  for (int e = lane; e < n; e += 64) y[e] = temp[e];
  __syncthreads();
#undef MM
  const long long t_2 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int e = lane; e < n; e += 64) d.x[e] = y[e];
  // pose updates (:560-576)
  for (int f = lane; f < d.n_frames; f += 64) {
    const int j = d.opt_index[f];
    if (j >= 0) sba_pose_update(d.T + 16 * (size_t)f, y + 6 * j);
  }
  // average pixel error of this iteration's linearisation point (:594-601)
  // (eight loads in flight per lane and step: one at a time, this loop alone took a third of the kernel)
  double e = 0.0;
  {
    int i = lane;
    for (; i + 7 * 64 < d.n_err; i += 8 * 64) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = d.err_part[i + 64 * q];
#pragma unroll
      for (int q = 0; q < 8; ++q) e += v[q];
    }
    for (; i < d.n_err; i += 64) e += d.err_part[i];
  }
  e = sba_wave_sum(e);
  if (lane == 0) {
    d.avg_err[iter] = sqrt(e / (double)(d.dyn ? d.dyn[1] : d.n_obs));
    if (e != e) atomicOr(d.flags, 2);
    // phase durations of the last iteration in 10 ns ticks (tests/measure/sbabench.py)
    d.flags[1] = (int)(t_1 - t_0);
    d.flags[2] = (int)(t_2 - t_1);
    d.flags[3] = (int)((long long)__builtin_amdgcn_s_memrealtime() - t_2);
  }
}

// ---- the same solve with the matrix in REGISTERS (n = N <= 64, compile-time) ---------------------------------------
// Lane i holds row i of the permuted lower triangle (entries 0..i) in N register pairs; every index below is a
// compile-time constant, so the factorisation is straight-line code: per (step k, column j < k) one lane broadcast of
// L(k,j) (two v_readlane), temp_j = D_j L(k,j), one multiply and one add per lane — no LDS round trip, no barrier, the
// same operations in the same order as sba_solve_kernel (and oracle_sba.c). 42 LDLT steps: 38 -> ~10 us. L is written
// to LDS once for the transposed sweep, which needs column access. Ties among the |diagonal| values (exact equality)
// take the sequential pivot pre-pass, as in the general kernel.
__device__ __forceinline__ double sba_rl(double v, int src) {
  int2 p = __builtin_bit_cast(int2, v);
  p.x = __builtin_amdgcn_readlane(p.x, src);
  p.y = __builtin_amdgcn_readlane(p.y, src);
  return __builtin_bit_cast(double, p);
}
// (one wavefront from here on: LDS accesses of a wavefront are executed in order, only the compiler has to be told)
#define SBA_WAVE_SYNC()                                  \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
    __builtin_amdgcn_wave_barrier();                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
  } while (0)
// a time stamp that the scheduler cannot move: taken after `dep` has been computed, nothing crosses it
__device__ __forceinline__ long long sba_stamp_after(double dep) {
  long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define SBA_SOLVE_WG 512
template <int N>
__global__ __launch_bounds__(SBA_SOLVE_WG) void sba_solve_reg_kernel(SbaDev d, int iter) {
  __shared__ double sL[N * N + N];  // first the reduced system (lower triangle, G(i,j) at [i * N + j]; rhs behind), later L
  __shared__ double s_dg[N];
  __shared__ int s_sig[N];
  __shared__ double s_Tpre[64 * 16];  // exp(log(T_f)) of the frames' poses, made by the third wavefront during the solve
  const int tid = threadIdx.x, lane = tid & 63;
  const long long t_0 = (long long)__builtin_amdgcn_s_memrealtime();
  // ---- the reduced system out of the partial sums (what sba_assemble_kernel does for the general kernel), by all
  // eight wavefronts straight into LDS: entry t of the packed lower triangle, then the right-hand side. An entry is two
  // sums of eight partials; a thread's (at most two) entries issue all their loads before anything waits: one exposure
  // of the memory latency instead of four.
  {
    constexpr int TRI = N * (N + 1) / 2, TOT = TRI + N, PER = (TOT + SBA_SOLVE_WG - 1) / SBA_SOLVE_WG;
    const int No = d.n_opt;
    const double *p0[PER], *p1[PER];
    int s1[PER], mode[PER], dst[PER];  // mode 0: -(sum p0); 1: sum p1 - sum p0; 2: the same with the damped diagonal; 3: rhs
    double v0[PER][SBA_SG], v1[PER][SBA_PG];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int t = tid + e * SBA_SOLVE_WG;
      mode[e] = -1;
      p0[e] = p1[e] = d.S;
      s1[e] = 0;
      dst[e] = 0;
      if (t < TRI) {
        int row = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        const int col = t - row * (row + 1) / 2;
        const int j = row / 6, r = row - 6 * j, u = col / 6, c = col - 6 * u;  // u <= j (sba_reduced_entry)
        p0[e] = d.S + 36 * (((size_t)u * No + j) * SBA_SG) + (c * 6 + r);      // block (u,j), read transposed
        p1[e] = d.Apart + 48 * ((size_t)j * SBA_PG) + (r * 6 + c);
        s1[e] = 48;
        mode[e] = j != u ? 0 : (r == c ? 2 : 1);
        dst[e] = row * N + col;
      } else if (t < TOT) {
        const int q = t - TRI, j = q / 6, r = q - 6 * j;
        p0[e] = d.Apart + 48 * ((size_t)j * SBA_PG) + 36 + r;  // a_j
        p1[e] = d.Apart + 48 * ((size_t)j * SBA_PG) + 42 + r;  // (B C^-1 b)_j
        s1[e] = 48;
        mode[e] = 3;
        dst[e] = N * N + q;
      }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int s0 = mode[e] == 3 ? 48 : 36;
#pragma unroll
      for (int g = 0; g < SBA_SG; ++g) v0[e][g] = p0[e][(size_t)s0 * g];
#pragma unroll
      for (int g = 0; g < SBA_PG; ++g) v1[e][g] = p1[e][(size_t)s1[e] * g];
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int g = 0; g < SBA_SG; ++g) a += v0[e][g];
#pragma unroll
      for (int g = 0; g < SBA_PG; ++g) b2 += v1[e][g];
      if (mode[e] == 2) b2 += d.lambda * b2;  // :433-441
      if (mode[e] >= 0) sL[dst[e]] = mode[e] == 0 ? -a : (mode[e] == 3 ? a - b2 : b2 - a);  // rhs: a - bcb (:508-509)
    }
  }
  const long long t_a = sba_stamp_after(0.0);
  __syncthreads();
  const long long t_b = sba_stamp_after(0.0);
  const bool pre_staged = d.n_frames <= 64;
  const bool deliver = d.res_host && pre_staged && iter == d.max_iter - 1;  // (uniform: see SbaDev::res_host)
  if (tid >= 192) return;
  if (tid >= 128) {
    // the third wavefront: the half of the pose update that does not need x — xi = log(T), exp(xi) (:563-566) — one
    // frame per lane, next to the factorisation; wavefront 0 picks the results up from LDS behind its solve
    if (pre_staged) {
      const int f = lane;
      if (f < d.n_frames && d.opt_index[f] >= 0) {
        double Tjw[16];
        sba_pose_pre(d.T + 16 * (size_t)f, Tjw);
#pragma unroll
        for (int q = 0; q < 16; ++q) s_Tpre[f * 16 + q] = Tjw[q];
      }
      __syncthreads();  // (with wavefront 0, below: the only two wavefronts that are left by then need not be — a
    }                   //  wavefront that has ended does not hold a barrier up)
    return;
  }
  if (tid >= 64) {
    // the second wavefront: average pixel error of this iteration's linearisation point (:594-601), next to the solve
    double e = 0.0;
    int i = lane;
    for (; i + 7 * 64 < d.n_err; i += 8 * 64) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = d.err_part[i + 64 * q];
#pragma unroll
      for (int q = 0; q < 8; ++q) e += v[q];
    }
    for (; i < d.n_err; i += 64) e += d.err_part[i];
    e = sba_wave_sum(e);
    if (lane == 0) {
      d.avg_err[iter] = sqrt(e / (double)(d.dyn ? d.dyn[1] : d.n_obs));
      if (e != e) atomicOr(d.flags, 2);
    }
    if (deliver) __syncthreads();  // (the error and its flag are part of what wavefront 0 sends to the host behind this barrier)
    return;
  }
  // ---- pivot order (see sba_solve_kernel): descending |diagonal| unless two are exactly equal
  {
    const double v = lane < N ? fabs(sL[lane * N + lane]) : -1.0;
    int rank = 0, tie = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double vj = sba_rl(v, j);
      rank += (vj > v) ? 1 : 0;
      tie |= (vj == v && j != lane) ? 1 : 0;
    }
    if (!__any(tie && lane < N)) {
      if (lane < N) s_sig[rank] = lane;
    } else {
      if (lane < N) {
        s_dg[lane] = v;
        s_sig[lane] = lane;
      }
      SBA_WAVE_SYNC();
      if (lane == 0)
        for (int k = 0; k < N; ++k) {  // Eigen's selection: the first of the largest among positions k.., then the swap
          int piv = k;
          for (int i = k + 1; i < N; ++i)
            if (s_dg[i] > s_dg[piv]) piv = i;
          const double t = s_dg[k];
          s_dg[k] = s_dg[piv];
          s_dg[piv] = t;
          const int ti = s_sig[k];
          s_sig[k] = s_sig[piv];
          s_sig[piv] = ti;
        }
    }
    SBA_WAVE_SYNC();
  }
  // ---- this lane's row of the permuted lower triangle, and its entry of P rhs
  // (sig[j] is lane j's `my`: a lane broadcast instead of an LDS read per column, and every load is issued before the
  // first one is waited for — the dependent LDS round trips of the first version were 4 us)
  const int my = lane < N ? s_sig[lane] : 0;
  double r[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int sj = __builtin_amdgcn_readlane(my, j);
    const int hi = my > sj ? my : sj, lo = my > sj ? sj : my;
    r[j] = sL[hi * N + lo];
  }
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = (lane < N && j <= lane) ? r[j] : 0.0;
  double y = lane < N ? sL[N * N + my] : 0.0;
  SBA_WAVE_SYNC();  // (sL is written again below)
  const long long t_1 = sba_stamp_after(r[0] + y);
  // ---- Eigen::LDLT (lower, in place), left-looking, fully unrolled. (temp[j] through LDS — one store by lane k, a
  // broadcast read by all — was measured slower than the lane broadcasts: 22 vs 17 us, two LDS round trips per step.)
  // Scheduled column by column: once column j is final its contribution L(i,j) * (D_j L(k,j)) goes into the accumulator of
  // every later column k — N - 1 - j independent multiply-adds per step instead of one chain of k dependent ones per
  // column. Each accumulator still receives its terms in increasing j and is subtracted from M(i,k) in one piece when
  // column k's turn comes: the same sums in the same order as the left-looking form (same bits), ~13 -> ~8 us.
  // (Finalising column j + 1 in front of the rest of column j's updates, to fill its chain of ~17 dependent instructions,
  // measured the same 9.9 us: 5 500 instructions of one wavefront at ~4.4 cycles each — two lane broadcasts, a wait state, a
  // multiply and an add per (j, k) — is what the factorisation costs.)
  double dgl = 0.0;  // this lane's final diagonal entry D(lane)
  double acc[N];     // acc[k] = sum_{j < k, final} M(lane,j) temp_k[j], temp_k[j] = D_j M(k,j)
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (j > 0 && lane >= j) r[j] -= acc[j];  // rows j (the diagonal entry) and j+1.. in one sweep
    const double ajj = sba_rl(r[j], j);
    if (lane == j) dgl = ajj;
    if (fabs(ajj) > 0.0 && lane > j) r[j] /= ajj;
    const double wj = ajj * r[j];  // lane k: D_j M(k,j) = temp[j] of step k
#pragma unroll
    for (int k = j + 1; k < N; ++k) acc[k] += r[j] * sba_rl(wj, k);
  }
  const long long t_f = sba_stamp_after(dgl + r[N - 1]);
  // ---- solve: x = P^T L^-T D^+ L^-1 (P rhs); the entry of lane i stays in a register
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double yj = sba_rl(y, j);
    if (lane > j && lane < N) y -= r[j] * yj;
  }
  const double tol = 2.2250738585072014e-308;
  if (lane < N) y = fabs(dgl) > tol ? y / dgl : 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (lane < N && j < lane) sL[lane * N + j] = r[j];
  SBA_WAVE_SYNC();
  for (int q = N - 1; q > 0; --q) {  // L^T sweep in DECREASING q, as sba_solve_kernel
    const double yq = sba_rl(y, q);  // (q is wave-uniform)
    if (lane < q) y -= sL[q * N + lane] * yq;
  }
  // P^T: scatter back through sig
  double *xs = s_dg;
  if (lane < N) xs[my] = y;
  SBA_WAVE_SYNC();
  const long long t_2 = sba_stamp_after(y);
  if (lane < N) d.x[lane] = xs[lane];
  // pose updates (:560-576)
  if (pre_staged) {
    __syncthreads();  // the third wavefront's exp(log(T_f)) are in LDS (it finished long ago)
    const int f = lane, j = f < d.n_frames ? d.opt_index[f] : -1;
    if (j >= 0) {
      double Tjw[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) Tjw[q] = s_Tpre[f * 16 + q];
      sba_pose_post(d.T + 16 * (size_t)f, Tjw, xs + 6 * j);
    }
  } else {
    for (int f = lane; f < d.n_frames; f += 64) {
      const int j = d.opt_index[f];
      if (j >= 0) sba_pose_update(d.T + 16 * (size_t)f, xs + 6 * j);
    }
  }
  if (deliver) {
    // the last iteration's solve: poses, errors, flags and counts are final (the launches behind this one move points only)
    SBA_WAVE_SYNC();  // (this wavefront's own pose stores above)
    for (int k = lane; k < d.res_words; k += 64)
      if (k != d.res_seq_word) d.res_host[k] = d.res_src[k];
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&d.res_host[d.res_seq_word], d.res_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (lane == 0) {
    d.flags[1] = (int)(t_1 - t_0);
    d.flags[2] = (int)(t_2 - t_1);
    d.flags[3] = (int)((long long)__builtin_amdgcn_s_memrealtime() - t_2);
    d.flags[4] = (int)(t_a - t_0);  // wavefront 0: its entries of the reduced system
    d.flags[5] = (int)(t_b - t_a);  // waiting for the other wavefronts
    d.flags[6] = (int)(t_1 - t_b);  // pivot order + rows into registers
    d.flags[7] = (int)(t_f - t_1);  // factorisation
  }
}

// ---- y_i and the point update (:537-556, :578-579), then C_i, b_i, C_i^-1 and the slot blocks (:207-471) ----------
// The point update of iteration k and the per-landmark linearisation of iteration k + 1 touch only landmark i (and the
// poses the solve of iteration k left): one launch instead of two. update != 0: X_i += y_i first; point != 0: then
// C_i, b_i (sums over the observations), damping, Eigen-order 3x3 LDLT inverse, C^-1 b, and per slot B_ji (Qij^T Rij of
// the LAST observation of landmark i in keyframe j, :315 / :410 assign), B_ji C_i^-1 (:471) and (B_ji C_i^-1) b_i (:473).
template <bool TLDS>
__global__ __launch_bounds__(64) void sba_update_point_kernel(SbaDev d, int update, int point) {
  __shared__ double sT[TLDS ? 16 * SBA_LDS_FRAMES : 2];
  __shared__ double sx[6 * SBA_MAX_OPT];
  const int lane = threadIdx.x, sub = lane & (SBA_LQ - 1);
  const long long st0 = SBA_TICK();
  const int M = d.dyn ? d.dyn[0] : d.M;
  if ((int)blockIdx.x * (64 / SBA_LQ) >= M) {  // (a launch sized by an upper bound of M)
    if (point && lane == 0) d.err_part[blockIdx.x] = 0.0;
    return;
  }
  const int i_raw = blockIdx.x * (64 / SBA_LQ) + lane / SBA_LQ;
  const bool live = i_raw < M;
  const int i = live ? i_raw : M - 1;  // a surplus quad repeats the last landmark and stores nothing (DPP needs all lanes)
  // (the landmark's own loads are issued next to the staging of poses and x: one round of latency instead of two)
  double X[3] = {d.X[3 * (size_t)i], d.X[3 * (size_t)i + 1], d.X[3 * (size_t)i + 2]};
  const int s0 = d.slot_ptr[i], s1 = d.slot_ptr[i + 1];
  const int o0 = d.obs_ptr[i], o1 = d.obs_ptr[i + 1];
  if (TLDS)
    for (int k = lane; k < 16 * d.n_frames; k += 64) sT[k] = d.T[k];
  if (update)
    for (int k = lane; k < 6 * d.n_opt; k += 64) sx[k] = d.x[k];
  // Everything a landmark's lane group will read next is known once the four pointers above are: the slot blocks of the
  // update, the landmark's observations (SBA_NPRE rounds of SBA_LQ: 24 observations cover a nine-keyframe stereo window) and
  // the slots' observation indices go out as ONE batch here — the first version met them as five to seven dependent rounds
  // of loads, each a microsecond on a wavefront with nothing else to do. Indices are clamped into the landmark's own range
  // (every landmark has observations; the arrays have a spare entry for a landmark without slots); what a lane does not own
  // is loaded and not used. The arithmetic and its order per lane are unchanged.
  constexpr int SBA_NPRE = 3;
  const int sA = s0 + sub;
  const bool hasA = sA < s1;
  const int sL = hasA ? sA : s0;  // (what is loaded for a lane without a slot: the landmark's first, or the spare entry)
  double preBC[18];
  int pre_j = 0, pre_bobs = o0;
  double preCb[3] = {0, 0, 0};
  if (update) {
    const double *BC = d.BCs + 18 * (size_t)sL;
#pragma unroll
    for (int k = 0; k < 18; ++k) preBC[k] = BC[k];
    pre_j = d.slot_j[sL];
#pragma unroll
    for (int c = 0; c < 3; ++c) preCb[c] = d.Cinvb[3 * (size_t)i + c];
  }
  double pre_px[SBA_NPRE][2];
  int pre_f[SBA_NPRE], pre_r[SBA_NPRE];
  if (point) {
#pragma unroll
    for (int q = 0; q < SBA_NPRE; ++q) {
      const int o = max(min(o0 + sub + SBA_LQ * q, o1 - 1), 0);
      pre_px[q][0] = d.obs_px[2 * o];
      pre_px[q][1] = d.obs_px[2 * o + 1];
      pre_f[q] = d.obs_frame[o];
      pre_r[q] = d.obs_right[o];
    }
    pre_bobs = d.slot_bobs[sL];
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);  // (the batch above stays above: the scheduler would sink the loads to their uses)
  const long long st1 = SBA_TICK();
  if (update) {
    double cbx[3] = {0, 0, 0};
    if (hasA) {
      const double *x = sx + 6 * pre_j;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) t += preBC[r * 3 + c] * x[r];
        cbx[c] += t;
      }
    }
    for (int s = sA + SBA_LQ; s < s1; s += SBA_LQ) {
      const double *BC = d.BCs + 18 * (size_t)s, *x = sx + 6 * d.slot_j[s];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) t += BC[r * 3 + c] * x[r];
        cbx[c] += t;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      X[c] += preCb[c] - sba_quad_sum(cbx[c]);
      if (live && sub == 0) d.X[3 * (size_t)i + c] = X[c];
    }
  }
  if (!point) return;
  const long long st2 = SBA_TICK();
  // the first slot's observation (index known since the batch): on its way while the observations are linearised
  const int ob = hasA ? pre_bobs : o0;
  const double slot_px[2] = {d.obs_px[2 * ob], d.obs_px[2 * ob + 1]};
  const int slot_f = d.obs_frame[ob], slot_r = d.obs_right[ob];
  __builtin_amdgcn_sched_barrier(0);
  double Cu[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, err = 0.0;  // C_i: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
  auto one_obs = [&](const double px[2], int f, int right) {
    SbaObs L;
    sba_linearize(d, TLDS ? sT + 16 * f : d.T + 16 * (size_t)f, X, px, right, L);
    // calc_Rij_t_Rij_weight (:911-930), b_i += -weight * (Rij^T rij)
    int q = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = r; c < 3; ++c) Cu[q++] += L.w * (L.R[r] * L.R[c] + L.R[3 + r] * L.R[3 + c]);
#pragma unroll
    for (int r = 0; r < 3; ++r) b[r] += -(L.w * (L.R[r] * L.r[0] + L.R[3 + r] * L.r[1]));
    err += L.r[0] * L.r[0] + L.r[1] * L.r[1];
  };
#pragma unroll
  for (int q = 0; q < SBA_NPRE; ++q)
    if (o0 + sub + SBA_LQ * q < o1) one_obs(pre_px[q], pre_f[q], pre_r[q]);
  for (int o = o0 + sub + SBA_LQ * SBA_NPRE; o < o1; o += SBA_LQ) {
    const double px[2] = {d.obs_px[2 * o], d.obs_px[2 * o + 1]};
    one_obs(px, d.obs_frame[o], d.obs_right[o]);
  }
  const long long st3 = SBA_TICK();
#pragma unroll
  for (int k = 0; k < 6; ++k) Cu[k] = sba_quad_sum(Cu[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) b[k] = sba_quad_sum(b[k]);
  err = sba_quad_sum(err);
  double C[9] = {Cu[0], Cu[1], Cu[2], Cu[1], Cu[3], Cu[4], Cu[2], Cu[4], Cu[5]};
#pragma unroll
  for (int k = 0; k < 3; ++k) C[k * 4] += d.lambda * C[k * 4];  // :454-456
  double Ci[9];
  sba_inv3_ldlt(C, Ci);
  if (live && sub == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      d.Cinvb[3 * (size_t)i + r] = Ci[r * 3] * b[0] + (Ci[r * 3 + 1] * b[1] + Ci[r * 3 + 2] * b[2]);
      d.b[3 * (size_t)i + r] = b[r];
    }
  }
  const long long st4 = SBA_TICK();
  auto one_slot = [&](int s, const double px[2], int f, int right) {
    SbaObs L;
    sba_linearize(d, TLDS ? sT + 16 * f : d.T + 16 * (size_t)f, X, px, right, L);
    double B[18], BC[18];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) B[r * 3 + c] = L.w * (L.Q[r] * L.R[c] + L.Q[6 + r] * L.R[3 + c]);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) BC[r * 3 + c] = B[r * 3] * Ci[c] + (B[r * 3 + 1] * Ci[3 + c] + B[r * 3 + 2] * Ci[6 + c]);  // :471
    if (live) {
      double *Bo = d.Bs + 18 * (size_t)s, *BCo = d.BCs + 18 * (size_t)s, *BCbo = d.BCb + 6 * (size_t)s;
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        Bo[k] = B[k];
        BCo[k] = BC[k];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) BCbo[r] = BC[r * 3] * b[0] + (BC[r * 3 + 1] * b[1] + BC[r * 3 + 2] * b[2]);  // :473
    }
  };
  if (hasA) one_slot(sA, slot_px, slot_f, slot_r);
  for (int s = sA + SBA_LQ; s < s1; s += SBA_LQ) {
    const int o = d.slot_bobs[s];
    const double px[2] = {d.obs_px[2 * o], d.obs_px[2 * o + 1]};
    one_slot(s, px, d.obs_frame[o], d.obs_right[o]);
  }
  const long long st5 = SBA_TICK();
  SBA_STAMP_MAX(10, st1, st0);
  SBA_STAMP_MAX(11, st2, st1);
  SBA_STAMP_MAX(12, st3, st2);
  SBA_STAMP_MAX(13, st4, st3);
  SBA_STAMP_MAX(14, st5, st4);
  const double e = sba_wave_sum(live && sub == 0 ? err : 0.0);
  if (lane == 0) d.err_part[blockIdx.x] = e;
}

// ---- host side ---------------------------------------------------------------------
struct vo_sba_state {
  void *dev;
  size_t cap;
  int phase_ticks[8];  // solve kernel of the last iteration: assembly, LDLT + solve, pose update; [3..5] parts of the assembly (10 ns ticks)
  void *stage;         // pinned host staging: the problem goes up in one copy, the results come back through it
  size_t stage_cap;
};
extern "C" int vo_debug_sba_phases(vo_ctx *c, int out[8]) {
  if (!c || !c->sba) return VO_ERR_INVALID;
  memcpy(out, c->sba->phase_ticks, sizeof(int) * 8);
  return VO_OK;
}

void vo_sba_free(vo_ctx *c) {
  if (c->sba) {
    if (c->sba->dev) (void)hipFree(c->sba->dev);
    if (c->sba->stage) (void)hipHostFree(c->sba->stage);
    delete c->sba;
    c->sba = nullptr;
  }
}

namespace {
struct Arena {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};
}  // namespace

size_t vo_sba_place_work(SbaDev *d, uint8_t *base, size_t off, size_t M, size_t ns, int No, int max_iter, int n_err) {
  Arena ar;
  ar.off = off;
  const size_t oCinvb = ar.take(sizeof(double) * 3 * (M + 1)), oB = ar.take(sizeof(double) * 3 * (M + 1));
  const size_t oErr = ar.take(sizeof(double) * (size_t)(n_err + 1));
  const size_t oBs = ar.take(sizeof(double) * 18 * (ns + 1)), oBCs = ar.take(sizeof(double) * 18 * (ns + 1));
  const size_t oBCb = ar.take(sizeof(double) * 6 * (ns + 1));
  const size_t oAp = ar.take(sizeof(double) * 48 * SBA_PG * (size_t)(No + 1));
  const size_t oS = ar.take(sizeof(double) * 36 * SBA_SG * ((size_t)No * No + 1));
  const size_t oG = ar.take(sizeof(double) * ((size_t)36 * No * No + 6 * No + 1));
  const size_t ox = ar.take(sizeof(double) * 6 * (size_t)(No + 1));
  (void)max_iter;
  if (base) {
    d->Cinvb = (double *)(base + oCinvb);
    d->b = (double *)(base + oB);
    d->err_part = (double *)(base + oErr);
    d->n_err = n_err;
    d->Bs = (double *)(base + oBs);
    d->BCs = (double *)(base + oBCs);
    d->BCb = (double *)(base + oBCb);
    d->Apart = (double *)(base + oAp);
    d->S = (double *)(base + oS);
    d->G = (double *)(base + oG);
    d->x = (double *)(base + ox);
  }
  return ar.off;
}

static bool sba_reg_solve(const vo_ctx *c, int No) { return No >= 1 && No <= 8 && !c->dbg[VO_DBG_SBA_LDS_SOLVE]; }
bool vo_sba_delivers_result(const vo_ctx *c, const SbaDev &d) {
  return d.res_host && d.max_iter > 0 && sba_reg_solve(c, d.n_opt) && d.n_frames <= 64;
}

// three launches per iteration in the steady-state window (four otherwise): [update of the previous iteration +
// per-landmark linearisation] -> [pose sums + Schur blocks] -> [assembly ->] solve; one last update behind the loop
int vo_sba_enqueue_iterations(vo_ctx *c, const SbaDev &d, int max_iter) {
  const int No = d.n_opt, n = 6 * No, n_err = d.n_err;
  hipStream_t s = c->stream;
  const size_t lds = sizeof(double) * ((size_t)n * n + 3 * (size_t)n) + sizeof(int) * 2 * (size_t)n + 64;
  if (lds > 64 * 1024)
    VO_CHECK_HIP(c, hipFuncSetAttribute((const void *)sba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool t_lds = d.n_frames <= SBA_LDS_FRAMES;
  auto launch_update_point = [&](int update, int point) {
    if (t_lds)
      hipLaunchKernelGGL(sba_update_point_kernel<true>, dim3(n_err), dim3(64), 0, s, d, update, point);
    else
      hipLaunchKernelGGL(sba_update_point_kernel<false>, dim3(n_err), dim3(64), 0, s, d, update, point);
  };
  // every window of up to ten keyframes (1..8 optimised poses) takes the register solve: three launches per iteration
  // (vo_debug_set VO_DBG_SBA_LDS_SOLVE: the general kernel for every n, an A/B switch)
  const bool reg_solve = sba_reg_solve(c, No);
  for (int iter = 0; iter < max_iter; ++iter) {
    launch_update_point(iter > 0 ? 1 : 0, 1);
    if (No > 0) {
      hipLaunchKernelGGL(sba_pose_schur_kernel, dim3(No * SBA_PG + (No * (No + 1) / 2) * SBA_SG), dim3(SBA_WG), 0, s, d);
      if (!reg_solve) hipLaunchKernelGGL(sba_assemble_kernel, dim3((n * n + n + 63) / 64), dim3(64), 0, s, d);
    }
    if (reg_solve) {  // (assembles the reduced system itself)
      switch (No) {
        case 1: hipLaunchKernelGGL(sba_solve_reg_kernel<6>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        case 2: hipLaunchKernelGGL(sba_solve_reg_kernel<12>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        case 3: hipLaunchKernelGGL(sba_solve_reg_kernel<18>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        case 4: hipLaunchKernelGGL(sba_solve_reg_kernel<24>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        case 5: hipLaunchKernelGGL(sba_solve_reg_kernel<30>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        case 6: hipLaunchKernelGGL(sba_solve_reg_kernel<36>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        case 7: hipLaunchKernelGGL(sba_solve_reg_kernel<42>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
        default: hipLaunchKernelGGL(sba_solve_reg_kernel<48>, dim3(1), dim3(SBA_SOLVE_WG), 0, s, d, iter); break;
      }
    } else
      hipLaunchKernelGGL(sba_solve_kernel, dim3(1), dim3(64), lds, s, d, iter);
  }
  if (max_iter > 0) launch_update_point(1, 0);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

extern "C" int vo_sba_solve(vo_ctx *c, const vo_sba_problem *p, double *T_jw, const int32_t *opt_index, double *X,
                            const int32_t *obs_ptr, const int32_t *obs_frame, const uint8_t *obs_right,
                            const double *obs_px, double *avg_err) {
  if (!c || !p || !T_jw || !opt_index || !X || !obs_ptr || !obs_frame || !obs_right || !obs_px) return VO_ERR_INVALID;
  static const bool trace = getenv("VO_SVO_TRACE") != nullptr;
  static double tt[8];
  static int n_calls;
  auto now_us = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
  };
  double t_last = trace ? now_us() : 0.0;
#define SBA_T(k)                    \
  do {                              \
    if (trace) {                    \
      const double now_ = now_us(); \
      tt[k] += now_ - t_last;       \
      t_last = now_;                \
    }                               \
  } while (0)
  const int Nf = p->n_frames, No = p->n_opt, M = p->n_points, nobs = p->n_obs;
  if (Nf <= 0 || No < 0 || M <= 0 || nobs <= 0 || p->max_iter < 0) VO_FAIL(c, VO_ERR_INVALID, "empty BA problem");
  if (No > SBA_MAX_OPT) VO_FAIL(c, VO_ERR_CAPACITY, "n_opt=%d exceeds %d optimised poses", No, SBA_MAX_OPT);
  // ---- validate the lists (device kernels index with them) and derive the gather lists ----
  if (obs_ptr[0] != 0 || obs_ptr[M] != nobs) VO_FAIL(c, VO_ERR_SIZE, "obs_ptr does not span n_obs");
  std::vector<int> seen_opt(No, 0);
  for (int f = 0; f < Nf; ++f) {
    if (opt_index[f] < -1 || opt_index[f] >= No) VO_FAIL(c, VO_ERR_INVALID, "opt_index[%d] out of range", f);
    if (opt_index[f] >= 0 && seen_opt[opt_index[f]]++) VO_FAIL(c, VO_ERR_INVALID, "optimised pose index used twice");
  }
  // Two passes over the observations, no per-list containers: pass 1 validates and counts (slots per landmark,
  // observations / slots per optimised pose, pairs per block), pass 2 writes every list at its final place inside ONE
  // pinned staging block laid out like the device arena, which then goes up in a single copy.
  std::vector<int> pose_obs_cnt(No + 1, 0), pose_slot_cnt(No + 1, 0), pair_cnt((size_t)No * No + 1, 0);
  int ns = 0;
  for (int i = 0; i < M; ++i) {
    if (obs_ptr[i + 1] < obs_ptr[i]) VO_FAIL(c, VO_ERR_SIZE, "obs_ptr not monotone");
    int lj[2 * SBA_MAX_OPT + 64], nl = 0;  // optimised-pose index of this landmark's slots, in list order
    for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
      if (obs_frame[o] < 0 || obs_frame[o] >= Nf) VO_FAIL(c, VO_ERR_INVALID, "obs_frame[%d] out of range", o);
      if (obs_right[o] && !p->stereo) VO_FAIL(c, VO_ERR_INVALID, "right-image observation in a mono problem");
      const int j = opt_index[obs_frame[o]];
      if (j >= 0) {
        ++pose_obs_cnt[j];
        if (!obs_right[o]) {
          if (nl >= (int)(sizeof(lj) / sizeof(lj[0]))) VO_FAIL(c, VO_ERR_CAPACITY, "a landmark with more than %d slots", nl);
          lj[nl++] = j;
          ++pose_slot_cnt[j];
        }
      }
    }
    for (int a2 = 0; a2 < nl; ++a2)
      for (int b2 = a2; b2 < nl; ++b2) ++pair_cnt[(size_t)lj[a2] * No + lj[b2]];
    ns += nl;
  }
  size_t n_pose_obs = 0, n_pairs = 0;
  for (int j = 0; j < No; ++j) n_pose_obs += (size_t)pose_obs_cnt[j];
  for (size_t jk = 0; jk < (size_t)No * No; ++jk) n_pairs += (size_t)pair_cnt[jk];

  SBA_T(0);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  // ---- device arena (inputs first: they are one contiguous upload) ----
  Arena ar;
  const size_t oT = ar.take(sizeof(double) * 16 * Nf), oOpt = ar.take(sizeof(int) * Nf), oX = ar.take(sizeof(double) * 3 * M);
  const size_t oOp = ar.take(sizeof(int) * (M + 1)), oOf = ar.take(sizeof(int) * nobs), oOr = ar.take((size_t)nobs);
  const size_t oPx = ar.take(sizeof(double) * 2 * nobs);
  const size_t oSp = ar.take(sizeof(int) * (M + 1)), oSo = ar.take(sizeof(int) * (ns + 1)), oSj = ar.take(sizeof(int) * (ns + 1));
  const size_t oSb = ar.take(sizeof(int) * (ns + 1)), oSl = ar.take(sizeof(int) * (ns + nobs + 1));
  const size_t oPop = ar.take(sizeof(int) * (No + 1)), oPo = ar.take(sizeof(int) * (n_pose_obs + 1));
  const size_t oPl = ar.take(sizeof(int) * (n_pose_obs + 1)), oOfr = ar.take(sizeof(int) * (No + 1));
  const size_t oPsp = ar.take(sizeof(int) * (No + 1)), oPs = ar.take(sizeof(int) * (ns + 1));
  const size_t oPp = ar.take(sizeof(int) * ((size_t)No * No + 1)), oPa = ar.take(sizeof(int) * (n_pairs + 1));
  const size_t oPb = ar.take(sizeof(int) * (n_pairs + 1));
  const size_t oFl = ar.take(sizeof(int) * 16), oAvg = ar.take(sizeof(double) * (p->max_iter + 1));
  const size_t in_bytes = ar.off;  // everything up to here comes from the host
  const int n_err = (M + 64 / SBA_LQ - 1) / (64 / SBA_LQ);  // workgroups of the point kernel
  SbaDev d;
  memset(&d, 0, sizeof(d));
  ar.off = vo_sba_place_work(&d, nullptr, ar.off, (size_t)M, (size_t)ns, No, p->max_iter, n_err);
  if (!c->sba) c->sba = new vo_sba_state{nullptr, 0, {0, 0, 0}, nullptr, 0};
  if (c->sba->cap < ar.off) {
    if (c->sba->dev) (void)hipFree(c->sba->dev);
    c->sba->dev = nullptr;
    c->sba->cap = 0;
    VO_CHECK_HIP(c, vo_dev_malloc(c, &c->sba->dev, ar.off + (ar.off >> 2)));
    c->sba->cap = ar.off + (ar.off >> 2);
  }
  const size_t out_bytes = sizeof(double) * (16 * (size_t)Nf + 3 * (size_t)M + (size_t)p->max_iter + 1) + 64;
  const size_t stage_need = in_bytes > out_bytes ? in_bytes : out_bytes;
  hipStream_t s = c->stream;
  // (the staging block is free: the previous solve ended with a synchronisation behind its read-back)
  if (c->sba->stage_cap < stage_need) {
    if (c->sba->stage) (void)hipHostFree(c->sba->stage);
    c->sba->stage = nullptr;
    c->sba->stage_cap = 0;
    VO_CHECK_HIP(c, vo_host_malloc(c, &c->sba->stage, stage_need + (stage_need >> 2), hipHostMallocDefault));
    c->sba->stage_cap = stage_need + (stage_need >> 2);
  }
  SBA_T(3);
  uint8_t *base = (uint8_t *)c->sba->dev;
  uint8_t *hs = (uint8_t *)c->sba->stage;
  memcpy(hs + oT, T_jw, sizeof(double) * 16 * Nf);
  memcpy(hs + oOpt, opt_index, sizeof(int) * Nf);
  memcpy(hs + oX, X, sizeof(double) * 3 * M);
  memcpy(hs + oOp, obs_ptr, sizeof(int) * (M + 1));
  memcpy(hs + oOf, obs_frame, sizeof(int) * nobs);
  memcpy(hs + oOr, obs_right, (size_t)nobs);
  memcpy(hs + oPx, obs_px, sizeof(double) * 2 * nobs);
  memset(hs + oFl, 0, sizeof(int) * 16);
  memset(hs + oAvg, 0, sizeof(double) * (p->max_iter + 1));
  SBA_T(4);
  int *h_slot_ptr = (int *)(hs + oSp), *h_slot_obs = (int *)(hs + oSo), *h_slot_j = (int *)(hs + oSj);
  int *h_slot_bobs = (int *)(hs + oSb), *h_slot_lm = (int *)(hs + oSl);
  int *h_pose_obs_ptr = (int *)(hs + oPop), *h_pose_obs = (int *)(hs + oPo), *h_pose_lm = (int *)(hs + oPl);
  int *h_opt_frame = (int *)(hs + oOfr);
  for (int f = 0; f < Nf; ++f)
    if (opt_index[f] >= 0) h_opt_frame[opt_index[f]] = f;
  int *h_pose_slot_ptr = (int *)(hs + oPsp), *h_pose_slot = (int *)(hs + oPs);
  int *h_pair_ptr = (int *)(hs + oPp), *h_pair_a = (int *)(hs + oPa), *h_pair_b = (int *)(hs + oPb);
  h_pose_obs_ptr[0] = h_pose_slot_ptr[0] = 0;
  for (int j = 0; j < No; ++j) {
    h_pose_obs_ptr[j + 1] = h_pose_obs_ptr[j] + pose_obs_cnt[j];
    h_pose_slot_ptr[j + 1] = h_pose_slot_ptr[j] + pose_slot_cnt[j];
  }
  h_pair_ptr[0] = 0;
  for (size_t jk = 0; jk < (size_t)No * No; ++jk) h_pair_ptr[jk + 1] = h_pair_ptr[jk] + pair_cnt[jk];
  {  // pass 2 (cursors start at the list heads; every list comes out in landmark order, as the nested containers gave it)
    std::vector<int> cur_po(h_pose_obs_ptr, h_pose_obs_ptr + No), cur_ps(h_pose_slot_ptr, h_pose_slot_ptr + No);
    std::vector<int> cur_pair(h_pair_ptr, h_pair_ptr + (size_t)No * No);
    int sidx = 0;
    h_slot_ptr[0] = 0;
    for (int i = 0; i < M; ++i) {
      const int s0 = sidx;
      for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
        h_slot_lm[ns + o] = i;  // (second half of the array: the landmark of every observation)
        const int j = opt_index[obs_frame[o]];
        if (j < 0) continue;
        h_pose_lm[cur_po[j]] = i;
        h_pose_obs[cur_po[j]++] = o;
        if (obs_right[o]) continue;
        // slot: left observation in an optimised keyframe; its B block is that of the LAST observation of this
        // landmark in the same keyframe (:315 / :410 assign, they do not accumulate)
        int last = o;
        for (int o2 = o + 1; o2 < obs_ptr[i + 1]; ++o2)
          if (opt_index[obs_frame[o2]] == j) last = o2;
        h_pose_slot[cur_ps[j]++] = sidx;
        h_slot_obs[sidx] = o;
        h_slot_j[sidx] = j;
        h_slot_bobs[sidx] = last;
        h_slot_lm[sidx] = i;
        ++sidx;
      }
      // pairs (:475-491): slots a, b of one landmark with list position b >= a contribute to block (j_a, j_b)
      for (int a2 = s0; a2 < sidx; ++a2)
        for (int b2 = a2; b2 < sidx; ++b2) {
          const int q = cur_pair[(size_t)h_slot_j[a2] * No + h_slot_j[b2]]++;
          h_pair_a[q] = a2;
          h_pair_b[q] = b2;
        }
      h_slot_ptr[i + 1] = sidx;
    }
  }
  SBA_T(5);
  VO_CHECK_HIP(c, hipMemcpyAsync(base, hs, in_bytes, hipMemcpyHostToDevice, s));
  SBA_T(6);
  vo_sba_place_work(&d, base, in_bytes, (size_t)M, (size_t)ns, No, p->max_iter, n_err);
  d.n_frames = Nf;
  d.n_opt = No;
  d.M = M;
  d.n_obs = nobs;
  d.n_slots = ns;
  d.stereo = p->stereo;
  d.max_iter = p->max_iter;
  for (int k = 0; k < 4; ++k) {
    d.Kl[k] = p->Kl[k];
    d.Kr[k] = p->stereo ? p->Kr[k] : p->Kl[k];
  }
  // geometry::inverseSE3(T_lr) (:175, geometry_library.cpp:561-567)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) d.R_rl[i * 3 + j] = p->stereo ? p->T_lr[j * 4 + i] : (i == j ? 1.0 : 0.0);
  for (int i = 0; i < 3; ++i)
    d.t_rl[i] = p->stereo ? (-d.R_rl[i * 3]) * p->T_lr[3] + ((-d.R_rl[i * 3 + 1]) * p->T_lr[7] + (-d.R_rl[i * 3 + 2]) * p->T_lr[11])
                          : 0.0;
  d.thres_huber = p->thres_huber;
  d.lambda = 0.00001;  // :186
  d.T = (double *)(base + oT);
  d.opt_index = (const int *)(base + oOpt);
  d.X = (double *)(base + oX);
  d.obs_ptr = (const int *)(base + oOp);
  d.obs_frame = (const int *)(base + oOf);
  d.obs_right = base + oOr;
  d.obs_px = (const double *)(base + oPx);
  d.slot_ptr = (const int *)(base + oSp);
  d.slot_obs = (const int *)(base + oSo);
  d.slot_j = (const int *)(base + oSj);
  d.slot_bobs = (const int *)(base + oSb);
  d.slot_lm = (const int *)(base + oSl);
  d.pose_obs_ptr = (const int *)(base + oPop);
  d.pose_obs = (const int *)(base + oPo);
  d.pose_lm = (const int *)(base + oPl);
  d.opt_frame = (const int *)(base + oOfr);
  d.pose_slot_ptr = (const int *)(base + oPsp);
  d.pose_slot = (const int *)(base + oPs);
  d.pair_ptr = (const int *)(base + oPp);
  d.pair_a = (const int *)(base + oPa);
  d.pair_b = (const int *)(base + oPb);
  d.avg_err = (double *)(base + oAvg);
  d.flags = (int *)(base + oFl);
  SBA_T(1);
  vo_prof_begin(c, VO_K_AUX);
  {
    const int rc_it = vo_sba_enqueue_iterations(c, d, p->max_iter);
    if (rc_it < 0) return rc_it;
  }
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  int flags[16] = {0};  // [0] error bits, [1..3] phase ticks of the solve kernel, [4..] -DSBA_STAMP phase maxima
  std::vector<double> errs(p->max_iter + 1, 0.0);
  {  // results through the pinned staging block (its upload half has been consumed: the copies are stream-ordered)
    double *o_T = (double *)hs, *o_X = o_T + 16 * (size_t)Nf, *o_e = o_X + 3 * (size_t)M;
    int *o_f = (int *)(o_e + p->max_iter + 1);
    VO_CHECK_HIP(c, hipMemcpyAsync(o_T, base + oT, sizeof(double) * 16 * Nf, hipMemcpyDeviceToHost, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(o_X, base + oX, sizeof(double) * 3 * M, hipMemcpyDeviceToHost, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(o_e, base + oAvg, sizeof(double) * (p->max_iter + 1), hipMemcpyDeviceToHost, s));
    VO_CHECK_HIP(c, hipMemcpyAsync(o_f, base + oFl, sizeof(flags), hipMemcpyDeviceToHost, s));
    VO_CHECK_HIP(c, hipStreamSynchronize(s));
    memcpy(T_jw, o_T, sizeof(double) * 16 * Nf);
    memcpy(X, o_X, sizeof(double) * 3 * M);
    memcpy(errs.data(), o_e, sizeof(double) * (p->max_iter + 1));
    memcpy(flags, o_f, sizeof(flags));
  }
  SBA_T(2);
  if (trace && (++n_calls % 10) == 0)
    fprintf(stderr, "[sba] per call (us): lists %.0f  upload %.0f  kernels+d2h %.0f | solve kernel, last iteration (us): pivot+assembly %.1f  "
                    "LDLT+solve %.1f  pose update+error %.1f\n", tt[0] / n_calls, tt[1] / n_calls, tt[2] / n_calls, flags[1] * 0.01, flags[2] * 0.01,
            flags[3] * 0.01);
#ifdef SBA_STAMP
  if (trace && (n_calls % 10) == 0)
    fprintf(stderr, "[sba] stamps, max over workgroups (us): pose head %.1f loop %.1f reduce %.1f | schur head %.1f loop %.1f reduce %.1f | point head %.1f update %.1f obs %.1f inv %.1f slots %.1f\n",
            flags[4] * 0.01, flags[5] * 0.01, flags[6] * 0.01, flags[7] * 0.01, flags[8] * 0.01, flags[9] * 0.01, flags[10] * 0.01,
            flags[11] * 0.01, flags[12] * 0.01, flags[13] * 0.01, flags[14] * 0.01);
#endif
  if (trace && (n_calls % 10) == 0)
    fprintf(stderr, "[sba] upload split (us): arena %.0f  memcpy to staging %.0f  lists pass 2 + order %.0f  h2d call %.0f  rest %.0f  (pairs %zu, slots %d, %zu KB)\n",
            tt[3] / n_calls, tt[4] / n_calls, tt[5] / n_calls, tt[6] / n_calls, tt[1] / n_calls, n_pairs, ns, in_bytes >> 10);
  if (avg_err)
    for (int k = 0; k < p->max_iter; ++k) avg_err[k] = errs[k];
  memcpy(c->sba->phase_ticks, flags + 1, sizeof(int) * 8);
  if (flags[0] & 1) VO_FAIL(c, VO_ERR_LBA_NAN, "In LBA, pose becomes nan!");
  if (flags[0] & 2) VO_FAIL(c, VO_ERR_LBA_NAN, "Local BA NAN!");
  if (p->max_iter == 0) return 1;
  return errs[p->max_iter - 1] <= 1.0 ? 1 : 0;  // THRES_SUCCESS_AVG_ERROR (:158, :599-601)
}
