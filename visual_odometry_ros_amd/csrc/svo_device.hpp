// svo_device.hpp — what the StereoVO loop shares between its driver (stereo_vo.hip) and the BA launch (gn_pose.hip), whose
// epilogue builds the next track set: the device-side structs and mapping::triangulateDLT (triangulate_3d.cpp:91-130).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SvoCam {  // what mapping::triangulateDLT needs (triangulate_3d.cpp:91-130), pixel-independent part precomputed
  float P10[12];  // [K1 * R10, K1 * t10], row-major 3x4
  float R10[9], t10[3];
  float K0[4], K1[4];
};

struct SvoTrackSet {  // device: stframe->getPtsSeen() (left / right) + related landmarks, one entry per landmark
  float *pts_l, *pts_r;  // [cap][2]
  float *Xw;             // [cap][3] lm->get3DPoint() (world frame)
  uint8_t *flags;        // [cap] VO_LM_TRIANGULATED | VO_LM_DROPPED | VO_LM_KF_MEMBER
  int32_t *ids;          // [cap] lm->getID()
};

struct SvoHdr {  // written by the BA launch's epilogue (gn_pose.hip): device copy and pinned host copy
  int n_surv, n_kf_tracked, n_new, n_next, n_emit, overflow;
  int seq;       // host copy: written last
  int id_min;    // first (= smallest) landmark id of the next track set
};


// What the BA launch needs to leave the NEXT track set behind (stereo_vo.cpp:670 lmtrack_final in index order, :714-739 the
// new landmarks of step [10], :752 setStereoPtsSeenAndRelatedLandmarks): filled by the StereoVO driver, consumed by
// gn_pose_kernel<true> in frame mode. The DLT depth test of every bin's candidate runs on extra workgroups of the same
// launch (speculatively, next to the iterations); the epilogue looks the emitted candidates up.
struct VoAdvArgs {
  int on;
  SvoTrackSet cur, nxt;
  int *acc_bin;      // [bins] workers -> epilogue: the candidate of bin j passes `Xl(2) > 0 && Xr(2) > 0` (write-through)
  uint8_t *accept;   // [emitted] out: became a landmark (inspection: vo_svo_get_new_points)
  int *dlt_done;     // cumulative count of finished worker wavefronts; dlt_target = what it reads when this frame's have
  int dlt_target;
  SvoCam cam;
  int id_base, cap;
  SvoHdr *hdr_dev, *hdr_host;
};

#ifdef __HIPCC__
// ---- device: Eigen::JacobiSVD<MatrixXf>(M, ComputeFullV) of a 4x4 and the DLT around it -----------------------------
// Same operations in the same order as oracle/oracle_vo.c (which says what of Eigen 3.4.0 it restates); one lane per
// point, the two 4x4 matrices in registers (every index below is a compile-time constant after unrolling).
__device__ __forceinline__ void svo_rot(float &x, float &y, float c, float s) {
  const float xi = x, yi = y;
  x = c * xi + s * yi;
  y = -s * xi + c * yi;
}

// V's column that belongs to the smallest singular value, as JacobiSVD leaves it in column 3 after its sort
static __device__ void svo_svd4_nullvec(const float (&M)[16], float (&v)[4]) {
  const float FMIN = 1.17549435e-38f, FEPS = 1.1920929e-07f, FMAX = 3.40282347e+38f;
  float W[16], V[16];
  float scale = 0.0f;
  bool finite = true;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float a = fabsf(M[i]);
    if (!(a <= FMAX)) finite = false;
    if (a > scale) scale = a;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  if (!finite) {  // Eigen: InvalidInput, V unset; here the identity (as the oracle)
    v[0] = v[1] = v[2] = 0.0f;
    v[3] = 1.0f;
    return;
  }
  if (scale == 0.0f) scale = 1.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) W[i] = M[i] / scale;
  const float precision = 2.0f * FEPS;
  float max_diag = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (fabsf(W[i * 5]) > max_diag) max_diag = fabsf(W[i * 5]);
  for (int sweeps = 1;; ++sweeps) {
    bool finished = true;
#pragma unroll
    for (int p = 1; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < p; ++q) {
        const float pm = precision * max_diag;
        const float threshold = FMIN > pm ? FMIN : pm;
        if (fabsf(W[p * 4 + q]) > threshold || fabsf(W[q * 4 + p]) > threshold) {
          finished = false;
          float m00 = W[p * 4 + p], m01 = W[p * 4 + q], m10 = W[q * 4 + p], m11 = W[q * 4 + q];
          float c1, s1;
          const float t = m00 + m11;
          const float d = m10 - m01;
          if (fabsf(d) < FMIN) {
            s1 = 0.0f;
            c1 = 1.0f;
          } else {
            const float u = t / d;
            const float tmp = sqrtf(1.0f + u * u);
            s1 = 1.0f / tmp;
            c1 = u / tmp;
          }
          if (!(c1 == 1.0f && s1 == 0.0f)) {
            svo_rot(m00, m10, c1, s1);
            svo_rot(m01, m11, c1, s1);
          }
          // j_right.makeJacobi(m, 0, 1)
          float cr, sr;
          const float deno = 2.0f * fabsf(m01);
          if (deno < FMIN) {
            cr = 1.0f;
            sr = 0.0f;
          } else {
            const float tau = (m00 - m11) / deno;
            const float w = sqrtf(tau * tau + 1.0f);
            float tt;
            if (tau > 0.0f)
              tt = 1.0f / (tau + w);
            else
              tt = 1.0f / (tau - w);
            const float sign_t = tt > 0.0f ? 1.0f : -1.0f;
            const float n = 1.0f / sqrtf(tt * tt + 1.0f);
            sr = -sign_t * (m01 / fabsf(m01)) * fabsf(tt) * n;
            cr = n;
          }
          // j_left = rot1 * j_right.transpose(); transpose = (c, -s)
          const float ct = cr, st = -sr;
          const float cl = c1 * ct - s1 * st;
          const float sl = c1 * st + s1 * ct;
          if (!(cl == 1.0f && sl == 0.0f)) {  // rows p and q of W
#pragma unroll
            for (int k = 0; k < 4; ++k) svo_rot(W[p * 4 + k], W[q * 4 + k], cl, sl);
          }
          if (!(ct == 1.0f && st == 0.0f)) {  // columns p and q of W and of V
#pragma unroll
            for (int k = 0; k < 4; ++k) svo_rot(W[k * 4 + p], W[k * 4 + q], ct, st);
#pragma unroll
            for (int k = 0; k < 4; ++k) svo_rot(V[k * 4 + p], V[k * 4 + q], ct, st);
          }
          const float a = fabsf(W[p * 4 + p]), b = fabsf(W[q * 4 + q]);
          const float mx = a > b ? a : b;
          if (mx > max_diag) max_diag = mx;
        }
      }
    if (finished || sweeps > 1000) break;
  }
  float sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sv[i] = fabsf(W[i * 5]) * scale;
  // selection sort, descending, first of equal maxima; only the column that ends in position 3 is needed
  int col[4] = {0, 1, 2, 3};
  bool stop = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int pos = i;
    float best = sv[i];
#pragma unroll
    for (int k = i + 1; k < 4; ++k)
      if (sv[k] > best) {
        best = sv[k];
        pos = k;
      }
    if (best == 0.0f) stop = true;
    if (!stop) {
#pragma unroll
      for (int k = i + 1; k < 4; ++k)
        if (pos == k) {
          const float ts = sv[i];
          sv[i] = sv[k];
          sv[k] = ts;
          const int tc = col[i];
          col[i] = col[k];
          col[k] = tc;
        }
    }
  }
  const int c3 = col[3];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = c3 == 0 ? V[r * 4 + 0] : (c3 == 1 ? V[r * 4 + 1] : (c3 == 2 ? V[r * 4 + 2] : V[r * 4 + 3]));
}

// mapping::triangulateDLT(pt0, pt1, R10, t10, cam0, cam1, X0, X1), triangulate_3d.cpp:91-130
static __device__ void svo_triangulate(const SvoCam &cam, float u0, float v0, float u1, float v1, float (&X0)[3], float (&X1)[3]) {
  float M[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) M[i] = 0.0f;
  M[0] = -cam.K0[0];
  M[5] = -cam.K0[1];
  M[2] = u0 - cam.K0[2];
  M[6] = v0 - cam.K0[3];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    M[8 + c] = u1 * cam.P10[8 + c] - cam.P10[0 + c];
    M[12 + c] = v1 * cam.P10[8 + c] - cam.P10[4 + c];
  }
  float v[4];
  svo_svd4_nullvec(M, v);
  X0[0] = v[0] / v[3];
  X0[1] = v[1] / v[3];
  X0[2] = v[2] / v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    X1[i] = (cam.R10[i * 3 + 0] * X0[0] + (cam.R10[i * 3 + 1] * X0[1] + cam.R10[i * 3 + 2] * X0[2])) + cam.t10[i];
}

#endif  // __HIPCC__
