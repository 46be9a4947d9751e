// misc_kernels.hip — ORB Hamming distance, landmark mask compaction and the
// per-landmark prior projection.
//
//  orb_hamming / orb_match : FeatureExtractor::descriptorDistance
//        (core/visual_odometry/feature_extractor.cpp:338-357) for whole descriptor
//        sets, and the nearest / second-nearest + ratio rule sketched in
//        test/test_orbmatching.cpp:87-137.
//  compact                 : StereoLandmarkTracking(src, mask) / LandmarkTracking(src, mask)
//        (core/visual_odometry/landmark.cpp:291-332, :194-231): stable compaction.
//  calc_prior              : FeatureTracker::calcPrior (feature_tracker.cpp:208-234).
//  stereo_prior            : the prior loop of StereoVO::trackStereoImages
//        (core/visual_odometry/stereo_vo/stereo_vo.cpp:483-522) with
//        Camera::projectToPixel / inImage (camera.cpp:208-229).
//
// Hamming is byte/integer work bound by the na x nb x 2 B distance matrix write
// (HBM): train descriptors sit in registers (8 dwords per lane), the 16 query
// descriptors of a lane group are broadcast from LDS, and 64 consecutive lanes
// write 64 consecutive uint16 (128 B per wave-instruction row).
#include "vo_internal.hpp"
#include "vo_kernels.hpp"

#define HM_TQ 64  // queries per block
#define HM_TT 64  // trains per block

__global__ __launch_bounds__(256) void orb_hamming_kernel(const uint32_t *__restrict__ a, int na,
                                                          const uint32_t *__restrict__ b, int nb,
                                                          uint16_t *__restrict__ dist) {
  __shared__ uint32_t s_q[HM_TQ * 8];
  const int t = threadIdx.x;
  const int q0 = blockIdx.y * HM_TQ, j0 = blockIdx.x * HM_TT;
  for (int i = t; i < HM_TQ * 8; i += 256) {
    const int q = q0 + (i >> 3);
    s_q[i] = q < na ? a[(size_t)q * 8 + (i & 7)] : 0u;
  }
  const int j = j0 + (t & 63);
  uint32_t tr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) tr[k] = j < nb ? b[(size_t)j * 8 + k] : 0u;
  __syncthreads();
  const int g = t >> 6;  // 4 groups of 16 queries
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int ql = g * 16 + k;
    const int q = q0 + ql;
    int d = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) d += __popc(s_q[ql * 8 + w] ^ tr[w]);
    if (q < na && j < nb) dist[(size_t)q * nb + j] = (uint16_t)d;
  }
}

// one wavefront per query: lanes stride over the train set
__global__ __launch_bounds__(64) void orb_match_kernel(const uint32_t *__restrict__ a, int na,
                                                       const uint32_t *__restrict__ b, int nb, int th_low,
                                                       float ratio, int32_t *best_idx, uint16_t *best_dist,
                                                       uint16_t *second_dist) {
  const int q = blockIdx.x;
  if (q >= na) return;
  const int lane = threadIdx.x;
  uint32_t qd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) qd[k] = a[(size_t)q * 8 + k];
  int bd = 256, bd2 = 256, bi = 0x7fffffff;
  for (int j = lane; j < nb; j += 64) {
    int d = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) d += __popc(qd[w] ^ b[(size_t)j * 8 + w]);
    if (d < bd) {
      bd2 = bd;
      bd = d;
      bi = j;
    } else if (d < bd2) {
      bd2 = d;
    }
  }
  // merge (best, index, second): min, first index on ties, second order statistic
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int obd = __shfl_xor(bd, off), obi = __shfl_xor(bi, off), obd2 = __shfl_xor(bd2, off);
    const int nsec = min(max(bd, obd), min(bd2, obd2));
    if (obd < bd || (obd == bd && obi < bi)) {
      bd = obd;
      bi = obi;
    }
    bd2 = nsec;
  }
  if (lane == 0) {
    best_dist[q] = (uint16_t)bd;
    second_dist[q] = (uint16_t)bd2;
    const bool have = bi != 0x7fffffff;
    best_idx[q] = (have && bd <= th_low && (float)bd < ratio * (float)bd2) ? bi : -1;
  }
}

int vo_hamming_enqueue(vo_ctx *c, const uint8_t *d_a, int na, const uint8_t *d_b, int nb, uint16_t *d_dist) {
  if (na <= 0 || nb <= 0) return VO_OK;
  dim3 grid((nb + HM_TT - 1) / HM_TT, (na + HM_TQ - 1) / HM_TQ);
  vo_prof_begin(c, VO_K_HAMMING);
  hipLaunchKernelGGL(orb_hamming_kernel, grid, dim3(256), 0, c->stream, (const uint32_t *)d_a, na,
                     (const uint32_t *)d_b, nb, d_dist);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

int vo_match_enqueue(vo_ctx *c, const uint8_t *d_a, int na, const uint8_t *d_b, int nb, int th_low, float ratio,
                     int32_t *d_best, uint16_t *d_bd, uint16_t *d_sd) {
  if (na <= 0) return VO_OK;
  vo_prof_begin(c, VO_K_HAMMING);
  hipLaunchKernelGGL(orb_match_kernel, dim3(na), dim3(64), 0, c->stream, (const uint32_t *)d_a, na,
                     (const uint32_t *)d_b, nb, th_low, ratio, d_best, d_bd, d_sd);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// ---- stable compaction --------------------------------------------------------
// One workgroup of 1024 lanes walks the input in chunks of 1024; per chunk a
// ballot + popcount-below gives each survivor its slot. Survivors of every
// attached array are gathered in the same pass (pixel pairs as float2, 3-D
// points as 3 floats, scalars, original indices).
struct CompactArgs {
  const uint8_t *mask;
  const uint8_t *alive;    // optional
  const uint8_t *tracked;  // optional
  const uint8_t *lm_flags; // optional: drop where (lm_flags[i] & lm_reject) != 0
  int lm_reject;
  int n;
  const int *d_n;
  int32_t *index_valid;    // optional out
  int *d_n_out;
  // gathers (all optional): up to 4 float2 arrays, 1 float3, 1 float, 1 int
  const float *in2[4];
  float *out2[4];
  const float *in3;
  float *out3;
  const float *in1;
  float *out1;
  const int32_t *in_i;
  int32_t *out_i;
  // stage bookkeeping (optional): stage[orig[i]] = stage_val for survivors
  uint8_t *stage;
  int stage_val;
  // optional scatter of a float2 array back to original index space for ALL i < n
  const float *sc_src;
  float *sc_dst;
  // optional gate of stereo_vo.cpp:653-668: keep only if ((y > 660) ? 100 : 0) < thres_sampson
  const float *gate_pts;
  float gate_thres;
  // optional fused trackWithPrior validity (feature_tracker.cpp:191-197): when klt_status is set the
  // keep flag is status>0 && 0<x<W && 0<y<H && err<=thr (AND mask when mask is non-null)
  const uint8_t *klt_status;
  const float *klt_err;
  const float *klt_pts;
  float klt_thres_err;
  int klt_W, klt_H;
};

__global__ __launch_bounds__(1024) void compact_kernel(CompactArgs a) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int n = a.d_n ? *a.d_n : a.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + tid;
    bool keep = false;
    if (i < n) {
      keep = (!a.mask || a.mask[i]) && (!a.alive || a.alive[i]) && (!a.tracked || a.tracked[i]);
      if (a.lm_flags) keep = keep && !(a.lm_flags[i] & a.lm_reject);
      if (a.klt_status) {
        const float x = a.klt_pts[2 * i], y = a.klt_pts[2 * i + 1];
        keep = keep && a.klt_status[i] > 0 && x > 0 && x < a.klt_W && y > 0 && y < a.klt_H;
        keep = keep && a.klt_err[i] <= a.klt_thres_err;
      }
      if (a.gate_pts) keep = keep && ((a.gate_pts[2 * i + 1] > 660 ? 100.f : 0.f) < a.gate_thres);
      if (a.sc_src) {
        const int o = a.in_i ? a.in_i[i] : i;
        a.sc_dst[2 * o] = a.sc_src[2 * i];
        a.sc_dst[2 * o + 1] = a.sc_src[2 * i + 1];
      }
    }
    const unsigned long long bal = __ballot(keep);
    const int below = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const int base = s_base;
    if (keep) {
      const int o = base + woff + below;
      if (a.index_valid) a.index_valid[o] = i;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (a.in2[k]) {
          a.out2[k][2 * o] = a.in2[k][2 * i];
          a.out2[k][2 * o + 1] = a.in2[k][2 * i + 1];
        }
      if (a.in3) {
        a.out3[3 * o] = a.in3[3 * i];
        a.out3[3 * o + 1] = a.in3[3 * i + 1];
        a.out3[3 * o + 2] = a.in3[3 * i + 2];
      }
      if (a.in1) a.out1[o] = a.in1[i];
      int orig = i;
      if (a.in_i) {
        orig = a.in_i[i];
        a.out_i[o] = orig;
      }
      if (a.stage) a.stage[orig] = (uint8_t)a.stage_val;
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base = base + tot;
    }
    __syncthreads();
  }
  if (tid == 0) *a.d_n_out = s_base;
}

int vo_compact_enqueue(vo_ctx *c, const CompactArgsHost &h) {
  CompactArgs a;
  memset(&a, 0, sizeof(a));
  a.mask = h.mask;
  a.alive = h.alive;
  a.tracked = h.tracked;
  a.lm_flags = h.lm_flags;
  a.lm_reject = h.lm_reject;
  a.n = h.n;
  a.d_n = h.d_n;
  a.index_valid = h.index_valid;
  a.d_n_out = h.d_n_out;
  for (int k = 0; k < 4; ++k) {
    a.in2[k] = h.in2[k];
    a.out2[k] = h.out2[k];
  }
  a.in3 = h.in3;
  a.out3 = h.out3;
  a.in1 = h.in1;
  a.out1 = h.out1;
  a.in_i = h.in_i;
  a.out_i = h.out_i;
  a.stage = h.stage;
  a.stage_val = h.stage_val;
  a.sc_src = h.sc_src;
  a.sc_dst = h.sc_dst;
  a.gate_pts = h.gate_pts;
  a.gate_thres = h.gate_thres;
  a.klt_status = h.klt_status;
  a.klt_err = h.klt_err;
  a.klt_pts = h.klt_pts;
  a.klt_thres_err = h.klt_thres_err;
  a.klt_W = h.klt_W;
  a.klt_H = h.klt_H;
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// ---- FeatureTracker::calcPrior -------------------------------------------------
struct PriorArgs {
  const float *pts0;
  int n_pts0;
  const float *Xw;
  int n;
  float T1w[16];
  float K[9];
  float *out;
};
__global__ __launch_bounds__(256) void calc_prior_kernel(PriorArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n_pts0) {
    a.out[2 * i] = a.pts0[2 * i];
    a.out[2 * i + 1] = a.pts0[2 * i + 1];
  }
  if (i >= a.n || i >= a.n_pts0) return;
  const float *Xi = a.Xw + 3 * i;
  float X[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    X[r] = ((a.T1w[r * 4 + 0] * Xi[0] + a.T1w[r * 4 + 1] * Xi[1]) + a.T1w[r * 4 + 2] * Xi[2]) + a.T1w[r * 4 + 3];
  const float nrm = sqrtf((X[0] * X[0] + X[1] * X[1]) + X[2] * X[2]);
  if (nrm > 0) {
    a.out[2 * i] = a.K[0] * X[0] / X[2] + a.K[2];
    a.out[2 * i + 1] = a.K[4] * X[1] / X[2] + a.K[5];
  }
}

int vo_calc_prior_enqueue(vo_ctx *c, const float *d_pts0, int n_pts0, const float *d_Xw, int n,
                          const float T1w[16], const float K[9], float *d_out) {
  if (n_pts0 <= 0) return VO_OK;
  PriorArgs a;
  a.pts0 = d_pts0;
  a.n_pts0 = n_pts0;
  a.Xw = d_Xw;
  a.n = n;
  memcpy(a.T1w, T1w, sizeof(a.T1w));
  memcpy(a.K, K, sizeof(a.K));
  a.out = d_out;
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(calc_prior_kernel, dim3((n_pts0 + 255) / 256), dim3(256), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// ---- stereo prior (stereo_vo.cpp:483-522), previous-left-camera coordinates ----
struct StereoPriorArgs {
  const float *Xp;
  const float *pts_l0;
  const float *pts_r0;
  const uint8_t *flags;  // bit 0 = landmark triangulated; null = all
  int n;
  float T_cp[16], T_rl[16];
  float Kl[4], Kr[4];
  int W, H;
  float *pts_l1, *pts_r1, *scale;
  int32_t *orig;
  uint8_t *stage;
};
__device__ __forceinline__ bool in_image_dev(float x, float y, int W, int H) {
  const float offset = 3.0f;
  return !(x < offset || y < offset || x >= W - offset || y >= H - offset);
}
__global__ __launch_bounds__(256) void stereo_prior_kernel(StereoPriorArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  a.orig[i] = i;
  a.stage[i] = 0;
  if (a.flags && !(a.flags[i] & VO_LM_TRIANGULATED)) {  // stereo_vo.cpp:515-519
    a.scale[i] = 1.0f;
    a.pts_l1[2 * i] = a.pts_l0[2 * i];
    a.pts_l1[2 * i + 1] = a.pts_l0[2 * i + 1];
    a.pts_r1[2 * i] = a.pts_r0[2 * i];
    a.pts_r1[2 * i + 1] = a.pts_r0[2 * i + 1];
    return;
  }
  const float *Xi = a.Xp + 3 * i;
  float Xl[3], Xr[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    Xl[r] = ((a.T_cp[r * 4 + 0] * Xi[0] + a.T_cp[r * 4 + 1] * Xi[1]) + a.T_cp[r * 4 + 2] * Xi[2]) + a.T_cp[r * 4 + 3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    Xr[r] = ((a.T_rl[r * 4 + 0] * Xl[0] + a.T_rl[r * 4 + 1] * Xl[1]) + a.T_rl[r * 4 + 2] * Xl[2]) + a.T_rl[r * 4 + 3];
  a.scale[i] = Xi[2] / Xl[2];
  const float izl = 1.0f / Xl[2], izr = 1.0f / Xr[2];
  const float plx = a.Kl[0] * Xl[0] * izl + a.Kl[2], ply = a.Kl[1] * Xl[1] * izl + a.Kl[3];
  const float prx = a.Kr[0] * Xr[0] * izr + a.Kr[2], pry = a.Kr[1] * Xr[1] * izr + a.Kr[3];
  if (!in_image_dev(plx, ply, a.W, a.H) || !in_image_dev(prx, pry, a.W, a.H) || (double)Xl[2] < 0.1 ||
      (double)Xr[2] < 0.1) {
    a.pts_l1[2 * i] = a.pts_l0[2 * i];
    a.pts_l1[2 * i + 1] = a.pts_l0[2 * i + 1];
    a.pts_r1[2 * i] = a.pts_r0[2 * i];
    a.pts_r1[2 * i + 1] = a.pts_r0[2 * i + 1];
  } else {
    a.pts_l1[2 * i] = plx;
    a.pts_l1[2 * i + 1] = ply;
    a.pts_r1[2 * i] = prx;
    a.pts_r1[2 * i + 1] = pry;
  }
}

int vo_stereo_prior_enqueue(vo_ctx *c, const float *d_Xp, const float *d_pl0, const float *d_pr0,
                            const uint8_t *d_flags, int n,
                            const float T_cp[16], const float T_rl[16], const float Kl[4], const float Kr[4],
                            int W, int H, float *d_pl1, float *d_pr1, float *d_scale, int32_t *d_orig,
                            uint8_t *d_stage) {
  if (n <= 0) return VO_OK;
  StereoPriorArgs a;
  a.Xp = d_Xp;
  a.pts_l0 = d_pl0;
  a.pts_r0 = d_pr0;
  a.flags = d_flags;
  a.n = n;
  memcpy(a.T_cp, T_cp, sizeof(a.T_cp));
  memcpy(a.T_rl, T_rl, sizeof(a.T_rl));
  memcpy(a.Kl, Kl, sizeof(a.Kl));
  memcpy(a.Kr, Kr, sizeof(a.Kr));
  a.W = W;
  a.H = H;
  a.pts_l1 = d_pl1;
  a.pts_r1 = d_pr1;
  a.scale = d_scale;
  a.orig = d_orig;
  a.stage = d_stage;
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(stereo_prior_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// ---- epipolar gates: MotionEstimator::calcSampsonDistance (motion_estimator.cpp:572-599) and
// calcSymmetricEpipolarDistance (:621-653), per-point part on a given F10 ------------------------
// Eigen evaluates its 3-term products as e0 + (e1 + e2) (unrolled redux); the written sums of the
// reference are left to right.
struct EpiArgs {
  const float *pts0, *pts1;
  int n;
  float F[9];
  int mode;  // 0 Sampson, 1 symmetric epipolar
  float *dist;
};
__device__ __forceinline__ float eig_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);
}
__global__ __launch_bounds__(256) void epi_distance_kernel(EpiArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float x0 = a.pts0[2 * i], y0 = a.pts0[2 * i + 1], x1 = a.pts1[2 * i], y1 = a.pts1[2 * i + 1];
  float p[3], q[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) p[r] = eig_dot3(a.F[r * 3 + 0], x0, a.F[r * 3 + 1], y0, a.F[r * 3 + 2], 1.0f);
#pragma unroll
  for (int r = 0; r < 3; ++r) q[r] = eig_dot3(a.F[0 * 3 + r], x1, a.F[1 * 3 + r], y1, a.F[2 * 3 + r], 1.0f);
  float num = eig_dot3(x1, p[0], y1, p[1], 1.0f, p[2]);
  if (a.mode == 0) {
    num *= num;
    const float den = ((p[0] * p[0] + p[1] * p[1]) + q[0] * q[0]) + q[1] * q[1];
    a.dist[i] = num / den;
  } else {
    num = fabsf(num);
    const float den = 1.0f / sqrtf(p[0] * p[0] + p[1] * p[1]) + 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1]);
    a.dist[i] = num * den;
  }
}
int vo_epi_distance_enqueue(vo_ctx *c, int mode, const float *d_pts0, const float *d_pts1, int n, const float F10[9],
                            float *d_dist) {
  if (n <= 0) return VO_OK;
  EpiArgs a;
  a.pts0 = d_pts0;
  a.pts1 = d_pts1;
  a.n = n;
  memcpy(a.F, F10, sizeof(a.F));
  a.mode = mode;
  a.dist = d_dist;
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(epi_distance_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// ---- feature bucketing: WeightBin::reset/update (core/visual_odometry/feature_extractor.h:120-135) and
// the arg-max-per-bin branch of extractORBwithBinning_fast (feature_extractor.cpp:241-277) ------------
struct BinArgs {
  const float *xy;        // points (update) / keypoints (arg-max)
  const float *response;  // arg-max only
  int n;
  int u_step, v_step;     // update: integer steps, the reference divides
  float inv_u, inv_v;     // arg-max: the reference multiplies by the inverse steps
  int n_bins_u, n_bins_v;
  int32_t *weight;        // update: out ; arg-max: in
  unsigned long long *key;  // arg-max: one 64-bit key per bin, 0 = empty
  float *pts_out;
  int32_t *idx_out;
  int *n_out;
  const int *d_n;         // arg-max: keypoint count on the device (overrides n when set)
};
__global__ __launch_bounds__(256) void weight_bin_update_kernel(BinArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int u_idx = (int)floorf(a.xy[2 * i] / (float)a.u_step);
  const int v_idx = (int)floorf(a.xy[2 * i + 1] / (float)a.v_step);
  const int bin_idx = v_idx * a.n_bins_u + u_idx;  // only the flattened index is tested (:130)
  if (bin_idx >= 0 && bin_idx < a.n_bins_u * a.n_bins_v) a.weight[bin_idx] = 0;
}
// "max_score < response" keeps the FIRST keypoint among equal responses: a 64-bit atomicMax on
// (order-preserving response bits, ~index) selects exactly that one, in any execution order.
__global__ __launch_bounds__(256) void bucket_key_kernel(BinArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (a.d_n ? *a.d_n : a.n)) return;
  const unsigned u = (unsigned)(int)floorf(a.xy[2 * i] * a.inv_u);
  const unsigned v = (unsigned)(int)floorf(a.xy[2 * i + 1] * a.inv_v);
  if (u >= (unsigned)a.n_bins_u || v >= (unsigned)a.n_bins_v) return;
  const int bin = (int)(v * (unsigned)a.n_bins_u + u);
  if (a.weight && a.weight[bin] == 0) return;  // (no weights: the best keypoint of EVERY bin, gated by the consumer)
  float r = a.response[i];
  if (!(-1.0f < r)) return;  // never beats the initial max_score of -1 (NaN included)
  r = r + 0.0f;              // -0 -> +0: the reference's "<" does not tell them apart
  const unsigned bits = __float_as_uint(r);
  const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
  const unsigned long long key = ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
  atomicMax(&a.key[bin], key);
}
__global__ __launch_bounds__(1024) void bucket_emit_kernel(BinArgs a) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = a.n_bins_u * a.n_bins_v;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < total; c0 += 1024) {
    const int j = c0 + tid;
    const unsigned long long key = j < total ? a.key[j] : 0ull;
    const bool keep = key != 0ull && a.weight[j < total ? j : 0] > 0;
    const unsigned long long bal = __ballot(keep);
    const int below = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const int base = s_base;
    if (keep) {
      const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      const int o = base + woff + below;
      a.pts_out[2 * o] = a.xy[2 * idx];
      a.pts_out[2 * o + 1] = a.xy[2 * idx + 1];
      a.idx_out[o] = idx;
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base = base + tot;
    }
    __syncthreads();
  }
  if (tid == 0) *a.n_out = s_base;
}
// per-bin table instead of the compacted list: has[bin], xy[bin] of the bin's first keypoint of largest response
__global__ __launch_bounds__(256) void bucket_table_kernel(BinArgs a, uint8_t *has) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.n_bins_u * a.n_bins_v) return;
  const unsigned long long key = a.key[j];
  has[j] = key != 0ull ? 1 : 0;
  float x = 0.f, y = 0.f;
  if (key != 0ull) {
    const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    x = a.xy[2 * idx];
    y = a.xy[2 * idx + 1];
  }
  a.pts_out[2 * j] = x;
  a.pts_out[2 * j + 1] = y;
}
int vo_bucket_table_enqueue(vo_ctx *c, const float *d_xy, const float *d_response, int n_max, float inv_u, float inv_v,
                            int n_bins_u, int n_bins_v, unsigned long long *d_key, float *d_tab_xy, uint8_t *d_tab_has,
                            const int *d_n) {
  const int total = n_bins_u * n_bins_v;
  VO_CHECK_HIP(c, hipMemsetAsync(d_key, 0, sizeof(unsigned long long) * (size_t)total, c->stream));
  BinArgs a;
  memset(&a, 0, sizeof(a));
  a.xy = d_xy;
  a.response = d_response;
  a.n = n_max;
  a.inv_u = inv_u;
  a.inv_v = inv_v;
  a.n_bins_u = n_bins_u;
  a.n_bins_v = n_bins_v;
  a.weight = nullptr;
  a.key = d_key;
  a.pts_out = d_tab_xy;
  a.d_n = d_n;
  vo_prof_begin(c, VO_K_AUX);
  if (n_max > 0) hipLaunchKernelGGL(bucket_key_kernel, dim3((n_max + 255) / 256), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(bucket_table_kernel, dim3((total + 255) / 256), dim3(256), 0, c->stream, a, d_tab_has);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

int vo_weight_bin_update_enqueue(vo_ctx *c, const float *d_pts, int n, int u_step, int v_step, int n_bins_u,
                                 int n_bins_v, int32_t *d_weight) {
  const int total = n_bins_u * n_bins_v;
  // reset(): every weight 1 (0x00000001 is not a byte pattern: a tiny fill kernel would do; memset D32 does)
  VO_CHECK_HIP(c, hipMemsetD32Async((hipDeviceptr_t)d_weight, 1, (size_t)total, c->stream));
  if (n <= 0) return VO_OK;
  BinArgs a;
  memset(&a, 0, sizeof(a));
  a.xy = d_pts;
  a.n = n;
  a.u_step = u_step;
  a.v_step = v_step;
  a.n_bins_u = n_bins_u;
  a.n_bins_v = n_bins_v;
  a.weight = d_weight;
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(weight_bin_update_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
int vo_bucket_argmax_enqueue(vo_ctx *c, const float *d_xy, const float *d_response, int n, float inv_u, float inv_v,
                             int n_bins_u, int n_bins_v, const int32_t *d_weight, unsigned long long *d_key,
                             float *d_pts_out, int32_t *d_idx_out, int *d_n_out, const int *d_n) {
  const int total = n_bins_u * n_bins_v;
  VO_CHECK_HIP(c, hipMemsetAsync(d_key, 0, sizeof(unsigned long long) * (size_t)total, c->stream));
  BinArgs a;
  memset(&a, 0, sizeof(a));
  a.xy = d_xy;
  a.response = d_response;
  a.n = n;
  a.inv_u = inv_u;
  a.inv_v = inv_v;
  a.n_bins_u = n_bins_u;
  a.n_bins_v = n_bins_v;
  a.weight = (int32_t *)d_weight;
  a.key = d_key;
  a.pts_out = d_pts_out;
  a.idx_out = d_idx_out;
  a.n_out = d_n_out;
  a.d_n = d_n;
  vo_prof_begin(c, VO_K_AUX);
  if (n > 0) hipLaunchKernelGGL(bucket_key_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(bucket_emit_kernel, dim3(1), dim3(1024), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
