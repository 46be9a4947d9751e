// mvo_device.hpp — MonoVO's device-resident track set and its advance step (mono_vo.hip), shared with the mono frame's BA
// launch (mono_gate.hpp), whose epilogue builds the next track set of a steady-state frame.
// Reference: core/visual_odometry/mono_vo/mono_vo.cpp:964-1018 (lmtrack_final -> addObservationAndRelatedFrame, the new
// landmarks Landmark(p0_new, frame_prev_) + observation (p1_new, frame_curr)), core/visual_odometry/landmark.cpp:76-135 (age,
// parallax against the oldest observation).
#pragma once
#include "svo_device.hpp"
#include "vo_internal.hpp"

#define MVO_FRAME_RING (1 << 14)
#define MVO_COS_NONE 2.0f  // no parallax yet (last_parallax_ = 0): never passes the threshold

struct MvoSet {
  SvoTrackSet t;     // pts_l = pts_r = the pixel seen; Xw; flags; ids (what stereo_vo_lba.hip reads)
  float *p_first;    // [cap][2] observations_.front()
  int32_t *f_first;  // [cap]    index of related_frames_.front()
  int32_t *age;      // [cap]
  float *cos_last;   // [cap]    cos of last_parallax_ (MVO_COS_NONE: none)
  int32_t *n_kf;     // [cap]    observations_on_keyframes_.size()
  float *p_kf_first; // [cap][2] observations_on_keyframes_.front()
  int32_t *kf_first; // [cap]    frame index of related_keyframes_.front()
};
struct MvoHdr {
  int n_surv, n_new, n_next, n_kf_tracked, id_min, overflow, n_recon, pad;
  uint32_t seq;
};

// Matrix4f * Matrix4f in Eigen's evaluation order (as svo_mul44), rows 0..2 only
__device__ __forceinline__ void mvo_mul34(const float *A, const float *B, float (&C)[12]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r = A[i * 4 + 0] * B[0 * 4 + j];
#pragma unroll
      for (int k = 1; k < 4; ++k) r = A[i * 4 + k] * B[k * 4 + j] + r;
      C[i * 4 + j] = r;
    }
}
__device__ __forceinline__ float mvo_dot3(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

// landmark.cpp:100-116: cos of the parallax between the oldest observation p0 (frame of inverse pose Tcw0) and the newest p1
// (frame of pose Twc1), pushed inside (-1, 1)
__device__ __forceinline__ float mvo_parallax_cos(float p0x, float p0y, float p1x, float p1y, const float K[4], const float *Tcw0,
                                                  const float *Twc1) {
  float T01[12];
  mvo_mul34(Tcw0, Twc1, T01);
  const float fxinv = 1.0f / K[0], fyinv = 1.0f / K[1];
  const float x0[3] = {(p0x - K[2]) * fxinv, (p0y - K[3]) * fyinv, 1.0f};
  const float x1[3] = {(p1x - K[2]) * fxinv, (p1y - K[3]) * fyinv, 1.0f};
  float r[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) r[i] = mvo_dot3(T01[i * 4 + 0], x1[0], T01[i * 4 + 1], x1[1], T01[i * 4 + 2], x1[2]);
  const float dot = mvo_dot3(x0[0], r[0], x0[1], r[1], x0[2], r[2]);
  const float n0 = sqrtf(mvo_dot3(x0[0], x0[0], x0[1], x0[1], x0[2], x0[2]));
  const float n1 = sqrtf(mvo_dot3(r[0], r[0], r[1], r[1], r[2], r[2]));
  float c = dot / (n0 * n1);
  if (c >= 1.0f) c = 0.99999f;
  if (c <= -1.0f) c = -0.99999f;
  return c;
}

// exclusive scan of one int per thread over a workgroup of NW wavefronts; every thread gets the total
template <int NW>
__device__ __forceinline__ int mvo_block_scan(int v, int *s_w, int &total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int c = s_w[k];
    before += k < wave ? c : 0;
    tot += c;
  }
  total = tot;
  return before + inc - v;
}

// ---- the next track set: lmtrack_final in index order (addObservationAndRelatedFrame for each: age, parallax), then the new
// landmarks Landmark(p0_new, frame_prev_) + observation (p1_new, frame_curr) (mono_vo.cpp:964-1018) ----------------------
struct MvoAdvArgs {
  MvoSet cur, nxt;
  int n, cap;
  const uint8_t *stage;    // [n] 4 = in lmtrack_final
  const float *pts1;       // [n][2] pixel in the current image
  const float *cand1, *cand0;  // new points: pixel in I1, back-tracked pixel in I0
  const uint8_t *mnew;
  int m;                   // candidates emitted
  int id_base, f;          // first new landmark id, index of the current frame
  float K[4];
  float T_obs[16];         // pose of the current frame as the survivors' observation sees it (identity at initialisation)
  float T_wc[16], T_cw[16];  // pose of the current frame (what the new landmarks see, and the frame table's entry f)
  float *frameT;           // [ring][32]
  MvoHdr *hdr_dev, *hdr_host;
  uint32_t seq;
  int *pos_s, *pos_n;      // [cap] scratch of the two-launch form: where survivor k / new point j goes (-1: nowhere)
  float T_wp[16];          // inside the BA launch: the previous frame's pose; T_wc = T_wp * dT01 is formed on the device
};

// one survivor (entry k of the current set, pixel (px, py) in the current image) -> entry pos of the next set
__device__ __forceinline__ void mvo_put_survivor(const MvoAdvArgs &a, int k, int pos, float px, float py, const float *T_obs) {
  a.nxt.t.pts_l[2 * pos] = px;
  a.nxt.t.pts_l[2 * pos + 1] = py;
  a.nxt.t.ids[pos] = a.cur.t.ids[k];
  a.nxt.t.flags[pos] = a.cur.t.flags[k];
  a.nxt.t.Xw[3 * pos] = a.cur.t.Xw[3 * k];
  a.nxt.t.Xw[3 * pos + 1] = a.cur.t.Xw[3 * k + 1];
  a.nxt.t.Xw[3 * pos + 2] = a.cur.t.Xw[3 * k + 2];
  const float p0x = a.cur.p_first[2 * k], p0y = a.cur.p_first[2 * k + 1];
  const int f0 = a.cur.f_first[k];
  a.nxt.p_first[2 * pos] = p0x;
  a.nxt.p_first[2 * pos + 1] = p0y;
  a.nxt.f_first[pos] = f0;
  a.nxt.age[pos] = a.cur.age[k] + 1;
  a.nxt.n_kf[pos] = a.cur.n_kf[k];
  a.nxt.p_kf_first[2 * pos] = a.cur.p_kf_first[2 * k];
  a.nxt.p_kf_first[2 * pos + 1] = a.cur.p_kf_first[2 * k + 1];
  a.nxt.kf_first[pos] = a.cur.kf_first[k];
  // (f0 < f: that entry of the frame table was written by an earlier launch)
  a.nxt.cos_last[pos] = mvo_parallax_cos(p0x, p0y, px, py, a.K, a.frameT + (size_t)(f0 & (MVO_FRAME_RING - 1)) * 32 + 16, T_obs);
}
// one new landmark (pixel (p0x, p0y) in the previous image, (p1x, p1y) in the current one) -> entry r of the next set
__device__ __forceinline__ void mvo_put_new(const MvoAdvArgs &a, int r, int id, float p0x, float p0y, float p1x, float p1y,
                                            const float *T_wc) {
  a.nxt.t.pts_l[2 * r] = p1x;
  a.nxt.t.pts_l[2 * r + 1] = p1y;
  a.nxt.t.ids[r] = id;
  a.nxt.t.flags[r] = 0;
  a.nxt.t.Xw[3 * r] = a.nxt.t.Xw[3 * r + 1] = a.nxt.t.Xw[3 * r + 2] = 0.0f;
  a.nxt.p_first[2 * r] = p0x;
  a.nxt.p_first[2 * r + 1] = p0y;
  a.nxt.f_first[r] = a.f - 1;
  a.nxt.age[r] = 2;
  a.nxt.n_kf[r] = 0;
  a.nxt.p_kf_first[2 * r] = a.nxt.p_kf_first[2 * r + 1] = 0.0f;
  a.nxt.kf_first[r] = -1;
  a.nxt.cos_last[r] = mvo_parallax_cos(p0x, p0y, p1x, p1y, a.K, a.frameT + (size_t)((a.f - 1) & (MVO_FRAME_RING - 1)) * 32 + 16, T_wc);
}

// The advance step as the last thing the mono frame's BA launch does (mono_gate.hpp): NW wavefronts of ONE workgroup, every
// store of the frame's results is behind a barrier. `skip`: the frame needs the 5-point fallback or failed — the host finishes
// it and runs the advance as launches of its own; only the header (pad = 1) and the sequence word go out. dT01: the pose-only
// BA's result (device memory); m: candidates emitted. Ends with the header in pinned host memory and the sequence word the
// host polls — behind everything this workgroup sent to the host before (the frame's own result block).
template <int NW>
__device__ __forceinline__ void mvo_advance_body(const MvoAdvArgs &a, const float *dT01, bool skip, int m, int tid) {
  __shared__ int s_w[NW];
  __shared__ int s_kft[NW];
  __shared__ float s_T[32];
  __shared__ int s_idmin, s_old;
  constexpr int NT = NW * 64;
  if (skip) {
    if (tid == 0) {
      MvoHdr h;
      h.n_surv = h.n_new = h.n_next = h.n_kf_tracked = h.id_min = h.overflow = h.n_recon = 0;
      h.pad = 1;
      h.seq = 0;
      *a.hdr_dev = h;
      *a.hdr_host = h;
    }
  } else {
    if (tid < 16) {  // frame_curr->setPose(Twc_prev * dT01) (mono_vo.cpp:883): svo_mul44, element by element
      const int i = tid >> 2, j = tid & 3;
      float r = a.T_wp[i * 4 + 0] * dT01[0 * 4 + j];
#pragma unroll
      for (int k = 1; k < 4; ++k) r = a.T_wp[i * 4 + k] * dT01[k * 4 + j] + r;
      s_T[tid] = r;
    }
    if (tid == 0) {
      s_idmin = a.id_base;
      s_old = 0;
    }
    __syncthreads();
    if (tid < 16) {  // svo_inv_se3 of it; both into the frame table
      const int i = tid >> 2, j = tid & 3;
      float v;
      if (i == 3)
        v = j == 3 ? 1.0f : 0.0f;
      else if (j < 3)
        v = s_T[j * 4 + i];
      else
        v = ((-s_T[0 * 4 + i]) * s_T[3] + (-s_T[1 * 4 + i]) * s_T[7]) + (-s_T[2 * 4 + i]) * s_T[11];
      s_T[16 + tid] = v;
      a.frameT[(size_t)(a.f & (MVO_FRAME_RING - 1)) * 32 + tid] = s_T[tid];
      a.frameT[(size_t)(a.f & (MVO_FRAME_RING - 1)) * 32 + 16 + tid] = v;
    }
    __syncthreads();
    int base = 0, kft = 0;
    for (int c0 = 0; c0 < a.n; c0 += NT) {
      const int k = c0 + tid;
      const int ok = (k < a.n && a.stage[k] == 4) ? 1 : 0;
      int total;
      const int pos = base + mvo_block_scan<NW>(ok, s_w, total);
      if (ok) {
        kft += (a.cur.t.flags[k] & VO_LM_KF_MEMBER) ? 1 : 0;
        if (pos == 0) s_idmin = a.cur.t.ids[k];
        // the frame-pose ring holds MVO_FRAME_RING frames: an older first observation would read another frame's pose
        if (a.f - a.cur.f_first[k] >= MVO_FRAME_RING) s_old = 1;
        if (pos < a.cap) mvo_put_survivor(a, k, pos, a.pts1[2 * k], a.pts1[2 * k + 1], s_T);
      }
      base += total;
    }
    const int n_surv = base;
    for (int c0 = 0; c0 < m; c0 += NT) {
      const int j = c0 + tid;
      const int ok = (j < m && a.mnew[j]) ? 1 : 0;
      int total;
      const int r = base + mvo_block_scan<NW>(ok, s_w, total);
      if (ok && r < a.cap) mvo_put_new(a, r, a.id_base + (r - n_surv), a.cand0[2 * j], a.cand0[2 * j + 1], a.cand1[2 * j], a.cand1[2 * j + 1], s_T);
      base += total;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kft += __shfl_down(kft, off);
    if ((tid & 63) == 0) s_kft[tid >> 6] = kft;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < NW; ++k) t += s_kft[k];
      MvoHdr h;
      h.n_surv = n_surv;
      h.n_new = base - n_surv;
      h.n_next = base < a.cap ? base : a.cap;
      h.n_kf_tracked = t;
      h.overflow = (base > a.cap ? 1 : 0) | (s_old ? 2 : 0);
      h.n_recon = 0;
      h.pad = 0;
      h.seq = 0;
      h.id_min = s_idmin;  // (the next set's first id; the first new id when nothing survived)
      *a.hdr_dev = h;
      *a.hdr_host = h;
    }
  }
  // the host polls the sequence word: everything this workgroup stored to host memory must be visible before it
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&a.hdr_host->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
