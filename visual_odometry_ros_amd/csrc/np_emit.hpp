// np_emit.hpp — the tail of the closed step [10]: extractor_->updateWeightBin(final pixels) and the emission of the
// bucketed candidates of the bins left empty (stereo_vo.cpp:691-711, mono_vo.cpp:977-1001), as a device function that
// runs inside the BA launch's epilogue (gn_pose.hip: stereo frame; mono_gate.hpp: mono frame).
#pragma once
#include "vo_internal.hpp"

struct VoNpArgs {
  int bins;                 // > 0: enabled; n_bins_u * n_bins_v
  int bins_u, u_step, v_step;
  const uint8_t *has;       // [bins] table: the bin holds a keypoint
  const float *xy;          // [bins][2] table: its pixel
  const float *bin_r;       // [bins][2] frame kernel: tracked position of the bin's candidate in the other image
  const uint8_t *bin_m;     // [bins]    frame kernel: trackBidirection mask of the bin's candidate
  float *out_l, *out_r;     // compacted, bins ascending (inside the result block)
  uint8_t *out_m;
  float *host_l, *host_r;   // the same places in the pinned host block, or null (the caller copies the block)
  uint8_t *host_m;
  const int *cand_done;     // non-null: the candidates are tracked by a launch of their own that may still be RUNNING (the
  int cand_target;          // synchronous call): wait until *cand_done has reached cand_target (cumulative, bounded), then read
                            // has / xy / bin_r / bin_m past the caches — they were written while this kernel ran
  const int *acc_bin;       // StereoVO: [bins] DLT depth test of every bin's candidate (written through by the workers), or null
  uint8_t *out_acc;         // StereoVO: [emitted] trackBidirection mask && depth test = the candidate becomes a landmark
};

// `nthr` threads of ONE workgroup (a multiple of 64, at most 1024). final(i): feature i is in lmtrack_final; its pixel
// is pix[2i], pix[2i+1]. s_occ: >= bins bytes of LDS, s_wv: >= 4 * nthr / 64 ints of LDS. Returns (to thread 0's
// *count) the number of candidates emitted. feature_extractor.h:116-135 (updateWeightBin: reset to 1, then 0 for every
// bin that holds a final feature; only the flattened index is range-tested, :130), feature_extractor.cpp:262-277
// (bins ascending, weight > 0).
// bounded device-side join on the candidates' launch (one lane polls; true = arrived)
__device__ __forceinline__ bool vo_np_wait_candidates(const VoNpArgs &a) {
  int polls = 0;
  while ((int)(__hip_atomic_load(a.cand_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.cand_target) < 0) {
    if (++polls > (1 << 17)) return false;  // ~0.1 s: the side stream does not run next to this kernel
    __builtin_amdgcn_s_sleep(16);
  }
  return true;
}
__device__ __forceinline__ int vo_np_ld8(const VoNpArgs &a, const uint8_t *p) {
  return a.cand_done ? (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (int)*p;
}
__device__ __forceinline__ float vo_np_ldf(const VoNpArgs &a, const float *p) {
  return a.cand_done ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

template <typename Final>
__device__ __forceinline__ void vo_np_emit(const VoNpArgs &a, int n, const float *pix, Final final, int tid, int nthr,
                                           uint8_t *s_occ, int *s_wv, int *count) {
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  for (int j = tid; j < a.bins; j += nthr) s_occ[j] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += nthr)
    if (final(i)) {
      const int u_idx = (int)floorf(pix[2 * i] / (float)a.u_step);
      const int v_idx = (int)floorf(pix[2 * i + 1] / (float)a.v_step);
      const int bin_idx = v_idx * a.bins_u + u_idx;
      if (bin_idx >= 0 && bin_idx < a.bins) s_occ[bin_idx] = 1;
    }
  __syncthreads();
  // compaction, bins ascending: four chunks of nthr bins per round (one round for <= 4 * nthr bins), two barriers a
  // round — every thread adds up the per-wavefront counts of the round itself instead of waiting for one that does
  constexpr int NCH = 4;
  int base = 0;  // entries emitted by the rounds before this one (the same in every thread)
  for (int c0 = 0; c0 < a.bins; c0 += NCH * nthr) {
    bool keep[NCH];
    int below[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int j = c0 + q * nthr + tid;
      keep[q] = j < a.bins && vo_np_ld8(a, &a.has[j >= a.bins ? 0 : j]) && !s_occ[j];
      const unsigned long long bal = __ballot(keep[q]);
      below[q] = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wv[q * nw + wave] = __popcll(bal);
    }
    __syncthreads();
    int off = base;  // entries in front of this thread's first chunk's wavefront
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      int woff = 0, tot = 0;
      for (int w = 0; w < nw; ++w) {
        const int cw = s_wv[q * nw + w];
        woff += w < wave ? cw : 0;
        tot += cw;
      }
      if (keep[q]) {
        const int j = c0 + q * nthr + tid;
        const int o = off + woff + below[q];
        const float lx = vo_np_ldf(a, &a.xy[2 * j]), ly = vo_np_ldf(a, &a.xy[2 * j + 1]);
        const float rx = vo_np_ldf(a, &a.bin_r[2 * j]), ry = vo_np_ldf(a, &a.bin_r[2 * j + 1]);
        const uint8_t mk = (uint8_t)vo_np_ld8(a, &a.bin_m[j]);
        a.out_l[2 * o] = lx;
        a.out_l[2 * o + 1] = ly;
        a.out_r[2 * o] = rx;
        a.out_r[2 * o + 1] = ry;
        a.out_m[o] = mk;
        if (a.acc_bin) a.out_acc[o] = (mk && __hip_atomic_load(&a.acc_bin[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1 : 0;
        if (a.host_l) {  // (straight into the pinned host block: only the emitted entries cross the bus)
          a.host_l[2 * o] = lx;
          a.host_l[2 * o + 1] = ly;
          a.host_r[2 * o] = rx;
          a.host_r[2 * o + 1] = ry;
          a.host_m[o] = mk;
        }
      }
      off += tot;
    }
    base = off;
    __syncthreads();  // s_wv is reused by the next round
  }
  if (tid == 0) *count = base;

  __syncthreads();
}
