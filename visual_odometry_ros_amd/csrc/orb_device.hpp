// orb_device.hpp — device code of the keypoint detector shared by the per-stage kernels (orb_detect.hip: vo_orb_detect,
// the general path) and the tile kernels of the ingestion chain (orb_tile.hpp): FAST-9/16 score, non-max test, Harris
// response, the order-preserving key and the rank selection of retainBest. Restates cv::ORB as oracle/oracle_orb.c does
// (extractor_orb_->detect, core/visual_odometry/feature_extractor.cpp:241; OpenCV 4 features2d orb.cpp / fast.cpp /
// fast_score.cpp / keypoint.cpp — third party, not in the reference tree).
//
// Plain C++ apart from __device__ / __shared__ / __syncthreads / atomics; the includer provides
//   int orb_wave_count(bool)   number of lanes of the caller's wavefront whose argument is true, the same value in every
//                              lane (on the device: the population count of the compare's lane mask — scalar instructions)
//   int orb_wave_rank(bool, int *n)   number of lower lanes whose argument is true; *n: all lanes' (every lane calls it)
//   int orb_wave_first(int v, bool)   v of the lowest lane whose second argument is true (every lane calls it)
// so that tests/emu/ can run the same text on CPU threads.
#pragma once
#include <stdint.h>

#define ORB_MAX_LEVELS 12

struct OrbLevel {
  const uint8_t *img;  // level image
  int w, h, stride;
  uint8_t *score;      // w x h
  int *row_count;      // h
  int *row_off;        // h
  int cand_base;       // first candidate slot of the level
  int quota;           // n_l
  float scale;
};
struct OrbDev {
  int n_levels, edge, fast_thr, cand_cap;
  OrbLevel L[ORB_MAX_LEVELS];
  int *hist;        // n_levels x 256
  int *lvl_total;   // n_levels: candidates after NMS + border
  int *lvl_cut;     // n_levels: FAST score cut
  unsigned *lvl_rcut;  // n_levels: ordered-uint Harris cut (0 = keep all)
  short *cx, *cy;   // candidate coordinates
  uint8_t *cs;      // candidate FAST score
  float *cr;        // candidate Harris response (valid when score >= cut)
  float *out_xy, *out_resp;
  int32_t *out_oct;
  int *out_n;
  int max_out;
  int *flags;       // bit 0: candidate capacity exceeded, bit 1: output capacity exceeded
};


// ---- FAST-9/16 --------------------------------------------------------------------------------------
// 0 when the pixel is not a corner; else cornerScore<16>: max over the 16 arcs of 9 contiguous circle pixels of
// min(v - x) and of min(x - v), floored at the threshold, minus 1
__device__ __forceinline__ int orb_fast_score(const uint8_t *__restrict__ p, int stride, int t) {
  const int v = p[0];
  // the four compass points first (fast.cpp's quick reject): 9 contiguous pixels contain at least two of them
  const int c0 = v - p[3 * stride], c4 = v - p[3], c8 = v - p[-3 * stride], c12 = v - p[-3];
  const int nd = (c0 > t) + (c4 > t) + (c8 > t) + (c12 > t), nb = (c0 < -t) + (c4 < -t) + (c8 < -t) + (c12 < -t);
  if (nd < 2 && nb < 2) return 0;
  int d[16];
  d[0] = c0;
  d[1] = v - p[1 + 3 * stride];
  d[2] = v - p[2 + 2 * stride];
  d[3] = v - p[3 + stride];
  d[4] = c4;
  d[5] = v - p[3 - stride];
  d[6] = v - p[2 - 2 * stride];
  d[7] = v - p[1 - 3 * stride];
  d[8] = c8;
  d[9] = v - p[-1 - 3 * stride];
  d[10] = v - p[-2 - 2 * stride];
  d[11] = v - p[-3 - stride];
  d[12] = c12;
  d[13] = v - p[-3 + stride];
  d[14] = v - p[-2 + 2 * stride];
  d[15] = v - p[-1 + 3 * stride];
  // min / max over every arc of 9 by doubling: windows of 2, 4, 8 (cyclic), then one more element
  int lo[16], hi[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int e = d[(k + 1) & 15];
    lo[k] = d[k] < e ? d[k] : e;
    hi[k] = d[k] > e ? d[k] : e;
  }
  int lo4[16], hi4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo4[k] = lo[k] < lo[(k + 2) & 15] ? lo[k] : lo[(k + 2) & 15];
    hi4[k] = hi[k] > hi[(k + 2) & 15] ? hi[k] : hi[(k + 2) & 15];
  }
  int best_lo = -1000, best_hi = 1000;  // max over arcs of min(d) ; min over arcs of max(d)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int mn = lo4[k] < lo4[(k + 4) & 15] ? lo4[k] : lo4[(k + 4) & 15];  // d[k .. k+7]
    int mx = hi4[k] > hi4[(k + 4) & 15] ? hi4[k] : hi4[(k + 4) & 15];
    const int e = d[(k + 8) & 15];
    mn = mn < e ? mn : e;
    mx = mx > e ? mx : e;
    best_lo = best_lo > mn ? best_lo : mn;
    best_hi = best_hi < mx ? best_hi : mx;
  }
  // corner iff some arc is entirely > t (best_lo > t) or entirely < -t (best_hi < -t)
  if (!(best_lo > t || best_hi < -t)) return 0;
  int a0 = t;
  a0 = a0 > best_lo ? a0 : best_lo;
  int b0 = -a0;
  b0 = b0 < best_hi ? b0 : best_hi;
  return -b0 - 1;
}
// strictly greater than the 8 neighbours' scores (FAST_t's non-max suppression); x, y at least 1 from the edge
__device__ __forceinline__ int orb_is_max(const uint8_t *__restrict__ s, int w, int x, int y) {
  const uint8_t *p = s + (size_t)y * w + x;
  const int c = p[0];
  if (!c) return 0;
  return c > p[-1] && c > p[1] && c > p[-w - 1] && c > p[-w] && c > p[-w + 1] && c > p[w - 1] && c > p[w] && c > p[w + 1];
}

// HarrisResponses (orb.cpp), blockSize 7, k = 0.04. The 9x9 neighbourhood is read once (81 byte loads instead of
// 8 per tap); the sums are integers, so the order does not matter.
__device__ __forceinline__ float orb_harris(const uint8_t *__restrict__ img, int stride, int x0, int y0) {
  int px[9][9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const uint8_t *p = img + (size_t)(y0 - 4 + r) * stride + (x0 - 4);
#pragma unroll
    for (int q = 0; q < 9; ++q) px[r][q] = p[q];
  }
  int a = 0, b = 0, c = 0;
#pragma unroll
  for (int r = 1; r < 8; ++r)
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const int Ix = (px[r][q + 1] - px[r][q - 1]) * 2 + (px[r - 1][q + 1] - px[r - 1][q - 1]) + (px[r + 1][q + 1] - px[r + 1][q - 1]);
      const int Iy = (px[r + 1][q] - px[r - 1][q]) * 2 + (px[r + 1][q - 1] - px[r - 1][q - 1]) + (px[r + 1][q + 1] - px[r - 1][q + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

// the same from an image in LDS whose rows start at 4-byte-aligned addresses (orb_tile.hpp): the nine bytes of a row come
// from three ALIGNED dword reads (a misaligned wide LDS read is several times slower: vo_layout.hpp, vo_bytes4)
__device__ __forceinline__ float orb_harris_lds(const uint8_t *__restrict__ img4, int stride, int x0, int y0) {
  int px[9][9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int off = (y0 - 4 + r) * stride + (x0 - 4), sh = off & 3;
    const uint32_t *w = (const uint32_t *)(img4 + (off & ~3));
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    const uint32_t lo = VO_ALIGNBYTE(w1, w0, sh), hi = VO_ALIGNBYTE(w2, w1, sh);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      px[r][q] = (int)((lo >> (8 * q)) & 255u);
      px[r][4 + q] = (int)((hi >> (8 * q)) & 255u);
    }
    px[r][8] = (int)((w2 >> (8 * sh)) & 255u);
  }
  int a = 0, b = 0, c = 0;
#pragma unroll
  for (int r = 1; r < 8; ++r)
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const int Ix = (px[r][q + 1] - px[r][q - 1]) * 2 + (px[r - 1][q + 1] - px[r - 1][q - 1]) + (px[r + 1][q + 1] - px[r + 1][q - 1]);
      const int Iy = (px[r + 1][q] - px[r - 1][q]) * 2 + (px[r + 1][q - 1] - px[r - 1][q - 1]) + (px[r + 1][q + 1] - px[r - 1][q + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

// float -> unsigned that orders the same way (NaN aside)
__device__ __forceinline__ unsigned orb_ord(float r) {
  const unsigned bits = __float_as_uint(r);
  return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
// ... and back (exact: the map is a bijection on the bit patterns)
__device__ __forceinline__ float orb_unord(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
// rank-th largest bin of a 256-bin histogram in LDS: returns the bin, *above = entries in higher bins
__device__ __forceinline__ int orb_hist_rank(const int *hist, int rank, int *above) {
  int acc = 0, b = 255;
  for (; b > 0; --b) {
    if (acc + hist[b] >= rank) break;
    acc += hist[b];
  }
  *above = acc;
  return b;
}

// ---- retainBest: the value of rank k --------------------------------------------------------------------------------
// keypoint.cpp KeyPointsFilter::retainBest(n) keeps everything >= the n-th largest value (ties stay). With a level's
// candidates in the registers of one workgroup (NQ per thread, key 0 = no candidate) the n-th largest key is found by
// bisection on the VALUE: count(key >= t) over the workgroup per step — no sort, no atomics on the data, any order of the
// candidates. Round 5: one barrier per step instead of two (three rotating count words), the per-thread compare loop
// sized by the level's candidate count (NQ = 4, 8, 16, 32), and an early exit — as soon as the undecided value interval
// holds at most 64 keys they are gathered and ranked against each other by one wavefront (typically after 10-14 of the
// 32 steps of a float key). Round 4's form took 28.6 us per image at 1241 x 376 (profiles/r04_a_*: orb_select_kernel).
// 512 lanes: two wavefronts per SIMD. (1024 lanes = four per SIMD: at more than 56 registers per lane such a workgroup cannot be
// placed on a compute unit where a wavefront of the strict-border replay pool — 276 registers — waits for its frame kernel,
// and while that pool waits it has one on EVERY compute unit: orb_finish_kernel at 79 registers held a mono frame back for the
// 134 ms of the pool's bounded waits, then the frame was re-issued. The compiler offers no way to cap the registers below the
// occupancy-8 budget of 64; tests/test_kernel_resources.py checks lanes / 256 x registers <= 236 for the side-chain kernels.)
#define ORB_ST 512       // threads of a selecting workgroup
#define ORB_RC 32        // candidates per thread it can hold in registers
#define ORB_GATHER 64

struct OrbSelShared {
  int cnt[3];            // rotating count words of orb_count_ge
  int n_list;
  unsigned list[ORB_GATHER];
  unsigned ans;
};

// count(key >= t) over the workgroup; `phase` advances by one per call (uniform). One barrier.
template <int NQ>
__device__ __forceinline__ int orb_count_ge(const unsigned (&key)[NQ], unsigned t, OrbSelShared *S, int &phase) {
  int c = 0;  // (wave-uniform: a compare and a scalar population count per register, no cross-lane reduction)
#pragma unroll
  for (int q = 0; q < NQ; ++q) c += orb_wave_count(key[q] >= t);
  const int slot = phase % 3;
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&S->cnt[slot], c);
  __syncthreads();
  const int tot = S->cnt[slot];
  // the word of call phase + 2 was last read in call phase - 1: every thread is past those reads (it is past this call's
  // barrier), and nobody adds to it before the barrier of call phase + 1
  if (threadIdx.x == 0) S->cnt[(phase + 2) % 3] = 0;
  ++phase;
  return tot;
}

// the k-th largest key (1 <= k <= kept = number of non-zero keys); `nbits`: keys are below 2^nbits
template <int NQ>
__device__ __forceinline__ unsigned orb_kth_largest(const unsigned (&key)[NQ], int k, int kept, int nbits, OrbSelShared *S, int &phase) {
  unsigned t = 0;
  int cnt_t = kept, above = 0;  // count(key >= max(t, 1)), count(key >= t + 2^(bit + 1))
#pragma unroll 1  // (unrolled, the 9 + 32 steps of four instantiations were 18 000 instructions of straight-line code that ran
                  //  once: every step an instruction-cache miss, ~1 us per step instead of ~0.3)
  for (int bit = nbits - 1; bit >= 0; --bit) {
    const unsigned cand = t | (1u << bit);
    const int c = orb_count_ge<NQ>(key, cand, S, phase);
    if (c >= k) {
      t = cand;
      cnt_t = c;
    } else {
      above = c;
    }
    // the answer is the (k - above)-th largest of the cnt_t - above keys in [max(t, 1), t + 2^bit)
    if (bit > 0 && cnt_t - above <= ORB_GATHER) {
      const unsigned lo = t ? t : 1u, span = 1u << bit;
      if (threadIdx.x == 0) S->n_list = 0;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (key[q] >= lo && key[q] - t < span) S->list[atomicAdd(&S->n_list, 1)] = key[q];
      __syncthreads();
      const int m = S->n_list, r = k - above;
      if ((int)threadIdx.x < m) {
        const unsigned e = S->list[threadIdx.x];
        int g = 0, eq = 0;
        for (int j = 0; j < m; ++j) {
          g += S->list[j] > e;
          eq += S->list[j] == e;
        }
        if (g < r && r <= g + eq) S->ans = e;  // (every lane that qualifies holds the same value)
      }
      __syncthreads();
      return S->ans;
    }
  }
  return t;
}

// both cuts of one level with its candidates in registers. cs / cr: the level's candidate scores and responses, n of
// them (n <= NQ * ORB_ST). On return key[q] = ordered response (orb_ord; orb_unord gives the response back) of candidate
// tid + q * ORB_ST if it survives both cuts, else 0; *cut, *rcut, *surv as orb_select_kernel always reported them.
// Register budget: ONE array of NQ live across the barriers — the scores during the first cut, the ordered responses (loaded
// behind it) during the second. With scores, responses and coordinates all held from the start a 1024-lane workgroup needed
// 79 registers per lane and no longer fitted next to a wavefront of the strict-border replay pool (orb_tile.hpp).
template <int NQ>
__device__ __forceinline__ void orb_select_regs(const uint8_t *__restrict__ cs, const float *__restrict__ cr, int n, int quota,
                                                OrbSelShared *S, unsigned (&key)[NQ], int *cut_out, unsigned *rcut_out, int *surv_out) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + q * ORB_ST;
    key[q] = i < n ? (unsigned)cs[i] + 1u : 0u;  // pass 1: score + 1 (0 = no candidate)
  }
  if (tid < 3) S->cnt[tid] = 0;
  __syncthreads();
  int phase = 0;
  // (1) retainBest(2 n_l) on the FAST score
  int cut = 0;
  const int keep = 2 * quota;
  if (n > keep) {
    if (keep == 0)
      cut = 256;
    else
      cut = (int)orb_kth_largest<NQ>(key, keep, n, 9, S, phase) - 1;
  }
  // (2) retainBest(n_l) on the Harris response of what is left
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + q * ORB_ST;
    const bool keep_q = key[q] != 0u && (int)key[q] - 1 >= cut;
    key[q] = keep_q ? orb_ord(cr[i]) : 0u;  // (i < n where keep_q holds)
  }
  const int kept = orb_count_ge<NQ>(key, 1u, S, phase);
  unsigned rcut = 0u;  // 0 = retainBest leaves the set alone
  int surv = kept;
  if (kept > quota) {
    if (quota == 0) {
      rcut = 0xFFFFFFFFu;
    } else {
      rcut = orb_kth_largest<NQ>(key, quota, kept, 32, S, phase);
    }
    surv = orb_count_ge<NQ>(key, rcut, S, phase);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (key[q] < rcut) key[q] = 0u;
  }
  *cut_out = cut;
  *rcut_out = rcut;
  *surv_out = surv;
}
