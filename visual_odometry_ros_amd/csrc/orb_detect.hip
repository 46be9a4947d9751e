// orb_detect.hip — keypoint detection of FeatureExtractor::extractORBwithBinning_fast on the device
// (SURVEY.md §8f #1): `extractor_orb_->detect(img, fts)` (core/visual_odometry/feature_extractor.cpp:241) with
// the parameters of initParams (:30-57), chained with the bucketing of :244-277 (misc_kernels.hip).
//
// cv::ORB is OpenCV 4 features2d and is not in the reference tree; its algorithm is restated from the upstream
// sources as known to the author (orb.cpp computeKeyPoints / HarrisResponses, fast.cpp FAST_t<16>,
// fast_score.cpp cornerScore<16>, keypoint.cpp runByImageBorder / retainBest, resize.cpp INTER_LINEAR_EXACT)
// — see oracle/oracle_orb.c for the statement of what is reproduced and of what could not be verified.
//
//   orb_resize_kernel   level l from level l-1, exact fixed-point bilinear (8.8 coefficients from host tables,
//                       16.16 vertical pass, +0.5 rounding); one lane per destination pixel; 7 dependent launches
//   orb_score_kernel    every level in one launch (blockIdx.z): FAST-9/16 test and cornerScore per pixel
//   orb_count_kernel    one wavefront per image row inside the border: non-max suppression (strict 3x3 maximum of
//                       the score), survivors per row
//   orb_plan_kernel     one workgroup per level: row offsets (exclusive scan)
//   orb_emit_kernel     one wavefront per row: candidates in raster order (ballot compaction at the row offset)
//   orb_harris_kernel   one lane per candidate: 7x7 block of 3x3 derivative sums, integer, as HarrisResponses
//   orb_select_kernel   one workgroup per level: FAST-score cut of retainBest(2 n_l) from an LDS histogram, then the
//                       response of rank n_l by 4-pass radix select (retainBest(n_l) keeps everything >= it)
//   orb_output_kernel   one workgroup per level: ordered compaction -> (x, y) * scale, response, octave
// Everything is deterministic (integer atomics only for counts and histograms); the keypoint order is level,
// then raster order — cv's own order after nth_element is unspecified, and the reference depends on it only
// through exact ties of float responses in the bucketing.
#include <stdlib.h>
#include <cmath>
#include <vector>

#include "vo_internal.hpp"
#include "vo_kernels.hpp"
#include "frame_state.hpp"

// what orb_device.hpp / orb_tile.hpp ask their includer for (tests/emu/ provides CPU stand-ins of the same names)
__device__ __forceinline__ int orb_wave_count(bool p) { return __popcll(__ballot(p)); }
__device__ __forceinline__ int orb_wave_rank(bool p, int *n) {
  const unsigned long long m = __ballot(p);
  *n = __popcll(m);
  return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}
__device__ __forceinline__ int orb_wave_first(int v, bool p) {
  const unsigned long long m = __ballot(p);
  return m ? __shfl(v, __ffsll((long long)m) - 1) : 0;
}
#define ORB_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
#define ORB_SET_PRIO()  // (s_setprio 3 for these short kernels next to the frame kernel: measured, no change — DESIGN.md facts table)
#define ORB_LD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_ST_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_ATOMIC_INC_AGENT(p) __hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_ATOMIC_ADD_AGENT(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define ORB_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#include "orb_tile.hpp"

// ---- pyramid level: resize INTER_LINEAR_EXACT ------------------------------------------------------
struct OrbResizeArgs {
  const uint8_t *src;
  int sw, sh, sstride;
  uint8_t *dst;
  int dw, dh;
  const int *ox, *cx, *oy, *cy;  // per destination column / row: source offset and 8-bit weight of the next sample
};
__global__ __launch_bounds__(256) void orb_resize_kernel(OrbResizeArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= a.dw) return;
  const int o = a.ox[x], a1 = a.cx[x], a0 = 256 - a1;
  const int yo = a.oy[y], b1 = a.cy[y], b0 = 256 - b1;
  const uint8_t *r0 = a.src + (size_t)yo * a.sstride + o, *r1 = r0 + a.sstride;
  const unsigned h0 = (unsigned)a0 * r0[0] + (unsigned)a1 * r0[1];  // horizontal pass, 8.8
  const unsigned h1 = (unsigned)a0 * r1[0] + (unsigned)a1 * r1[1];
  const unsigned v = (unsigned)b0 * h0 + (unsigned)b1 * h1;          // vertical pass, 16.16
  const unsigned r = (v + 32768u) >> 16;
  a.dst[(size_t)y * a.dw + x] = (uint8_t)(r > 255u ? 255u : r);
}

__global__ __launch_bounds__(256) void orb_score_kernel(OrbDev d) {
  const OrbLevel &L = d.L[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (y >= L.h || x >= L.w) return;
  int s = 0;
  if (x >= 3 && x < L.w - 3 && y >= 3 && y < L.h - 3) s = orb_fast_score(L.img + (size_t)y * L.stride + x, L.stride, d.fast_thr);
  L.score[(size_t)y * L.w + x] = (uint8_t)s;
}

// rows inside the border: blockIdx.x = row - edge, blockIdx.z = level
__global__ __launch_bounds__(64) void orb_count_kernel(OrbDev d) {
  const OrbLevel &L = d.L[blockIdx.z];
  const int y = (int)blockIdx.x + d.edge, lane = threadIdx.x;
  if (L.w <= 2 * d.edge || y >= L.h - d.edge) return;  // runByImageBorder clears everything on a too-small image
  int cnt = 0;
  for (int x0 = d.edge; x0 < L.w - d.edge; x0 += 64) {
    const int x = x0 + lane;
    const int m = x < L.w - d.edge ? orb_is_max(L.score, L.w, x, y) : 0;
    cnt += __popcll(__ballot(m));
  }
  if (lane == 0) L.row_count[y] = cnt;
}

// per level: exclusive scan of the row counts, level total
__global__ __launch_bounds__(256) void orb_plan_kernel(OrbDev d) {
  __shared__ int s_wsum[4];
  __shared__ int s_run;
  const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const OrbLevel &L = d.L[l];
  const int y0 = d.edge, y1 = (L.w > 2 * d.edge) ? L.h - d.edge : y0;  // rows [y0, y1)
  if (tid == 0) s_run = 0;
  __syncthreads();
  for (int c0 = y0; c0 < y1; c0 += 256) {
    const int y = c0 + tid;
    const int v = y < y1 ? L.row_count[y] : 0;
    int inc = v;  // inclusive scan within the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wsum[w];
    const int base = s_run;
    if (y < y1) L.row_off[y] = base + woff + inc - v;
    __syncthreads();
    if (tid == 255) s_run = base + woff + inc;
    __syncthreads();
  }
  if (tid == 0) {
    d.lvl_total[l] = s_run;
    if (s_run > d.cand_cap) atomicOr(d.flags, 1);
  }
}

__global__ __launch_bounds__(64) void orb_emit_kernel(OrbDev d) {
  const int l = blockIdx.z;
  const OrbLevel &L = d.L[l];
  const int y = (int)blockIdx.x + d.edge, lane = threadIdx.x;
  if (L.w <= 2 * d.edge || y >= L.h - d.edge) return;
  if (d.lvl_total[l] > d.cand_cap) return;  // flagged by orb_plan_kernel
  int off = L.cand_base + L.row_off[y];
  for (int x0 = d.edge; x0 < L.w - d.edge; x0 += 64) {
    const int x = x0 + lane;
    const int m = x < L.w - d.edge ? orb_is_max(L.score, L.w, x, y) : 0;
    const unsigned long long bal = __ballot(m);
    if (m) {
      const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
      const int s = L.score[(size_t)y * L.w + x];
      d.cx[o] = (short)x;
      d.cy[o] = (short)y;
      d.cs[o] = (uint8_t)s;
    }
    off += __popcll(bal);
  }
}

// Harris response of every candidate: one lane each (the candidate lists are dense, the corners are not)
__global__ __launch_bounds__(256) void orb_harris_kernel(OrbDev d) {
  const int l = blockIdx.y;
  const OrbLevel &L = d.L[l];
  const int n = d.lvl_total[l] > d.cand_cap ? 0 : d.lvl_total[l];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = L.cand_base + i;
  d.cr[o] = orb_harris(L.img, L.stride, d.cx[o], d.cy[o]);
}

// per level: the two retainBest cuts. (1) FAST score: nothing is dropped unless there are more than 2 n_l
// candidates, else everything >= the score of rank 2 n_l stays; (2) Harris response among those: everything >= the
// response of rank n_l. Also the number of survivors. Levels of up to 16 384 candidates (every level of a KITTI-sized
// image): orb_select_regs (orb_device.hpp), the candidates in registers; beyond that an LDS histogram and a 4-pass radix
// select on the ordered key.
template <int NQ>
__device__ __forceinline__ void orb_select_fast(const OrbDev &d, int l, int n, OrbSelShared *S) {
  const OrbLevel &L = d.L[l];
  unsigned key[NQ];
  int cut, surv;
  unsigned rcut;
  orb_select_regs<NQ>(d.cs + L.cand_base, d.cr + L.cand_base, n, L.quota, S, key, &cut, &rcut, &surv);
  if (threadIdx.x == 0) {
    d.lvl_cut[l] = cut;
    d.lvl_rcut[l] = rcut;
    d.hist[l] = surv;
  }
}
__global__ __launch_bounds__(ORB_ST) void orb_select_kernel(OrbDev d) {
  __shared__ int s_hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_rank, s_cut, s_kept, s_surv;
  __shared__ OrbSelShared s_sel;
  const int l = blockIdx.x, tid = threadIdx.x;
  const OrbLevel &L = d.L[l];
  const int n = d.lvl_total[l] > d.cand_cap ? 0 : d.lvl_total[l];
  const uint8_t *cs = d.cs + L.cand_base;
  const float *cr = d.cr + L.cand_base;
  if (n <= ORB_RC * ORB_ST) {
    if (n <= 4 * ORB_ST)
      orb_select_fast<4>(d, l, n, &s_sel);
    else if (n <= 8 * ORB_ST)
      orb_select_fast<8>(d, l, n, &s_sel);
    else if (n <= 16 * ORB_ST)
      orb_select_fast<16>(d, l, n, &s_sel);
    else
      orb_select_fast<ORB_RC>(d, l, n, &s_sel);
    return;
  }
  if (tid < 256) s_hist[tid] = 0;
  if (tid == 0) s_surv = 0;
  __syncthreads();
  #pragma unroll 4
  for (int i = tid; i < n; i += ORB_ST) atomicAdd(&s_hist[cs[i]], 1);
  __syncthreads();
  if (tid == 0) {
    int cut = 0, kept = n;
    const int keep = 2 * L.quota;
    if (n > keep) {
      if (keep == 0) {
        cut = 256;
        kept = 0;
      } else {
        int above;
        cut = orb_hist_rank(s_hist, keep, &above);
        kept = above + s_hist[cut];
      }
    }
    s_cut = cut;
    s_kept = kept;
    d.lvl_cut[l] = cut;
  }
  __syncthreads();
  const int cut = s_cut, kept = s_kept;
  unsigned rcut = 0u;  // 0 = retainBest leaves the set alone
  if (kept > L.quota) {
    if (L.quota == 0) {
      rcut = 0xFFFFFFFFu;
    } else {
      if (tid == 0) {
        s_prefix = 0;
        s_rank = L.quota;
      }
      __syncthreads();
      for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
#pragma unroll 4
        for (int i = tid; i < n; i += ORB_ST) {
          if (cs[i] < cut) continue;
          const unsigned k = orb_ord(cr[i]);
          if ((k & himask) == (prefix & himask)) atomicAdd(&s_hist[(k >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
          int above;
          const int b = orb_hist_rank(s_hist, s_rank, &above);
          s_rank -= above;
          s_prefix = prefix | ((unsigned)b << shift);
        }
        __syncthreads();
      }
      rcut = s_prefix;
    }
  }
  int mine = 0;
#pragma unroll 4
  for (int i = tid; i < n; i += ORB_ST) mine += cs[i] >= cut && (rcut == 0u || orb_ord(cr[i]) >= rcut);
  atomicAdd(&s_surv, mine);
  __syncthreads();
  if (tid == 0) {
    d.lvl_rcut[l] = rcut;
    d.hist[l] = s_surv;  // survivors of the level (the output kernel's offsets)
  }
}

// ordered compaction of every level's survivors (one workgroup per level; level, then raster order)
__global__ __launch_bounds__(1024) void orb_output_kernel(OrbDev d) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = blockIdx.x;
  const OrbLevel &L = d.L[l];
  if (tid == 0) {
    int base = 0;
    for (int q = 0; q < l; ++q) base += d.hist[q];
    s_base = base;
    if (l == d.n_levels - 1) {
      const int total = base + d.hist[l];
      if (total > d.max_out) atomicOr(d.flags, 2);
      *d.out_n = total < d.max_out ? total : d.max_out;
    }
  }
  __syncthreads();
  const int n = d.lvl_total[l] > d.cand_cap ? 0 : d.lvl_total[l], cut = d.lvl_cut[l];
  const unsigned rcut = d.lvl_rcut[l];
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + tid;
    bool keep = false;
    float r = 0.f;
    if (i < n) {
      r = d.cr[L.cand_base + i];
      keep = d.cs[L.cand_base + i] >= cut && (rcut == 0u || orb_ord(r) >= rcut);
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const int base = s_base;
    if (keep) {
      const int o = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
      if (o < d.max_out) {
        const float x = (float)d.cx[L.cand_base + i], y = (float)d.cy[L.cand_base + i];
        d.out_xy[2 * o] = l ? x * L.scale : x;  // keypoints[i].pt *= scale for level != firstLevel
        d.out_xy[2 * o + 1] = l ? y * L.scale : y;
        d.out_resp[o] = r;
        d.out_oct[o] = l;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base = base + tot;
    }
    __syncthreads();
  }
}

// ---- host side ---------------------------------------------------------------------
struct vo_orb_state {
  int w = 0, h = 0, n_levels = 0, nfeatures = 0, edge = 0;
  double scale_factor = 0;
  uint8_t *arena = nullptr;
  size_t cap = 0;
  // layout for the current configuration
  int lw[ORB_MAX_LEVELS], lh[ORB_MAX_LEVELS], quota[ORB_MAX_LEVELS];
  float lscale[ORB_MAX_LEVELS];
  size_t o_img[ORB_MAX_LEVELS], o_score[ORB_MAX_LEVELS], o_rc[ORB_MAX_LEVELS], o_ro[ORB_MAX_LEVELS];
  size_t o_tab[ORB_MAX_LEVELS][4];
  size_t o_hist, o_total, o_cut, o_rcut, o_cx, o_cy, o_cs, o_cr, o_oxy, o_oresp, o_ooct, o_on, o_flags;
  size_t o_keys, o_bpts, o_bidx, o_bn, o_weight;
  int cand_cap_level = 0, cand_cap = 0, max_out = 0, max_bins = 0, harris_blocks = 0;
  // asynchronous use (vo_extract_orb_with_binning_enqueue / _result): side stream, pinned result block
  hipEvent_t ev_start = nullptr, ev_done = nullptr;
  uint8_t *h_res = nullptr;
  size_t h_cap = 0;
  int pending_bins = 0;
  bool pending = false;
  vo_cand_table tab[2];  // closed step [10]: double-buffered so that frame k reads one while the detection of k+1 fills the other
  // the tile kernels of the per-bin table (orb_tile.hpp): plan of the current configuration, its tables in the arena
  OrbTilePlan plan;
  size_t o_gx = 0, o_gy = 0, o_tx[ORB_MAX_LEVELS] = {0}, o_ty[ORB_MAX_LEVELS] = {0}, o_surv = 0, o_done = 0, o_devflags = 0, o_thist = 0, o_cidx = 0, o_lcnt = 0;
  int finish_parts = 1, hist_copies = 1;
  bool tile_clean = false;  // lvl_total / done / keys are zero (the tile kernels leave them so; the per-stage kernels do not)
};

void vo_orb_free(vo_ctx *c) {
  if (c->orb) {
    if (c->orb->ev_start) (void)hipEventDestroy(c->orb->ev_start);
    if (c->orb->ev_done) (void)hipEventDestroy(c->orb->ev_done);
    if (c->orb->h_res) (void)hipHostFree(c->orb->h_res);
    if (c->orb->arena) (void)hipFree(c->orb->arena);
    for (vo_cand_table &t : c->orb->tab) {
      if (t.xy) (void)hipFree(t.xy);
      if (t.has) (void)hipFree(t.has);
      if (t.ready) (void)hipEventDestroy(t.ready);
      if (t.h_flags) (void)hipHostFree(t.h_flags);
    }
    delete c->orb;
    c->orb = nullptr;
  }
}

static int orb_prepare(vo_ctx *c, int w, int h, const vo_orb_params *p, int max_bins) {
  if (!c->orb) c->orb = new vo_orb_state();
  vo_orb_state *S = c->orb;
  const bool same = S->arena && S->w == w && S->h == h && S->n_levels == p->n_levels && S->nfeatures == p->nfeatures &&
                    S->edge == p->edge_threshold && S->scale_factor == p->scale_factor && S->max_bins >= max_bins;
  if (same) return VO_OK;
  S->w = w;
  S->h = h;
  S->n_levels = p->n_levels;
  S->nfeatures = p->nfeatures;
  S->edge = p->edge_threshold;
  S->scale_factor = p->scale_factor;
  S->max_bins = max_bins;
  // ORB_Impl::detectAndCompute level sizes and computeKeyPoints quotas (orb.cpp)
  orb_level_layout(w, h, p->n_levels, p->scale_factor, p->nfeatures, S->lw, S->lh, S->lscale, S->quota);
  for (int l = 0; l < p->n_levels; ++l)
    if (S->lw[l] < 8 || S->lh[l] < 8) VO_FAIL(c, VO_ERR_INVALID, "ORB level %d of a %dx%d image is too small", l, w, h);
  // arena
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  S->cand_cap_level = (w * (size_t)h / 16 > 65536) ? (int)(w * (size_t)h / 16) : 65536;  // NMS corners are far sparser
  S->cand_cap_level = (S->cand_cap_level + 15) & ~15;  // (orb_finish_kernel reads a level's scores four to a word)
  S->cand_cap = S->cand_cap_level;
  // the Harris launch covers the candidates a level can realistically hold; beyond that the tail of its grid idles
  S->harris_blocks = S->cand_cap_level;
  S->max_out = p->nfeatures + 4096;  // retainBest keeps ties: a little more than nfeatures can come out
  for (int l = 0; l < p->n_levels; ++l) {
    const size_t px = (size_t)S->lw[l] * S->lh[l];
    S->o_img[l] = l ? take(px) : 0;
    S->o_score[l] = take(px);
    S->o_rc[l] = take(sizeof(int) * S->lh[l]);
    S->o_ro[l] = take(sizeof(int) * S->lh[l]);
    if (l) {
      S->o_tab[l][0] = take(sizeof(int) * S->lw[l]);
      S->o_tab[l][1] = take(sizeof(int) * S->lw[l]);
      S->o_tab[l][2] = take(sizeof(int) * S->lh[l]);
      S->o_tab[l][3] = take(sizeof(int) * S->lh[l]);
    }
  }
  const size_t nc = (size_t)S->cand_cap_level * p->n_levels;
  S->o_hist = take(sizeof(int) * 256 * p->n_levels);
  S->o_total = take(sizeof(int) * p->n_levels);
  S->o_cut = take(sizeof(int) * p->n_levels);
  S->o_rcut = take(sizeof(unsigned) * p->n_levels);
  S->o_flags = take(sizeof(int) * 4);
  S->o_cx = take(sizeof(short) * nc);
  S->o_cy = take(sizeof(short) * nc);
  S->o_cs = take(nc);
  S->o_cr = take(sizeof(float) * nc);
  S->o_oxy = take(sizeof(float) * 2 * S->max_out);
  S->o_oresp = take(sizeof(float) * S->max_out);
  S->o_ooct = take(sizeof(int32_t) * S->max_out);
  S->o_on = take(sizeof(int) * 4);
  S->o_keys = take(sizeof(unsigned long long) * (size_t)(max_bins + 1));
  S->o_bpts = take(sizeof(float) * 2 * (size_t)(max_bins + 1));
  S->o_bidx = take(sizeof(int32_t) * (size_t)(max_bins + 1));
  S->o_bn = take(sizeof(int) * 4);
  S->o_weight = take(sizeof(int32_t) * (size_t)(max_bins + 1));
  // the tile kernels' plan of this configuration (orb_plan.hpp); 48 x 32 level-0 pixels per workgroup: 312 workgroups at
  // 1241 x 376, 36 KB of LDS each
  orb_tile_plan(S->lw, S->lh, p->n_levels, p->edge_threshold, 48, 32, 64 * 1024, &S->plan);
  if (S->plan.ok && S->plan.nx * S->plan.ny > 1024) {
    // a large image has workgroups to spare: larger tiles recompute less halo (staged / owned pixels 2.9 instead of 3.8 at
    // 3840 x 2160; 80 KB of LDS, two workgroups per compute unit): 360 against 425 us for the two launches (tools/tileprobe.hip)
    OrbTilePlan big;
    orb_tile_plan(S->lw, S->lh, p->n_levels, p->edge_threshold, 56, 48, 96 * 1024, &big);
    if (big.ok) S->plan = big;
  }
  if (S->plan.ok && S->plan.lds_bytes > 64 * 1024)
    VO_CHECK_HIP(c, hipFuncSetAttribute((const void *)orb_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S->plan.lds_bytes));
  if (S->plan.ok) {
    S->o_gx = take(sizeof(OrbSpan) * S->plan.gx.size());
    S->o_gy = take(sizeof(OrbSpan) * S->plan.gy.size());
    for (int l = 1; l < p->n_levels; ++l) {
      S->o_tx[l] = take(sizeof(int) * S->plan.tabx[l].size());
      S->o_ty[l] = take(sizeof(int) * S->plan.taby[l].size());
    }
    S->o_surv = take(sizeof(int) * ORB_MAX_LEVELS);
    S->hist_copies = S->plan.nx * S->plan.ny / 64;
    if (S->hist_copies < 1) S->hist_copies = 1;
    if (S->hist_copies > 64) S->hist_copies = 64;
    S->o_thist = take(sizeof(int) * 256 * ORB_MAX_LEVELS * (size_t)S->hist_copies);
    S->o_cidx = take(sizeof(int) * (size_t)ORB_RC * ORB_ST * ORB_MAX_LEVELS);
    S->o_lcnt = take(sizeof(int) * 2 * ORB_MAX_LEVELS);
    // workgroups per level of orb_finish_kernel: a slice of at most ~8 000 candidates each on the fullest level one can expect
    // (a 3 x 3 maximum per ~40 pixels on a textured image): 1 at 1241 x 376, 26 at 3840 x 2160
    S->finish_parts = (int)((w * (size_t)h / 40 + 8191) / 8192);
    if (S->finish_parts < 1) S->finish_parts = 1;
    if (S->finish_parts > 32) S->finish_parts = 32;
    S->o_done = take(sizeof(int) * 4);
    S->o_devflags = take(sizeof(int) * 4);
  }
  S->tile_clean = false;
  if (off > S->cap) {
    if (S->arena) (void)hipFree(S->arena);
    S->arena = nullptr;
    S->cap = 0;
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&S->arena, off));
    S->cap = off;
  }
  // coefficient tables
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  for (int l = 1; l < p->n_levels; ++l) {
    std::vector<int> ox, cx, oy, cy;
    orb_linear_exact_coeffs(S->lw[l - 1], S->lw[l], ox, cx);
    orb_linear_exact_coeffs(S->lh[l - 1], S->lh[l], oy, cy);
    VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_tab[l][0], ox.data(), sizeof(int) * ox.size(), hipMemcpyHostToDevice));
    VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_tab[l][1], cx.data(), sizeof(int) * cx.size(), hipMemcpyHostToDevice));
    VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_tab[l][2], oy.data(), sizeof(int) * oy.size(), hipMemcpyHostToDevice));
    VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_tab[l][3], cy.data(), sizeof(int) * cy.size(), hipMemcpyHostToDevice));
  }
  if (S->plan.ok) {
    VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_gx, S->plan.gx.data(), sizeof(OrbSpan) * S->plan.gx.size(), hipMemcpyHostToDevice));
    VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_gy, S->plan.gy.data(), sizeof(OrbSpan) * S->plan.gy.size(), hipMemcpyHostToDevice));
    for (int l = 1; l < p->n_levels; ++l) {
      VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_tx[l], S->plan.tabx[l].data(), sizeof(int) * S->plan.tabx[l].size(), hipMemcpyHostToDevice));
      VO_CHECK_HIP(c, hipMemcpy(S->arena + S->o_ty[l], S->plan.taby[l].data(), sizeof(int) * S->plan.taby[l].size(), hipMemcpyHostToDevice));
    }
  }
  return VO_OK;
}

// The per-bin candidate table of the image in `slot` by the two tile kernels (orb_tile.hpp), on c->stream. Needs
// S->plan.ok. Leaves lvl_total / done / keys zeroed, as it needs them.
// src != null: level 0 is read from that image (device memory, src_stride bytes per row, w x h) instead of the slot's plane
static int orb_tile_enqueue(vo_ctx *c, int slot, const vo_bin_params *bp, vo_cand_table &T, const uint8_t *src = nullptr,
                            int src_stride = 0, int src_w = 0, int src_h = 0) {
  const vo_orb_params *p = &bp->orb;
  if (!src && (slot < 0 || slot >= c->cfg.n_slots || c->slots[slot].n_levels <= 0)) VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
  const int img_w = src ? src_w : c->slots[slot].w, img_h = src ? src_h : c->slots[slot].h;
  const int total = bp->n_bins_u * bp->n_bins_v;
  int rc = orb_prepare(c, img_w, img_h, p, total);
  if (rc) return rc;
  vo_orb_state *S = c->orb;
  if (!S->plan.ok) return 1;  // (the caller takes the per-stage kernels)
  if (!src && vo_slot_acquire(c, slot) < 0) return VO_ERR_HIP;
  hipStream_t s = c->stream;
  uint8_t *A = S->arena;
  if (!S->tile_clean) {  // (once after the per-stage kernels ran on this arena, never in a steady stream)
    VO_CHECK_HIP(c, hipMemsetAsync(A + S->o_total, 0, sizeof(int) * p->n_levels, s));
    VO_CHECK_HIP(c, hipMemsetAsync(A + S->o_done, 0, sizeof(int) * 4, s));
    VO_CHECK_HIP(c, hipMemsetAsync(A + S->o_thist, 0, sizeof(int) * 256 * ORB_MAX_LEVELS * (size_t)S->hist_copies, s));
    VO_CHECK_HIP(c, hipMemsetAsync(A + S->o_lcnt, 0, sizeof(int) * 2 * ORB_MAX_LEVELS, s));
    VO_CHECK_HIP(c, hipMemsetAsync(A + S->o_keys, 0, sizeof(unsigned long long) * (size_t)(S->max_bins + 1), s));
    S->tile_clean = true;
  }
  OrbTileArgs a;
  memset(&a, 0, sizeof(a));
  if (src) {
    a.img = src;
    a.img_end = src + (size_t)src_stride * (size_t)(src_h - 1) + (size_t)src_w;
    a.stride = src_stride;
  } else {
    const vo_pyramid &P = c->slots[slot];
    a.img = P.lv[0].origin();
    a.img_end = P.lv[0].base + (size_t)P.lv[0].stride * (size_t)(P.lv[0].h + 2 * VO_PAD);
    a.stride = P.lv[0].stride;
  }
  a.n_levels = p->n_levels;
  a.nx = S->plan.nx;
  a.ny = S->plan.ny;
  a.fast_thr = p->fast_threshold;
  a.cand_cap = S->cand_cap_level;
  a.stash_off = S->plan.stash_off;
  a.stash_cap = S->plan.stash_cap;
  a.gx = (const OrbSpan *)(A + S->o_gx);
  a.gy = (const OrbSpan *)(A + S->o_gy);
  for (int l = 0; l < p->n_levels; ++l) {
    OrbTileLevel &L = a.L[l];
    L.w = S->lw[l];
    L.h = S->lh[l];
    L.lds_off = S->plan.lds_off[l];
    L.lds_stride = S->plan.lds_stride[l];
    L.sc_off = S->plan.sc_off[l];
    L.sc_stride = S->plan.sc_stride[l];
    L.cand_base = l * S->cand_cap_level;
    L.tx_off = S->plan.tx_off[l];
    L.ty_off = S->plan.ty_off[l];
    L.tabx = l ? (const int *)(A + S->o_tx[l]) : nullptr;
    L.taby = l ? (const int *)(A + S->o_ty[l]) : nullptr;
  }
  a.lvl_total = (int *)(A + S->o_total);
  a.hist = (int *)(A + S->o_thist);
  a.hist_copies = S->hist_copies;
  a.cx = (short *)(A + S->o_cx);
  a.cy = (short *)(A + S->o_cy);
  a.cs = A + S->o_cs;
  a.cr = (float *)(A + S->o_cr);
  OrbFinishArgs f;
  memset(&f, 0, sizeof(f));
  f.n_levels = p->n_levels;
  f.cand_cap = S->cand_cap_level;
  f.max_out = S->max_out;
  for (int l = 0; l < p->n_levels; ++l) {
    f.cand_base[l] = l * S->cand_cap_level;
    f.quota[l] = S->quota[l];
    f.scale[l] = S->lscale[l];
  }
  f.lvl_total = a.lvl_total;
  f.parts = S->finish_parts;
  f.cidx_cap = ORB_RC * ORB_ST;
  f.hist = a.hist;
  f.hist_copies = a.hist_copies;
  f.cidx = (int *)(A + S->o_cidx);
  f.lvl_cnt = (int *)(A + S->o_lcnt);
  f.lvl_done = f.lvl_cnt + ORB_MAX_LEVELS;
  f.cx = a.cx;
  f.cy = a.cy;
  f.cs = a.cs;
  f.cr = a.cr;
  f.surv = (int *)(A + S->o_surv);
  f.done = (int *)(A + S->o_done);
  f.key = (unsigned long long *)(A + S->o_keys);
  f.n_bins_u = bp->n_bins_u;
  f.n_bins_v = bp->n_bins_v;
  f.inv_u = bp->inv_u_step;
  f.inv_v = bp->inv_v_step;
  f.tab_xy = T.xy;
  f.tab_has = T.has;
  f.host_flags = T.h_flags;
  f.dev_flags = (int *)(A + S->o_devflags);
  vo_prof_begin(c, VO_K_AUX);
  hipLaunchKernelGGL(orb_tile_kernel, dim3(a.nx * a.ny), dim3(ORB_TILE_NT), (size_t)S->plan.lds_bytes, s, a);
  hipLaunchKernelGGL(orb_finish_kernel, dim3(p->n_levels * f.parts), dim3(ORB_ST), 0, s, f);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// enqueue the detection of the image in `slot`; results stay on the device (S->o_oxy, ...)
static int orb_enqueue(vo_ctx *c, int slot, const vo_orb_params *p, int max_bins) {
  if (slot < 0 || slot >= c->cfg.n_slots || c->slots[slot].n_levels <= 0) VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
  if (p->n_levels < 1 || p->n_levels > ORB_MAX_LEVELS || p->nfeatures < 0 || p->edge_threshold < 4 ||
      !(p->scale_factor > 1.0) || p->fast_threshold < 0 || p->fast_threshold > 254)
    VO_FAIL(c, VO_ERR_INVALID, "ORB parameters out of range (1..%d levels, edge threshold >= 4, scale factor > 1)", ORB_MAX_LEVELS);
  const vo_pyramid &P = c->slots[slot];
  if (P.w > 32767 || P.h > 32767) VO_FAIL(c, VO_ERR_CAPACITY, "image too large for 16-bit candidate coordinates");
  int rc = orb_prepare(c, P.w, P.h, p, max_bins);
  if (rc) return rc;
  if (vo_slot_acquire(c, slot) < 0) return VO_ERR_HIP;
  vo_orb_state *S = c->orb;
  hipStream_t s = c->stream;
  uint8_t *A = S->arena;
  OrbDev d;
  memset(&d, 0, sizeof(d));
  d.n_levels = p->n_levels;
  d.edge = p->edge_threshold;
  d.fast_thr = p->fast_threshold;
  d.cand_cap = S->cand_cap_level;
  for (int l = 0; l < p->n_levels; ++l) {
    OrbLevel &L = d.L[l];
    L.img = l ? A + S->o_img[l] : P.lv[0].origin();
    L.w = S->lw[l];
    L.h = S->lh[l];
    L.stride = l ? S->lw[l] : P.lv[0].stride;
    L.score = A + S->o_score[l];
    L.row_count = (int *)(A + S->o_rc[l]);
    L.row_off = (int *)(A + S->o_ro[l]);
    L.cand_base = l * S->cand_cap_level;
    L.quota = S->quota[l];
    L.scale = S->lscale[l];
  }
  d.hist = (int *)(A + S->o_hist);
  d.lvl_total = (int *)(A + S->o_total);
  d.lvl_cut = (int *)(A + S->o_cut);
  d.lvl_rcut = (unsigned *)(A + S->o_rcut);
  d.flags = (int *)(A + S->o_flags);
  d.cx = (short *)(A + S->o_cx);
  d.cy = (short *)(A + S->o_cy);
  d.cs = A + S->o_cs;
  d.cr = (float *)(A + S->o_cr);
  d.out_xy = (float *)(A + S->o_oxy);
  d.out_resp = (float *)(A + S->o_oresp);
  d.out_oct = (int32_t *)(A + S->o_ooct);
  d.out_n = (int *)(A + S->o_on);
  d.max_out = S->max_out;
  // hist .. flags are contiguous in the arena: one memset
  VO_CHECK_HIP(c, hipMemsetAsync(A + S->o_hist, 0, S->o_cx - S->o_hist, s));
  S->tile_clean = false;  // (level totals and, with the bucketing behind this, the keys are left as they come out)
  vo_prof_begin(c, VO_K_AUX);
  for (int l = 1; l < p->n_levels; ++l) {
    OrbResizeArgs a;
    a.src = d.L[l - 1].img;
    a.sw = d.L[l - 1].w;
    a.sh = d.L[l - 1].h;
    a.sstride = d.L[l - 1].stride;
    a.dst = A + S->o_img[l];
    a.dw = S->lw[l];
    a.dh = S->lh[l];
    a.ox = (const int *)(A + S->o_tab[l][0]);
    a.cx = (const int *)(A + S->o_tab[l][1]);
    a.oy = (const int *)(A + S->o_tab[l][2]);
    a.cy = (const int *)(A + S->o_tab[l][3]);
    hipLaunchKernelGGL(orb_resize_kernel, dim3((a.dw + 255) / 256, a.dh), dim3(256), 0, s, a);
  }
  const int rows = P.h - 2 * p->edge_threshold;
  hipLaunchKernelGGL(orb_score_kernel, dim3((P.w + 255) / 256, P.h, p->n_levels), dim3(256), 0, s, d);
  if (rows > 0) hipLaunchKernelGGL(orb_count_kernel, dim3(rows, 1, p->n_levels), dim3(64), 0, s, d);
  hipLaunchKernelGGL(orb_plan_kernel, dim3(p->n_levels), dim3(256), 0, s, d);
  if (rows > 0) hipLaunchKernelGGL(orb_emit_kernel, dim3(rows, 1, p->n_levels), dim3(64), 0, s, d);
  hipLaunchKernelGGL(orb_harris_kernel, dim3((S->harris_blocks + 255) / 256, p->n_levels), dim3(256), 0, s, d);
  hipLaunchKernelGGL(orb_select_kernel, dim3(p->n_levels), dim3(ORB_ST), 0, s, d);
  hipLaunchKernelGGL(orb_output_kernel, dim3(p->n_levels), dim3(1024), 0, s, d);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

static int orb_check_flags(vo_ctx *c, int flags) {
  if (flags & 1) VO_FAIL(c, VO_ERR_CAPACITY, "more than %d FAST corners on one pyramid level", c->orb->cand_cap_level);
  if (flags & 2) VO_FAIL(c, VO_ERR_CAPACITY, "more keypoints than the output buffer holds");
  return VO_OK;
}

extern "C" int vo_orb_detect(vo_ctx *c, int slot, const vo_orb_params *p, float *kp_xy, float *kp_response,
                             int32_t *kp_octave, int max_kp, int *n_out) {
  if (!c || !p || !n_out || max_kp < 0) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = orb_enqueue(c, slot, p, 0);
  if (rc) return rc;
  vo_orb_state *S = c->orb;
  int n = 0, flags = 0;
  VO_CHECK_HIP(c, hipMemcpyAsync(&n, S->arena + S->o_on, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipMemcpyAsync(&flags, S->arena + S->o_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  rc = orb_check_flags(c, flags);
  if (rc) return rc;
  if (n > max_kp) VO_FAIL(c, VO_ERR_CAPACITY, "%d keypoints, the caller's buffers hold %d", n, max_kp);
  if (n > 0) {
    if (kp_xy) VO_CHECK_HIP(c, hipMemcpy(kp_xy, S->arena + S->o_oxy, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost));
    if (kp_response) VO_CHECK_HIP(c, hipMemcpy(kp_response, S->arena + S->o_oresp, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    if (kp_octave) VO_CHECK_HIP(c, hipMemcpy(kp_octave, S->arena + S->o_ooct, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
  }
  *n_out = n;
  return VO_OK;
}

// test hook: level image `level` of the last detection (tightly packed)
extern "C" int vo_orb_get_level(vo_ctx *c, int level, uint8_t *host, int *width, int *height) {
  if (!c || !c->orb || !c->orb->arena || level < 1 || level >= c->orb->n_levels) return VO_ERR_INVALID;
  vo_orb_state *S = c->orb;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  if (host) VO_CHECK_HIP(c, hipMemcpy(host, S->arena + S->o_img[level], (size_t)S->lw[level] * S->lh[level], hipMemcpyDeviceToHost));
  if (width) *width = S->lw[level];
  if (height) *height = S->lh[level];
  return VO_OK;
}

// FeatureExtractor::extractORBwithBinning_fast with flag_nonmax_ (feature_extractor.cpp:211-277): detection and
// the per-bin arg-max chained on the device; only the bucketed pixels come back
extern "C" int vo_extract_orb_with_binning(vo_ctx *c, int slot, const vo_orb_params *p, float inv_u_step,
                                           float inv_v_step, int n_bins_u, int n_bins_v, const int32_t *weight,
                                           float *pts_out, int *n_out, int *n_detected) {
  if (!c || !p || !weight || !pts_out || !n_out || n_bins_u <= 0 || n_bins_v <= 0) return VO_ERR_INVALID;
  const int total = n_bins_u * n_bins_v;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  int rc = orb_enqueue(c, slot, p, total);
  if (rc) return rc;
  vo_orb_state *S = c->orb;
  uint8_t *A = S->arena;
  VO_CHECK_HIP(c, hipMemcpyAsync(A + S->o_weight, weight, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, c->stream));
  rc = vo_bucket_argmax_enqueue(c, (const float *)(A + S->o_oxy), (const float *)(A + S->o_oresp), S->max_out, inv_u_step,
                                inv_v_step, n_bins_u, n_bins_v, (const int32_t *)(A + S->o_weight),
                                (unsigned long long *)(A + S->o_keys), (float *)(A + S->o_bpts), (int32_t *)(A + S->o_bidx),
                                (int *)(A + S->o_bn), (const int *)(A + S->o_on));
  if (rc < 0) return rc;
  int m = 0, n = 0, flags = 0;
  VO_CHECK_HIP(c, hipMemcpyAsync(&m, A + S->o_bn, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipMemcpyAsync(&n, A + S->o_on, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipMemcpyAsync(&flags, A + S->o_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  rc = orb_check_flags(c, flags);
  if (rc) return rc;
  if (m > 0) VO_CHECK_HIP(c, hipMemcpy(pts_out, A + S->o_bpts, sizeof(float) * 2 * (size_t)m, hipMemcpyDeviceToHost));
  *n_out = m;
  if (n_detected) *n_detected = n;
  return VO_OK;
}

// The same, asynchronous and off the main chain: the kernels run on the context's side stream behind the last
// pyramid build, so that detection overlaps the frame operator of the same image pair (which may have been enqueued
// before it). One detection in flight per context; `slot` must not be rebuilt before _result() returned.
extern "C" int vo_extract_orb_with_binning_enqueue(vo_ctx *c, int slot, const vo_orb_params *p, float inv_u_step,
                                                   float inv_v_step, int n_bins_u, int n_bins_v, const int32_t *weight) {
  if (!c || !p || !weight || n_bins_u <= 0 || n_bins_v <= 0) return VO_ERR_INVALID;
  const int total = n_bins_u * n_bins_v;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (c->orb && c->orb->pending) VO_FAIL(c, VO_ERR_INVALID, "a detection is already in flight");
  if (!c->orb) c->orb = new vo_orb_state();
  vo_orb_state *S = c->orb;
  if (!S->ev_start) {
    VO_CHECK_HIP(c, hipEventCreateWithFlags(&S->ev_start, hipEventDisableTiming));
    VO_CHECK_HIP(c, hipEventCreateWithFlags(&S->ev_done, hipEventDisableTiming));
  }
  const size_t need = 64 + sizeof(int32_t) * (size_t)total + sizeof(float) * 2 * (size_t)total;
  if (need > S->h_cap) {
    if (S->h_res) (void)hipHostFree(S->h_res);
    S->h_res = nullptr;
    S->h_cap = 0;
    VO_CHECK_HIP(c, vo_host_malloc(c, (void **)&S->h_res, need, hipHostMallocDefault));
    S->h_cap = need;
  }
  int32_t *h_w = (int32_t *)(S->h_res + 64);
  float *h_pts = (float *)(S->h_res + 64 + sizeof(int32_t) * (size_t)total);
  memcpy(h_w, weight, sizeof(int32_t) * (size_t)total);
  hipStream_t main_stream = c->stream;
  // the side stream waits for the slot's own build only (vo_slot_acquire in orb_enqueue) — not for what the main
  // stream has enqueued since (the frame operator)
  c->stream = c->stream2;  // every launcher below enqueues on ctx->stream
  int rc = orb_enqueue(c, slot, p, total);
  if (rc == VO_OK) {
    uint8_t *A = S->arena;
    hipError_t e = hipMemcpyAsync(A + S->o_weight, h_w, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
      rc = vo_bucket_argmax_enqueue(c, (const float *)(A + S->o_oxy), (const float *)(A + S->o_oresp), S->max_out, inv_u_step,
                                    inv_v_step, n_bins_u, n_bins_v, (const int32_t *)(A + S->o_weight),
                                    (unsigned long long *)(A + S->o_keys), (float *)(A + S->o_bpts),
                                    (int32_t *)(A + S->o_bidx), (int *)(A + S->o_bn), (const int *)(A + S->o_on));
    if (e == hipSuccess && rc == VO_OK) e = hipMemcpyAsync(S->h_res, A + S->o_bn, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == VO_OK) e = hipMemcpyAsync(S->h_res + 4, A + S->o_on, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == VO_OK) e = hipMemcpyAsync(S->h_res + 8, A + S->o_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == VO_OK)
      e = hipMemcpyAsync(h_pts, A + S->o_bpts, sizeof(float) * 2 * (size_t)total, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == VO_OK) e = hipEventRecord(S->ev_done, c->stream);
    if (e != hipSuccess) {
      snprintf(c->err, sizeof(c->err), "vo_extract_orb_with_binning_enqueue: %s", hipGetErrorString(e));
      rc = VO_ERR_HIP;
    }
  }
  c->stream = main_stream;
  if (rc < 0) return rc;
  S->pending = true;
  S->pending_bins = total;
  return VO_OK;
}

extern "C" int vo_extract_orb_with_binning_result(vo_ctx *c, float *pts_out, int *n_out, int *n_detected) {
  if (!c || !c->orb || !c->orb->pending || !pts_out || !n_out) return VO_ERR_INVALID;
  vo_orb_state *S = c->orb;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_CHECK_HIP(c, hipEventSynchronize(S->ev_done));
  S->pending = false;
  const int *hdr = (const int *)S->h_res;
  int rc = orb_check_flags(c, hdr[2]);
  if (rc) return rc;
  const int m = hdr[0];
  const float *h_pts = (const float *)(S->h_res + 64 + sizeof(int32_t) * (size_t)S->pending_bins);
  if (m > 0) memcpy(pts_out, h_pts, sizeof(float) * 2 * (size_t)m);
  *n_out = m;
  if (n_detected) *n_detected = hdr[1];
  return VO_OK;
}

// ---- closed step [10]: candidates for every bin, ahead of the frame -------------------------------------------
// `extractor_orb_->detect(I1_left)` and the per-bin arg-max of extractORBwithBinning_fast (feature_extractor.cpp:
// 241-277) depend on the image only — which bins they are needed for (updateWeightBin(lmtrack_final.pts_l1),
// stereo_vo.cpp:692) is known after the frame's BA. So the best keypoint of EVERY bin is found as soon as the image
// is on the device (side stream, overlapping the previous frame), the frame kernel tracks all of them speculatively
// next to the features (+22 us at 1500 bins against ~100 us for a dependent launch behind the BA), and the BA launch's
// epilogue emits those whose bin lmtrack_final left empty.
const vo_cand_table *vo_orb_cand_table(vo_ctx *c, int table) {
  if (!c->orb || table < 0 || table > 1 || c->orb->tab[table].n_bins <= 0) return nullptr;
  return &c->orb->tab[table];
}

static int cand_table_enqueue(vo_ctx *c, int slot, const vo_bin_params *bp, int table, const uint8_t *src, int src_stride, int src_w,
                              int src_h);
extern "C" int vo_new_point_candidates_enqueue(vo_ctx *c, int slot, const vo_bin_params *bp, int table) {
  return cand_table_enqueue(c, slot, bp, table, nullptr, 0, 0, 0);
}
int vo_new_point_candidates_enqueue_image(vo_ctx *c, const uint8_t *dev_img, int stride, int w, int h, const vo_bin_params *bp, int table) {
  if (!dev_img || stride < w || w <= 0 || h <= 0) return VO_ERR_INVALID;
  return cand_table_enqueue(c, -1, bp, table, dev_img, stride, w, h);
}
static int cand_table_enqueue(vo_ctx *c, int slot, const vo_bin_params *bp, int table, const uint8_t *src, int src_stride, int src_w,
                              int src_h) {
  if (!c || !bp || table < 0 || table > 1 || bp->n_bins_u <= 0 || bp->n_bins_v <= 0) return VO_ERR_INVALID;
  const int total = bp->n_bins_u * bp->n_bins_v;
  if (total > 32768) VO_FAIL(c, VO_ERR_CAPACITY, "%d bins: the closed step [10] handles at most 32768", total);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (!c->orb) c->orb = new vo_orb_state();
  vo_orb_state *S = c->orb;
  vo_cand_table &T = S->tab[table];
  if (c->frame && c->frame->pending && c->frame->table == &T)
    VO_FAIL(c, VO_ERR_INVALID, "candidate table %d is read by the closed frame in flight: collect its result first", table);
  if (T.n_bins != total) {
    if (T.xy) (void)hipFree(T.xy);
    if (T.has) (void)hipFree(T.has);
    T.xy = nullptr;
    T.has = nullptr;
    T.n_bins = 0;
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&T.xy, sizeof(float) * 2 * (size_t)total));
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&T.has, (size_t)total));
    if (!T.ready) VO_CHECK_HIP(c, hipEventCreateWithFlags(&T.ready, hipEventDisableTiming));
    if (!T.h_flags) VO_CHECK_HIP(c, vo_host_malloc(c, (void **)&T.h_flags, 64, hipHostMallocDefault));
    T.n_bins = total;
  }
  // MEASUREMENT ONLY (vo_debug_set VO_DBG_SKIP_DETECT): a table filled twice keeps its content — what the frame costs
  // without the detection under it (results are those of a stale table)
  if (c->dbg[VO_DBG_SKIP_DETECT] && T.dbg_filled++ >= 2) {
    VO_CHECK_HIP(c, hipEventRecord(T.ready, c->stream2));
    return VO_OK;
  }
  if (src && c->dbg[VO_DBG_STAGED_DETECT]) return 1;  // (the per-stage kernels read a slot)
  hipStream_t caller = c->stream;
  c->stream = c->stream2;  // every launcher below enqueues on ctx->stream
  // two launches (orb_tile.hpp) wherever the configuration fits them — every configuration of the reference does;
  // VO_DBG_STAGED_DETECT forces the per-stage kernels (19 launches), the same table bit for bit
  int rc = 1;
  if (!c->dbg[VO_DBG_STAGED_DETECT]) {
    rc = orb_tile_enqueue(c, slot, bp, T, src, src_stride, src_w, src_h);
    if (rc == VO_OK) {
      hipError_t e = hipEventRecord(T.ready, c->stream);
      if (e != hipSuccess) {
        snprintf(c->err, sizeof(c->err), "vo_new_point_candidates_enqueue: %s", hipGetErrorString(e));
        rc = VO_ERR_HIP;
      }
    }
  }
  if (rc <= 0 || src) {  // (src: only the tile kernels read an image that is not a slot)
    c->stream = caller;
    return rc;
  }
  rc = orb_enqueue(c, slot, &bp->orb, total);
  if (rc == VO_OK) {
    uint8_t *A = S->arena;
    rc = vo_bucket_table_enqueue(c, (const float *)(A + S->o_oxy), (const float *)(A + S->o_oresp), S->max_out,
                                 bp->inv_u_step, bp->inv_v_step, bp->n_bins_u, bp->n_bins_v,
                                 (unsigned long long *)(A + S->o_keys), T.xy, T.has, (const int *)(A + S->o_on));
    hipError_t e = hipSuccess;
    if (rc == VO_OK) e = hipMemcpyAsync(T.h_flags, A + S->o_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (rc == VO_OK && e == hipSuccess)
      e = hipMemcpyAsync(T.h_flags + 1, A + S->o_on, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (rc == VO_OK && e == hipSuccess) e = hipEventRecord(T.ready, c->stream);
    if (e != hipSuccess) {
      snprintf(c->err, sizeof(c->err), "vo_new_point_candidates_enqueue: %s", hipGetErrorString(e));
      rc = VO_ERR_HIP;
    }
  }
  c->stream = caller;
  return rc;
}

// test hook: the table as the frame kernel will see it (waits for its detection)
extern "C" int vo_new_point_candidates_get(vo_ctx *c, int table, float *xy, uint8_t *has, int *n_detected) {
  const vo_cand_table *T = c ? vo_orb_cand_table(c, table) : nullptr;
  if (!T) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_CHECK_HIP(c, hipEventSynchronize(T->ready));
  int rc = orb_check_flags(c, T->h_flags[0]);
  if (rc) return rc;
  if (xy) VO_CHECK_HIP(c, hipMemcpy(xy, T->xy, sizeof(float) * 2 * (size_t)T->n_bins, hipMemcpyDeviceToHost));
  if (has) VO_CHECK_HIP(c, hipMemcpy(has, T->has, (size_t)T->n_bins, hipMemcpyDeviceToHost));
  if (n_detected) *n_detected = T->h_flags[1];
  return VO_OK;
}
