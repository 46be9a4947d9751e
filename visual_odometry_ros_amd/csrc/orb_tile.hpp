// orb_tile.hpp — the keypoint detection of the closed step [10] in TWO launches (round 5):
//
//   orb_tile_kernel     one workgroup per image tile, no workgroup waits for another: the tile's region of level 0 is
//                       staged in LDS; levels 1 .. n-1 of cv::ORB's 1.2x pyramid (INTER_LINEAR_EXACT) are produced in
//                       LDS for the region each needs (the halo is recomputed instead of exchanged); FAST-9/16 score
//                       on the owned pixels + one ring; strict 3x3 non-max test; Harris response of the survivors;
//                       the candidates go to the level's list in HBM (one atomic per workgroup and level reserves the
//                       slots). Only pixels inside runByImageBorder's rectangle are owned, so no image border is touched.
//   orb_finish_kernel   one workgroup per level: both retainBest cuts with the level's candidates in registers
//                       (orb_device.hpp), then the per-bin arg-max of extractORBwithBinning_fast straight from the
//                       registers — 64-bit atomicMax on (response, ~(level, y, x)): the FIRST keypoint of largest
//                       response in (level, raster) order, i.e. the order of the reference's keypoint vector; the
//                       last workgroup to finish turns the keys into the per-bin table, reports flags and count to
//                       the pinned page and leaves every counter zeroed for the next image (no memset, no copy).
//
// Replaces, for the per-bin candidate table (vo_new_point_candidates_enqueue), rounds 1-4's chain of 7 orb_resize +
// orb_score + orb_count + orb_plan + orb_emit + orb_harris + orb_select + orb_output + bucket_key + bucket_table
// launches, a memset and two copies (19 launches, ~165 us per image next to a frame in flight; profiles/r04_a_*).
// The candidate lists are no longer in raster order (slots are reserved per workgroup): nothing downstream depends on
// the order — the cuts are rank statistics, the arg-max carries the position in its key. vo_orb_detect (the ordered
// keypoint list of the class surface) keeps the per-stage kernels of orb_detect.hip.
//
// Reference: extractor_orb_->detect + the arg-max-per-bin branch of FeatureExtractor::extractORBwithBinning_fast
// (core/visual_odometry/feature_extractor.cpp:241-277); cv::ORB restated as in oracle/oracle_orb.c.
//
// Plain C++ apart from the HIP keywords; the includer provides orb_wave_count (orb_device.hpp), ORB_DYN_LDS (the
// dynamic LDS array), __umulhi and the agent-scope load / store / fence spellings below — tests/emu/ runs both kernels
// on CPU threads against the oracle.
#pragma once
#include "vo_layout.hpp"
#include "orb_device.hpp"
#include "orb_plan.hpp"

#define ORB_TILE_NT 256

struct OrbTileLevel {
  int w, h;
  int lds_off, lds_stride;   // image region of the level in LDS
  int sc_off, sc_stride;     // score tile (owned pixels + one ring)
  int cand_base;             // first slot of the level's candidate list
  int tx_off, ty_off;        // level >= 1: where the region's slices of tabx / taby are staged in LDS
  const int *tabx, *taby;    // level >= 1: (source offset << 16) | weight of the next sample, per destination column / row
};
struct OrbTileArgs {
  const uint8_t *img;        // level 0 (any stride; the pyramid slot's padded plane or the caller's image)
  const uint8_t *img_end;    // one past the last byte that may be read (16-byte pieces stop there)
  int stride;
  int n_levels, nx, ny, fast_thr, cand_cap;
  int stash_off, stash_cap;
  const OrbSpan *gx, *gy;    // [level * nx + tile column], [level * ny + tile row]
  OrbTileLevel L[ORB_MAX_LEVELS];
  int *lvl_total;            // per level: candidates so far (zero before the launch)
  short *cx, *cy;
  uint8_t *cs;
  float *cr;
};

__global__ __launch_bounds__(ORB_TILE_NT) void orb_tile_kernel(OrbTileArgs a) {
  ORB_DYN_LDS(lds);
  __shared__ OrbSpan s_x[ORB_MAX_LEVELS], s_y[ORB_MAX_LEVELS];
  __shared__ unsigned s_mreg[ORB_MAX_LEVELS], s_msc[ORB_MAX_LEVELS], s_mown[ORB_MAX_LEVELS];  // magic reciprocals of the widths
  __shared__ int s_pre_sc[ORB_MAX_LEVELS + 1], s_pre_own[ORB_MAX_LEVELS + 1];  // running pixel counts: score rectangles, owned rectangles
  __shared__ int s_cnt[ORB_MAX_LEVELS], s_base[ORB_MAX_LEVELS];
  __shared__ int s_nstash;
  __shared__ int s_pre_tab[2 * ORB_MAX_LEVELS + 1];  // running entry counts of the table slices: (level, x), (level, y), ...

  const int tid = threadIdx.x, nl = a.n_levels;
  const int ti = (int)blockIdx.x % a.nx, tj = (int)blockIdx.x / a.nx;
  if (tid < nl) {
    const OrbSpan x = a.gx[tid * a.nx + ti], y = a.gy[tid * a.ny + tj];
    s_x[tid] = x;
    s_y[tid] = y;
    s_cnt[tid] = 0;
    s_mreg[tid] = vo_magic(x.reg1 - x.reg0);
    s_msc[tid] = vo_magic(x.own1 - x.own0 + 2);
    s_mown[tid] = vo_magic(x.own1 - x.own0);
  }
  if (tid == 0) s_nstash = 0;
  __syncthreads();
  if (tid == 0) {
    int psc = 0, pown = 0;
    for (int l = 0; l < nl; ++l) {
      s_pre_sc[l] = psc;
      s_pre_own[l] = pown;
      const int ow = s_x[l].own1 - s_x[l].own0, oh = s_y[l].own1 - s_y[l].own0;
      if (ow > 0 && oh > 0) {
        psc += (ow + 2) * (oh + 2);
        pown += ow * oh;
      }
    }
    s_pre_sc[nl] = psc;
    s_pre_own[nl] = pown;
    int pt = 0;
    for (int l = 1; l < nl; ++l) {
      s_pre_tab[2 * l - 2] = pt;
      pt += s_x[l].reg1 > s_x[l].reg0 ? s_x[l].reg1 - s_x[l].reg0 : 0;
      s_pre_tab[2 * l - 1] = pt;
      pt += s_y[l].reg1 > s_y[l].reg0 ? s_y[l].reg1 - s_y[l].reg0 : 0;
    }
    s_pre_tab[nl > 1 ? 2 * nl - 2 : 0] = pt;
  }
  // ---- level 0: the region from the image, in 16-byte pieces — every load of a thread in flight before its first LDS
  // store (with about one workgroup per compute unit a dependent load costs its whole latency: the first version, one
  // byte per load in a loop, spent 25 of its 60 us here and as much on the coefficient tables read inside the loops below)
  {
    const int x0 = s_x[0].reg0, y0 = s_y[0].reg0, rw = s_x[0].reg1 - x0, rh = s_y[0].reg1 - y0;
    if (rw > 0 && rh > 0) {
      uint8_t *D = lds + a.L[0].lds_off;
      const int ds = a.L[0].lds_stride;
      const int nch = (rw + 15) >> 4, total = nch * rh;
      const unsigned m = vo_magic(nch);
      const uint8_t *__restrict__ src = a.img + (size_t)y0 * a.stride + x0;
      for (int i0 = 0; i0 < total; i0 += 4 * ORB_TILE_NT) {
        vo_u128 v[4];
        int dst[4], tail[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // no branch between the loads: a piece that must not be read reads the image's first bytes
          const int i = i0 + tid + q * ORB_TILE_NT;
          int ry, c;
          vo_divmod(i < total ? i : 0, nch, m, ry, c);
          const uint8_t *p = src + (size_t)ry * a.stride + 16 * c;
          const bool safe = i < total && p + 16 <= a.img_end;
          dst[q] = safe ? ry * ds + 16 * c : -1;
          tail[q] = (i < total && !safe) ? i : -1;
          const vo_u128_unaligned u = *(const vo_u128_unaligned *)(safe ? p : a.img);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q].v[e] = u.v[e];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dst[q] >= 0) *(vo_u128 *)(D + dst[q]) = v[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // pieces that reach past the last readable byte (a caller's image, its last rows): byte by byte
          if (tail[q] < 0) continue;
          int ry, c;
          vo_divmod(tail[q], nch, m, ry, c);
          const uint8_t *p = src + (size_t)ry * a.stride + 16 * c;
          for (int e = 0; e < 16 && p + e < a.img_end; ++e) D[ry * ds + 16 * c + e] = p[e];
        }
      }
    }
  }
  __syncthreads();
  if (s_pre_own[nl] == 0) return;  // (a tile inside the border strip owns nothing on any level)
  // ---- the regions' slices of the resize coefficient tables -> LDS (one batch of loads, as above) ---------------------------
  {
    const int total = s_pre_tab[nl > 1 ? 2 * nl - 2 : 0];
    for (int i0 = 0; i0 < total; i0 += 4 * ORB_TILE_NT) {
      int val[4], *dstp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + tid + q * ORB_TILE_NT;
        dstp[q] = nullptr;
        val[q] = 0;
        if (i < total) {
          int sgm = 0;
          while (i >= s_pre_tab[sgm + 1]) ++sgm;
          const int l = (sgm >> 1) + 1, k = i - s_pre_tab[sgm];
          if (sgm & 1) {
            val[q] = a.L[l].taby[s_y[l].reg0 + k];
            dstp[q] = (int *)(lds + a.L[l].ty_off) + k;
          } else {
            val[q] = a.L[l].tabx[s_x[l].reg0 + k];
            dstp[q] = (int *)(lds + a.L[l].tx_off) + k;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (dstp[q]) *dstp[q] = val[q];
    }
  }
  __syncthreads();

  // ---- levels 1 .. n-1: cv::resize INTER_LINEAR_EXACT of the previous level's region, in LDS -------------------------------
  for (int l = 1; l < nl; ++l) {
    const int x0 = s_x[l].reg0, y0 = s_y[l].reg0, rw = s_x[l].reg1 - x0, rh = s_y[l].reg1 - y0;
    if (rw > 0 && rh > 0) {
      const uint8_t *S = lds + a.L[l - 1].lds_off;
      uint8_t *D = lds + a.L[l].lds_off;
      const int ss = a.L[l - 1].lds_stride, ds = a.L[l].lds_stride;
      const int sx0 = s_x[l - 1].reg0, sy0 = s_y[l - 1].reg0;
      const int *tabx = (const int *)(lds + a.L[l].tx_off), *taby = (const int *)(lds + a.L[l].ty_off);
      const unsigned m = s_mreg[l];
      for (int i = tid; i < rw * rh; i += ORB_TILE_NT) {
        int ry, rx;
        vo_divmod(i, rw, m, ry, rx);
        const int tx = tabx[rx], ty = taby[ry];
        const int a1 = tx & 0xFFFF, a0 = 256 - a1, b1 = ty & 0xFFFF, b0 = 256 - b1;
        const uint8_t *r0 = S + ((ty >> 16) - sy0) * ss + ((tx >> 16) - sx0), *r1 = r0 + ss;
        const unsigned h0 = (unsigned)a0 * r0[0] + (unsigned)a1 * r0[1];  // horizontal pass, 8.8
        const unsigned h1 = (unsigned)a0 * r1[0] + (unsigned)a1 * r1[1];
        const unsigned v = (unsigned)b0 * h0 + (unsigned)b1 * h1;          // vertical pass, 16.16
        const unsigned r = (v + 32768u) >> 16;
        D[ry * ds + rx] = (uint8_t)(r > 255u ? 255u : r);
      }
    }
    __syncthreads();
  }

  // ---- FAST score of the owned pixels and one ring around them, all levels in one pass ---------------------------------------
  {
    const int total = s_pre_sc[nl];
    int l = 0;
    for (int i = tid; i < total; i += ORB_TILE_NT) {
      while (i >= s_pre_sc[l + 1]) ++l;
      const int sw = s_x[l].own1 - s_x[l].own0 + 2;
      int sy, sx;
      vo_divmod(i - s_pre_sc[l], sw, s_msc[l], sy, sx);
      const int x = s_x[l].own0 - 1 + sx, y = s_y[l].own0 - 1 + sy;
      const int st = a.L[l].lds_stride;
      const uint8_t *p = lds + a.L[l].lds_off + (y - s_y[l].reg0) * st + (x - s_x[l].reg0);
      lds[a.L[l].sc_off + sy * a.L[l].sc_stride + sx] = (uint8_t)orb_fast_score(p, st, a.fast_thr);
    }
  }
  __syncthreads();

  // ---- strict 3x3 maxima of the score among the owned pixels -> the workgroup's stash -------------------------------------------
  unsigned *stash = (unsigned *)(lds + a.stash_off);
  {
    const int total = s_pre_own[nl];
    int l = 0;
    for (int i = tid; i < total; i += ORB_TILE_NT) {
      while (i >= s_pre_own[l + 1]) ++l;
      const int ow = s_x[l].own1 - s_x[l].own0;
      int oy, ox;
      vo_divmod(i - s_pre_own[l], ow, s_mown[l], oy, ox);
      const int scs = a.L[l].sc_stride;
      const uint8_t *p = lds + a.L[l].sc_off + (oy + 1) * scs + (ox + 1);
      const int c = p[0];
      if (c && c > p[-1] && c > p[1] && c > p[-scs - 1] && c > p[-scs] && c > p[-scs + 1] && c > p[scs - 1] && c > p[scs] && c > p[scs + 1]) {
        const int rank = atomicAdd(&s_cnt[l], 1);
        const int e = atomicAdd(&s_nstash, 1);
        if (e < a.stash_cap) {  // (cannot fail: the stash holds one entry per 2x2 cell of every owned rectangle)
          stash[2 * e] = (unsigned)(s_x[l].own0 + ox) | ((unsigned)(s_y[l].own0 + oy) << 16);
          stash[2 * e + 1] = (unsigned)l | ((unsigned)c << 8) | ((unsigned)rank << 16);
        }
      }
    }
  }
  __syncthreads();
  if (tid < nl) s_base[tid] = s_cnt[tid] ? atomicAdd(&a.lvl_total[tid], s_cnt[tid]) : 0;
  __syncthreads();

  // ---- Harris response of every stashed candidate, then out to the level's list ------------------------------------------------
  {
    const int n = s_nstash < a.stash_cap ? s_nstash : a.stash_cap;
    for (int e = tid; e < n; e += ORB_TILE_NT) {
      const unsigned w0 = stash[2 * e], w1 = stash[2 * e + 1];
      const int x = (int)(w0 & 0xFFFFu), y = (int)(w0 >> 16), l = (int)(w1 & 0xFFu), c = (int)((w1 >> 8) & 0xFFu), rank = (int)(w1 >> 16);
      const int idx = s_base[l] + rank;
      if (idx >= a.cand_cap) continue;  // the level's list is full: orb_finish_kernel reports it (lvl_total > cand_cap)
      const float r = orb_harris(lds + a.L[l].lds_off, a.L[l].lds_stride, x - s_x[l].reg0, y - s_y[l].reg0);
      const int o = a.L[l].cand_base + idx;
      a.cx[o] = (short)x;
      a.cy[o] = (short)y;
      a.cs[o] = (uint8_t)c;
      a.cr[o] = r;
    }
  }
}

// ---- second launch: cuts, per-bin arg-max, table -----------------------------------------------------------------------------------
struct OrbFinishArgs {
  int n_levels, cand_cap, max_out;
  int cand_base[ORB_MAX_LEVELS], quota[ORB_MAX_LEVELS];
  float scale[ORB_MAX_LEVELS];
  int *lvl_total;            // in: candidates per level; zeroed on exit
  const short *cx, *cy;
  const uint8_t *cs;
  const float *cr;
  int *surv;                 // [n_levels] survivors per level (scratch)
  int *done;                 // workgroups finished (zero before the launch, zeroed on exit)
  unsigned long long *key;   // [n_bins] (zero before the launch, zeroed on exit)
  int n_bins_u, n_bins_v;
  float inv_u, inv_v;
  float *tab_xy;             // [n_bins][2]
  uint8_t *tab_has;          // [n_bins]
  int *host_flags;           // pinned: [0] capacity flags (1 candidate lists, 2 keypoint count), [1] keypoints detected
  int *dev_flags;            // the same two words on the device (test hooks)
};

// one candidate that survived both cuts: its bin's key. xs / ys as orb_output_kernel writes keypoint coordinates
// (pt *= scale for level != 0), bin and response test as bucket_key_kernel (misc_kernels.hip)
__device__ __forceinline__ void orb_finish_vote(const OrbFinishArgs &a, int l, int x, int y, float r) {
  const float xs = l ? (float)x * a.scale[l] : (float)x, ys = l ? (float)y * a.scale[l] : (float)y;
  const unsigned u = (unsigned)(int)floorf(xs * a.inv_u), v = (unsigned)(int)floorf(ys * a.inv_v);
  if (u >= (unsigned)a.n_bins_u || v >= (unsigned)a.n_bins_v) return;
  if (!(-1.0f < r)) return;  // never beats the initial max_score of -1 (NaN included)
  r = r + 0.0f;              // -0 -> +0: the reference's "<" does not tell them apart
  const unsigned ord = orb_ord(r);
  // low word: larger for EARLIER keypoints in (level, row, column) order — the order of cv::ORB's keypoint vector as
  // restated (level, then raster): among equal responses the first one wins, as "max_score < response" keeps it
  const unsigned pos = ((unsigned)l << 28) | ((unsigned)y << 14) | (unsigned)x;
  atomicMax(&a.key[v * (unsigned)a.n_bins_u + u], ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - pos));
}

template <int NQ>
__device__ __forceinline__ int orb_finish_level(const OrbFinishArgs &a, int l, int n, OrbSelShared *S) {
  unsigned key[NQ];
  float resp[NQ];
  int xy[NQ];  // the candidates' coordinates: loaded with their scores and responses, in one batch in front of the selection
  int cut, surv;
  unsigned rcut;
  const int base = a.cand_base[l];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = (int)threadIdx.x + q * ORB_ST;
    xy[q] = i < n ? ((int)(unsigned short)a.cx[base + i] | ((int)(unsigned short)a.cy[base + i] << 16)) : 0;
  }
  orb_select_regs<NQ>(a.cs + base, a.cr + base, n, a.quota[l], S, key, resp, &cut, &rcut, &surv);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (key[q] != 0u) orb_finish_vote(a, l, xy[q] & 0xFFFF, (int)((unsigned)xy[q] >> 16), resp[q]);
  return surv;
}

__global__ __launch_bounds__(ORB_ST) void orb_finish_kernel(OrbFinishArgs a) {
  __shared__ OrbSelShared s_sel;
  __shared__ int s_hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_rank, s_cut, s_kept, s_surv, s_last;
  const int l = blockIdx.x, tid = threadIdx.x;
  const int total = a.lvl_total[l];
  const int n = total > a.cand_cap ? 0 : total;
  int surv = 0;
  if (n <= 2 * ORB_ST) {
    surv = orb_finish_level<2>(a, l, n, &s_sel);
  } else if (n <= 4 * ORB_ST) {
    surv = orb_finish_level<4>(a, l, n, &s_sel);
  } else if (n <= 8 * ORB_ST) {
    surv = orb_finish_level<8>(a, l, n, &s_sel);
  } else if (n <= ORB_RC * ORB_ST) {
    surv = orb_finish_level<ORB_RC>(a, l, n, &s_sel);
  } else {
    // a level of more than 16 384 candidates (4K images): LDS histogram for the score cut, 4-pass radix select on the
    // ordered response (the general path's orb_select_kernel, unchanged), then the votes from memory
    const uint8_t *cs = a.cs + a.cand_base[l];
    const float *cr = a.cr + a.cand_base[l];
    const int quota = a.quota[l];
    if (tid < 256) s_hist[tid] = 0;
    if (tid == 0) s_surv = 0;
    __syncthreads();
    for (int i = tid; i < n; i += ORB_ST) atomicAdd(&s_hist[cs[i]], 1);
    __syncthreads();
    if (tid == 0) {
      int cut = 0, kept = n;
      const int keep = 2 * quota;
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
    }
    __syncthreads();
    const int cut = s_cut, kept = s_kept;
    unsigned rcut = 0u;
    if (kept > quota) {
      if (quota == 0) {
        rcut = 0xFFFFFFFFu;
      } else {
        if (tid == 0) {
          s_prefix = 0;
          s_rank = quota;
        }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
          if (tid < 256) s_hist[tid] = 0;
          __syncthreads();
          const unsigned prefix = s_prefix;
          const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
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
    for (int i = tid; i < n; i += ORB_ST) {
      const float r = cr[i];
      if (cs[i] >= cut && (rcut == 0u || orb_ord(r) >= rcut)) {
        ++mine;
        orb_finish_vote(a, l, a.cx[a.cand_base[l] + i], a.cy[a.cand_base[l] + i], r);
      }
    }
    if (mine) atomicAdd(&s_surv, mine);
    __syncthreads();
    surv = s_surv;
  }
  // ---- the last workgroup to get here turns the keys into the table -----------------------------------------------------------
  ORB_FENCE_RELEASE();  // every thread: its votes have been performed ...
  __syncthreads();      // ... before thread 0 takes the workgroup's ticket
  if (tid == 0) {
    ORB_ST_AGENT(&a.surv[l], surv);
    ORB_FENCE_RELEASE();
    s_last = (ORB_ATOMIC_INC_AGENT(a.done) == a.n_levels - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  ORB_FENCE_ACQUIRE();
  // (every load below is issued before the first one is used: the other workgroups' results come from memory)
  const int nb = a.n_bins_u * a.n_bins_v;
  unsigned long long kk[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) kk[q] = (tid + q * ORB_ST) < nb ? ORB_LD_AGENT(&a.key[tid + q * ORB_ST]) : 0ull;
  if (tid < a.n_levels) {
    s_hist[tid] = ORB_LD_AGENT(&a.surv[tid]);
    s_hist[ORB_MAX_LEVELS + tid] = ORB_LD_AGENT(&a.lvl_total[tid]);
  }
  __syncthreads();
  if (tid == 0) {
    int tot = 0, flags = 0;
    for (int q = 0; q < a.n_levels; ++q) {
      tot += s_hist[q];
      if (s_hist[ORB_MAX_LEVELS + q] > a.cand_cap) flags |= 1;  // more corners on a level than its list holds
    }
    if (tot > a.max_out) flags |= 2;  // (the general path's output buffer: kept so that both paths report alike)
    const int n_out = tot < a.max_out ? tot : a.max_out;
    a.host_flags[0] = flags;
    a.host_flags[1] = n_out;
    a.dev_flags[0] = flags;
    a.dev_flags[1] = n_out;
    ORB_ST_AGENT(a.done, 0);
  }
  if (tid < a.n_levels) ORB_ST_AGENT(&a.lvl_total[tid], 0);
  for (int j0 = 0; j0 < nb; j0 += 2 * ORB_ST) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = j0 + tid + q * ORB_ST;
      if (j >= nb) continue;
      const unsigned long long k = kk[q];
      float x = 0.f, y = 0.f;
      if (k != 0ull) {
        const unsigned pos = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        const int lv = (int)(pos >> 28), py = (int)((pos >> 14) & 0x3FFFu), px = (int)(pos & 0x3FFFu);
        x = lv ? (float)px * a.scale[lv] : (float)px;
        y = lv ? (float)py * a.scale[lv] : (float)py;
        ORB_ST_AGENT(&a.key[j], 0ull);
      }
      a.tab_has[j] = k != 0ull ? 1 : 0;
      a.tab_xy[2 * j] = x;
      a.tab_xy[2 * j + 1] = y;
    }
    if (j0 + 2 * ORB_ST < nb) {  // (more than 2048 bins: the next two keys of this thread)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = j0 + 2 * ORB_ST + tid + q * ORB_ST;
        kk[q] = j < nb ? ORB_LD_AGENT(&a.key[j]) : 0ull;
      }
    }
  }
}
