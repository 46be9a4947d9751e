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
// Plain C++ apart from the HIP keywords; the includer provides orb_wave_count (orb_device.hpp), orb_wave_rank / orb_wave_first (below), ORB_DYN_LDS (the
// dynamic LDS array), ORB_SET_PRIO, __umulhi / __mul24 / __umul24 and the agent-scope load / store / fence spellings below — tests/emu/ runs both kernels
// on CPU threads against the oracle.
#pragma once
#include "vo_layout.hpp"
#include "orb_device.hpp"
#include "orb_plan.hpp"

#ifndef TILE_STAMP_AT
#define TILE_STAMP_AT(k)  // (tools/tileprobe.hip defines it: phase time stamps of one workgroup, measurement builds only)
#endif

#ifndef ORB_TILE_NT
#define ORB_TILE_NT 512  // two wavefronts per SIMD: every phase is a chain of dependent LDS reads (with one: 52 us per workgroup,
                         // tools/tileprobe.hip); four would not fit next to the frame kernel's wavefronts on a SIMD
#endif

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
  int *hist;                 // [copy][level][256] candidates per FAST score (zero before the launch; orb_finish_kernel's first cut)
  int hist_copies;           // workgroup b adds to copy b % hist_copies: at 3840 x 2160 600 000 memory atomics on the few hundred
                             // words of ONE copy took 200 us of the launch (tools/tileprobe.hip), spread over 64 copies they are free
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
  __shared__ int s_stash_lvl[ORB_MAX_LEVELS];  // first stash entry of a level's part
  __shared__ int s_pre_tab[2 * ORB_MAX_LEVELS + 1];  // running entry counts of the table slices: (level, x), (level, y), ...

  // The per-level table of the kernel arguments, copied to LDS by ONE batch of loads: indexed by a level that is not a
  // compile-time constant, every access to it in the argument segment was a memory round trip of its own (1-1.5 us each —
  // per level in the resize and non-max loops, per pixel where the level differs between lanes: 30 of the first version's 40 us)
  __shared__ OrbTileLevel s_L[ORB_MAX_LEVELS];
  const int tid = threadIdx.x, nl = a.n_levels;
  const int ti = (int)blockIdx.x % a.nx, tj = (int)blockIdx.x / a.nx;
  ORB_SET_PRIO();  // a short kernel next to the frame kernel's long-lived wavefronts: its instructions go first
  TILE_STAMP_AT(0);
  if (tid < nl * (int)(sizeof(OrbTileLevel) / sizeof(int))) ((int *)s_L)[tid] = ((const int *)a.L)[tid];
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
  TILE_STAMP_AT(1);
  if (tid == 0) {
    int psc = 0, pown = 0, pst = 0;
    for (int l = 0; l < nl; ++l) {
      s_pre_sc[l] = psc;
      s_pre_own[l] = pown;
      s_stash_lvl[l] = pst;
      const int ow = s_x[l].own1 - s_x[l].own0, oh = s_y[l].own1 - s_y[l].own0;
      if (ow > 0 && oh > 0) {
        psc += (ow + 2) * (oh + 2);
        pown += ow * oh;
        pst += ((ow + 1) >> 1) * ((oh + 1) >> 1);  // strict 3x3 maxima in an ow x oh rectangle: at most one per 2x2 cell
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
      uint8_t *D = lds + s_L[0].lds_off;
      const int ds = s_L[0].lds_stride;
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
  TILE_STAMP_AT(2);
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
            val[q] = s_L[l].taby[s_y[l].reg0 + k];
            dstp[q] = (int *)(lds + s_L[l].ty_off) + k;
          } else {
            val[q] = s_L[l].tabx[s_x[l].reg0 + k];
            dstp[q] = (int *)(lds + s_L[l].tx_off) + k;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (dstp[q]) *dstp[q] = val[q];
    }
  }
  __syncthreads();
  TILE_STAMP_AT(3);

  // ---- levels 1 .. n-1: cv::resize INTER_LINEAR_EXACT of the previous level's region, in LDS -------------------------------
  for (int l = 1; l < nl; ++l) {
    const int x0 = s_x[l].reg0, y0 = s_y[l].reg0, rw = s_x[l].reg1 - x0, rh = s_y[l].reg1 - y0;
    if (rw > 0 && rh > 0) {
      const uint8_t *S = lds + s_L[l - 1].lds_off;
      uint8_t *D = lds + s_L[l].lds_off;
      const int ss = s_L[l - 1].lds_stride, ds = s_L[l].lds_stride;
      const int sx0 = s_x[l - 1].reg0, sy0 = s_y[l - 1].reg0;
      const int *tabx = (const int *)(lds + s_L[l].tx_off), *taby = (const int *)(lds + s_L[l].ty_off);
      // lanes along a row, wavefronts over the rows, four rows per lane and pass: a lane's column terms (source offset,
      // horizontal weights) are loop constants, a row's terms are the same in every lane, no division, products through the
      // 24-bit multiplier. (One flattened output at a time with a division each: 300 issue cycles per wavefront and output —
      // the phase was VALU-bound whatever the number of wavefronts, 15 of the kernel's 40 us.)
      const int lane = tid & 63, wave = tid >> 6;
      constexpr int NW = ORB_TILE_NT / 64;
      for (int c0 = 0; c0 < rw; c0 += 64) {
        const int c = c0 + lane;
        const bool col = c < rw;
        const int tx = tabx[col ? c : 0];
        const int xo = (tx >> 16) - sx0;
        const unsigned a1 = (unsigned)(tx & 0xFFFF), a0 = 256u - a1;
        for (int r0 = wave; r0 < rh; r0 += 4 * NW) {
          int ty[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) ty[q] = taby[r0 + q * NW < rh ? r0 + q * NW : 0];
          unsigned p00[4], p01[4], p10[4], p11[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (the two source pixels of a row by aligned dword reads: vo_bytes4)
            const int o = __mul24((ty[q] >> 16) - sy0, ss) + xo;
            const uint32_t v0 = vo_bytes4(S, o), v1 = vo_bytes4(S, o + ss);
            p00[q] = v0 & 255u;
            p01[q] = (v0 >> 8) & 255u;
            p10[q] = v1 & 255u;
            p11[q] = (v1 >> 8) & 255u;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ry = r0 + q * NW;
            const unsigned b1 = (unsigned)(ty[q] & 0xFFFF), b0 = 256u - b1;
            const unsigned h0 = __umul24(a0, p00[q]) + __umul24(a1, p01[q]);  // horizontal pass, 8.8
            const unsigned h1 = __umul24(a0, p10[q]) + __umul24(a1, p11[q]);
            const unsigned v = __umul24(b0, h0) + __umul24(b1, h1);            // vertical pass, 16.16 (b <= 256, h <= 65280: 24-bit products)
            const unsigned r = (v + 32768u) >> 16;
            if (col && ry < rh) D[__mul24(ry, ds) + c] = (uint8_t)(r > 255u ? 255u : r);
          }
        }
      }
    }
    __syncthreads();
    if (l == 1) TILE_STAMP_AT(4);
  }
  TILE_STAMP_AT(5);

  // ---- FAST score of the owned pixels and one ring around them, all levels together, in two passes: (a) the compass-point
  // test of fast.cpp on every pixel — most fail it and get score 0; the others are listed; (b) the 16-point test and
  // cornerScore on the listed pixels, one per lane with no idle lanes (in one pass a wavefront ran the ~150 instructions of
  // (b) in nearly every step for the few lanes that needed it: 10.5 of the kernel's 40 us)
  unsigned *plist = (unsigned *)(lds + a.stash_off);  // (the stash's place: it is not in use yet)
  {
    const int total = s_pre_sc[nl], thr = a.fast_thr;
    int l = 0;
    for (int i0 = 0; i0 < total; i0 += 4 * ORB_TILE_NT) {
      bool pass[4];
      unsigned ent[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + tid + q * ORB_TILE_NT;
        pass[q] = false;
        ent[q] = 0;
        if (i < total) {
          while (i >= s_pre_sc[l + 1]) ++l;
          const int sw = s_x[l].own1 - s_x[l].own0 + 2;
          int sy, sx;
          vo_divmod(i - s_pre_sc[l], sw, s_msc[l], sy, sx);
          const int st = s_L[l].lds_stride;
          const uint8_t *p = lds + s_L[l].lds_off + __mul24(s_y[l].own0 - 1 + sy - s_y[l].reg0, st) + (s_x[l].own0 - 1 + sx - s_x[l].reg0);
          const int v = p[0], c0 = v - p[3 * st], c4 = v - p[3], c8 = v - p[-3 * st], c12 = v - p[-3];
          const int nd = (c0 > thr) + (c4 > thr) + (c8 > thr) + (c12 > thr), nb = (c0 < -thr) + (c4 < -thr) + (c8 < -thr) + (c12 < -thr);
          pass[q] = nd >= 2 || nb >= 2;
          ent[q] = (unsigned)l | ((unsigned)sy << 8) | ((unsigned)sx << 20);
          if (!pass[q]) lds[s_L[l].sc_off + __mul24(sy, s_L[l].sc_stride) + sx] = 0;
        }
      }
      int r[4], tot = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int nq;
        r[q] = tot + orb_wave_rank(pass[q], &nq);
        tot += nq;
      }
      int base = 0;
      if ((tid & 63) == 0 && tot) base = atomicAdd(&s_nstash, tot);
      base = orb_wave_first(base, (tid & 63) == 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (pass[q]) plist[base + r[q]] = ent[q];
    }
  }
  __syncthreads();
  TILE_STAMP_AT(10);
  {
    const int n = s_nstash, thr = a.fast_thr;
    for (int e = tid; e < n; e += ORB_TILE_NT) {
      const unsigned w = plist[e];
      const int l = (int)(w & 0xFFu), sy = (int)((w >> 8) & 0xFFFu), sx = (int)(w >> 20);
      const int st = s_L[l].lds_stride;
      const uint8_t *p = lds + s_L[l].lds_off + __mul24(s_y[l].own0 - 1 + sy - s_y[l].reg0, st) + (s_x[l].own0 - 1 + sx - s_x[l].reg0);
      lds[s_L[l].sc_off + __mul24(sy, s_L[l].sc_stride) + sx] = (uint8_t)orb_fast_score(p, st, thr);
    }
  }
  __syncthreads();
  TILE_STAMP_AT(6);

  // ---- strict 3x3 maxima of the score among the owned pixels -> the workgroup's stash -------------------------------------------
  // all levels in one flattened pass (a pass per level left most lanes idle on the small levels: 8 x 2048 slots for 4 800
  // pixels); a maximum takes its rank within its level by an LDS atomic. The stash is partitioned by level (a level's part
  // holds one entry per 2x2 cell of its owned rectangle: cannot overflow), so the rank is the slot.
  unsigned *stash = (unsigned *)(lds + a.stash_off);
  {
    const int total = s_pre_own[nl];
    int l = 0;
    for (int i0 = 0; i0 < total; i0 += 4 * ORB_TILE_NT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + tid + q * ORB_TILE_NT;
        if (i >= total) continue;
        while (i >= s_pre_own[l + 1]) ++l;
        const int ow = s_x[l].own1 - s_x[l].own0, scs = s_L[l].sc_stride;
        int oy, ox;
        vo_divmod(i - s_pre_own[l], ow, s_mown[l], oy, ox);
        // the 3x3 neighbourhood by aligned dword reads (vo_bytes4): bytes 0..2 of each word are the row's three scores
        const uint8_t *sc = lds + s_L[l].sc_off;
        const int o = __mul24(oy + 1, scs) + ox;  // (the centre's left neighbour)
        const uint32_t ra = vo_bytes4(sc, o - scs), rb = vo_bytes4(sc, o), rc = vo_bytes4(sc, o + scs);
        const int c = (int)((rb >> 8) & 255u);
        const int m0 = (int)(ra & 255u), m1 = (int)((ra >> 8) & 255u), m2 = (int)((ra >> 16) & 255u), m3 = (int)(rb & 255u),
                  m4 = (int)((rb >> 16) & 255u), m5 = (int)(rc & 255u), m6 = (int)((rc >> 8) & 255u), m7 = (int)((rc >> 16) & 255u);
        if (c && c > m0 && c > m1 && c > m2 && c > m3 && c > m4 && c > m5 && c > m6 && c > m7) {
          const int e = s_stash_lvl[l] + atomicAdd(&s_cnt[l], 1);
          stash[2 * e] = (unsigned)(s_x[l].own0 + ox) | ((unsigned)(s_y[l].own0 + oy) << 16);
          stash[2 * e + 1] = (unsigned)l | ((unsigned)c << 8);
        }
      }
    }
  }
  __syncthreads();
  TILE_STAMP_AT(7);
  if (tid < nl) s_base[tid] = s_cnt[tid] ? atomicAdd(&a.lvl_total[tid], s_cnt[tid]) : 0;
  if (tid == 0) {  // running candidate counts: the flat index of the loop below -> (level, rank)
    int run = 0;
    for (int l = 0; l < nl; ++l) {
      s_pre_sc[l] = run;
      run += s_cnt[l];
    }
    s_pre_sc[nl] = run;
  }
  __syncthreads();
  TILE_STAMP_AT(8);

  // ---- Harris response of every stashed candidate, then out to the level's list ------------------------------------------------
  {
    const int n = s_pre_sc[nl];
    int l = 0;
    for (int i = tid; i < n; i += ORB_TILE_NT) {
      while (i >= s_pre_sc[l + 1]) ++l;
      const int rank = i - s_pre_sc[l], e = s_stash_lvl[l] + rank;
      const unsigned w0 = stash[2 * e], w1 = stash[2 * e + 1];
      const int x = (int)(w0 & 0xFFFFu), y = (int)(w0 >> 16), c = (int)((w1 >> 8) & 0xFFu);
      const int idx = s_base[l] + rank;
      if (idx >= a.cand_cap) continue;  // the level's list is full: orb_finish_kernel reports it (lvl_total > cand_cap)
      const float r = orb_harris_lds(lds + s_L[l].lds_off, s_L[l].lds_stride, x - s_x[l].reg0, y - s_y[l].reg0);
      const int o = s_L[l].cand_base + idx;
      a.cx[o] = (short)x;
      a.cy[o] = (short)y;
      a.cs[o] = (uint8_t)c;
      a.cr[o] = r;
#ifndef TILE_NO_HIST  // (measurement builds of tools/tileprobe.hip)
      // (no use of the returned value: a fire-and-forget memory atomic)
      ORB_ATOMIC_ADD_AGENT(&a.hist[(((int)blockIdx.x % a.hist_copies) * nl + l) * 256 + c], 1);
#endif
    }
  }
  TILE_STAMP_AT(9);
}

// ---- second launch: cuts, per-bin arg-max, table -----------------------------------------------------------------------------------
// `parts` workgroups per level. Every one of them knows the level's FIRST cut at once — the tile kernel left a histogram of the
// scores it emitted — and packs the indices of its slice's survivors into the level's list (counts by lane-mask population
// counts, one memory atomic per workgroup and round to reserve the places). The LAST of a level's workgroups to get there
// (a ticket; nobody waits) loads the survivors' responses into registers, finds the second cut by orb_kth_largest and votes;
// the last LEVEL to finish turns the keys into the table. A 3840 x 2160 image has > 100 000 candidates on level 0: one
// workgroup walking them five times (round 5's first version: histogram + four radix passes) took 650 us.
struct OrbFinishArgs {
  int n_levels, cand_cap, max_out;
  int parts;                 // workgroups per level
  int cidx_cap;              // survivors of the first cut a level's list holds (= what one workgroup's registers hold)
  int cand_base[ORB_MAX_LEVELS], quota[ORB_MAX_LEVELS];
  float scale[ORB_MAX_LEVELS];
  int *lvl_total;            // in: candidates per level; zeroed on exit
  int *hist;                 // in: [copy][level][256] candidates per score; zeroed on exit
  int hist_copies;
  int *cidx;                 // [level][cidx_cap] scratch: indices of the first cut's survivors
  int *lvl_cnt, *lvl_done;   // [level] survivors listed so far / workgroups of the level that have finished (zero; zeroed on exit)
  const short *cx, *cy;
  const uint8_t *cs;
  const float *cr;
  int *surv;                 // [n_levels] survivors per level (scratch)
  int *done;                 // levels finished (zero before the launch, zeroed on exit)
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

// The level's last workgroup: m survivors of the first cut are listed in ci[0, m) (any order). retainBest(quota) on their
// Harris responses — everything >= the quota-th largest stays — and the votes. Returns the number that stayed.
// One register array live across the barriers (the ordered responses); indices and coordinates are loaded again for the votes.
template <int NQ>
__device__ __forceinline__ int orb_finish_survivors(const OrbFinishArgs &a, int l, int m, const int *ci, OrbSelShared *S) {
  const int tid = threadIdx.x, base = a.cand_base[l], quota = a.quota[l];
  unsigned key[NQ];
  {
    constexpr int NB = NQ < 8 ? NQ : 8;  // (indices of eight survivors in flight, then their responses)
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += NB) {
      int idx[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int j = tid + (q0 + q) * ORB_ST;
        idx[q] = j < m ? ORB_LD_AGENT(&ci[j]) : -1;
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) key[q0 + q] = idx[q] >= 0 ? orb_ord(a.cr[base + idx[q]]) : 0u;
    }
  }
  if (tid < 3) S->cnt[tid] = 0;
  __syncthreads();
  int phase = 0;
  unsigned rcut = 0u;  // 0 = retainBest leaves the set alone
  int surv = orb_count_ge<NQ>(key, 1u, S, phase);  // (= m unless a response is NaN: key 0)
  if (surv > quota) {
    if (quota == 0)
      rcut = 0xFFFFFFFFu;
    else
      rcut = orb_kth_largest<NQ>(key, quota, surv, 32, S, phase);
    surv = orb_count_ge<NQ>(key, rcut, S, phase);
  }
  constexpr int NV = NQ < 4 ? NQ : 4;  // (four survivors' coordinates in flight at a time: registers)
#pragma unroll
  for (int q0 = 0; q0 < NQ; q0 += NV) {
    int x[NV], y[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const bool in = key[q0 + q] != 0u && key[q0 + q] >= rcut;
      const int idx = in ? ORB_LD_AGENT(&ci[tid + (q0 + q) * ORB_ST]) : 0;
      x[q] = in ? a.cx[base + idx] : 0;
      y[q] = in ? a.cy[base + idx] : 0;
    }
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (key[q0 + q] != 0u && key[q0 + q] >= rcut) orb_finish_vote(a, l, x[q], y[q], orb_unord(key[q0 + q]));
  }
  return surv;
}

__global__ __launch_bounds__(ORB_ST) void orb_finish_kernel(OrbFinishArgs a) {
  __shared__ OrbSelShared s_sel;
  __shared__ int s_hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_rank, s_cut, s_kept, s_surv, s_last, s_m;
  __shared__ int s_wtot[ORB_ST / 64], s_wbase[ORB_ST / 64];
  __shared__ float s_scale[ORB_MAX_LEVELS];
  const int l = (int)blockIdx.x / a.parts, part = (int)blockIdx.x - l * a.parts, tid = threadIdx.x;
  ORB_SET_PRIO();
  TILE_STAMP_AT(0);
  if (tid < a.n_levels) s_scale[tid] = a.scale[tid];  // (the table is decoded with a per-bin level: not from the argument segment)
  const int total = a.lvl_total[l];
  {  // the level's histogram: the sum of the tile kernel's copies (two threads per score, every load in flight together)
    int part_sum = 0;
    for (int k = tid >> 8; k < a.hist_copies; k += ORB_ST / 256) part_sum += a.hist[(k * a.n_levels + l) * 256 + (tid & 255)];
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    if (part_sum) atomicAdd(&s_hist[tid & 255], part_sum);
  }
  TILE_STAMP_AT(1);
  const int n = total > a.cand_cap ? 0 : total;
  const int quota = a.quota[l];
  const uint8_t *cs = a.cs + a.cand_base[l];
  const float *cr = a.cr + a.cand_base[l];
  int *ci = a.cidx + l * a.cidx_cap;
  __syncthreads();
  // (1) retainBest(2 n_l) on the FAST score: the value of rank 2 n_l from the histogram; kept = what is at or above it
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
  const bool listed = kept <= a.cidx_cap;  // (the same for every workgroup of the level: same inputs)
  if (listed) {
    // this workgroup's slice of the level's scores, four to a word (cand_base and cand_cap are multiples of 16); a round
    // = four words per thread
    const uint32_t *cs32 = (const uint32_t *)cs;
    const int words = (n + 3) >> 2;
    const int w0 = (int)(((long long)words * part) / a.parts), w1 = (int)(((long long)words * (part + 1)) / a.parts);
    const int wave = tid >> 6;
    for (int wb = w0; wb < w1; wb += 4 * ORB_ST) {
      uint32_t v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int w = wb + tid + q * ORB_ST;
        v[q] = w < w1 ? cs32[w] : 0u;
      }
      int wtot = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int w = wb + tid + q * ORB_ST;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool in = w < w1 && 4 * w + b < n && (int)((v[q] >> (8 * b)) & 255u) >= cut;
          wtot += orb_wave_count(in);
        }
      }
      if ((tid & 63) == 0) s_wtot[wave] = wtot;
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int k = 0; k < ORB_ST / 64; ++k) tot += s_wtot[k];
        int g = tot ? ORB_ATOMIC_ADD_AGENT(&a.lvl_cnt[l], tot) : 0;
        for (int k = 0; k < ORB_ST / 64; ++k) {
          s_wbase[k] = g;
          g += s_wtot[k];
        }
      }
      __syncthreads();
      int run = s_wbase[wave];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int w = wb + tid + q * ORB_ST;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool in = w < w1 && 4 * w + b < n && (int)((v[q] >> (8 * b)) & 255u) >= cut;
          int cnt;
          const int r = orb_wave_rank(in, &cnt);
          if (in && run + r < a.cidx_cap) ORB_ST_AGENT(&ci[run + r], 4 * w + b);
          run += cnt;
        }
      }
      // (no barrier here: s_wtot / s_wbase are written again behind the next round's first barrier, which every thread
      // reaches after it has read them)
    }
  }
  // ---- the level's ticket: the last of its workgroups goes on ---------------------------------------------------------------------
  ORB_FENCE_RELEASE();  // every thread: its list entries have been performed ...
  __syncthreads();      // ... before thread 0 takes the workgroup's ticket
  if (tid == 0) s_last = (ORB_ATOMIC_INC_AGENT(&a.lvl_done[l]) == a.parts - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  ORB_FENCE_ACQUIRE();
  int surv = 0;
  if (listed) {
    if (tid == 0) s_m = ORB_LD_AGENT(&a.lvl_cnt[l]);
    __syncthreads();
    const int m = s_m < a.cidx_cap ? s_m : a.cidx_cap;  // (= kept)
    if (m <= 4 * ORB_ST) {
      surv = orb_finish_survivors<4>(a, l, m, ci, &s_sel);
    } else if (m <= 8 * ORB_ST) {
      surv = orb_finish_survivors<8>(a, l, m, ci, &s_sel);
    } else if (m <= 16 * ORB_ST) {
      surv = orb_finish_survivors<16>(a, l, m, ci, &s_sel);
    } else {
      surv = orb_finish_survivors<ORB_RC>(a, l, m, ci, &s_sel);
    }
  } else {
    // more survivors of the first cut than a workgroup's registers hold (a score tie of tens of thousands): 4-pass radix
    // select on the ordered response straight from the level's arrays (the general path's orb_select_kernel), then the votes
    unsigned rcut = 0u;
    if (tid == 0) s_surv = 0;
    __syncthreads();
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
  // the level's words go back to zero for the next image (every workgroup of the level has read them: it holds the last ticket)
  for (int k = tid >> 8; k < a.hist_copies; k += ORB_ST / 256) ORB_ST_AGENT(&a.hist[(k * a.n_levels + l) * 256 + (tid & 255)], 0);
  if (tid == 0) {
    ORB_ST_AGENT(&a.lvl_cnt[l], 0);
    ORB_ST_AGENT(&a.lvl_done[l], 0);
  }
  // ---- the last LEVEL to get here turns the keys into the table ---------------------------------------------------------------
  TILE_STAMP_AT(2);
  ORB_FENCE_RELEASE();  // every thread: its votes have been performed ...
  __syncthreads();      // ... before thread 0 takes the level's ticket
  if (tid == 0) {
    ORB_ST_AGENT(&a.surv[l], surv);
    ORB_FENCE_RELEASE();
    s_last = (ORB_ATOMIC_INC_AGENT(a.done) == a.n_levels - 1) ? 1 : 0;
  }
  __syncthreads();
  TILE_STAMP_AT(3);
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
        x = lv ? (float)px * s_scale[lv] : (float)px;
        y = lv ? (float)py * s_scale[lv] : (float)py;
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
