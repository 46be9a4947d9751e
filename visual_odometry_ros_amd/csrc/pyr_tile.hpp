// pyr_tile.hpp — the whole PyrLK pyramid of an image (or a stereo pair) in ONE launch: pyr_build_kernel.
//
// What it replaces: cv::buildOpticalFlowPyramid inside every cv::calcOpticalFlowPyrLK call of the reference
// (core/visual_odometry/feature_tracker.cpp:29,60,69,108,117,186 -> OpenCV 4 video/lkpyramid.cpp, imgproc/pyramids.cpp:
// copyMakeBorder(BORDER_REFLECT_101) per level, pyrDown = 5-tap [1 4 6 4 1]/16 both ways, (sum + 128) >> 8) — and rounds
// 1-4 of this repository, which built a slot's pyramid with one launch per level (pad_level0_kernel + 4 x pyr_down_kernel:
// 5 dependent launches, 38 us per pair next to a frame in flight; profiles/r04_a_*).
//
// Level l+1 depends on level l only through a 5x5 neighbourhood, so a workgroup that owns a T x T tile of the TOP level
// can produce its piece of every level by itself: it stages the base-level region its tile depends on in LDS
// ((64 + 3 (2^nl - 1))^2 bytes at most: 109 x 109 for four levels), reduces it level by level in LDS — recomputing the
// halo its neighbours also compute instead of waiting for them — and writes, per level, the pixels it OWNS (a 64 >> k
// tile of level k, dword stores) plus their mirror images in the level's REFLECT_101 border (byte stores). No workgroup
// waits for another one, no level goes through HBM between its producer and its consumer, one launch.
//
// Bits: the same integer sums in any order — identical to pad_level0_kernel / pyr_down_kernel and oracle_klt.c.
//
// Plain C++ apart from __global__ / __shared__ / __syncthreads / threadIdx / blockIdx: tests/emu/ runs this file's
// kernel on CPU threads against the oracle's pyramid (borders included) in the CPU suite.
#pragma once
#include "vo_layout.hpp"

#ifndef TILE_STAMP_AT
#define TILE_STAMP_AT(k)  // (tools/tileprobe.hip defines it: phase time stamps of one workgroup, measurement builds only)
#endif

#ifndef PYR_SET_PRIO
#define PYR_SET_PRIO()  // (pyramid.hip: s_setprio 3)
#endif

#define PYR_NL_MAX 4   // levels produced above the launch's base level (deeper pyramids chain a second launch)
#define PYR_T0 64      // edge of the base-level tile a workgroup owns (top-level tile: PYR_T0 >> nl)
#ifndef PYR_NT
#define PYR_NT 512  // two wavefronts per SIMD (tools/tileprobe.hip: 26 us per launch with one, 17 with four — but a workgroup of four per SIMD does not fit next to a wavefront of the strict-border replay pool, which is resident on every compute unit while it waits for the frame kernel)
#endif
#define PYR_S0 112     // LDS row strides of the level regions (region edges 109, 53, 25, 11, 4 at most)
#define PYR_S1 56
#define PYR_S2 28
#define PYR_S3 12
#define PYR_S4 4

struct PyrTileArgs {
  const uint8_t *src[2];            // base-level image: the caller's image, or the origin of an already padded level
  int sstride[2];
  vo_level L[2][PYR_NL_MAX + 1];    // [image][0] the base level's plane (written when write_base), [1..nl] the levels produced
  int nl;                           // levels above the base (0..PYR_NL_MAX)
  int write_base;                   // 1: the base level is copied into L[.][0], border included
  int tiles_x, tiles_y;             // tiles of the top level, T x T each
  int T;                            // PYR_T0 >> nl
};

// geometry of one axis of one workgroup: per level k (0 = base) the region held in LDS [lo, lo + size) and the owned
// interval [o0, o1) (clipped to the image)
struct PyrAxis {
  int lo[PYR_NL_MAX + 1], size[PYR_NL_MAX + 1], o0[PYR_NL_MAX + 1], o1[PYR_NL_MAX + 1];
};
__device__ inline void pyr_axis(int tile, int T, int nl, const int *dim /* per level */, PyrAxis *A) {
  int lo = tile * T, hi = lo + T;  // top level: exactly the owned tile
  for (int k = nl; k >= 0; --k) {
    A->lo[k] = lo;
    A->size[k] = hi - lo;
    const int sh = nl - k;
    int o0 = (tile * T) << sh, o1 = ((tile + 1) * T) << sh;
    if (o0 > dim[k]) o0 = dim[k];
    if (o1 > dim[k]) o1 = dim[k];
    A->o0[k] = o0;
    A->o1[k] = o1;
    hi = 2 * hi + 1;  // taps 2x-2 .. 2x+2 of the last needed column
    lo = 2 * lo - 2;
  }
}

__device__ inline uint8_t *pyr_lds_level(uint8_t *a0, uint8_t *a1, uint8_t *a2, uint8_t *a3, uint8_t *a4, int k) {
  return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : (k == 3 ? a3 : a4)));
}
__device__ inline int pyr_lds_stride(int k) {
  return k == 0 ? PYR_S0 : (k == 1 ? PYR_S1 : (k == 2 ? PYR_S2 : (k == 3 ? PYR_S3 : PYR_S4)));
}

__global__ __launch_bounds__(PYR_NT) void pyr_build_kernel(PyrTileArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_l0[109 * PYR_S0 + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_l1[53 * PYR_S1 + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_l2[25 * PYR_S2 + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_l3[11 * PYR_S3 + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_l4[4 * PYR_S4 + 16];
  // mirror lists: the padded coordinates of a level's border whose REFLECT_101 source lies in this workgroup's owned
  // interval — (destination padded coordinate, source coordinate), per level and axis; a border is VO_PAD wide on each side
  __shared__ short s_mdst[PYR_NL_MAX + 1][2][2 * VO_PAD];
  __shared__ short s_msrc[PYR_NL_MAX + 1][2][2 * VO_PAD];
  __shared__ int s_mcnt[PYR_NL_MAX + 1][2];

  // the workgroup's geometry, per axis and level: in LDS because it is indexed by the level (as private arrays it went to
  // scratch memory: 224 bytes per lane)
  __shared__ PyrAxis s_axis[2];
  __shared__ int s_dim[2][PYR_NL_MAX + 1];
  // magic reciprocals (vo_divmod) of the run-time widths the loops below divide by, per level
  __shared__ unsigned s_m_size[PYR_NL_MAX + 1], s_m_ndw[PYR_NL_MAX + 1], s_m_wide[PYR_NL_MAX + 1], s_m_mx[PYR_NL_MAX + 1];

  // this image's level descriptors, copied from the kernel arguments by ONE batch of loads (indexed by image and level they
  // are memory accesses, one round trip each when made one after the other)
  __shared__ vo_level s_lv[PYR_NL_MAX + 1];
  const int tid = threadIdx.x, z = blockIdx.z;
  const int nl = a.nl;
  PYR_SET_PRIO();  // (a short kernel in front of, or next to, a frame's long-lived wavefronts)
  TILE_STAMP_AT(0);
  {
    constexpr int NW = (int)(sizeof(vo_level) / sizeof(int)) * (PYR_NL_MAX + 1);
    if (tid < NW) ((int *)s_lv)[tid] = ((const int *)a.L[z])[tid];
  }
  if (tid < 2 * (PYR_NL_MAX + 1)) s_mcnt[tid >> 1][tid & 1] = 0;
  __syncthreads();
  if (tid < 2) {
    const int tile = tid == 0 ? (int)blockIdx.x % a.tiles_x : (int)blockIdx.x / a.tiles_x;
    for (int k = 0; k <= nl; ++k) s_dim[tid][k] = tid == 0 ? s_lv[k].w : s_lv[k].h;
    pyr_axis(tile, a.T, nl, s_dim[tid], &s_axis[tid]);
  }
  __syncthreads();
  const PyrAxis &X = s_axis[0], &Y = s_axis[1];
  const int *wk = s_dim[0], *hk = s_dim[1];
  TILE_STAMP_AT(1);

  if (tid <= nl) s_m_size[tid] = vo_magic(X.size[tid]);
  // ---- stage the base-level region (only pixels inside the image are ever read back: taps are reflected first) -------
  // 16-byte pieces, every load of a thread in flight before its first LDS store: with one workgroup per compute unit a
  // dependent load costs its full latency (first version, one byte per load in a loop: 40 us of the kernel's 65)
  {
    const uint8_t *__restrict__ src = a.src[z];
    const int ss = a.sstride[z], x0 = X.lo[0], y0 = Y.lo[0], sx = X.size[0], sy = Y.size[0], w0 = wk[0], h0 = hk[0];
    const int nch = (sx + 15) >> 4, total = nch * sy;  // at most 7 x 109 pieces
    const unsigned m = vo_magic(nch);
    // (a) pieces that lie inside an image row: one 16-byte load each, no branch between a thread's loads (a piece that is
    //     not loaded reads a harmless address instead: the row start of an in-image row, or nothing when the image is
    //     narrower than a piece)
    constexpr int NP = (7 * 109 + PYR_NT - 1) / PYR_NT;  // pieces per thread
    vo_u128 v[NP];
    int dst[NP], edge[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int i = tid + q * PYR_NT;
      int ry, c;
      vo_divmod(i < total ? i : 0, nch, m, ry, c);
      const int y = y0 + ry, x = x0 + 16 * c;
      const bool row = i < total && y >= 0 && y < h0;
      const bool in = row && x >= 0 && x + 16 <= w0;
      dst[q] = in ? ry * PYR_S0 + 16 * c : -1;
      edge[q] = (row && !in && x + 16 > 0 && x < w0) ? i : -1;  // straddles the first or last column: (b)
      const uint8_t *p = src + (ptrdiff_t)(row ? y : 0) * ss + (in ? x : 0);
      if (w0 >= 16) {
        const vo_u128_unaligned u = *(const vo_u128_unaligned *)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q].v[e] = u.v[e];
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (dst[q] >= 0) *(vo_u128 *)(s_l0 + dst[q]) = v[q];
    // (b) the pieces at the image's first and last column (tiles at the image edge only), byte by byte; a byte outside the
    //     image is never read back (taps are reflected into the image first)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (edge[q] < 0) continue;
      int ry, c;
      vo_divmod(edge[q], nch, m, ry, c);
      const int y = y0 + ry, x = x0 + 16 * c;
      const uint8_t *rowp = src + (ptrdiff_t)y * ss;
      uint8_t bytes[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int xb = x + e;
        xb = xb < 0 ? 0 : (xb >= w0 ? w0 - 1 : xb);  // (clamped: every load is a valid one, all sixteen go out together)
        bytes[e] = rowp[xb];
      }
      vo_u128 wv;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        wv.v[e] = (uint32_t)bytes[4 * e] | ((uint32_t)bytes[4 * e + 1] << 8) | ((uint32_t)bytes[4 * e + 2] << 16) | ((uint32_t)bytes[4 * e + 3] << 24);
      *(vo_u128 *)(s_l0 + ry * PYR_S0 + 16 * c) = wv;
    }
  }
  TILE_STAMP_AT(2);
  // ---- the mirror lists of every level (2 * VO_PAD border coordinates per level and axis) --------------------------------
  for (int i = tid; i < (nl + 1) * 2 * (2 * VO_PAD); i += PYR_NT) {
    const int k = i / (4 * VO_PAD), r = i - k * (4 * VO_PAD), axis = r / (2 * VO_PAD), q = r - axis * (2 * VO_PAD);
    const int dim = axis ? hk[k] : wk[k];
    const int p = q < VO_PAD ? q - VO_PAD : dim + (q - VO_PAD);  // -VO_PAD .. -1, dim .. dim + VO_PAD - 1
    const int s = vo_reflect101(p, dim);
    const int o0 = axis ? Y.o0[k] : X.o0[k], o1 = axis ? Y.o1[k] : X.o1[k];
    if (s >= o0 && s < o1) {
      const int e = atomicAdd(&s_mcnt[k][axis], 1);
      s_mdst[k][axis][e] = (short)p;
      s_msrc[k][axis][e] = (short)s;
    }
  }
  __syncthreads();
  if (tid <= nl) {  // (read by the write phase, behind at least one more barrier)
    const int nx = X.o1[tid] - X.o0[tid];
    s_m_ndw[tid] = vo_magic((nx + 3) >> 2);
    s_m_wide[tid] = vo_magic(nx + s_mcnt[tid][0]);
    s_m_mx[tid] = vo_magic(s_mcnt[tid][0]);
  }
  if (nl == 0) __syncthreads();
  TILE_STAMP_AT(3);

  // ---- level k + 1 from level k, in LDS ------------------------------------------------------------------------
  for (int k = 0; k < nl; ++k) {
    const uint8_t *S = pyr_lds_level(s_l0, s_l1, s_l2, s_l3, s_l4, k);
    uint8_t *D = pyr_lds_level(s_l0, s_l1, s_l2, s_l3, s_l4, k + 1);
    const int ssd = pyr_lds_stride(k), dsd = pyr_lds_stride(k + 1);
    const int sx = X.size[k + 1], sy = Y.size[k + 1], n = sx * sy;
    const int sw = wk[k], sh = hk[k], dw = wk[k + 1], dh = hk[k + 1];
    const int slx = X.lo[k], sly = Y.lo[k];
    const unsigned msx = s_m_size[k + 1];
    for (int i = tid; i < n; i += PYR_NT) {
      int ry, rx;
      vo_divmod(i, sx, msx, ry, rx);
      const int x = X.lo[k + 1] + rx, y = Y.lo[k + 1] + ry;
      if (x < 0 || x >= dw || y < 0 || y >= dh) continue;
      int cx[5], cy[5];
      if (2 * x - 2 >= 0 && 2 * x + 2 < sw) {
#pragma unroll
        for (int c = 0; c < 5; ++c) cx[c] = 2 * x - 2 + c - slx;
      } else {
#pragma unroll
        for (int c = 0; c < 5; ++c) cx[c] = vo_reflect101(2 * x - 2 + c, sw) - slx;
      }
      if (2 * y - 2 >= 0 && 2 * y + 2 < sh) {
#pragma unroll
        for (int r = 0; r < 5; ++r) cy[r] = (2 * y - 2 + r - sly) * ssd;
      } else {
#pragma unroll
        for (int r = 0; r < 5; ++r) cy[r] = (vo_reflect101(2 * y - 2 + r, sh) - sly) * ssd;
      }
      int s = 0;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int wgt = (r == 0 || r == 4) ? 1 : ((r == 1 || r == 3) ? 4 : 6);
        const uint8_t *row = S + cy[r];
        s += wgt * ((int)row[cx[0]] + 4 * (int)row[cx[1]] + 6 * (int)row[cx[2]] + 4 * (int)row[cx[3]] + (int)row[cx[4]]);
      }
      D[ry * dsd + rx] = (uint8_t)((s + 128) >> 8);
    }
    __syncthreads();
    if (k == 0) TILE_STAMP_AT(4);
  }
  TILE_STAMP_AT(5);

  // ---- write what this workgroup owns of every level ----------------------------------------------------------------
  for (int k = a.write_base ? 0 : 1; k <= nl; ++k) {
    const uint8_t *S = pyr_lds_level(s_l0, s_l1, s_l2, s_l3, s_l4, k);
    const int ssd = pyr_lds_stride(k);
    const vo_level Lv = s_lv[k];
    uint8_t *org = Lv.base + (size_t)VO_PAD * Lv.stride + VO_PAD;  // pixel (0, 0)
    const int x0 = X.o0[k], x1 = X.o1[k], y0 = Y.o0[k], y1 = Y.o1[k];
    const int nx = x1 - x0, ny = y1 - y0;
    if (nx <= 0 || ny <= 0) continue;
    const int lx = X.lo[k], ly = Y.lo[k];
    // (a) the owned pixels themselves: x0 is a multiple of 4 and the plane's pixel (0, 0) is 4-byte aligned -> dword stores
    const int ndw = (nx + 3) >> 2;
    const unsigned m_ndw = s_m_ndw[k], m_wide = s_m_wide[k], m_mx = s_m_mx[k];
    for (int i = tid; i < ndw * ny; i += PYR_NT) {
      int ry, j;
      vo_divmod(i, ndw, m_ndw, ry, j);
      const int x = x0 + 4 * j, y = y0 + ry;
      const uint8_t *sp = S + (y - ly) * ssd + (x - lx);
      uint8_t *dp = org + (ptrdiff_t)y * Lv.stride + x;
      if (x + 3 < x1) {
        *(uint32_t *)dp = vo_bytes4(S, (y - ly) * ssd + (x - lx));  // (aligned dword reads: the region's column offset is even, not a multiple of 4)
      } else {  // the image's last columns: the bytes behind them are border pixels (another interval's mirror images)
        for (int q = 0; x + q < x1; ++q) dp[q] = sp[q];
      }
    }
    // (b) their mirror images in the border: mirror rows x (owned + mirror columns), owned rows x mirror columns
    const int mx = s_mcnt[k][0], my = s_mcnt[k][1];
    const int wide = nx + mx;
    for (int i = tid; i < my * wide; i += PYR_NT) {
      int e, c;
      vo_divmod(i, wide, m_wide, e, c);
      const int py = s_mdst[k][1][e], sy = s_msrc[k][1][e];
      const int px = c < nx ? x0 + c : s_mdst[k][0][c - nx], sx = c < nx ? x0 + c : s_msrc[k][0][c - nx];
      org[(ptrdiff_t)py * Lv.stride + px] = S[(sy - ly) * ssd + (sx - lx)];
    }
    for (int i = tid; i < ny * mx; i += PYR_NT) {
      int ry, e;
      vo_divmod(i, mx, m_mx, ry, e);
      const int y = y0 + ry, px = s_mdst[k][0][e], sx = s_msrc[k][0][e];
      org[(ptrdiff_t)y * Lv.stride + px] = S[(y - ly) * ssd + (sx - lx)];
    }
    if (k == 0) TILE_STAMP_AT(6);
  }
  TILE_STAMP_AT(7);
}
