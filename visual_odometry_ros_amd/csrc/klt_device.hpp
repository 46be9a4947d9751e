// klt_device.hpp — pyramidal Lucas-Kanade for one feature point on one gfx950 wavefront
// (device code shared by klt_track.hip and the fused frame kernel; see klt_track.hip for the
// mapping and the reference citations).
#pragma once
#include "vo_internal.hpp"

#define KLT_W_BITS 14

template <int WIN>
struct KltCfg {
  static constexpr int rl_for() {
    for (int r = 1; r <= WIN; ++r)
      if (((WIN + r - 1) / r) * WIN <= 64) return r;
    return WIN;
  }
  static constexpr int RL = rl_for();             // samples per lane
  static constexpr int RPR = (WIN + RL - 1) / RL;  // runs per window row
  static constexpr int SPAN = RPR * RL;            // columns covered (>= WIN)
  // template tile: rows -1..WIN+1, cols (aligned) covering -1..SPAN+1
  static constexpr int TT_H = WIN + 3;
  static constexpr int TT_WD = (SPAN + 3 + 3 + 3) / 4;  // dwords per row (3 align slack)
  // search tile
  static constexpr int M = (WIN <= 21) ? 4 : 3;
  static constexpr int TJ_H = WIN + 1 + 2 * M;
  static constexpr int TJ_WD = (SPAN + 1 + 2 * M + 3 + 3 + 3) / 4;  // ceil, incl. 3 align slack each side
  // per-lane register rows
  static constexpr int NB_T = RL + 3;                 // bytes per template row incl. halo
  static constexpr int ND_T = (NB_T + 3) / 4 + 1;     // dwords to read for any byte alignment
  static constexpr int NB_J = RL + 1;
  static constexpr int ND_J = (NB_J + 3) / 4 + 1;
};

__device__ __forceinline__ int descale_dev(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// a*w00 + b*w01 + c*w10 + d*w11 (+ acc) for |a..d| <= 4080 and weights in [-1, 2^14]: every operand
// fits the signed 24-bit multiplier. Written as v_mad_i32_i24 (full rate, multiply and add in one
// instruction); left to the compiler these become quarter-rate v_mul_lo_u32 plus separate adds.
__device__ __forceinline__ int klt_mad24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int klt_dot4(int a, int b, int c, int d, int w00, int w01, int w10, int w11, int acc) {
  return klt_mad24(d, w11, klt_mad24(c, w10, klt_mad24(b, w01, klt_mad24(a, w00, acc))));
}
__device__ __forceinline__ int klt_descale_dot4(int a, int b, int c, int d, int w00, int w01, int w10, int w11, int n) {
  return klt_dot4(a, b, c, d, w00, w01, w10, w11, 1 << (n - 1)) >> n;
}

// Packed 16-bit forms of the same integer arithmetic (every sum below is exact in int32, so the grouping of the
// terms does not change a bit of the result): pixels are 0..255, weights -1..2^14, template values and differences
// fit int16, so one v_dot2c_i32_i16 does two multiply-adds.
typedef short klt_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int klt_dot2(uint32_t a, uint32_t b, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(klt_s2, a), __builtin_bit_cast(klt_s2, b), acc, false);
}
// low halves of two ints as one packed pair (v_perm_b32)
__device__ __forceinline__ uint32_t klt_pack16(int lo, int hi) {
  return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}
// bytes k and k+1 (k = 0..3) of the eight bytes {hi:lo}, zero-extended to a packed 16-bit pair
__device__ __forceinline__ uint32_t klt_byte_pair(uint32_t lo, uint32_t hi, int k) {
  return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)((k + 1) << 16) | (uint32_t)k);
}
// bytes j, j+1 of a little-endian dword array
template <int NW>
__device__ __forceinline__ uint32_t klt_pair_at(const uint32_t (&w)[NW], int j) {
  const int d = j >> 2;
  return klt_byte_pair(w[d], w[d + 1 < NW ? d + 1 : d], j & 3);
}

// byte k of a little-endian dword array starting at byte offset `sh` (0..3) of w[0]
template <int ND>
__device__ __forceinline__ void align_row(const uint32_t (&w)[ND], int sh, uint32_t (&out)[ND - 1]) {
#pragma unroll
  for (int i = 0; i < ND - 1; ++i) out[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
}
template <int NW>
__device__ __forceinline__ int byte_at(const uint32_t (&w)[NW], int k) {
  return (int)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
}

__device__ __forceinline__ float uniform_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

struct KltResult {
  float x, y;  // nextPts
  int status;
  float err;
#ifdef FRAME_STAMP
  int iters;   // diagnostic: iterations over all levels
#endif
};

// One point of cv::calcOpticalFlowPyrLK on one wavefront (all 64 lanes must call it together).
// I / J: pyramid levels 0..max_level of the previous / next image; (ix, iy): initial nextPts
// (used when flags has VO_KLT_USE_INITIAL_FLOW); s_tt / s_tj: LDS tiles of KltCfg<WIN> size.
template <int WIN>
__device__ __forceinline__ KltResult klt_point(const vo_level *I, const vo_level *J, int max_level, int flags,
                                               int max_count, double epsilon, float min_eig, float p0x, float p0y,
                                               float ix, float iy, uint32_t *s_tt, uint32_t *s_tj, int lane) {
  using C = KltCfg<WIN>;
  constexpr int RL = C::RL;
  const int row = lane / C::RPR;
  const int x0 = (lane % C::RPR) * RL;
  const bool lane_on = row < WIN;
  // samples of this lane: columns x0 .. x0+nx-1 of window row `row`
  const int nx = lane_on ? ((x0 + RL <= WIN) ? RL : (WIN - x0 > 0 ? WIN - x0 : 0)) : 0;

  const float halfWin = (WIN - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  float npx = ix, npy = iy;  // "nextPts[ptidx]"
#ifdef FRAME_STAMP
  int dbg_iters = 0;
#endif
  int status = 1;
  float errv = 0.f;

  // Tile prefetch. The template tile of a level depends only on p0, so it is fetched into
  // registers one level ahead (in flight during the previous level's iterations); the search tile
  // depends on the previous level's result, so it is fetched at level entry and lands while the
  // template is being computed. Either way a level exposes the memory latency once, not twice.
  constexpr int NT = (C::TT_H * C::TT_WD + 63) / 64;
  constexpr int NJ = (C::TJ_H * C::TJ_WD + 63) / 64;
  uint32_t rt[NT];
  auto fetch_template_tile = [&](int lv) {
    const vo_level L = I[lv];
    const float ls = (float)(1. / (1 << lv));
    const int fx = __builtin_amdgcn_readfirstlane((int)floorf(p0x * ls - halfWin));
    const int fy = __builtin_amdgcn_readfirstlane((int)floorf(p0y * ls - halfWin));
    if (fx < -WIN || fx >= L.w || fy < -WIN || fy >= L.h) return;  // the level will be skipped
    const uint8_t *g = L.origin() + (ptrdiff_t)(fy - 1) * L.stride + ((fx - 1) & ~3);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int i = lane + 64 * q;
      const int ii = i < C::TT_H * C::TT_WD ? i : 0;
      const int r = ii / C::TT_WD, cdw = ii - r * C::TT_WD;
      rt[q] = *(const uint32_t *)(g + (ptrdiff_t)r * L.stride + cdw * 4);
    }
  };
#pragma unroll
  for (int q = 0; q < NT; ++q) rt[q] = 0;
  fetch_template_tile(max_level);

  for (int level = max_level; level >= 0; --level) {
    const vo_level LI = I[level];
    const vo_level LJ = J[level];
    const float lscale = (float)(1. / (1 << level));
    float prevx = p0x * lscale, prevy = p0y * lscale;
    float nextx, nexty;
    if (level == max_level) {
      if (flags & VO_KLT_USE_INITIAL_FLOW) {
        nextx = npx * lscale;
        nexty = npy * lscale;
      } else {
        nextx = prevx;
        nexty = prevy;
      }
    } else {
      nextx = npx * 2.f;
      nexty = npy * 2.f;
    }
    npx = nextx;
    npy = nexty;
    prevx -= halfWin;
    prevy -= halfWin;
    const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(prevx));
    const int ipy = __builtin_amdgcn_readfirstlane((int)floorf(prevy));
    if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
      if (level == 0) {
        status = 0;
        errv = 0.f;
      }
      if (level > 0) fetch_template_tile(level - 1);
      continue;
    }
    // search tile around the initial estimate of this level (the first iteration's window)
    uint32_t rs[NJ];
    int tjx = 0, tjy = 0;  // image coords of search-tile byte (0,0)
    bool tile_ok = false;
    {
      const int sx = __builtin_amdgcn_readfirstlane((int)floorf(nextx - halfWin));
      const int sy = __builtin_amdgcn_readfirstlane((int)floorf(nexty - halfWin));
#pragma unroll
      for (int q = 0; q < NJ; ++q) rs[q] = 0;
      if (!(sx < -WIN || sx >= LJ.w || sy < -WIN || sy >= LJ.h)) {
        tjx = (sx - C::M) & ~3;
        tjy = sy - C::M;
        tile_ok = true;
        const uint8_t *g = LJ.origin() + (ptrdiff_t)tjy * LJ.stride + tjx;
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
          const int i = lane + 64 * q;
          const int ii = i < C::TJ_H * C::TJ_WD ? i : 0;
          const int r = ii / C::TJ_WD, cdw = ii - r * C::TJ_WD;
          rs[q] = *(const uint32_t *)(g + (ptrdiff_t)r * LJ.stride + cdw * 4);
        }
      }
    }
    float fa = prevx - ipx, fb = prevy - ipy;
    int iw00 = (int)rintf((1.f - fa) * (1.f - fb) * (1 << KLT_W_BITS));
    int iw01 = (int)rintf(fa * (1.f - fb) * (1 << KLT_W_BITS));
    int iw10 = (int)rintf((1.f - fa) * fb * (1 << KLT_W_BITS));
    int iw11 = (1 << KLT_W_BITS) - iw00 - iw01 - iw10;

    // ---- stage the template tile (rows ipy-1.., cols aligned down from ipx-1) ----
    const int tx0 = (ipx - 1) & ~3;  // VO_PAD is a multiple of 4, so this is 4-byte aligned in memory
    const int tsh = (ipx - 1) - tx0;
    {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const int i = lane + 64 * q;
        if (i < C::TT_H * C::TT_WD) s_tt[i] = rt[q];  // fetched one level ahead
      }
      __syncthreads();
    }
    // ---- per-lane template: I, Ix, Iy at RL samples (Scharr on the fly), as packed 16-bit pairs ----
    // Every quantity fits int16 (pixels <= 255, column sums <= 4080, derivatives within +-4080, template <= 8160), and
    // every sum is exact, so pairs of neighbouring columns go through v_pk_* / v_dot2 two at a time — the same
    // integers as cv::Scharr + the bilinear taps of calcOpticalFlowPyrLK, half the instructions.
    constexpr int NP = (RL + 1) / 2;  // pairs of samples
    uint32_t tIp[NP], tXp[NP], tYp[NP];
    int pA11 = 0, pA12 = 0, pA22 = 0;
    {
      // tile rows row..row+3 <-> image rows ipy+row-1 .. ipy+row+2 ; bytes from column (x0) incl. halo
      uint32_t rb[4][C::ND_T - 1];
      const int boff = tsh + x0;  // byte offset of image column ipx+x0-1 within the tile row
      const int dwo = boff >> 2, sh = boff & 3;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint32_t w[C::ND_T];
        const int trow = lane_on ? row + rr : rr;
        const char *rowp = (const char *)s_tt + klt_mad24(trow, C::TT_WD * 4, 0);
#pragma unroll
        for (int d = 0; d < C::ND_T; ++d) {
          const int cd = dwo + d;
          w[d] = *(const uint32_t *)(rowp + (cd < C::TT_WD ? cd : C::TT_WD - 1) * 4);
        }
        align_row<C::ND_T>(w, sh, rb[rr]);
      }
      constexpr int PC = (RL + 4) / 2;  // pairs of columns (RL + 3 of them, halo included)
      constexpr int QD = (RL + 2) / 2;  // pairs of derivative samples (RL + 1 of them)
      klt_s2 U[4][PC];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int m = 0; m < PC; ++m) U[rr][m] = __builtin_bit_cast(klt_s2, klt_pair_at(rb[rr], 2 * m));
      // derivative samples k = 0..RL of this lane sit at image columns ipx + x0 + k: the derivative plane is
      // zero-padded, so samples outside the image are cleared (bit k of xbits: inside)
      const int xb = ipx + x0;
      const int kmin = min(max(-xb, 0), RL + 1), kmax = min(max(LI.w - xb, 0), RL + 1);
      const unsigned xbits = ((1u << kmax) - 1u) & ~((1u << kmin) - 1u);
      uint32_t DX[2][QD], DY[2][QD];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        klt_s2 t0[PC], t1[PC];
#pragma unroll
        for (int m = 0; m < PC; ++m) {
          t0[m] = (U[r][m] + U[r + 2][m]) * (short)3 + U[r + 1][m] * (short)10;
          t1[m] = U[r + 2][m] - U[r][m];
        }
        const bool yin = (unsigned)(ipy + row + r) < (unsigned)LI.h;
        const int bits = yin ? (int)xbits : 0;
#pragma unroll
        for (int m = 0; m < QD; ++m) {
          const klt_s2 dx = t0[m + 1] - t0[m];
          const klt_s2 mid = __builtin_bit_cast(
              klt_s2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, t1[m + 1]), __builtin_bit_cast(uint32_t, t1[m]), 16));
          const klt_s2 dy = (t1[m + 1] + t1[m]) * (short)3 + mid * (short)10;
          // 0xFFFF per half whose sample is inside: sign-extended one-bit fields, merged
          const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe(bits, 2 * m, 1), hi = (uint32_t)__builtin_amdgcn_sbfe(bits, 2 * m + 1, 1);
          const uint32_t msk = (lo & 0xFFFFu) | (hi & 0xFFFF0000u);
          DX[r][m] = __builtin_bit_cast(uint32_t, dx) & msk;
          DY[r][m] = __builtin_bit_cast(uint32_t, dy) & msk;
        }
      }
      const uint32_t w0p = klt_pack16(iw00, iw01), w1p = klt_pack16(iw10, iw11);
      int vI[2 * NP], vX[2 * NP], vY[2 * NP];
#pragma unroll
      for (int j = 0; j < RL; ++j) {
        // samples j, j+1 of a packed row: the pair itself (j even) or the halves of two neighbouring pairs (j odd)
        auto pair_of = [&](const uint32_t(&a)[QD]) -> uint32_t {
          return (j & 1) ? __builtin_amdgcn_alignbit(a[(j + 1) / 2], a[j / 2], 16) : a[j / 2];
        };
        const uint32_t i0 = ((j + 1) & 1) ? klt_pair_at(rb[1], j + 1) : __builtin_bit_cast(uint32_t, U[1][(j + 1) / 2]);
        const uint32_t i1 = ((j + 1) & 1) ? klt_pair_at(rb[2], j + 1) : __builtin_bit_cast(uint32_t, U[2][(j + 1) / 2]);
        vI[j] = klt_dot2(i1, w1p, klt_dot2(i0, w0p, 1 << (KLT_W_BITS - 5 - 1))) >> (KLT_W_BITS - 5);
        const int ixval = klt_dot2(pair_of(DX[1]), w1p, klt_dot2(pair_of(DX[0]), w0p, 1 << (KLT_W_BITS - 1))) >> KLT_W_BITS;
        const int iyval = klt_dot2(pair_of(DY[1]), w1p, klt_dot2(pair_of(DY[0]), w0p, 1 << (KLT_W_BITS - 1))) >> KLT_W_BITS;
        const bool on = j < nx;
        vX[j] = on ? ixval : 0;  // (all three fit int16: the reference's (short) casts change nothing)
        vY[j] = on ? iyval : 0;
      }
      if (RL & 1) vI[2 * NP - 1] = vX[2 * NP - 1] = vY[2 * NP - 1] = 0;
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        tIp[m] = klt_pack16(vI[2 * m], vI[2 * m + 1]);
        tXp[m] = klt_pack16(vX[2 * m], vX[2 * m + 1]);
        tYp[m] = klt_pack16(vY[2 * m], vY[2 * m + 1]);
        pA11 = klt_dot2(tXp[m], tXp[m], pA11);
        pA12 = klt_dot2(tXp[m], tYp[m], pA12);
        pA22 = klt_dot2(tYp[m], tYp[m], pA22);
      }
    }
    if (level > 0) fetch_template_tile(level - 1);  // in flight during this level's iterations
    float sA11, sA12, sA22;
    wave_sum3_i32_to_f32(pA11, pA12, pA22, sA11, sA12, sA22);
    const float A11 = sA11 * FLT_SCALE;
    const float A12 = sA12 * FLT_SCALE;
    const float A22 = sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    if (minEig < min_eig || D < 1.19209290e-07f) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    nextx -= halfWin;
    nexty -= halfWin;
    float pdx = 0.f, pdy = 0.f;
    if (tile_ok) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NJ; ++q) {
        const int i = lane + 64 * q;
        if (i < C::TJ_H * C::TJ_WD) s_tj[i] = rs[q];
      }
      __syncthreads();
    }

    if (!tile_ok) tjx = 0x40000000;  // (no tile yet: the first window test below reloads)

    // bilinear difference of the current window against the template, per lane: dp[m] = packed pair of
    // (J sample - template sample) for samples 2m, 2m+1 (each within +-8160)
    auto eval_diffs = [&](int inx, int iny, int w00, int w01, int w10, int w11, uint32_t (&dp)[NP], bool check = true) {
      // window inside the staged tile <=> 0 <= inx - tjx <= slack_x and 0 <= iny - tjy <= slack_y
      if (check && __builtin_expect((unsigned)inx - (unsigned)tjx > (unsigned)(C::TJ_WD * 4 - 3 - C::SPAN - 1) ||
                               (unsigned)iny - (unsigned)tjy > (unsigned)(C::TJ_H - WIN - 1),
                           0)) {
        tjx = (inx - C::M) & ~3;
        tjy = iny - C::M;
        const uint8_t *g = LJ.origin() + (ptrdiff_t)tjy * LJ.stride + tjx;
        __syncthreads();
        for (int i = lane; i < C::TJ_H * C::TJ_WD; i += 64) {
          const int r = i / C::TJ_WD, cdw = i - r * C::TJ_WD;
          s_tj[i] = *(const uint32_t *)(g + (ptrdiff_t)r * LJ.stride + cdw * 4);
        }
        __syncthreads();
      }
      const int boff = (inx - tjx) + x0;
      const int dwo = boff >> 2, sh = boff & 3;
      const int trow = (iny - tjy) + (lane_on ? row : 0);
      // (byte offset of the tile row; as v_mad_i32_i24 — the compiler's v_mul_lo_u32 is quarter rate)
      const char *rowp = (const char *)s_tj + klt_mad24(trow, C::TJ_WD * 4, 0);
      uint32_t r0[C::ND_J - 1], r1[C::ND_J - 1];
      {
        uint32_t w0[C::ND_J], w1[C::ND_J];
#pragma unroll
        for (int d = 0; d < C::ND_J; ++d) {
          const int cd = dwo + d < C::TJ_WD ? dwo + d : C::TJ_WD - 1;
          w0[d] = *(const uint32_t *)(rowp + cd * 4);
          w1[d] = *(const uint32_t *)(rowp + (C::TJ_WD + cd) * 4);
        }
        align_row<C::ND_J>(w0, sh, r0);
        align_row<C::ND_J>(w1, sh, r1);
      }
      const uint32_t w0p = klt_pack16(w00, w01), w1p = klt_pack16(w10, w11);
      int v[2 * NP];
#pragma unroll
      for (int j = 0; j < RL; ++j)
        v[j] = klt_dot2(klt_pair_at(r1, j), w1p, klt_dot2(klt_pair_at(r0, j), w0p, 1 << (KLT_W_BITS - 5 - 1))) >>
               (KLT_W_BITS - 5);
      if (RL & 1) v[2 * NP - 1] = 0;
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const klt_s2 d = __builtin_bit_cast(klt_s2, klt_pack16(v[2 * m], v[2 * m + 1])) - __builtin_bit_cast(klt_s2, tIp[m]);
        dp[m] = __builtin_bit_cast(uint32_t, d);
      }
    };

    for (int j = 0; j < max_count; ++j) {
#ifdef FRAME_STAMP
      ++dbg_iters;
#endif
      // (nextx / nexty come from wave sums: the compiler knows they are wave-uniform and branches on them with scalar
      // code by itself; forcing them into SGPRs here costs a dozen moves and hazard nops per iteration)
      const int inx = (int)floorf(nextx);
      const int iny = (int)floorf(nexty);
      // inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h
      if (__builtin_expect((unsigned)inx + (unsigned)WIN >= (unsigned)(LJ.w + WIN) || (unsigned)iny + (unsigned)WIN >= (unsigned)(LJ.h + WIN), 0)) {
        if (level == 0) status = 0;
        break;
      }
      fa = nextx - inx;
      fb = nexty - iny;
      iw00 = (int)rintf((1.f - fa) * (1.f - fb) * (1 << KLT_W_BITS));
      iw01 = (int)rintf(fa * (1.f - fb) * (1 << KLT_W_BITS));
      iw10 = (int)rintf((1.f - fa) * fb * (1 << KLT_W_BITS));
      iw11 = (1 << KLT_W_BITS) - iw00 - iw01 - iw10;
      uint32_t dp[NP];
#ifdef KLT_UNSAFE_SKIP_WINDOW_TEST_AFTER  // MEASUREMENT ONLY: an upper bound for what predicting an
      // oscillating run and dropping its search-tile window test could save; not correct when the window leaves the tile
      eval_diffs(inx, iny, iw00, iw01, iw10, iw11, dp, j < KLT_UNSAFE_SKIP_WINDOW_TEST_AFTER);
#else
      eval_diffs(inx, iny, iw00, iw01, iw10, iw11, dp);
#endif
      int pb1 = 0, pb2 = 0;
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        pb1 = klt_dot2(dp[m], tXp[m], pb1);
        pb2 = klt_dot2(dp[m], tYp[m], pb2);
      }
      float sb1, sb2;
      wave_sum2_i32_to_f32(pb1, pb2, sb1, sb2);
      const float b1 = sb1 * FLT_SCALE;
      const float b2 = sb2 * FLT_SCALE;
      const float dx = (float)((A12 * b2 - A22 * b1) * D);
      const float dy = (float)((A12 * b1 - A11 * b2) * D);
      nextx += dx;
      nexty += dy;
      npx = nextx + halfWin;
      npy = nexty + halfWin;
      if (__builtin_expect((double)dx * dx + (double)dy * dy <= epsilon, 0)) break;
      // (double)|f| < 0.01 for a float f  <=>  |f| <= 0.01f : 0.01f = 0.00999999977... is the largest
      // float below 0.01 (the next one is 0.0100000007...)
      if (__builtin_expect(j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f, 0)) {
        npx -= dx * 0.5f;
        npy -= dy * 0.5f;
        break;
      }
      pdx = dx;
      pdy = dy;
    }

    if (status && level == 0) {
      const float ex = npx - halfWin, ey = npy - halfWin;
      const int inx = __builtin_amdgcn_readfirstlane((int)floorf(ex));
      const int iny = __builtin_amdgcn_readfirstlane((int)floorf(ey));
      if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
        status = 0;
      } else {
        const float aa = ex - inx, bb = ey - iny;
        iw00 = (int)rintf((1.f - aa) * (1.f - bb) * (1 << KLT_W_BITS));
        iw01 = (int)rintf(aa * (1.f - bb) * (1 << KLT_W_BITS));
        iw10 = (int)rintf((1.f - aa) * bb * (1 << KLT_W_BITS));
        iw11 = (1 << KLT_W_BITS) - iw00 - iw01 - iw10;
        uint32_t dp[NP];
        eval_diffs(inx, iny, iw00, iw01, iw10, iw11, dp);
        int pe = 0;
#pragma unroll
        for (int k = 0; k < RL; ++k) {
          const int dk = (k & 1) ? ((int)dp[k >> 1] >> 16) : (int)(short)(dp[k >> 1] & 0xFFFFu);
          pe += (k < nx) ? abs(dk) : 0;
        }
        const float errval = (float)wave_sum_i32(pe);  // < 2^24: exact
        errv = errval * 1.f / (float)(32 * WIN * WIN);
      }
    }
  }
  KltResult res;
  res.x = npx;
  res.y = npy;
  res.status = status;
  res.err = errv;
#ifdef FRAME_STAMP
  res.iters = dbg_iters;
#endif
  return res;
}

