// ic_device.hpp — device code of the scale-compensated inverse-compositional refinement
// (shared by ic_refine.hip and the fused frame kernel; see ic_refine.hip for the design notes
// and the reference citations).
#pragma once
#include "vo_internal.hpp"

#define IC_HALF 11
#define IC_NELEM 264
#define IC_T 64    // lanes per point (one wavefront)
#define IC_K 5     // taps per lane: lane l owns taps l + 64 k
#define IC_TW 8    // template tile dwords per row (32 bytes)
#define IC_TH 27   // template tile rows
#define IC_TN ((IC_TH * IC_TW + IC_T - 1) / IC_T)  // template tile dwords per lane
#define IC_JW 11   // I1 search tile: dwords per row (44 bytes)
#define IC_JH 40   // I1 search tile rows
#define IC_JN ((IC_JH * IC_JW + IC_T - 1) / IC_T)  // search tile dwords per lane
#define IC_MW 9        // mask words per point (264 bits)
#define IC_MAXRUN 192  // longest run the parallel strict replay handles (else sequential fallback)
#define IC_CAND 320    // predecessors examined for the nearest clean point
#define IC_JAC_OVF 15
#define IC_JAC_NT 14
#define IC_JAC_VER 13    // replay kernel: number of record publications so far
#define IC_JAC_P1 12     // frame kernel: features whose pass-1 record (and list entry) is complete
#define IC_JAC_SLOTS 16  // replay kernel: one "idle at version" word per workgroup
#define IC_JGRID 512     // workgroups of the replay kernel (2 per CU: all co-resident); each strides over the touched list
#define IC_CONC_GRID 256  // workgroups of the concurrent replay (one per CU, resident next to the frame kernel from its start)
#define IC_JAC_WORDS (IC_JAC_SLOTS + IC_JGRID)
#define IC_JAC_BYTES (((IC_JAC_WORDS * 4 + 63) / 64) * 64)
#define IC_MAX_PASSES (1 << 14)  // a look per version bump is normal; this only guards against a hang
#define IC_SPIN_LIMIT (1 << 17)  // idle polls before a workgroup gives up (~0.2 s) and requests the sequential replay
#define IC_DBG_OFF 4096  // IC_STAMP builds: debug words in the unused tail of tlist (needs max_points >= 4200)
#ifndef IC_MAX_ITER
#define IC_MAX_ITER 30  // feature_tracker.cpp:290
#endif
#ifndef IC_WAIT_SLEEP
#define IC_WAIT_SLEEP 16  // s_sleep argument between two polls of a waiting feature's writers (~0.5 us)
#endif

struct IcArgs {
  vo_level I0, I1;
  const float *pts0;
  const float *scale;
  const float *pts_prior;  // initial pts_track
  float *pts_track;        // out (pre-set to the prior)
  const uint8_t *mask_in;  // phase 1: entry mask (null = all true); may alias mask
  uint8_t *mask;           // out
  uint8_t *touched;        // out (phase 1) / in (strict)
  uint8_t *cls;            // out (phase 1) / in (strict): 0 skipped, 1 template only, 2 iterated
  float *last_pu;          // out (phase 1) / in (strict): last evaluated pt_update
  int n;
  const int *d_n;
  int *flags;              // [0] |= 1 ax/ay NaN, |= 2 patch NaN, |= 4 update NaN
  // per-point tap records for the parallel strict replay (optional; see ic_jacobi_kernel)
  uint32_t *recW0, *recW1;  // [n][IC_MW] bit j: tap j written by the point's template / I1 samples
  float *recV0;             // [n][3][IC_NELEM] template values (I0, du, dv) of the written taps
  float *recV1;             // [n][IC_NELEM] last I1 value the point wrote per tap
  float *pre1;              // [n][IC_NELEM] I1 pre-state the point last ran with
  uint32_t *preM;           // [n][IC_MW] its mask
  int *pubc;                // [n] how often the feature's record was published (monotonic across frames)
  uint8_t *ready;           // [n] touched feature has published its first strict-state result (cleared by pass 1)
  int *jac;                 // control words of the replay: [IC_JAC_NT] #touched, [IC_JAC_OVF], [IC_JAC_VER], slots
  int *tlist;               // indices of the touched points (any order)
  // concurrent replay (ic_replay<true>: the kernel that writes the pass-1 records is still running)
  unsigned long long *tl2;  // [n] list entries {epoch << 32 | feature}, written through to memory
  int *p1e;                 // [n] epoch stamp: the feature's pass-1 record and outputs are complete in memory
  int epoch;                // this frame's stamp (never 0)
  int *p1_word;             // cumulative count of features past pass 1, in IC_P1_SHARDS shards on lines of their own
                            // (shard s at p1_word[s * IC_P1_STRIDE]; feature i counts in shard i % IC_P1_SHARDS) ...
  int p1_target;            // ... and what their sum reads when all of this frame's have passed
};

struct IcShared {
  uint32_t tt[IC_TH * IC_TW];
  uint32_t tj[IC_JH * IC_JW];
};

struct IcState {
  float I0[IC_K], du[IC_K], dv[IC_K], I1[IC_K];
  unsigned m;  // bit k: template tap k valid (mask_I0) ; bit 8+k: I1 tap k valid (mask_I1)
               // bit 16+k / 24+k: written by THIS point's template / I1 evaluations
};

// lane-constant tap offsets (feature_tracker.cpp:308-320): rows v = 0..22; even rows hold
// u = 1,3,..,21 (11 taps), odd rows u = 0,2,..,22 (12 taps)
struct IcTaps {
  float px[IC_K], py[IC_K];
  unsigned on;  // bit k: the lane owns tap lane + 64 k (< 264)
};
__device__ __forceinline__ void ic_tap_xy(int j, float &px, float &py) {
  const int p = j / 23, r = j - p * 23;
  int u, v;
  if (r < 11) {
    v = 2 * p;
    u = 1 + 2 * r;
  } else {
    v = 2 * p + 1;
    u = 2 * (r - 11);
  }
  px = (float)(u - IC_HALF);
  py = (float)(v - IC_HALF);
}
__device__ __forceinline__ IcTaps ic_make_taps(int lane) {
  IcTaps tp;
  tp.on = 0;
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const int j = lane + 64 * k;
    const bool on = j < IC_NELEM;
    if (on) tp.on |= 1u << k;
    ic_tap_xy(on ? j : 0, tp.px[k], tp.py[k]);
  }
  return tp;
}

// Slot k of a lane holds tap lane + 64 k: slots 0..3 always do (256 < IC_NELEM), slot 4 only on lanes 0..7. Written
// this way the hot loops carry one lane mask instead of five.
static_assert(IC_NELEM > 4 * IC_T && IC_NELEM <= 5 * IC_T && IC_K == 5, "tap ownership pattern");
__device__ __forceinline__ bool ic_tap_on(const IcTaps &tp, int k) { return k < 4 ? true : ((tp.on >> 4) & 1u) != 0; }
// a * b + c on 24-bit operands (LDS offsets): full rate, where the compiler's 64-bit multiply-add is not
__device__ __forceinline__ int ic_mad24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ float ic_bilin(float I1, float I2, float I3, float I4, float ax, float ay, float axay) {
  return ((axay * (((I1 - I2) - I3) + I4) + ax * (-I1 + I2)) + ay * (-I1 + I3)) + I1;
}

__device__ __forceinline__ int ic_safe_int(float v) {
  return (int)fminf(fmaxf(v, -1.0e6f), 1.0e6f);
}

// Wavefront sums of four per-lane partials in the canonical tree order; every lane gets the same bits.
__device__ __forceinline__ void ic_wave_sum4(float (&v)[4]) { wave_sum4_f32(v[0], v[1], v[2], v[3]); }

// ---- LDS tiles: the global loads are issued first (registers), committed to LDS later, so
// that both tiles of a point share one exposure of the memory latency ----
template <int N>
struct IcTileRegs {
  uint32_t r[N];
  int x0, y0;  // image coordinates of tile byte (0,0); x0 is 4-byte aligned in memory
};
template <int N, int TW, int TH>
__device__ __forceinline__ void ic_tile_fetch(const vo_level &L, int x0, int y0, int lane, IcTileRegs<N> &R) {
  R.x0 = x0;
  R.y0 = y0;
  const uint8_t *g = L.origin() + (ptrdiff_t)y0 * L.stride + x0;
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int i = lane + IC_T * q;
    const int ii = i < TW * TH ? i : 0;
    const int r = ii / TW, cdw = ii - r * TW;
    R.r[q] = *(const uint32_t *)(g + (ptrdiff_t)r * L.stride + cdw * 4);
  }
}
template <int N, int TW, int TH>
__device__ __forceinline__ void ic_tile_commit(const IcTileRegs<N> &R, int lane, uint32_t *dst) {
  __syncthreads();  // earlier readers of the tile are done (one wavefront: ordering only)
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int i = lane + IC_T * q;
    if (i < TW * TH) dst[i] = R.r[q];
  }
  __syncthreads();
}
typedef IcTileRegs<IC_TN> IcTRegs;
typedef IcTileRegs<IC_JN> IcJRegs;

__device__ __forceinline__ void ic_template_fetch(const vo_level &L0, float pt0x, float pt0y, int lane, IcTRegs &R) {
  const int W = L0.w, H = L0.h;
  const int cx = ic_safe_int(pt0x), cy = ic_safe_int(pt0y);
  int tox = (cx - 13) & ~3;
  int toy = cy - 12;  // rows cy-12 .. cy+14 cover every valid tap's 4x4 neighbourhood
  tox = max(-VO_PAD, min(tox, ((W + VO_PAD - IC_TW * 4) & ~3)));
  toy = max(-VO_PAD, min(toy, H + VO_PAD - IC_TH));
  ic_tile_fetch<IC_TN, IC_TW, IC_TH>(L0, tox, toy, lane, R);
}
// I1 search tile: 40 rows x 44 B of the current image around the prior position, staged once per
// point; taps that fall outside it (large drift or scale) fall back to global loads per lane.
__device__ __forceinline__ void ic_I1_fetch(const vo_level &L1, float cxf, float cyf, int lane, IcJRegs &R) {
  const int cx = ic_safe_int(cxf), cy = ic_safe_int(cyf);
  int x0 = (cx - 19) & ~3;
  int y0 = cy - 19;
  x0 = max(-VO_PAD, min(x0, ((L1.w + VO_PAD - IC_JW * 4) & ~3)));
  y0 = max(-VO_PAD, min(y0, L1.h + VO_PAD - IC_JH));
  ic_tile_fetch<IC_JN, IC_JW, IC_JH>(L1, x0, y0, lane, R);
}
struct IcTile {
  int x0, y0;
};

// one template tap: 4x4 neighbourhood (u0-1..u0+2, v0-1..v0+2) from the LDS tile
__device__ __forceinline__ void ic_template_tap(const uint32_t *s_t, int bx, int by, float ax, float ay, float axay,
                                                float &nI, float &nu, float &nv) {
  const int dwo = bx >> 2, sh = bx & 3;
  int b[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t w0 = s_t[(by + r) * IC_TW + dwo];
    const uint32_t w1 = s_t[(by + r) * IC_TW + min(dwo + 1, IC_TW - 1)];
    const uint32_t v = __builtin_amdgcn_alignbyte(w1, w0, sh);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) b[r][cc] = (int)((v >> (8 * cc)) & 0xFFu);
  }
  float Iv[2][2], du[2][2], dv[2][2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      Iv[jj][ii] = (float)b[1 + jj][1 + ii];
      du[jj][ii] = (float)((b[jj][ii + 2] - b[jj][ii]) + 2 * (b[jj + 1][ii + 2] - b[jj + 1][ii]) +
                           (b[jj + 2][ii + 2] - b[jj + 2][ii]));
      dv[jj][ii] = (float)((b[jj + 2][ii] - b[jj][ii]) + 2 * (b[jj + 2][ii + 1] - b[jj][ii + 1]) +
                           (b[jj + 2][ii + 2] - b[jj][ii + 2]));
    }
  nI = ic_bilin(Iv[0][0], Iv[0][1], Iv[1][0], Iv[1][1], ax, ay, axay);
  nu = ic_bilin(du[0][0], du[0][1], du[1][0], du[1][1], ax, ay, axay);
  nv = ic_bilin(dv[0][0], dv[0][1], dv[1][0], dv[1][1], ax, ay, axay);
}

// interpImage3SameRatio on the taps of this lane (tile already in LDS): writes state where the
// tap is valid. STRICT: mask bits are sticky (never reset); otherwise they are this evaluation's validity.
template <bool STRICT>
__device__ __forceinline__ void ic_template(const vo_level &L0, const IcTaps &tp, float pt0x, float pt0y, float ax,
                                            float ay, float axay, const IcTile &tile, const IcShared &sh, IcState &S,
                                            int &touched) {
  const int W = L0.w, H = L0.h;
  if (!STRICT) S.m &= ~0x1Fu;
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const bool on = ic_tap_on(tp, k);
    const float uc = pt0x + tp.px[k], vc = pt0y + tp.py[k];
    const int u0 = (int)uc, v0 = (int)vc;
    const bool valid = on && !(u0 < 1 || u0 >= W - 2 || v0 < 1 || v0 >= H - 2);
    if (on && !valid) touched = 1;
    float nI, nu, nv;
    ic_template_tap(sh.tt, valid ? (u0 - 1) - tile.x0 : 0, valid ? (v0 - 1) - tile.y0 : 0, ax, ay, axay, nI, nu, nv);
    S.I0[k] = valid ? nI : S.I0[k];
    S.du[k] = valid ? nu : S.du[k];
    S.dv[k] = valid ? nv : S.dv[k];
    if (valid) S.m |= 0x10001u << k;  // bit 16+k: written by THIS point
  }
}

// interpImageSameRatio on the taps of this lane (float compares, image_processing.cpp:79-118)
template <bool STRICT>
__device__ __forceinline__ void ic_sample_I1(const vo_level &L1, const IcTaps &tp, const float (&sx)[IC_K],
                                             const float (&sy)[IC_K], float pux, float puy, float ax, float ay,
                                             float axay, IcState &S, int &touched, const IcTile &tile,
                                             const IcShared &sh) {
  const uint8_t *sb = (const uint8_t *)sh.tj;
  if (!STRICT) S.m &= ~0x1F00u;
  const float fw = (float)(L1.w - 2), fh = (float)(L1.h - 2);
  float val[IC_K];
  unsigned vmask = 0, gmask = 0;
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const bool on = ic_tap_on(tp, k);
    const float uc = pux + sx[k], vc = puy + sy[k];
    const bool valid = on && !(uc < 1 || uc >= fw || vc < 1 || vc >= fh);
    const int u0 = (int)uc, v0 = (int)vc;
    const int lx = u0 - tile.x0, ly = v0 - tile.y0;
    const bool inside = valid && (unsigned)lx < (unsigned)(IC_JW * 4 - 1) && (unsigned)ly < (unsigned)(IC_JH - 1);
    const int off = ic_mad24(ly, IC_JW * 4, lx);  // (unconditional: inline asm under a select becomes a branch)
    const uint8_t *q = sb + (inside ? off : 0);
    val[k] = ic_bilin((float)q[0], (float)q[1], (float)q[IC_JW * 4], (float)q[IC_JW * 4 + 1], ax, ay, axay);
    if (valid) vmask |= 1u << k;
    if (valid && !inside) gmask |= 1u << k;
    if (on && !valid) touched = 1;
  }
  if (__any(gmask != 0)) {  // rare: the window left the staged tile
#pragma unroll
    for (int k = 0; k < IC_K; ++k)
      if ((gmask >> k) & 1u) {
        const int u0 = (int)(pux + sx[k]), v0 = (int)(puy + sy[k]);
        const uint8_t *p = L1.origin() + (ptrdiff_t)v0 * L1.stride + u0;
        val[k] = ic_bilin((float)p[0], (float)p[1], (float)p[L1.stride], (float)p[L1.stride + 1], ax, ay, axay);
      }
  }
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const bool valid = (vmask >> k) & 1u;
    S.I1[k] = valid ? val[k] : S.I1[k];
  }
  S.m |= vmask * 0x1000100u;  // bits 8+k and 24+k (written by THIS point)
}

__device__ __forceinline__ void ic_frac(float x, float y, float &ax, float &ay, float &axay) {
  // pt - floor(pt) (feature_tracker.cpp:355-356, :404-405); exact in float
  ax = x - floorf(x);
  ay = y - floorf(y);
  axay = ax * ay;
}

struct IcResult {
  int cls;       // 1: template only (range / determinant test failed), 2: iterated
  int ok;        // mask_valid
  float x, y;    // pts_track: refined position when ok, else the prior
  int err_flag;  // 1 ax/ay NaN, 2 patch NaN, 4 update NaN (the reference throws)
};

// What a feature needs before it can iterate: the template applied to the state, the inverse of the
// 2x2 system and the search tile in LDS. Independent of the I1 part of the state, so the strict
// replay prepares a feature while it still waits for its I1 inputs.
struct IcPrep {
  int cls;  // 1: rejected before iterating (range or determinant test), 2: ready to iterate
  float iD_A11, iD_A12, iD_A22;
  IcTile tile;  // search tile staged in sh.tj
};
template <bool STRICT>
__device__ __forceinline__ IcPrep ic_prepare(const vo_level &I0, const vo_level &I1, const IcTaps &tp, float pt0x,
                                             float pt0y, float pt1x, float pt1y, int lane, IcShared &sh, IcState &S,
                                             int &touched) {
  IcPrep pr;
  pr.cls = 1;
  pr.iD_A11 = pr.iD_A12 = pr.iD_A22 = 0.f;
  pr.tile.x0 = pr.tile.y0 = 0;
  float ax, ay, axay;
  ic_frac(pt0x, pt0y, ax, ay, axay);
  if (ax < 0 || ax > 1 || ay < 0 || ay > 1) return pr;
  IcTRegs rt;
  IcJRegs rj;
  ic_template_fetch(I0, pt0x, pt0y, lane, rt);
  ic_I1_fetch(I1, pt1x, pt1y, lane, rj);
  ic_tile_commit<IC_TN, IC_TW, IC_TH>(rt, lane, sh.tt);
  const IcTile tt = {rt.x0, rt.y0};
  ic_template<STRICT>(I0, tp, pt0x, pt0y, ax, ay, axay, tt, sh, S, touched);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const bool use = (S.m >> k) & 1u;
    acc[0] = use ? acc[0] + S.du[k] * S.du[k] : acc[0];
    acc[1] = use ? acc[1] + S.du[k] * S.dv[k] : acc[1];
    acc[2] = use ? acc[2] + S.dv[k] * S.dv[k] : acc[2];
  }
  ic_wave_sum4(acc);
  const float A11 = acc[0], A12 = acc[1], A22 = acc[2];
  const float D = A11 * A22 - A12 * A12;
  if (D < 1e-4f) return pr;
  const float invD = (float)(1.0 / (double)D);
  pr.iD_A11 = A11 * invD;
  pr.iD_A12 = A12 * invD;
  pr.iD_A22 = A22 * invD;
  ic_tile_commit<IC_JN, IC_JW, IC_JH>(rj, lane, sh.tj);
  pr.tile.x0 = rj.x0;
  pr.tile.y0 = rj.y0;
  pr.cls = 2;
  return pr;
}

// The iterations of one prepared feature (feature_tracker.cpp:398-503).
template <bool STRICT>
__device__ __forceinline__ IcResult ic_iterate(const vo_level &I1, const IcTaps &tp, const IcPrep &pr, float pt0x,
                                               float pt0y, float pt1x, float pt1y, float scale, int lane,
                                               const IcShared &sh, IcState &S, int &touched, float &last_pux,
                                               float &last_puy, int &n_iter) {
  IcResult res;
  res.cls = pr.cls;
  res.ok = 0;
  res.x = pt1x;
  res.y = pt1y;
  res.err_flag = 0;
  if (pr.cls != 2) return res;
  const float iD_A11 = pr.iD_A11, iD_A12 = pr.iD_A12, iD_A22 = pr.iD_A22;
  const IcTile tile = pr.tile;
  float ax, ay, axay;
  float err_curr = 0.f, err_prev = 1e12f;
  float tx = pt1x - pt0x, ty = pt1y - pt0y;
  int err_flag = 0;
  float sx[IC_K], sy[IC_K];
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    sx[k] = tp.px[k] * scale;
    sy[k] = tp.py[k] * scale;
  }
  // ax = x - floorf(x) lies in [0, 1] for every finite x, so the reference's range test (:407-411)
  // can only be "failed" by a NaN, which its next test reports; and a NaN in ax/ay, in a used tap
  // or in the update reaches dtu + dtv (all other inputs are finite u8-derived values). One
  // wave-uniform test per iteration therefore covers the three throw sites (:412, :438-448, :466);
  // which one it was is sorted out after the loop, off the hot path.
  float b1 = 0.f, b2 = 0.f, e2 = 0.f;
  bool nan_exit = false;
  for (int iter = 0; iter < IC_MAX_ITER; ++iter) {
    const float pux = pt0x + tx, puy = pt0y + ty;
    ic_frac(pux, puy, ax, ay, axay);
    last_pux = pux;
    last_puy = puy;
    ic_sample_I1<STRICT>(I1, tp, sx, sy, pux, puy, ax, ay, axay, S, touched, tile, sh);
    float v[4] = {0.f, 0.f, 0.f, 0.f};  // b1, b2, sum r^2, count
    const unsigned both = S.m & (S.m >> 8);
#pragma unroll
    for (int k = 0; k < IC_K; ++k) {
      // An unused tap adds (+-0) * finite = +-0 to sums that are never -0 (they start at +0 and x + (-x) = +0): the
      // same bits as skipping the addition, one select instead of four.
      const bool use = (both >> k) & 1u;
      const float r = use ? S.I1[k] - S.I0[k] : 0.f;
      v[0] = v[0] + S.du[k] * r;
      v[1] = v[1] + S.dv[k] * r;
      v[2] = v[2] + r * r;
      v[3] = v[3] + (use ? 1.0f : 0.f);
    }
    ic_wave_sum4(v);
    ++n_iter;
    b1 = v[0];
    b2 = v[1];
    e2 = v[2];
    const float dtu = (-iD_A22 * b1 + iD_A12 * b2);
    const float dtv = (iD_A12 * b1 - iD_A11 * b2);
    const float e = sqrtf(e2 / v[3]);
    const float err_rate = fabsf(err_prev - e) / err_prev;
    const float dt_norm = dtu * dtu + dtv * dtv;
    const bool is_nan = (int)isnan(dtu + dtv) | (int)isnan(ax + ay);  // bitwise on purpose: no branch
    const bool conv = (iter > 1) & ((err_rate <= 1e-3f) | (dt_norm <= 1e-4f));
    // every lane holds the same values: make the exit a scalar branch
    const int ex = __builtin_amdgcn_readfirstlane((is_nan ? 2 : 0) | (conv ? 1 : 0));
    if (ex & 2) {
      nan_exit = true;
      break;
    }
    tx += dtu;
    ty += dtv;
    err_curr = e;
    if (ex) break;
    err_prev = e;
  }
  if (nan_exit) {
    // the reference would have stopped before this iteration's sampling (ax/ay NaN), at the
    // tap test (patch NaN) or at the update test; err_curr keeps the previous iteration's value
    err_flag = isnan(ax + ay) ? 1 : ((isnan(b1) || isnan(b2) || isnan(e2)) ? 2 : 4);
  }
  res.cls = 2;
  res.err_flag = err_flag;
  if (!err_flag && !isnan(err_curr) && err_curr <= 30) {
    res.ok = 1;
    res.x = pt0x + tx;
    res.y = pt0y + ty;
  }
  return res;
}

// One point, feature_tracker.cpp:336-503: template at pt0 in I0, refinement in I1 from the prior
// pt1. All 64 lanes call it together with the same arguments; nothing is written to memory.
template <bool STRICT>
__device__ __forceinline__ IcResult ic_point(const vo_level &I0, const vo_level &I1, const IcTaps &tp, float pt0x,
                                             float pt0y, float pt1x, float pt1y, float scale, int lane, IcShared &sh,
                                             IcState &S, int &touched, float &last_pux, float &last_puy, int &n_iter) {
  const IcPrep pr = ic_prepare<STRICT>(I0, I1, tp, pt0x, pt0y, pt1x, pt1y, lane, sh, S, touched);
  return ic_iterate<STRICT>(I1, tp, pr, pt0x, pt0y, pt1x, pt1y, scale, lane, sh, S, touched, last_pux, last_puy, n_iter);
}

// ic_point on array operands: inputs of point `pt` from a.pts0 / a.pts_prior / a.scale, results to
// a.mask / a.pts_track (the prior when the refinement is rejected) / a.flags.
template <bool STRICT>
__device__ __forceinline__ IcResult ic_point_io(const IcArgs &a, const IcTaps &tp, int pt, int lane, IcShared &sh,
                                                IcState &S, int &touched, float &last_pux, float &last_puy,
                                                int &n_iter) {
  const IcResult r = ic_point<STRICT>(a.I0, a.I1, tp, a.pts0[2 * pt], a.pts0[2 * pt + 1], a.pts_prior[2 * pt],
                                      a.pts_prior[2 * pt + 1], a.scale[pt], lane, sh, S, touched, last_pux, last_puy,
                                      n_iter);
  if (lane == 0) {
    if (r.err_flag) atomicOr(a.flags, r.err_flag);
    a.pts_track[2 * pt] = r.x;
    a.pts_track[2 * pt + 1] = r.y;
    a.mask[pt] = (uint8_t)r.ok;
  }
  return r;
}

__device__ __forceinline__ void ic_state_clear(IcState &S) {
#pragma unroll
  for (int k = 0; k < IC_K; ++k) S.I0[k] = S.du[k] = S.dv[k] = S.I1[k] = 0.f;
  S.m = 0;
}

// Stores that another kernel running CONCURRENTLY on another XCD reads (or overwrites) go through to memory: an
// agent-scope atomic store is a plain store with sc1 set — no read-modify-write — and leaves no dirty line behind in
// this XCD's L2 (each XCD has its own; a release fence would write the WHOLE L2 back, once per wavefront).
template <bool COH, typename T>
__device__ __forceinline__ void ic_store(T *p, T v) {
  if (COH)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}

// ---- tap records (who wrote which tap, and what) -----------------------------------
// 264 mask bits: taps 64k..64k+63 (k < 4) go to words 2k, 2k+1; taps 256..263 to word 8.
// `bits`: bit k of the lane = predicate of tap lane + 64 k.
template <bool COH = false>
__device__ __forceinline__ void ic_store_mask(uint32_t *dst, unsigned bits, int lane) {
  // lane w < 9 stores word w: ONE store instruction for the nine words (nine single-lane write-through stores are
  // nine fabric writes in a row on the publishing path of every replayed feature)
  uint32_t wv = 0;
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const unsigned long long m = __ballot((bits >> k) & 1u);
    if (k < 4) {
      wv = lane == 2 * k ? (uint32_t)m : wv;
      wv = lane == 2 * k + 1 ? (uint32_t)(m >> 32) : wv;
    } else {
      wv = lane == 8 ? (uint32_t)m : wv;  // taps 256..263 (lanes 8.. own no tap there)
    }
  }
  if (lane < IC_MW) ic_store<COH>(&dst[lane], wv);
}
// bit of tap lane + 64 k in a 9-word mask
__device__ __forceinline__ bool ic_bit_k(const uint32_t *w, int lane, int k) {
  return (w[k < 4 ? 2 * k + (lane >> 5) : 8] >> (lane & 31)) & 1u;
}

template <bool COH = false>
__device__ __forceinline__ void ic_store_records(const IcArgs &a, int pt, int lane, const IcTaps &tp, const IcState &S,
                                                 int cls) {
  if (!a.recW0) return;
  const bool processed = cls >= 1, iterated = cls == 2;
  ic_store_mask<COH>(a.recW0 + (size_t)pt * IC_MW, processed ? ((S.m >> 16) & tp.on) : 0u, lane);
  ic_store_mask<COH>(a.recW1 + (size_t)pt * IC_MW, iterated ? ((S.m >> 24) & tp.on) : 0u, lane);
  if (lane == 0) ic_store<COH>(&a.ready[pt], (uint8_t)0);
  float *v0 = a.recV0 + (size_t)pt * 3 * IC_NELEM;
  float *v1 = a.recV1 + (size_t)pt * IC_NELEM;
#pragma unroll
  for (int k = 0; k < IC_K; ++k)
    if ((tp.on >> k) & 1u) {
      const int j = lane + 64 * k;
      ic_store<COH>(&v0[j], S.I0[k]);
      ic_store<COH>(&v0[IC_NELEM + j], S.du[k]);
      ic_store<COH>(&v0[2 * IC_NELEM + j], S.dv[k]);
      ic_store<COH>(&v1[j], S.I1[k]);
    }
}

// ---- pass 2a: parallel fixed-point replay of the touched points ------------------------
// The state a touched point P sees is, per tap, the value written by the NEAREST earlier point
// that wrote that tap (template taps: a static function of pts0; I1 taps: depends on that
// point's own trajectory). Measured on forward-driving streams the runs of consecutive touched
// points are ~100 long but the true value-dependency depth is <= 6, so instead of replaying a
// run sequentially every touched point is recomputed in parallel from the current records of its
// predecessors, again and again, until nothing changes any more. That fixed point is unique and
// equals the sequential (reference) result: by induction over the index order, a point whose
// predecessors' records are final computes its final record the next time it looks.
//
// The iteration is asynchronous ("chaotic relaxation"): one launch, each workgroup (one
// wavefront) owns one or more touched points and loops  look -> (recompute, publish)  without
// waiting for the others, so a slow point (30 iterations) only delays the points that really
// depend on it.
//   publish : write the record, release fence, bump the global `version` counter;
//   look    : read `version` (acquire), rebuild the pre-state from the predecessors' records and
//             compare the taps the point can observe with the pre-state of its last run;
//   idle    : a workgroup whose pass changed nothing stores version+1 in its slot and polls;
//             it looks again as soon as `version` moves.
// Reads may race with a concurrent publish; a torn read can only cause an extra recomputation,
// because every publish ends in a version bump that makes every reader look again.
// Termination: version == v before and after seeing every slot at v+1 means every workgroup
// finished a full pass at version v and nobody can publish any more. The workgroups that own
// list entries are co-resident by construction (IC_JGRID = 2 per CU, ~17 KB of LDS each). All
// waits are bounded (IC_SPIN_LIMIT polls, IC_MAX_PASSES passes); on overflow, or when a run
// exceeds IC_MAXRUN, IC_JAC_OVF is raised and ic_strict_kernel replays sequentially.
__device__ __forceinline__ int ic_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ic_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ic_ld8(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ic_st8(uint8_t *p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float ic_ldf(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ic_ldu(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a load of something the pass-1 kernel wrote: that kernel may still be running (CONC) — then only a load that goes
// past this CU's L1 (and whatever this XCD's L2 holds from the previous frame) may be used
template <bool CONC> __device__ __forceinline__ float ic_in(const float *p) { return CONC ? ic_ldf(p) : *p; }
template <bool CONC> __device__ __forceinline__ uint32_t ic_in(const uint32_t *p) { return CONC ? ic_ldu(p) : *p; }
template <bool CONC> __device__ __forceinline__ int ic_in(const uint8_t *p) { return CONC ? ic_ld8(p) : (int)*p; }
template <bool CONC> __device__ __forceinline__ int ic_in(const int *p) { return CONC ? ic_ld(p) : *p; }

// The pass-1 count is polled by the whole replay pool while 1500 wavefronts of the frame kernel add to it: as ONE word
// it is one hot line, and an add then takes tens of microseconds to come back — time the adding wavefront spends in
// its next s_waitcnt (measured: step [5] of every feature 1.5x slower). 64 shards, one per polling lane.
#define IC_P1_SHARDS 64
#define IC_P1_STRIDE 32  // ints: 128 bytes
__device__ __forceinline__ int ic_p1_count(const IcArgs &a, int lane) {
  return wave_sum_i32(ic_ld(&a.p1_word[(lane & (IC_P1_SHARDS - 1)) * IC_P1_STRIDE]));
}

struct IcReplayShared {
  IcShared sh;
  uint32_t w[IC_MAXRUN * IC_MW];  // tap masks of the predecessors (W0 while the statics are built, else W1)
  uint8_t cls[IC_MAXRUN];
  int src[IC_NELEM];              // nearest earlier writer per wanted tap (-1: none)
  int pub[IC_MAXRUN];             // publication counts of the predecessors at the last look
  uint32_t fw[64 * 6];            // ic_find_writers: per (group, word) lane the met taps and five step-number planes
  uint32_t s0[(4 * IC_K + 1) * IC_T];  // the feature's static pre-state (kept here, not in 21 registers, between looks)
};
__device__ __forceinline__ void ic_state_put(uint32_t *dst, const IcState &S, int lane) {
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    dst[(4 * k + 0) * IC_T + lane] = __float_as_uint(S.I0[k]);
    dst[(4 * k + 1) * IC_T + lane] = __float_as_uint(S.du[k]);
    dst[(4 * k + 2) * IC_T + lane] = __float_as_uint(S.dv[k]);
    dst[(4 * k + 3) * IC_T + lane] = __float_as_uint(S.I1[k]);
  }
  dst[4 * IC_K * IC_T + lane] = S.m;
}
__device__ __forceinline__ void ic_state_get(const uint32_t *src, IcState &S, int lane) {
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    S.I0[k] = __uint_as_float(src[(4 * k + 0) * IC_T + lane]);
    S.du[k] = __uint_as_float(src[(4 * k + 1) * IC_T + lane]);
    S.dv[k] = __uint_as_float(src[(4 * k + 2) * IC_T + lane]);
    S.I1[k] = __uint_as_float(src[(4 * k + 3) * IC_T + lane]);
  }
  S.m = src[4 * IC_K * IC_T + lane];
}

// For every tap flagged in `want` (bit k of a lane = tap lane + 64 k): the nearest predecessor
// r in [0, L) (largest r) with rs.cls[r] >= need whose mask rs.w[r] has the tap's bit; result lo + r
// (or -1) in rs.src[tap].
// ONE walk over the predecessors, nearest first, for all wanted taps together. Seven groups of nine lanes share the
// predecessors: group g walks the g-th seventh of them (nearest first), lane w of a group owns word w of the 9-word
// masks (words 2k / 2k+1 = taps 64k.. / 64k+32.., word 8 = taps 256..263) and the wanted taps of that word it has not
// met yet. A tap is met at most once per group, so WHERE it was met is kept bit-sliced: five 32-bit planes hold the
// step number of every bit of the word (plane p collects the hits of the steps whose number has bit p set) — a few
// ORs per step, no loop over the hit bits. (Earlier forms: a ballot loop over the predecessors per tap, ~5 us for the
// ~70 observable taps of a bottom-row feature; a per-bit LDS max inside the walk, whose divergent loops ran once per
// hit bit and group — 3.5-6 us for the same feature, on every link of the deepest dependency chains.) Each lane then
// reads its own taps' answer: the first group, nearest first, that met the tap, and the step from that group's planes.
#define IC_FW_GROUPS 7
__device__ __forceinline__ void ic_find_writers(IcReplayShared &rs, unsigned want, int need, int lo, int L, int lane) {
  unsigned long long b[IC_K];
#pragma unroll
  for (int k = 0; k < IC_K; ++k) b[k] = __ballot((want >> k) & 1u);
  const int g = lane / IC_MW, w = lane - g * IC_MW;
  const bool act = g < IC_FW_GROUPS;
  unsigned ww = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ww = w == 2 * k ? (unsigned)b[k] : ww;
    ww = w == 2 * k + 1 ? (unsigned)(b[k] >> 32) : ww;
  }
  ww = w == 8 ? (unsigned)b[4] : ww;
  ww = act ? ww : 0u;
  const int chunk = (L + IC_FW_GROUPS - 1) / IC_FW_GROUPS;  // <= 28 steps for L <= IC_MAXRUN: five planes
  static_assert((IC_MAXRUN + IC_FW_GROUPS - 1) / IC_FW_GROUPS <= 32, "step number needs five bits");
  const int hi = L - 1 - g * chunk;  // nearest predecessor of this group
  const int lo_r = hi - chunk + 1 > 0 ? hi - chunk + 1 : 0;
  unsigned F = 0, P0 = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0;
  for (int c0 = 0; c0 < chunk; c0 += 4) {
    unsigned m[4], h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = hi - c0 - i;
      const bool ok = act && r >= lo_r && rs.cls[r >= 0 ? r : 0] >= need;
      m[i] = ok ? rs.w[(r >= 0 ? r : 0) * IC_MW + w] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = ww & m[i];
      ww &= ~m[i];
    }
    const unsigned hall = (h[0] | h[1]) | (h[2] | h[3]);
    F |= hall;
    P0 |= h[1] | h[3];
    P1 |= h[2] | h[3];
    if (c0 & 4) P2 |= hall;
    if (c0 & 8) P3 |= hall;
    if (c0 & 16) P4 |= hall;
  }
  if (act) {
    uint32_t *o = rs.fw + lane * 6;
    o[0] = F;
    o[1] = P0;
    o[2] = P1;
    o[3] = P2;
    o[4] = P3;
    o[5] = P4;
  }
  __syncthreads();  // (one wavefront per workgroup: orders the stores above before the reads below)
#pragma unroll
  for (int k = 0; k < IC_K; ++k)
    if ((want >> k) & 1u) {
      const int wj = k < 4 ? 2 * k + (lane >> 5) : 8, bj = lane & 31;
      unsigned f[IC_FW_GROUPS];
#pragma unroll
      for (int g2 = 0; g2 < IC_FW_GROUPS; ++g2) f[g2] = rs.fw[(g2 * IC_MW + wj) * 6];
      int gs = -1;
#pragma unroll
      for (int g2 = IC_FW_GROUPS - 1; g2 >= 0; --g2) gs = ((f[g2] >> bj) & 1u) ? g2 : gs;  // nearest group last: it wins
      int src = -1;
      if (gs >= 0) {
        const uint32_t *o = rs.fw + (gs * IC_MW + wj) * 6;
        const int step = (int)((o[1] >> bj) & 1u) | (int)((o[2] >> bj) & 1u) << 1 | (int)((o[3] >> bj) & 1u) << 2 |
                         (int)((o[4] >> bj) & 1u) << 3 | (int)((o[5] >> bj) & 1u) << 4;
        src = lo + (L - 1 - gs * chunk) - step;
      }
      rs.src[lane + 64 * k] = src;
    }
}

// Body of the replay kernel. Returns the number of workgroups P that own list entries (this
// workgroup handles entries blockIdx.x, blockIdx.x + P, ..), 0 if this workgroup owns none, or -1
// when the sequential fallback was requested (IC_JAC_OVF).
// `after_run(pt, result)` is called after every (re)computation of a feature, once its record is
// published: the frame kernel continues with the feature's next step there instead of waiting for
// the whole relaxation (a later recomputation, rare, calls it again).
//
// CONC: the replay runs NEXT TO the kernel that produces the pass-1 records (the frame kernel), as a pool of
// workgroups that are resident from the start. List entries appear while it runs ({epoch, feature} words), a feature's
// pass-1 data are complete in memory once its epoch stamp is (a.p1e), and a.p1_word tells when every feature has
// passed. A workgroup touches an entry only when the entry, the feature's own stamp and the stamps of the
// predecessors it has to look at are there; until then the entry is `pending` and the workgroup comes back to it.
// Everything the producer wrote is read with agent-scope loads (it was stored write-through). Nothing here makes the
// producer wait, and a waiting workgroup only ever waits for entries with a lower feature index or for the producer,
// so the dependency chains start as soon as their members are through pass 1 instead of after the producer's last
// wavefront. The quiescence vote additionally requires that the producer had finished before the voter's pass began.
template <bool CONC = false, typename AfterRun>
__device__ __forceinline__ int ic_replay(const IcArgs &a, IcReplayShared &rs, int lane, AfterRun after_run) {
  IcShared &sh = rs.sh;
  int n_touched = CONC ? 0 : a.jac[IC_JAC_NT];
  if (!CONC && n_touched == 0) return 0;  // nothing left the image: pass 1 already is the reference result
  const int P = CONC ? (int)gridDim.x : min((int)gridDim.x, n_touched);
  if ((int)blockIdx.x >= P) return 0;
  // one feature per workgroup (the usual case): everything static about it is computed once
  bool single = CONC ? true : n_touched <= P;
  int *const ver = &a.jac[IC_JAC_VER];
  int *const ovf = &a.jac[IC_JAC_OVF];
  int *const slots = a.jac + IC_JAC_SLOTS;
  const IcTaps tp = ic_make_taps(lane);
  const float fw1 = (float)(a.I1.w - 2), fh1 = (float)(a.I1.h - 2);
  int polls = 0;
#ifdef IC_STAMP
  if (lane == 0 && blockIdx.x == 0) a.tlist[IC_DBG_OFF + 31] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#endif
  // statics of the current feature: run bounds, template pre-state, observable I1 taps
  int lo = 0, L = 0;
  bool have_statics = false, skip_pt = false;
  unsigned seen = 0;
  IcPrep prep;  // single mode: the feature is prepared (template, 2x2 inverse, search tile) ahead of its inputs
  prep.cls = 0;
  prep.iD_A11 = prep.iD_A12 = prep.iD_A22 = 0.f;
  prep.tile.x0 = prep.tile.y0 = 0;
  bool mine_ran = false;  // single mode: this workgroup's feature has published (== ready[pt], without the round trip)
  int result = P;
  bool producer_done = !CONC;  // (as of the beginning of the current pass)
  if (CONC) {
    // The control block this kernel uses is reset by the launch that ENDS the previous frame, on the producer's stream;
    // the producer of this frame is ordered behind that launch, this kernel is not. Its first sign of life — a
    // feature through pass 1 — is therefore what this kernel starts on.
    const int base = a.p1_target - a.n;
    while ((int)(ic_p1_count(a, lane) - base) <= 0) {
      if (++polls > IC_SPIN_LIMIT) {
        // The producer has not shown up (kernels serialised across the queues by a tool, most likely). The control
        // block cannot carry the news (it may be reset after this), so a word next to the shards does: the BA launch
        // reports it as an error instead of using pass-1 results as if nothing had to be replayed.
        if (lane == 0) atomicAdd(&a.p1_word[-IC_P1_STRIDE + 2], 1);
        return -1;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  for (int pass = 0;; ++pass) {
    // ---- look ----
    int v = ic_ld(ver);
    if (ic_ld(ovf)) {
      result = -1;
      break;
    }
    if (pass >= IC_MAX_PASSES) {
      if (lane == 0) atomicExch(ovf, 1);
      result = -1;
      break;
    }
    v = __builtin_amdgcn_readfirstlane(v);
    int any_change = 0;
    bool pending = false;  // CONC: an entry of this workgroup could not be looked at yet
    if (CONC) {
      // "done" is read before the list length: a count read behind a finished producer is final
      producer_done = (int)(ic_p1_count(a, lane) - a.p1_target) >= 0;
      __builtin_amdgcn_s_waitcnt(0);
      n_touched = __builtin_amdgcn_readfirstlane(ic_ld(&a.jac[IC_JAC_NT]));
      if (single && n_touched > P) {  // more entries than workgroups: from here on every look rebuilds its statics
        single = false;
        have_statics = false;
      }
    }
    for (int li = blockIdx.x; li < n_touched; li += P) {
      int pt;
      if (CONC) {
        const unsigned long long e = __hip_atomic_load(&a.tl2[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pe = __builtin_amdgcn_readfirstlane((int)(e >> 32));
        pt = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
        // the entry itself, then the feature's own pass-1 data (the list store may overtake them)
        if (pe != a.epoch || __builtin_amdgcn_readfirstlane(ic_ld(&a.p1e[pt])) != a.epoch) {
          pending = true;
          continue;
        }
      } else {
        pt = a.tlist[li];
      }
      __syncthreads();  // LDS of the previous list entry is free
      const float p0x = a.pts0[2 * pt], p0y = a.pts0[2 * pt + 1];
      if (!single || !have_statics) {
        skip_pt = false;
        // nearest clean (untouched, iterated) predecessor: 64 candidates at a time
        int dist = -1;
        bool early = false;  // CONC: a predecessor nearer than the nearest clean one is not through pass 1 yet
        for (int c0 = 0; c0 < IC_CAND; c0 += 64) {
          const int q = pt - 1 - c0 - lane;
          const int qq = q >= 0 ? q : 0;
          const bool there = !CONC || ic_ld(&a.p1e[qq]) == a.epoch;
          const bool is_clean = q >= 0 && there && ic_in<CONC>(&a.cls[qq]) == 2 && !ic_in<CONC>(&a.touched[qq]);
          const unsigned long long bal = __ballot(is_clean);
          const unsigned long long nb = __ballot(q >= 0 && !there);
          if (bal) {
            const int f = __ffsll((long long)bal) - 1;
            dist = c0 + f;
            if (nb & ((1ull << f) - 1ull)) early = true;
            break;
          }
          if (nb) {
            early = true;
            break;
          }
          if (pt - 1 - c0 - 63 <= 0) break;  // ran past index 0
        }
        if (CONC && early) {
          pending = true;
          continue;
        }
        have_statics = true;
        if (dist < 0 && pt > IC_CAND) skip_pt = true;  // no clean point among the candidates
        lo = dist < 0 ? 0 : pt - 1 - dist;
        L = pt - lo;  // predecessors lo .. pt-1
        if (L > IC_MAXRUN) skip_pt = true;
        if (skip_pt) {
          if (lane == 0) atomicExch(ovf, 1);
          continue;
        }
        // template pre-state: only taps the feature's own template evaluation does not write
        unsigned tseen = 0;
#pragma unroll
        for (int k = 0; k < IC_K; ++k) {
          const int u0 = (int)(p0x + tp.px[k]), v0 = (int)(p0y + tp.py[k]);
          const bool valid = !(u0 < 1 || u0 >= a.I0.w - 2 || v0 < 1 || v0 >= a.I0.h - 2);
          if (((tp.on >> k) & 1u) && !valid) tseen |= 1u << k;
        }
        if (!CONC) __threadfence();  // (acquire for pass 0: pass-1 records come from the previous launch anyway)
        for (int i = lane; i < L * IC_MW; i += IC_T) rs.w[i] = ic_in<CONC>(&a.recW0[(size_t)lo * IC_MW + i]);
        for (int i = lane; i < L; i += IC_T) rs.cls[i] = (uint8_t)ic_in<CONC>(&a.cls[lo + i]);
        __syncthreads();
        ic_find_writers(rs, tseen, 1, lo, L, lane);
        __syncthreads();
        IcState S0;
        ic_state_clear(S0);
#pragma unroll
        for (int k = 0; k < IC_K; ++k)
          if ((tseen >> k) & 1u) {
            const int j = lane + 64 * k;
            const int src = rs.src[j];
            if (src >= 0) {
              const float *v0 = a.recV0 + (size_t)src * 3 * IC_NELEM;
              S0.I0[k] = ic_in<CONC>(&v0[j]);
              S0.du[k] = ic_in<CONC>(&v0[IC_NELEM + j]);
              S0.dv[k] = ic_in<CONC>(&v0[2 * IC_NELEM + j]);
              S0.m |= 1u << k;
            }
          }
        // Only taps that are outside the image at the feature's FIRST I1 evaluation (the prior
        // position: static) can show their pre-state to it; every other tap is overwritten by that
        // evaluation before anything reads it.
        const float pux = p0x + (ic_in<CONC>(&a.pts_prior[2 * pt]) - p0x), puy = p0y + (ic_in<CONC>(&a.pts_prior[2 * pt + 1]) - p0y);
        const float sc = ic_in<CONC>(&a.scale[pt]);
        seen = 0;
#pragma unroll
        for (int k = 0; k < IC_K; ++k) {
          const float uc = pux + tp.px[k] * sc, vc = puy + tp.py[k] * sc;
          if (((tp.on >> k) & 1u) && (uc < 1 || uc >= fw1 || vc < 1 || vc >= fh1)) seen |= 1u << k;
        }
        __syncthreads();
        if (single) {
          // S0 becomes the state after the feature's own template evaluation (static, like its inputs)
          int dummy_t = 0;
          prep = ic_prepare<true>(a.I0, a.I1, tp, p0x, p0y, ic_in<CONC>(&a.pts_prior[2 * pt]),
                                  ic_in<CONC>(&a.pts_prior[2 * pt + 1]), lane, sh, S0, dummy_t);
        }
        ic_state_put(rs.s0, S0, lane);
      } else if (skip_pt) {
        continue;
      }
      // words of the 9-word tap masks that hold observable taps (usually 2 or 3): only those are staged
      unsigned wordmask = 0;
      uint32_t seenw[IC_MW];  // the observable taps as mask words (wave-uniform)
#pragma unroll
      for (int k = 0; k < IC_K; ++k) {
        const unsigned long long b = __ballot((seen >> k) & 1u);
        if (k < 4) {
          seenw[2 * k] = (uint32_t)b;
          seenw[2 * k + 1] = (uint32_t)(b >> 32);
          if ((uint32_t)b) wordmask |= 1u << (2 * k);
          if ((uint32_t)(b >> 32)) wordmask |= 1u << (2 * k + 1);
        } else {
          seenw[8] = (uint32_t)b;
          if (b) wordmask |= 1u << 8;
        }
      }
      IcState S;  // (filled from rs.s0 when the run is about to start)
      bool give_up = false, unchanged = false;
#ifdef IC_STAMP
      int dbg_t_att = 0, dbg_t_fw = 0, dbg_n_att = 0, dbg_t_ld = 0;
#endif
      float pv[IC_K] = {0.f, 0.f, 0.f, 0.f, 0.f};  // values of the predicted writers (see below)
      int psrc[IC_K] = {-1, -1, -1, -1, -1};        // and who they are
      bool same_writers = false;
      for (int attempt = 0;; ++attempt) {
#ifdef IC_STAMP
        dbg_t_att = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
        ++dbg_n_att;
#endif
        // publication counts of the predecessors, read BEFORE their records (a count is bumped after
        // the record's release): unchanged counts => unchanged records => nothing to do
        int pc[(IC_MAXRUN + 63) / 64];
        // a look that follows a wait (attempt > 0) knows that something moved: it skips the counts — one round trip
        // less on the critical path of the chain; the stale counts only make a later idle look take place once more
        const bool counts = attempt == 0;
        bool moved = pass == 0 || !single || attempt > 0 || (single ? !mine_ran : !ic_ld8(&a.ready[pt]));
        if (counts) {
#pragma unroll
          for (int q = 0; q < (IC_MAXRUN + 63) / 64; ++q) {
            const int i = lane + 64 * q;
            pc[q] = i < L ? ic_ld(&a.pubc[lo + i]) : 0;
            if (i < L && pc[q] != rs.pub[i]) moved = true;
          }
        }
        if (!__any(moved)) {
          unchanged = true;
          break;
        }
        // (the records behind the counts just read are loaded with agent-scope loads below: they go past this XCD's
        // L2 lines that may be stale, which an acquire fence would have to invalidate wholesale, once per look)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (counts) {
#pragma unroll
          for (int q = 0; q < (IC_MAXRUN + 63) / 64; ++q) {
            const int i = lane + 64 * q;
            if (i < L) rs.pub[i] = pc[q];
          }
        }
        // A look that follows the wait for this feature's writers (attempt > 0) expects to find what the previous look
        // found, plus their new values: the values of the writers found THEN are loaded together with the masks, and
        // when no mask (and no class) of a predecessor has changed, the writers are the same, they are all ready (the
        // wait just saw that) and those values are the pre-state — three round trips less between a writer's
        // publication and the start of this feature's run, on every link of every chain.
        const bool predicted = attempt > 0;
        same_writers = false;
        {
          // every load of the look in flight together: one exposure of the memory latency (a loop that loads and
          // stores word by word pays it once per word and 64 predecessors — six round trips for a bottom-row feature)
          uint32_t mw[(IC_MAXRUN + 63) / 64][IC_MW];
          int cl[(IC_MAXRUN + 63) / 64];
#pragma unroll
          for (int q = 0; q < (IC_MAXRUN + 63) / 64; ++q) {
            const int i = lane + 64 * q;
            const bool in = i < L;
            cl[q] = in ? ic_ld8(&a.cls[lo + (in ? i : 0)]) : 0;
#pragma unroll
            for (int w = 0; w < IC_MW; ++w)
              mw[q][w] = (in && ((wordmask >> w) & 1u))
                             ? __hip_atomic_load(&a.recW1[(size_t)(lo + (in ? i : 0)) * IC_MW + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : 0u;
          }
#pragma unroll
          for (int k = 0; k < IC_K; ++k) {
            pv[k] = 0.f;
            psrc[k] = -1;
            if (predicted && ((seen >> k) & 1u)) {
              const int j = lane + 64 * k;
              const int src = rs.src[j];
              psrc[k] = src;
              if (src >= 0) pv[k] = __hip_atomic_load(&a.recV1[(size_t)src * IC_NELEM + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
#ifdef IC_STAMP
          __builtin_amdgcn_s_waitcnt(0);
          dbg_t_ld = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#endif
          bool chg = !predicted;
#pragma unroll
          for (int q = 0; q < (IC_MAXRUN + 63) / 64; ++q) {
            const int i = lane + 64 * q;
            if (i < L) {
              if (predicted && rs.cls[i] != (uint8_t)cl[q]) chg = true;
              rs.cls[i] = (uint8_t)cl[q];
#pragma unroll
              for (int w = 0; w < IC_MW; ++w)
                if ((wordmask >> w) & 1u) {
                  // (only the observable taps' bits decide who this feature's writers are)
                  if (predicted && ((rs.w[i * IC_MW + w] ^ mw[q][w]) & seenw[w])) chg = true;
                  rs.w[i * IC_MW + w] = mw[q][w];
                }
            }
          }
          same_writers = !__any(chg);
        }
#ifdef IC_STAMP
        if (same_writers) {
          dbg_t_fw = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
          if (lane == 0) atomicAdd(&a.tlist[IC_DBG_OFF + 34], 1);
        }
#endif
        if (same_writers) break;  // rs.src, and pv, stand
        __syncthreads();
        ic_find_writers(rs, seen, 2, lo, L, lane);
        __syncthreads();
        if (predicted) {
          // masks changed, writers did not: the same shortcut
          bool other = false;
#pragma unroll
          for (int k = 0; k < IC_K; ++k)
            if (((seen >> k) & 1u) && rs.src[lane + 64 * k] != psrc[k]) other = true;
          if (!__any(other)) {
            same_writers = true;
            break;
          }
        }
#ifdef IC_STAMP
        dbg_t_fw = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#endif
        // Dataflow gate: while a touched feature that this one observes has not produced its first
        // strict-state result, a run here would only compute from pass-1 data that is about to
        // change, and would keep this wavefront busy when the real input arrives. Poll exactly those
        // features' flags (cheap), then look again: their publication may have changed who writes what.
        bool wait = false;
#pragma unroll
        for (int k = 0; k < IC_K; ++k)
          if ((seen >> k) & 1u) {
            const int src = rs.src[lane + 64 * k];
            const int sq = src >= 0 ? src : 0;
            const int tq = ic_in<CONC>(&a.touched[sq]), rq = ic_ld8(&a.ready[sq]);  // both loads in flight together
            if (src >= 0 && tq && !rq) wait = true;
          }
        if (!__any(wait)) break;
        if (CONC && !single) {  // this workgroup may own the awaited writer itself: come back in the next pass
          give_up = true;
          pending = true;
          break;
        }
        if (attempt >= 64) {
          give_up = true;  // fall back to the version-driven wait
          break;
        }
        for (;;) {
          bool w2 = false;
#pragma unroll
          for (int k = 0; k < IC_K; ++k)
            if ((seen >> k) & 1u) {
              const int src = rs.src[lane + 64 * k];
              const int sq = src >= 0 ? src : 0;
              const int tq = ic_in<CONC>(&a.touched[sq]), rq = ic_ld8(&a.ready[sq]);
              if (src >= 0 && tq && !rq) w2 = true;
            }
          if (!__any(w2)) break;
          if (__builtin_amdgcn_readfirstlane(ic_ld(ovf)) || ++polls > IC_SPIN_LIMIT) {
            give_up = true;
            break;
          }
          __builtin_amdgcn_s_sleep(IC_WAIT_SLEEP);  // a hundred waiting wavefronts must not flood the fabric
        }
        if (give_up) break;
      }
      if (unchanged || give_up) continue;
      ic_state_get(rs.s0, S, lane);
#pragma unroll
      for (int k = 0; k < IC_K; ++k)
        if ((seen >> k) & 1u) {
          const int j = lane + 64 * k;
          const int src = rs.src[j];
          if (src >= 0) {
            S.I1[k] = same_writers ? pv[k]
                                   : __hip_atomic_load(&a.recV1[(size_t)src * IC_NELEM + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.m |= 0x100u << k;
          }
        }
      // skip when the observable I1 pre-state is exactly the one this feature last ran with
      {
        float *p1 = a.pre1 + (size_t)pt * IC_NELEM;
        uint32_t *pm = a.preM + (size_t)pt * IC_MW;
        // the first strict-state run is unconditional (single mode knows without asking memory: every global round
        // trip between "my writers are final" and "my record is published" is on the critical path of a chain)
        int diff = single ? (mine_ran ? 0 : 1) : __builtin_amdgcn_readfirstlane(!ic_ld8(&a.ready[pt]));
        if (!diff) {
#pragma unroll
          for (int k = 0; k < IC_K; ++k)
            if ((seen >> k) & 1u) {
              const bool b0 = (S.m >> (8 + k)) & 1u;
              const int j = lane + 64 * k;
              if (b0 != ic_bit_k(pm, lane, k) || (b0 && __float_as_uint(p1[j]) != __float_as_uint(S.I1[k]))) diff = 1;
            }
        }
        if (!__any(diff)) continue;
#ifdef IC_STAMP
        if (lane == 0) atomicAdd(&a.tlist[IC_DBG_OFF + 32], 1);
#endif
#pragma unroll
        for (int k = 0; k < IC_K; ++k)
          if ((tp.on >> k) & 1u) p1[lane + 64 * k] = S.I1[k];
        ic_store_mask(pm, (S.m >> 8) & tp.on, lane);
      }
      int dummy = 0, n_iter = 0;
      float lx = 0.f, ly = 0.f;
#ifdef IC_STAMP
      if (lane == 0 && li < 256) a.tlist[IC_DBG_OFF + 64 + 8 * li + 0] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#endif
      IcResult res_pt;
      {
        const float q1x = ic_in<CONC>(&a.pts_prior[2 * pt]), q1y = ic_in<CONC>(&a.pts_prior[2 * pt + 1]);
        IcPrep pr = prep;
        if (!single) pr = ic_prepare<true>(a.I0, a.I1, tp, p0x, p0y, q1x, q1y, lane, sh, S, dummy);
        res_pt = ic_iterate<true>(a.I1, tp, pr, p0x, p0y, q1x, q1y, ic_in<CONC>(&a.scale[pt]), lane, sh, S, dummy, lx, ly, n_iter);
#ifdef IC_REPLAY_EXTRA_SLEEP  // measurement build: every replayed iteration made LONGER by s_sleep(IC_REPLAY_EXTRA_SLEEP) — 64 cycles
        // each — to read off how much of an iteration's cost is on the frame's critical path
        for (int q = 0; q < n_iter; ++q) __builtin_amdgcn_s_sleep(IC_REPLAY_EXTRA_SLEEP);
#endif
        if (lane == 0) {
          if (res_pt.err_flag) atomicOr(a.flags, res_pt.err_flag);
          a.pts_track[2 * pt] = res_pt.x;
          a.pts_track[2 * pt + 1] = res_pt.y;
          a.mask[pt] = (uint8_t)res_pt.ok;
        }
      }
#ifdef IC_STAMP
      if (lane == 0 && li < 256) {
        a.tlist[IC_DBG_OFF + 64 + 8 * li + 4] = dbg_t_att;
        a.tlist[IC_DBG_OFF + 64 + 8 * li + 5] = dbg_t_fw;
        a.tlist[IC_DBG_OFF + 64 + 8 * li + 7] = dbg_t_ld;
        (void)dbg_n_att;
        a.tlist[IC_DBG_OFF + 64 + 8 * li + 1] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
        a.tlist[IC_DBG_OFF + 64 + 8 * li + 2] = n_iter;
        a.tlist[IC_DBG_OFF + 64 + 8 * li + 3] = pt;
      }
#endif
      const int cls = res_pt.cls;
      // own I1 writes of this run of the feature vs the stored record
      const bool iterated = cls == 2;
      const unsigned o = iterated ? ((S.m >> 24) & tp.on) : 0u;
      uint32_t *w1 = a.recW1 + (size_t)pt * IC_MW;
      float *v1 = a.recV1 + (size_t)pt * IC_NELEM;
      const bool first_run = single ? !mine_ran : !ic_ld8(&a.ready[pt]);
      int changed = 0;
      if (!first_run) {  // (a first run publishes whatever it found: no need to read the pass-1 record back)
        // (the record this workgroup published itself, written through: read back the same way)
        changed = ic_ld8(&a.cls[pt]) != cls;
#pragma unroll
        for (int k = 0; k < IC_K; ++k)
          if ((tp.on >> k) & 1u) {
            const bool ok = (o >> k) & 1u;
            const bool was = (ic_ldu(&w1[k < 4 ? 2 * k + (lane >> 5) : 8]) >> (lane & 31)) & 1u;
            if (ok != was || (ok && __float_as_uint(ic_ldf(&v1[lane + 64 * k])) != __float_as_uint(S.I1[k]))) changed = 1;
          }
      }
      if (first_run || __any(changed)) {
        // publish: record (write-through stores: a release fence would write this XCD's whole L2 back, once per
        // hand-over of every chain), wait for the stores, count, version
#pragma unroll
        for (int k = 0; k < IC_K; ++k)
          if ((tp.on >> k) & 1u) ic_store<true>(&v1[lane + 64 * k], S.I1[k]);
        ic_store_mask<true>(w1, o, lane);
        if (lane == 0) ic_store<true>(&a.cls[pt], (uint8_t)cls);
        __builtin_amdgcn_s_waitcnt(0);
        mine_ran = true;
        if (lane == 0) {
          // count, ready flag and version are issued back to back: each becomes visible after the record (the wait
          // above), and their order among themselves does not matter — a reader that sees the flag or the version
          // before the count only looks once more than it had to
          __hip_atomic_fetch_add(&a.pubc[pt], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ic_st8(&a.ready[pt], 1);
          __hip_atomic_fetch_add(ver, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef IC_STAMP
          atomicAdd(&a.tlist[IC_DBG_OFF + 33], 1);
          if (li < 256) a.tlist[IC_DBG_OFF + 64 + 8 * li + 6] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#endif
        }
        any_change = 1;
      }
      after_run(pt, res_pt);
    }  // touched list
    if (any_change) continue;  // look again at once: the version moved at least by our own publish

    // ---- idle at version v: vote, then poll until the version moves or everybody is idle ----
    // (CONC: a vote says "nothing left for me, and the producer had finished before I looked"; without that the
    // workgroup only waits for news: a new version, a longer list, the producer's end)
    const bool may_vote = !CONC || (producer_done && !pending);
    if (may_vote && lane == 0) ic_st(&slots[blockIdx.x], v + 1);
    int res;
    for (;;) {
      const int cur = __builtin_amdgcn_readfirstlane(ic_ld(ver));
      if (__builtin_amdgcn_readfirstlane(ic_ld(ovf))) {
        res = 2;
        break;
      }
      if (cur != v) {
        res = 1;
        break;
      }
      if (CONC && !may_vote) {
        const bool done_now = (int)(ic_p1_count(a, lane) - a.p1_target) >= 0;
        const int nt_now = __builtin_amdgcn_readfirstlane(ic_ld(&a.jac[IC_JAC_NT]));
        if (pending || nt_now != n_touched || done_now != producer_done) {
          // (a pending entry is re-examined at the polling rate: its stamps are the only news it waits for)
          __builtin_amdgcn_s_sleep(IC_WAIT_SLEEP);
          if (++polls > IC_SPIN_LIMIT) {
            if (lane == 0) atomicExch(ovf, 1);
            res = 2;
            break;
          }
          res = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(IC_WAIT_SLEEP);
        if (++polls > IC_SPIN_LIMIT) {
          if (lane == 0) atomicExch(ovf, 1);
          res = 2;
          break;
        }
        continue;
      }
      if ((polls & 3) == 3) {  // the termination test is the expensive part of a poll: every 4th
        bool ok = true;
        for (int k = lane; k < P; k += 64) ok = ok && (ic_ld(&slots[k]) == v + 1);
        const bool all_idle = __all(ok);
        if (all_idle && __builtin_amdgcn_readfirstlane(ic_ld(ver)) == v) {
          res = 0;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(2);
      if (++polls > IC_SPIN_LIMIT) {
        if (lane == 0) atomicExch(ovf, 1);
        res = 2;
        break;
      }
    }
    if (res == 2) result = -1;
    if (res != 1) break;  // 0: quiescent -> done ; 2: sequential fallback requested
  }  // passes
#ifdef IC_STAMP
  if (lane == 0 && blockIdx.x == 0) a.tlist[IC_DBG_OFF + 0] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
#endif
  return result;
}

// ---- sequential replay of the run of features that starts at touched feature `pt` (the fallback of
// ic_replay): one wavefront walks the run, the carried tap state lives in registers. Returns at once
// unless `pt` is the first touched feature of its run. `after_point(p, result)` is called for every
// touched feature replayed.
template <typename AfterPoint>
__device__ __forceinline__ void ic_strict_run(const IcArgs &a, IcShared &sh, int pt, int n, int lane,
                                              AfterPoint after_point) {
  if (!a.touched[pt]) return;
  // head test: walk back over skipped / template-only untouched points
  int start = 0, clean = -1;
  for (int j = pt - 1; j >= 0; --j) {
    const int cj = a.cls[j];
    if (cj == 0) continue;
    if (a.touched[j]) return;  // an earlier touched point owns this run
    if (cj == 2) {
      clean = j;
      break;
    }
  }
  start = clean + 1;
  IcState S;
  ic_state_clear(S);
  const IcTaps tp = ic_make_taps(lane);
  int dummy = 0, n_iter = 0;
  if (clean >= 0) {
    // state left behind by an untouched, iterated point: its template and its last I1 patch
    float ax, ay, axay;
    const float cx = a.pts0[2 * clean], cy = a.pts0[2 * clean + 1];
    ic_frac(cx, cy, ax, ay, axay);
    IcTRegs rt;
    IcJRegs rj;
    const float pux = a.last_pu[2 * clean], puy = a.last_pu[2 * clean + 1];
    ic_template_fetch(a.I0, cx, cy, lane, rt);
    ic_I1_fetch(a.I1, pux, puy, lane, rj);
    ic_tile_commit<IC_TN, IC_TW, IC_TH>(rt, lane, sh.tt);
    const IcTile tt = {rt.x0, rt.y0};
    ic_template<true>(a.I0, tp, cx, cy, ax, ay, axay, tt, sh, S, dummy);
    ic_frac(pux, puy, ax, ay, axay);
    ic_tile_commit<IC_JN, IC_JW, IC_JH>(rj, lane, sh.tj);
    const IcTile tile = {rj.x0, rj.y0};
    float sx[IC_K], sy[IC_K];
    const float sc = a.scale[clean];
#pragma unroll
    for (int k = 0; k < IC_K; ++k) {
      sx[k] = tp.px[k] * sc;
      sy[k] = tp.py[k] * sc;
    }
    ic_sample_I1<true>(a.I1, tp, sx, sy, pux, puy, ax, ay, axay, S, dummy, tile, sh);
  }
  for (int p = start; p < n; ++p) {
    const int cp = a.cls[p];
    if (cp == 0) continue;
    const int is_touched = a.touched[p];
    if (!is_touched) {
      if (cp == 2) break;  // next clean point: end of the run
      // untouched, failed the determinant test: it only rewrote the template state
      float ax, ay, axay;
      const float cx = a.pts0[2 * p], cy = a.pts0[2 * p + 1];
      ic_frac(cx, cy, ax, ay, axay);
      IcTRegs rt;
      ic_template_fetch(a.I0, cx, cy, lane, rt);
      ic_tile_commit<IC_TN, IC_TW, IC_TH>(rt, lane, sh.tt);
      const IcTile tt = {rt.x0, rt.y0};
      ic_template<true>(a.I0, tp, cx, cy, ax, ay, axay, tt, sh, S, dummy);
      continue;
    }
    float lx, ly;
    const IcResult r = ic_point_io<true>(a, tp, p, lane, sh, S, dummy, lx, ly, n_iter);
    after_point(p, r);
  }
}

