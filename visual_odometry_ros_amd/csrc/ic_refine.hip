// ic_refine.hip — scale-compensated inverse-compositional patch refinement,
// one gfx950 wavefront per feature point.
//
// Replaces FeatureTracker::trackWithScale (core/visual_odometry/feature_tracker.cpp:236-504)
// with its samplers image_processing::interpImage3SameRatio / interpImageSameRatio
// (core/util/image_processing.cpp:268-331, :79-118), the two img.convertTo(CV_32FC1)
// calls (:297-299) and the cv::Sobel(CV_32FC1, ksize 3) pair the drivers run first
// (core/visual_odometry/stereo_vo/stereo_vo.cpp:549-552, mono_vo.cpp:779-782): the
// f32 image planes and both derivative planes are never materialised — the u8
// level-0 planes already resident for the KLT tracker are read, Sobel is
// evaluated at the four bilinear corners from a 27-row u8 LDS tile.
//
// Mapping: 264 checkerboard taps of the 23x23 window; lane l owns taps
// j = l, l+64, l+128, l+192 (+256 for l < 8). The template (I0, du0, dv0 per tap)
// stays in registers for all iterations. Float sums use the canonical order
// "lane partial over its taps ascending, then balanced tree over the 64 lanes"
// (oracle VO_SUM_TREE).
//
// Border semantics, two kernels:
//  * ic_refine_kernel (all points in parallel): a tap whose footprint leaves the
//    valid region is excluded from that evaluation (oracle VO_IC_BORDER_MASKED);
//    such points are reported in `touched`. Every point that never touches the
//    border gets exactly the reference result.
//  * ic_strict_kernel (optional second pass): the reference allocates its tap
//    value / mask vectors once per call and never resets them
//    (feature_tracker.cpp:324-333, image_processing.cpp:88-89,276-279), so an
//    out-of-image tap keeps the value and mask bit of the most recent earlier
//    evaluation in which it was inside. Only touched points
//    see that state, and the state after an untouched, iterated ("clean") point
//    is fully determined by that point alone. So each run of points between two
//    clean points that contains a touched point is replayed sequentially by one
//    wavefront, the carried state living in registers (5 taps x 4 values per
//    lane); runs replay in parallel.
#include "vo_internal.hpp"
#include "vo_kernels.hpp"

#define IC_HALF 11
#define IC_NELEM 264
#define IC_K 5
#define IC_TW 8    // tile dwords per row (32 bytes)
#define IC_TH 27   // tile rows

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
};

struct IcState {
  float I0[IC_K], du[IC_K], dv[IC_K], I1[IC_K];
  bool m0[IC_K], m1[IC_K];
};

__device__ __forceinline__ float ic_bilin(float I1, float I2, float I3, float I4, float ax, float ay, float axay) {
  return ((axay * (((I1 - I2) - I3) + I4) + ax * (-I1 + I2)) + ay * (-I1 + I3)) + I1;
}

__device__ __forceinline__ void ic_tap_offset(int j, float &px, float &py) {
  // feature_tracker.cpp:308-320: rows v = 0..22; even rows hold u = 1,3,..,21 (11 taps),
  // odd rows u = 0,2,..,22 (12 taps)
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

__device__ __forceinline__ int ic_safe_int(float v) {
  return (int)fminf(fmaxf(v, -1.0e6f), 1.0e6f);
}

// interpImage3SameRatio on the taps of this lane: writes state where the tap is valid.
// STRICT: mask bits are sticky (never reset); otherwise they are this evaluation's validity.
template <bool STRICT>
__device__ __forceinline__ void ic_template(const vo_level &L0, float pt0x, float pt0y, float ax, float ay,
                                            float axay, int lane, uint32_t *s_t, IcState &S, int &touched) {
  const int W = L0.w, H = L0.h;
  const int cx = ic_safe_int(pt0x), cy = ic_safe_int(pt0y);
  int tox = (cx - 13) & ~3;
  int toy = cy - 12;  // rows cy-12 .. cy+14 cover every valid tap's 4x4 neighbourhood
  tox = max(-VO_PAD, min(tox, ((W + VO_PAD - IC_TW * 4) & ~3)));
  toy = max(-VO_PAD, min(toy, H + VO_PAD - IC_TH));
  {
    const uint8_t *g = L0.origin() + (ptrdiff_t)toy * L0.stride + tox;
    __syncthreads();
    for (int i = lane; i < IC_TH * IC_TW; i += 64) {
      const int r = i / IC_TW, cdw = i - r * IC_TW;
      s_t[i] = *(const uint32_t *)(g + (ptrdiff_t)r * L0.stride + cdw * 4);
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const int j = lane + 64 * k;
    const bool on = j < IC_NELEM;
    float px = 0.f, py = 0.f;
    if (on) ic_tap_offset(j, px, py);
    const float uc = pt0x + px, vc = pt0y + py;
    const int u0 = (int)uc, v0 = (int)vc;
    const bool valid = on && !(u0 < 1 || u0 >= W - 2 || v0 < 1 || v0 >= H - 2);
    if (on && !valid) touched = 1;
    if (!STRICT) S.m0[k] = valid;
    if (valid) {
      // 4x4 neighbourhood (u0-1..u0+2, v0-1..v0+2)
      const int bx = (u0 - 1) - tox, by = (v0 - 1) - toy;
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
      S.I0[k] = ic_bilin(Iv[0][0], Iv[0][1], Iv[1][0], Iv[1][1], ax, ay, axay);
      S.du[k] = ic_bilin(du[0][0], du[0][1], du[1][0], du[1][1], ax, ay, axay);
      S.dv[k] = ic_bilin(dv[0][0], dv[0][1], dv[1][0], dv[1][1], ax, ay, axay);
      S.m0[k] = true;
    }
  }
}

// interpImageSameRatio (float compares) on the taps of this lane
template <bool STRICT>
__device__ __forceinline__ void ic_sample_I1(const vo_level &L1, float pux, float puy, float scale, float ax,
                                             float ay, float axay, int lane, IcState &S, int &touched) {
  const int W = L1.w, H = L1.h;
  const uint8_t *o1 = L1.origin();
  const int st1 = L1.stride;
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    const int j = lane + 64 * k;
    const bool on = j < IC_NELEM;
    float px = 0.f, py = 0.f;
    if (on) ic_tap_offset(j, px, py);
    const float uc = pux + px * scale, vc = puy + py * scale;
    const bool valid = on && !(uc < 1 || uc >= (float)(W - 2) || vc < 1 || vc >= (float)(H - 2));
    if (on && !valid) touched = 1;
    if (!STRICT) S.m1[k] = valid;
    if (valid) {
      const int u0 = (int)uc, v0 = (int)vc;
      const uint8_t *p = o1 + (ptrdiff_t)v0 * st1 + u0;
      S.I1[k] = ic_bilin((float)p[0], (float)p[1], (float)p[st1], (float)p[st1 + 1], ax, ay, axay);
      S.m1[k] = true;
    }
  }
}

__device__ __forceinline__ void ic_frac(float x, float y, float &ax, float &ay, float &axay) {
  ax = (float)((double)x - floor((double)x));
  ay = (float)((double)y - floor((double)y));
  axay = ax * ay;
}

// One point, feature_tracker.cpp:336-503. Returns cls (1 template only, 2 iterated).
template <bool STRICT>
__device__ int ic_point(const IcArgs &a, int pt, int lane, uint32_t *s_t, IcState &S, int &touched,
                        float &last_pux, float &last_puy) {
  const float pt0x = a.pts0[2 * pt], pt0y = a.pts0[2 * pt + 1];
  const float pt1x = a.pts_prior[2 * pt], pt1y = a.pts_prior[2 * pt + 1];
  const float scale = a.scale[pt];
  float ax, ay, axay;
  ic_frac(pt0x, pt0y, ax, ay, axay);
  if (ax < 0 || ax > 1 || ay < 0 || ay > 1) {
    if (lane == 0) a.mask[pt] = 0;
    return 1;
  }
  ic_template<STRICT>(a.I0, pt0x, pt0y, ax, ay, axay, lane, s_t, S, touched);
  float pA11 = 0.f, pA12 = 0.f, pA22 = 0.f;
#pragma unroll
  for (int k = 0; k < IC_K; ++k)
    if (S.m0[k]) {
      pA11 += S.du[k] * S.du[k];
      pA12 += S.du[k] * S.dv[k];
      pA22 += S.dv[k] * S.dv[k];
    }
  const float A11 = wave_sum_f32(pA11);
  const float A12 = wave_sum_f32(pA12);
  const float A22 = wave_sum_f32(pA22);
  const float D = A11 * A22 - A12 * A12;
  if (D < 1e-4f) {
    if (lane == 0) a.mask[pt] = 0;
    return 1;
  }
  const float invD = (float)(1.0 / (double)D);
  const float iD_A11 = A11 * invD, iD_A12 = A12 * invD, iD_A22 = A22 * invD;

  float err_curr = 0.f, err_prev = 1e12f;
  float tx = pt1x - pt0x, ty = pt1y - pt0y;
  int err_flag = 0;
  for (int iter = 0; iter < 30; ++iter) {
    const float pux = pt0x + tx, puy = pt0y + ty;
    ic_frac(pux, puy, ax, ay, axay);
    if (ax < 0 || ax > 1 || ay < 0 || ay > 1) break;  // :407-411 (mask is overwritten below, as in the reference)
    if (isnan(ax + ay)) {
      err_flag |= 1;
      break;
    }
    last_pux = pux;
    last_puy = puy;
    ic_sample_I1<STRICT>(a.I1, pux, puy, scale, ax, ay, axay, lane, S, touched);
    float pb1 = 0.f, pb2 = 0.f, pe = 0.f;
    int cnt = 0, nanp = 0;
#pragma unroll
    for (int k = 0; k < IC_K; ++k)
      if (S.m0[k] && S.m1[k]) {
        if (isnan(S.I0[k]) || isnan(S.I1[k]) || isnan(S.du[k]) || isnan(S.dv[k])) nanp = 1;
        const float r = S.I1[k] - S.I0[k];
        pb1 += S.du[k] * r;
        pb2 += S.dv[k] * r;
        pe += r * r;
        ++cnt;
      }
    if (__any(nanp)) {
      err_flag |= 2;
      break;
    }
    const float b1 = wave_sum_f32(pb1);
    const float b2 = wave_sum_f32(pb2);
    err_curr = wave_sum_f32(pe);
    const int cnt_valid = wave_sum_i32(cnt);
    const float dtu = (-iD_A22 * b1 + iD_A12 * b2);
    const float dtv = (iD_A12 * b1 - iD_A11 * b2);
    if (isnan(dtu + dtv)) {
      err_flag |= 4;
      break;
    }
    tx += dtu;
    ty += dtv;
    err_curr /= (float)cnt_valid;
    err_curr = sqrtf(err_curr);
    const float err_rate = fabsf(err_prev - err_curr) / err_prev;
    const float dt_norm = dtu * dtu + dtv * dtv;
    if (iter > 1) {
      if (err_rate <= 1e-3f || dt_norm <= 1e-4f) break;
    }
    err_prev = err_curr;
  }
  if (lane == 0) {
    if (err_flag) {
      atomicOr(a.flags, err_flag);
      a.mask[pt] = 0;
    } else if (isnan(err_curr)) {
      a.mask[pt] = 0;
    } else if (err_curr <= 30) {
      a.pts_track[2 * pt] = pt0x + tx;
      a.pts_track[2 * pt + 1] = pt0y + ty;
      a.mask[pt] = 1;
    } else {
      a.mask[pt] = 0;
    }
  }
  return 2;
}

__device__ __forceinline__ void ic_state_clear(IcState &S) {
#pragma unroll
  for (int k = 0; k < IC_K; ++k) {
    S.I0[k] = S.du[k] = S.dv[k] = S.I1[k] = 0.f;
    S.m0[k] = S.m1[k] = false;
  }
}

// ---- pass 1: every point in parallel -------------------------------------------
__global__ __launch_bounds__(64) void ic_refine_kernel(IcArgs a) {
  __shared__ uint32_t s_t[IC_TH * IC_TW];
  const int n = a.d_n ? *a.d_n : a.n;
  const int pt = blockIdx.x;
  if (pt >= n) return;
  const int lane = threadIdx.x;
  if (lane == 0) {
    a.pts_track[2 * pt] = a.pts_prior[2 * pt];
    a.pts_track[2 * pt + 1] = a.pts_prior[2 * pt + 1];
  }
  int cls = 0, touched = 0;
  float lpx = 0.f, lpy = 0.f;
  const bool entry = a.mask_in ? a.mask_in[pt] != 0 : true;
  if (entry) {
    IcState S;
    ic_state_clear(S);
    cls = ic_point<false>(a, pt, lane, s_t, S, touched, lpx, lpy);
  } else if (lane == 0) {
    a.mask[pt] = 0;
  }
  const int any_t = __any(touched);
  if (lane == 0) {
    if (a.touched) a.touched[pt] = (uint8_t)(any_t ? 1 : 0);
    if (a.cls) a.cls[pt] = (uint8_t)cls;
    if (a.last_pu) {
      a.last_pu[2 * pt] = lpx;
      a.last_pu[2 * pt + 1] = lpy;
    }
  }
}

// ---- pass 2: sequential replay of the runs that contain touched points --------------
__global__ __launch_bounds__(64) void ic_strict_kernel(IcArgs a) {
  __shared__ uint32_t s_t[IC_TH * IC_TW];
  const int n = a.d_n ? *a.d_n : a.n;
  const int pt = blockIdx.x;
  if (pt >= n) return;
  const int lane = threadIdx.x;
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
  int dummy = 0;
  if (clean >= 0) {
    // state left behind by an untouched, iterated point: its template and its last I1 patch
    float ax, ay, axay;
    ic_frac(a.pts0[2 * clean], a.pts0[2 * clean + 1], ax, ay, axay);
    ic_template<true>(a.I0, a.pts0[2 * clean], a.pts0[2 * clean + 1], ax, ay, axay, lane, s_t, S, dummy);
    const float pux = a.last_pu[2 * clean], puy = a.last_pu[2 * clean + 1];
    ic_frac(pux, puy, ax, ay, axay);
    ic_sample_I1<true>(a.I1, pux, puy, a.scale[clean], ax, ay, axay, lane, S, dummy);
  }
  for (int p = start; p < n; ++p) {
    const int cp = a.cls[p];
    if (cp == 0) continue;
    const int tp = a.touched[p];
    if (!tp) {
      if (cp == 2) break;  // next clean point: end of the run
      // untouched, failed the determinant test: it only rewrote the template state
      float ax, ay, axay;
      ic_frac(a.pts0[2 * p], a.pts0[2 * p + 1], ax, ay, axay);
      ic_template<true>(a.I0, a.pts0[2 * p], a.pts0[2 * p + 1], ax, ay, axay, lane, s_t, S, dummy);
      continue;
    }
    if (lane == 0) {
      a.pts_track[2 * p] = a.pts_prior[2 * p];
      a.pts_track[2 * p + 1] = a.pts_prior[2 * p + 1];
    }
    float lx, ly;
    (void)ic_point<true>(a, p, lane, s_t, S, dummy, lx, ly);
  }
}

static int ic_args(vo_ctx *c, int slot0, int slot1, IcArgs &a, int *d_flags) {
  if (slot0 < 0 || slot0 >= c->cfg.n_slots || slot1 < 0 || slot1 >= c->cfg.n_slots)
    VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  const vo_pyramid &P0 = c->slots[slot0], &P1 = c->slots[slot1];
  if (P0.n_levels <= 0 || P1.n_levels <= 0) VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
  if (P0.w != P1.w || P0.h != P1.h) VO_FAIL(c, VO_ERR_SIZE, "image size mismatch");
  a.I0 = P0.lv[0];
  a.I1 = P1.lv[0];
  a.flags = d_flags ? d_flags : c->d_flags;
  return VO_OK;
}

// pass 1. d_prior and d_pts_track must be different buffers.
int vo_ic_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_scale,
                  const float *d_prior, float *d_pts_track, const uint8_t *d_mask_in, uint8_t *d_mask,
                  uint8_t *d_touched, uint8_t *d_cls, float *d_last_pu, int n_max, const int *d_n, int *d_flags) {
  if (n_max <= 0) return VO_OK;
  IcArgs a;
  memset(&a, 0, sizeof(a));
  int rc = ic_args(c, slot0, slot1, a, d_flags);
  a.mask_in = d_mask_in;
  if (rc) return rc;
  a.pts0 = d_pts0;
  a.scale = d_scale;
  a.pts_prior = d_prior;
  a.pts_track = d_pts_track;
  a.mask = d_mask;
  a.touched = d_touched;
  a.cls = d_cls;
  a.last_pu = d_last_pu;
  a.n = n_max;
  a.d_n = d_n;
  vo_prof_begin(c, VO_K_IC);
  hipLaunchKernelGGL(ic_refine_kernel, dim3(n_max), dim3(64), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// pass 2 (reference-exact border state); consumes pass 1's touched / cls / last_pu.
int vo_ic_strict_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_scale,
                         const float *d_prior, float *d_pts_track, uint8_t *d_mask, uint8_t *d_touched,
                         uint8_t *d_cls, float *d_last_pu, int n_max, const int *d_n, int *d_flags) {
  if (n_max <= 0) return VO_OK;
  IcArgs a;
  memset(&a, 0, sizeof(a));
  int rc = ic_args(c, slot0, slot1, a, d_flags);
  if (rc) return rc;
  a.pts0 = d_pts0;
  a.scale = d_scale;
  a.pts_prior = d_prior;
  a.pts_track = d_pts_track;
  a.mask = d_mask;
  a.touched = d_touched;
  a.cls = d_cls;
  a.last_pu = d_last_pu;
  a.n = n_max;
  a.d_n = d_n;
  vo_prof_begin(c, VO_K_IC);
  hipLaunchKernelGGL(ic_strict_kernel, dim3(n_max), dim3(64), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}
