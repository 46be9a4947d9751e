// vo_internal.hpp — context layout and device helpers shared by the gfx950 kernels.
// Wave width is 64 everywhere (CDNA4); reductions use DPP butterflies whose
// summation tree is the balanced binary tree over lanes in natural order
// (adjacent pairs first) — the order oracle/ calls VO_SUM_TREE.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vo_hip.h"

#define VO_ALIGNBYTE(hi, lo, n) __builtin_amdgcn_alignbyte((hi), (lo), (n))
#include "vo_layout.hpp"   // VO_PAD, VO_MAX_LEVELS, vo_level, vo_reflect101
#define VO_WAVE 64

struct vo_pyramid {
  vo_level lv[VO_MAX_LEVELS];
  int n_levels;    // levels built (0..n_levels-1 valid)
  int w, h;
  uint8_t *mem;    // one allocation for all levels
  size_t bytes;
  // ordering across the context's two streams: `ready` is recorded behind every build of the slot, on the stream
  // that built it; a consumer on the other stream waits for it once (vo_slot_acquire)
  hipEvent_t ready;
  uint8_t seen[2];  // [0] main stream / [1] side stream is already ordered behind the last build
  uint8_t *stage;  // device staging for an image that arrives from the host asynchronously (allocated on demand)
};

struct vo_gn_dev_info {
  int iterations;
  float err, delta_err, delta_norm;
  int cnt_invalid;
  int is_nan;
};

struct vo_prof_rec {
  hipEvent_t a, b;
  int cls;
};

struct vo_ctx {
  vo_config cfg;
  int device;
  hipStream_t stream;      // the stream launchers enqueue on (= stream_main, except while a side-stream chain is built)
  hipStream_t stream_main;
  hipStream_t stream2;     // side stream: work that does not depend on the main chain of a frame
  hipStream_t stream3;     // the strict-border replay of the frame in flight (runs next to the frame kernel)
  int ingest_side;         // vo_set_ingest_side_stream: image ingestion (H2D, pyramids) runs on the side stream
  // StereoVO's synchronous call with host images: the detection of the LEFT image is started from its staging plane as soon
  // as that upload is queued (vo_set_stereo_pair_host_async), not behind the pair's pyramids; armed per call, consumed there
  const vo_bin_params *early_bins;
  int early_table, early_issued;
  hipEvent_t ev_fork, ev_join;
  hipEvent_t ev_pyr;       // recorded behind every pyramid build: what side-stream consumers of a slot wait for
  char err[512];
  vo_pyramid *slots;
  // per-point device buffers (capacity cfg.max_points)
  float *d_pts0, *d_pts1, *d_pts2, *d_pts3, *d_err, *d_err2, *d_scale, *d_X, *d_X2;
  uint8_t *d_status, *d_status2, *d_mask, *d_mask2;
  int32_t *d_idx;
  int *d_count;
  float *d_mat;            // small matrices / scalars
  vo_gn_dev_info *d_gninfo;
  int *d_flags;            // error flags raised by kernels
  // pinned host staging
  uint8_t *h_stage;
  size_t h_stage_bytes;
  uint8_t *d_img_stage;    // device staging for host images
  // hamming
  uint8_t *d_desc_a, *d_desc_b;
  uint16_t *d_dist;
  size_t desc_cap, dist_cap;
  // frame pipeline state
  void *ic_rec;            // tap records of the IC strict replay (ic_refine.hip)
  struct vo_frame_state *frame;
  int frame_strict_ic;     // replay border-touching points with the reference's sticky tap state (as requested: 0..4)
  int frame_strict_now;    // the stereo frame in flight: 4 (automatic) resolved to 1 or 3
  int frame_recoveries;    // frames issued again after such a time-out (vo_stereo_frame_recoveries)
  int frame_conc_off;      // a device-side join timed out once: the concurrent arrangements (3, 5, 4 -> 3) are off for good
  int frame_slots_busy;    // a frame is in flight and reads frame_slot[0..2]
  int frame_slot[3];
  // profiling
  vo_prof_rec *prof;
  int prof_cap, prof_n;
  unsigned prof_mask;      // bit per kernel class; 0 = all
  int prof_open;           // begin() recorded an event that end() must close
  int pyr_win_hint;        // > 0: build only the levels calcOpticalFlowPyrLK would use for this window
  // undistortion / rectification maps (rectify.hip): camera 0 = left or mono, 1 = right
  float *rect_u[2], *rect_v[2];
  int rect_w[2], rect_h[2];
  // per-stream ID counters: Landmark::landmark_counter_ (landmark.h:64) and Frame::frame_counter_ (frame.h:53) are
  // process-global in the reference; one pair per context keeps the streams of a batch from interleaving
  int32_t next_landmark_id, next_frame_id;
  struct vo_sba_state *sba;  // device arena of the sparse local BA (sba.hip)
  struct vo_orb_state *orb;  // pyramid, score planes and candidate lists of the keypoint detector (orb_detect.hip)
  // every device / pinned allocation made on behalf of this context (vo_dev_malloc / vo_host_malloc): what
  // vo_debug_allocation_count reports, so that a test can assert that a steady-state frame allocates nothing
  long long n_allocs;
  // vo_debug_set: test / measurement switches of THIS context (never read from the environment inside the library)
  int dbg[VO_DBG_COUNT];
};

static inline hipError_t vo_dev_malloc(vo_ctx *c, void **p, size_t bytes) {
  if (c) ++c->n_allocs;
  return hipMalloc(p, bytes);
}
static inline hipError_t vo_host_malloc(vo_ctx *c, void **p, size_t bytes, unsigned flags) {
  if (c) ++c->n_allocs;
  return hipHostMalloc(p, bytes, flags);
}

#define VO_CHECK_HIP(ctx, expr)                                                            \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, \
               hipGetErrorString(_e));                                                     \
      return VO_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

#define VO_FAIL(ctx, code, ...)                                \
  do {                                                         \
    snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__);     \
    return (code);                                             \
  } while (0)

// A call needs pyramid levels 0..eff of a slot. Fewer were built when vo_config.max_level is below the call's
// maxLevel or when vo_set_pyramid_window_hint() named a LARGER window than the call uses (a larger window stops
// the pyramid earlier): tracking with fewer levels than cv::buildOpticalFlowPyramid would build diverges from the
// reference, so it is an error, not a clamp.
#define VO_NEED_LEVELS(ctx, P, eff)                                                                                  \
  do {                                                                                                               \
    if ((eff) > (P).n_levels - 1)                                                                                    \
      VO_FAIL(ctx, VO_ERR_INVALID,                                                                                   \
              "the slot's pyramid holds levels 0..%d, this call needs 0..%d: raise vo_config.max_level, or give "    \
              "vo_set_pyramid_window_hint the SMALLEST window in use",                                               \
              (P).n_levels - 1, (eff));                                                                              \
  } while (0)

// Work about to be enqueued on c->stream reads the pyramid of `slot`: order it behind the slot's last build if that
// ran on the other stream (one hipStreamWaitEvent per build and stream, nothing when both are the same stream).
static inline int vo_slot_acquire(vo_ctx *c, int slot) {
  vo_pyramid &P = c->slots[slot];
  const int k = c->stream == c->stream2 ? 1 : 0;
  if (!P.seen[k]) {
    VO_CHECK_HIP(c, hipStreamWaitEvent(c->stream, P.ready, 0));
    P.seen[k] = 1;
  }
  return VO_OK;
}

// Image ingestion (H2D + pyramid chain) enqueues on the side stream when vo_set_ingest_side_stream is on: launchers
// use c->stream, which this scope points at the ingest stream and restores on exit.
struct vo_ingest_scope {
  vo_ctx *c;
  hipStream_t saved;
  explicit vo_ingest_scope(vo_ctx *ctx) : c(ctx), saved(ctx->stream) {
    if (c->ingest_side && c->stream == c->stream_main) c->stream = c->stream2;
  }
  ~vo_ingest_scope() { c->stream = saved; }
};

// The cumulative hand-shake targets (features past pass 1, finished fallback workgroups, candidate workgroups, DLT workers)
// grow by a few thousand per frame for the life of a stream and are compared by wrapped difference on the device; on the
// host they wrap the same way — through unsigned arithmetic, a signed `+=` would be undefined behaviour after ~5e5 frames.
static inline void vo_wrap_add(int &x, int n) { x = (int)((unsigned)x + (unsigned)n); }

// ---- profiling brackets (no-ops unless vo_profile_enable was called) --------
static inline void vo_prof_begin(vo_ctx *c, int cls) {
  c->prof_open = 0;
  if (c->prof && c->prof_n < c->prof_cap && (!c->prof_mask || (c->prof_mask & (1u << cls)))) {
    c->prof[c->prof_n].cls = cls;
    (void)hipEventRecord(c->prof[c->prof_n].a, c->stream);
    c->prof_open = 1;
  }
}
static inline void vo_prof_end(vo_ctx *c) {
  if (c->prof_open) {
    (void)hipEventRecord(c->prof[c->prof_n].b, c->stream);
    ++c->prof_n;
    c->prof_open = 0;
  }
}

#ifdef __HIPCC__
// ---- DPP cross-lane helpers ---------------------------------------------------
// dpp_ctrl: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E,
//           row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Sum over the 64 lanes; every lane returns the same bits.
// Tree: ((l0+l1)+(l2+l3)) ... within rows of 16 (quad_perm, half_mirror, mirror), then
// row_bcast15 (rows 1,3 += previous row), row_bcast31 (rows 2,3 += row 1) and one readlane(63):
// (r3+r2)+(r1+r0), the same additions as the balanced tree (r0+r1)+(r2+r3) since + commutes.
__device__ __forceinline__ float wave_sum_f32(float v) {
  v = v + dpp_f32<0xB1>(v);
  v = v + dpp_f32<0x4E>(v);
  v = v + dpp_f32<0x141>(v);
  v = v + dpp_f32<0x140>(v);
  v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Four independent sums, stepped together so that each DPP's wait states are filled by the other
// three chains (a lone wavefront otherwise stalls on every dependent DPP). Same tree as wave_sum_f32.
__device__ __forceinline__ void wave_sum4_f32(float &a, float &b, float &c, float &d) {
#define VO_STEP4(CTRL)                                    \
  {                                                       \
    const float ta = dpp_f32<CTRL>(a), tb = dpp_f32<CTRL>(b), tc = dpp_f32<CTRL>(c), td = dpp_f32<CTRL>(d); \
    a = a + ta;                                           \
    b = b + tb;                                           \
    c = c + tc;                                           \
    d = d + td;                                           \
  }
  VO_STEP4(0xB1)
  VO_STEP4(0x4E)
  VO_STEP4(0x141)
  VO_STEP4(0x140)
#undef VO_STEP4
#define VO_BC4(CTRL, RM)                                                                                   \
  {                                                                                                        \
    const float ta = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, RM, 0xF, false)); \
    const float tb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), CTRL, RM, 0xF, false)); \
    const float tc = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), CTRL, RM, 0xF, false)); \
    const float td = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), CTRL, RM, 0xF, false)); \
    a = a + ta;                                                                                            \
    b = b + tb;                                                                                            \
    c = c + tc;                                                                                            \
    d = d + td;                                                                                            \
  }
  VO_BC4(0x142, 0xA)
  VO_BC4(0x143, 0xC)
#undef VO_BC4
  a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
  b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
  c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
  d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 63));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  v = v + dpp_i32<0xB1>(v);
  v = v + dpp_i32<0x4E>(v);
  v = v + dpp_i32<0x141>(v);
  v = v + dpp_i32<0x140>(v);
  v = v + __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v = v + __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void wave_sum4_i32(int &a, int &b, int &c, int &d) {
#define VO_STEP4I(CTRL)                                                                       \
  {                                                                                           \
    const int ta = dpp_i32<CTRL>(a), tb = dpp_i32<CTRL>(b), tc = dpp_i32<CTRL>(c), td = dpp_i32<CTRL>(d); \
    a += ta;                                                                                  \
    b += tb;                                                                                  \
    c += tc;                                                                                  \
    d += td;                                                                                  \
  }
  VO_STEP4I(0xB1)
  VO_STEP4I(0x4E)
  VO_STEP4I(0x141)
  VO_STEP4I(0x140)
#undef VO_STEP4I
#define VO_BC4I(CTRL, RM)                                                  \
  {                                                                        \
    const int ta = __builtin_amdgcn_update_dpp(0, a, CTRL, RM, 0xF, false); \
    const int tb = __builtin_amdgcn_update_dpp(0, b, CTRL, RM, 0xF, false); \
    const int tc = __builtin_amdgcn_update_dpp(0, c, CTRL, RM, 0xF, false); \
    const int td = __builtin_amdgcn_update_dpp(0, d, CTRL, RM, 0xF, false); \
    a += ta;                                                               \
    b += tb;                                                               \
    c += tc;                                                               \
    d += td;                                                               \
  }
  VO_BC4I(0x142, 0xA)
  VO_BC4I(0x143, 0xC)
#undef VO_BC4I
  a = __builtin_amdgcn_readlane(a, 63);
  b = __builtin_amdgcn_readlane(b, 63);
  c = __builtin_amdgcn_readlane(c, 63);
  d = __builtin_amdgcn_readlane(d, 63);
}
// two exact (float)(int64 sum) at once: 16/16 split of both partials, one interleaved 4-way butterfly
__device__ __forceinline__ void wave_sum2_i32_to_f32(int p, int q, float &fp, float &fq) {
  // When every lane's partials are below 2^25 in magnitude (the usual case) the 64-lane sums fit int32: two chains
  // instead of four, and (float)(int32) rounds once exactly like (float)(int64).
  if (__builtin_expect(!__any((((unsigned)p + (1u << 25)) | ((unsigned)q + (1u << 25))) >> 26), 1)) {
#define VO_STEP2I(EXPR_P, EXPR_Q) \
  {                               \
    const int tp = EXPR_P, tq = EXPR_Q; \
    p += tp;                      \
    q += tq;                      \
  }
    VO_STEP2I(dpp_i32<0xB1>(p), dpp_i32<0xB1>(q))
    VO_STEP2I(dpp_i32<0x4E>(p), dpp_i32<0x4E>(q))
    VO_STEP2I(dpp_i32<0x141>(p), dpp_i32<0x141>(q))
    VO_STEP2I(dpp_i32<0x140>(p), dpp_i32<0x140>(q))
    VO_STEP2I(__builtin_amdgcn_update_dpp(0, p, 0x142, 0xA, 0xF, false), __builtin_amdgcn_update_dpp(0, q, 0x142, 0xA, 0xF, false))
    VO_STEP2I(__builtin_amdgcn_update_dpp(0, p, 0x143, 0xC, 0xF, false), __builtin_amdgcn_update_dpp(0, q, 0x143, 0xC, 0xF, false))
#undef VO_STEP2I
    fp = (float)__builtin_amdgcn_readlane(p, 63);
    fq = (float)__builtin_amdgcn_readlane(q, 63);
    return;
  }
  int plo = p & 0xFFFF, phi = p >> 16, qlo = q & 0xFFFF, qhi = q >> 16;
  wave_sum4_i32(plo, phi, qlo, qhi);
  fp = (float)((double)phi * 65536.0 + (double)plo);
  fq = (float)((double)qhi * 65536.0 + (double)qlo);
}
// three at once: six interleaved chains (16/16 halves of p, q, r), same butterfly
__device__ __forceinline__ void wave_sum3_i32_to_f32(int p, int q, int r, float &fp, float &fq, float &fr) {
  int v[6] = {p & 0xFFFF, p >> 16, q & 0xFFFF, q >> 16, r & 0xFFFF, r >> 16};
#define VO_STEP6(CTRL)                                      \
  {                                                         \
    int t[6];                                               \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) t[k] = dpp_i32<CTRL>(v[k]); \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) v[k] += t[k];               \
  }
  VO_STEP6(0xB1)
  VO_STEP6(0x4E)
  VO_STEP6(0x141)
  VO_STEP6(0x140)
#undef VO_STEP6
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] += __builtin_amdgcn_update_dpp(0, v[k], 0x142, 0xA, 0xF, false);
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] += __builtin_amdgcn_update_dpp(0, v[k], 0x143, 0xC, 0xF, false);
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = __builtin_amdgcn_readlane(v[k], 63);
  fp = (float)((double)v[1] * 65536.0 + (double)v[0]);
  fq = (float)((double)v[3] * 65536.0 + (double)v[2]);
  fr = (float)((double)v[5] * 65536.0 + (double)v[4]);
}
// Exact 64-bit sum of 64 int32 lane values (each |v| < 2^30): split 16/16 so the
// two int32 wave sums cannot overflow, recombine in int64.
__device__ __forceinline__ long long wave_sum_i32_to_i64(int v) {
  int lo = v & 0xFFFF;
  int hi = v >> 16;
  int slo = wave_sum_i32(lo);
  int shi = wave_sum_i32(hi);
  return (long long)shi * 65536LL + (long long)slo;
}
// (float)(exact int64 sum): the sum is < 2^53, so hi*65536 + lo is exact in double and the
// double -> float conversion rounds once, exactly like (float)(int64_t).
__device__ __forceinline__ float wave_sum_i32_to_f32(int v) {
  int lo = v & 0xFFFF;
  int hi = v >> 16;
  int slo = wave_sum_i32(lo);
  int shi = wave_sum_i32(hi);
  return (float)((double)shi * 65536.0 + (double)slo);
}

__device__ __forceinline__ int reflect101_dev(int p, int n) { return vo_reflect101(p, n); }
#endif  // __HIPCC__

