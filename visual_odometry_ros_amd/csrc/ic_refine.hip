// ic_refine.hip — scale-compensated inverse-compositional patch refinement,
// one gfx950 wavefront per feature point, five checkerboard taps per lane.
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
// The kernel is latency-bound (a dependent sample -> reduce -> 2x2 solve chain per
// iteration, up to 30 iterations per point). Measured on gfx950 (tools/clockprobe5.hip): a
// cross-wavefront exchange of the four sums (LDS + s_barrier) costs 500-750 cycles per
// iteration once two wavefronts of a workgroup share a SIMD, three times the 4-way DPP
// butterfly itself, so a point is ONE wavefront: lane l owns taps l, l+64, .., l+256
// (264 taps), the five bilinear taps of an iteration are independent chains the
// scheduler interleaves, the four sums are per-lane partials (taps in ascending order)
// followed by one 4-way interleaved DPP butterfly, and there is no LDS exchange and
// no barrier in the loop. Float sums use the canonical order "64 strided partials
// (tap j -> partial j mod 64, ascending j), balanced binary tree over the partials" —
// the oracle's VO_SUM_TREE for this operator.
//
// Border semantics, three kernels:
//  * ic_refine_kernel (all points in parallel): a tap whose footprint leaves the
//    valid region is excluded from that evaluation (oracle VO_IC_BORDER_MASKED);
//    such points are reported in `touched`. Every point that never touches the
//    border gets exactly the reference result.
//  * The reference allocates its tap value / mask vectors once per call and never
//    resets them (feature_tracker.cpp:324-333, image_processing.cpp:88-89,276-279),
//    so an out-of-image tap keeps the value and mask bit of the most recent earlier
//    evaluation in which it was inside. Only touched points see
//    that state, and the state after an untouched, iterated ("clean") point is
//    fully determined by that point alone.
//    ic_jacobi_kernel recomputes all touched points in parallel from per-point tap
//    records until nothing changes any more (see the comment at the kernel);
//  * ic_strict_kernel is the sequential replay (one wavefront walks a run of points
//    between two clean points, the carried state living in registers); it only
//    runs when the parallel replay asks for it.
#include "ic_device.hpp"
#include "vo_kernels.hpp"

// ---- pass 1: every point in parallel -------------------------------------------
__global__ __launch_bounds__(IC_T) void ic_refine_kernel(IcArgs a) {
  __shared__ IcShared sh;
  const int n = a.d_n ? *a.d_n : a.n;
  const int pt = blockIdx.x;
  if (pt >= n) return;
  const int lane = threadIdx.x;
  int cls = 0, touched = 0, n_iter = 0;
  float lpx = 0.f, lpy = 0.f;
  const IcTaps tp = ic_make_taps(lane);
  const bool entry = a.mask_in ? a.mask_in[pt] != 0 : true;
#ifdef IC_STAMP
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  IcState S;
  ic_state_clear(S);
  if (entry) {
    cls = ic_point_io<false>(a, tp, pt, lane, sh, S, touched, lpx, lpy, n_iter).cls;
  } else if (lane == 0) {
    a.pts_track[2 * pt] = a.pts_prior[2 * pt];
    a.pts_track[2 * pt + 1] = a.pts_prior[2 * pt + 1];
    a.mask[pt] = 0;
  }
  ic_store_records(a, pt, lane, tp, S, cls);
#ifdef IC_STAMP
  if (lane == 0 && a.pre1) {
    float *dbg = a.pre1 + (size_t)pt * IC_NELEM;
    const unsigned long long wg_t1 = __builtin_amdgcn_s_memrealtime();
    dbg[0] = (float)n_iter;
    dbg[4] = (float)(wg_t0 & 0xFFFFFF);
    dbg[5] = (float)(wg_t1 & 0xFFFFFF);
  }
#endif
  const int any_t = __any(touched);
  if (lane == 0) {
    if (a.touched) a.touched[pt] = (uint8_t)(any_t ? 1 : 0);
    if (a.tlist && any_t) a.tlist[atomicAdd(&a.jac[IC_JAC_NT], 1)] = pt;
    if (a.cls) a.cls[pt] = (uint8_t)cls;
    if (a.last_pu) {
      a.last_pu[2 * pt] = lpx;
      a.last_pu[2 * pt + 1] = lpy;
    }
  }
}

// ---- pass 2a: parallel fixed-point replay of the touched points (ic_device.hpp: ic_replay) ----
__global__ __launch_bounds__(IC_T) void ic_jacobi_kernel(IcArgs a) {
  __shared__ IcReplayShared rs;
  (void)ic_replay(a, rs, threadIdx.x, [](int, const IcResult &) {});
}

// ---- pass 2: sequential replay of the runs that contain touched points (ic_device.hpp: ic_strict_run) ----
__global__ __launch_bounds__(IC_T) void ic_strict_kernel(IcArgs a) {
  __shared__ IcShared sh;
  const int n = a.d_n ? *a.d_n : a.n;
  const int pt = blockIdx.x;
  if (pt >= n) return;
  // with the parallel replay in front, this kernel only runs when that did not finish
  if (a.jac && a.jac[IC_JAC_OVF] == 0) return;
  ic_strict_run(a, sh, pt, n, threadIdx.x, [](int, const IcResult &) {});
}

static int ic_args(vo_ctx *c, int slot0, int slot1, IcArgs &a, int *d_flags) {
  if (slot0 < 0 || slot0 >= c->cfg.n_slots || slot1 < 0 || slot1 >= c->cfg.n_slots)
    VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  const vo_pyramid &P0 = c->slots[slot0], &P1 = c->slots[slot1];
  if (P0.n_levels <= 0 || P1.n_levels <= 0) VO_FAIL(c, VO_ERR_INVALID, "slot holds no image");
  if (P0.w != P1.w || P0.h != P1.h) VO_FAIL(c, VO_ERR_SIZE, "image size mismatch");
  if (vo_slot_acquire(c, slot0) < 0 || vo_slot_acquire(c, slot1) < 0) return VO_ERR_HIP;
  a.I0 = P0.lv[0];
  a.I1 = P1.lv[0];
  a.flags = d_flags ? d_flags : c->d_flags;
  return VO_OK;
}

// tap records of the parallel strict replay, allocated on first use (capacity cfg.max_points)
static int ic_records(vo_ctx *c, IcArgs &a) {
  if (!c->ic_rec) {
    const size_t N = (size_t)c->cfg.max_points;
    const size_t bytes = N * (4 + 4 + 4 + 2 * IC_MW * 4 + 3 * IC_NELEM * 4 + IC_NELEM * 4 + IC_NELEM * 4 + IC_MW * 4 + 8 + 4) + IC_JAC_BYTES;
    VO_CHECK_HIP(c, vo_dev_malloc(c, &c->ic_rec, bytes));
    VO_CHECK_HIP(c, hipMemsetAsync(c->ic_rec, 0, bytes, c->stream));
  }
  const size_t N = (size_t)c->cfg.max_points;
  uint8_t *p = (uint8_t *)c->ic_rec;
  a.jac = (int *)p;                 p += IC_JAC_BYTES;
  a.tlist = (int *)p;               p += N * 4;
  a.pubc = (int *)p;                p += N * 4;
  a.ready = (uint8_t *)p;           p += N * 4;
  a.recW0 = (uint32_t *)p;          p += N * IC_MW * 4;
  a.recW1 = (uint32_t *)p;          p += N * IC_MW * 4;
  a.preM = (uint32_t *)p;           p += N * IC_MW * 4;
  a.recV0 = (float *)p;             p += N * 3 * IC_NELEM * 4;
  a.recV1 = (float *)p;             p += N * IC_NELEM * 4;
  a.pre1 = (float *)p;              p += N * IC_NELEM * 4;
  a.tl2 = (unsigned long long *)p;  p += N * 8;  // (8-byte aligned: the arrays above take 5400 bytes per point)
  a.p1e = (int *)p;
  return VO_OK;
}

// IcArgs for the fused frame kernel: image levels and (strict border) the tap records. `ctl` is the
// frame's own control block: ctl[0] = error flags, ctl[16..] = replay control words. It is zeroed
// once at allocation and re-zeroed by the GN launch (frame mode prologue) at the end of every frame.
size_t vo_ic_ctl_bytes() { return 64 + IC_JAC_BYTES; }
int vo_ic_ctl_nt_word() { return IC_JAC_NT; }
int vo_ic_frame_args(vo_ctx *c, int slot0, int slot1, IcArgs *a, int *ctl, bool with_records) {
  memset(a, 0, sizeof(*a));
  int rc = ic_args(c, slot0, slot1, *a, ctl);
  if (rc) return rc;
  if (with_records) {
    rc = ic_records(c, *a);
    if (rc) return rc;
  }
  a->jac = ctl + 16;
  return VO_OK;
}
void vo_ic_strict_launch(vo_ctx *c, const IcArgs &a) {
  hipLaunchKernelGGL(ic_strict_kernel, dim3(a.n), dim3(IC_T), 0, c->stream, a);
}

// pass 1. d_prior and d_pts_track must be different buffers. with_records: also write the tap
// records the strict replay consumes.
int vo_ic_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_scale,
                  const float *d_prior, float *d_pts_track, const uint8_t *d_mask_in, uint8_t *d_mask,
                  uint8_t *d_touched, uint8_t *d_cls, float *d_last_pu, int n_max, const int *d_n, int *d_flags,
                  bool with_records) {
  if (n_max <= 0) return VO_OK;
  IcArgs a;
  memset(&a, 0, sizeof(a));
  int rc = ic_args(c, slot0, slot1, a, d_flags);
  if (rc) return rc;
  if (with_records) {
    rc = ic_records(c, a);
    if (rc) return rc;
    VO_CHECK_HIP(c, hipMemsetAsync(a.jac, 0, IC_JAC_BYTES, c->stream));
  }
  a.mask_in = d_mask_in;
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
  hipLaunchKernelGGL(ic_refine_kernel, dim3(n_max), dim3(IC_T), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

// pass 2 (reference-exact border state); consumes pass 1's touched / cls / last_pu and tap records:
// the parallel fixed-point replay, then the sequential replay, which only runs if the former asked for it.
int vo_ic_strict_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_scale,
                         const float *d_prior, float *d_pts_track, uint8_t *d_mask, uint8_t *d_touched,
                         uint8_t *d_cls, float *d_last_pu, int n_max, const int *d_n, int *d_flags,
                         bool sequential_only) {
  if (n_max <= 0) return VO_OK;
  IcArgs a;
  memset(&a, 0, sizeof(a));
  int rc = ic_args(c, slot0, slot1, a, d_flags);
  if (rc) return rc;
  rc = ic_records(c, a);
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
  if (sequential_only)  // validation mode: request the fallback up front
    VO_CHECK_HIP(c, hipMemsetAsync(&a.jac[IC_JAC_OVF], 1, sizeof(int), c->stream));
  else
    hipLaunchKernelGGL(ic_jacobi_kernel, dim3(n_max < IC_JGRID ? n_max : IC_JGRID), dim3(IC_T), 0, c->stream, a);
  hipLaunchKernelGGL(ic_strict_kernel, dim3(n_max), dim3(IC_T), 0, c->stream, a);
  vo_prof_end(c);
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

#ifdef IC_STAMP
// diagnostic (IC_STAMP builds): jac[16] + 64 debug words; clears the debug words
extern "C" int vo_debug_ic_jac(vo_ctx *c, int *dst) {
  if (!c || !c->ic_rec || !dst) return VO_ERR_INVALID;
  IcArgs a;
  memset(&a, 0, sizeof(a));
  ic_records(c, a);
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  VO_CHECK_HIP(c, hipMemcpy(dst, a.jac, 64, hipMemcpyDeviceToHost));
  VO_CHECK_HIP(c, hipMemcpy(dst + 16, a.tlist + IC_DBG_OFF, 256 + 4 * 4 * 512, hipMemcpyDeviceToHost));
  VO_CHECK_HIP(c, hipMemset(a.tlist + IC_DBG_OFF, 0, 256 + 4 * 4 * 512));
  return VO_OK;
}
// diagnostic (IC_STAMP builds): first `k` floats of each point's pre1 row
extern "C" int vo_debug_ic_rows(vo_ctx *c, float *dst, int k, int n) {
  if (!c || !c->ic_rec || !dst) return VO_ERR_INVALID;
  IcArgs a;
  memset(&a, 0, sizeof(a));
  ic_records(c, a);
  VO_CHECK_HIP(c, hipMemcpy2D(dst, sizeof(float) * k, a.pre1, sizeof(float) * IC_NELEM, sizeof(float) * k, n,
                              hipMemcpyDeviceToHost));
  return VO_OK;
}
#endif
