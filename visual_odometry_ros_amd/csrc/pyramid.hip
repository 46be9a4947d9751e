// pyramid.hip — device-resident image pyramids for the pyramidal LK tracker.
//
// Replaces what cv::calcOpticalFlowPyrLK rebuilds on EVERY call in the reference
// (buildOpticalFlowPyramid + pyrDown, OpenCV 4 modules/video/src/lkpyramid.cpp and
// modules/imgproc/src/pyramids.cpp; reference call sites
// core/visual_odometry/feature_tracker.cpp:29,60,69,108,117,186 — eight pyramid
// builds per stereo frame for three distinct images). Here a slot's pyramid is
// built once per image and stays in HBM/L2 for every tracker call that uses it;
// the left and right image of a stereo pair are built by the same launches
// (blockIdx.z selects the image).
//
// Layout: level l is a padded u8 plane, VO_PAD pixels of REFLECT_101 border on
// every side (OpenCV pads by winSize; a fixed VO_PAD >= winSize+9 lets the
// tracker stage 4-byte-aligned tiles with a halo and never leave the
// allocation), row stride a multiple of 64 B so that every row starts on a
// cache-line boundary and tile rows can be fetched with aligned dword loads.
//
// Round 5: one launch per pyramid — pyr_build_kernel (pyr_tile.hpp) stages a base-level region
// in LDS and produces its tile of EVERY level, border included; the per-level kernels of rounds
// 1-4 (pad_level0_kernel, pyr_down_kernel: five dependent launches per pair) are gone. The
// rectifying ingestion keeps its own level-0 kernel (remap_level0_kernel) in front of it.
#include "vo_internal.hpp"
#include "vo_kernels.hpp"
#define PYR_SET_PRIO()
#include "pyr_plan.hpp"

// level 0 of an image that goes through Camera::undistortImage / StereoCamera::rectifyStereoImages first
// (core/visual_odometry/camera.cpp:166-183, :300-336: convertTo(CV_32FC1), cv::remap with float maps,
// INTER_LINEAR, BORDER_CONSTANT 0; then convertTo(CV_8UC1), stereo_vo.cpp:420-421 / mono_vo.cpp:512):
// the remap is fused into the level-0 build, the float image never exists. cv::remap's arithmetic
// (OpenCV 4 imgproc/imgwarp.cpp) for this case: coordinates quantised to 1/32 px with cvRound
// (sx = round-half-even(mapx * 32)), weights (32-ay)(32-ax), (32-ay)ax, ay(32-ax), ay*ax over 1024, taps
// outside the source = 0; convertTo = round-half-even, saturated. With u8 samples every product and the
// sum are exact in float, so the integer form below gives the same bytes.
struct RemapArgs {
  const uint8_t *src[2];
  const float *mu[2], *mv[2];
  uint8_t *dst[2];
  int w, h, sstride, dstride;
};
__device__ __forceinline__ int remap_sample(const uint8_t *__restrict__ src, int w, int h, int sstride, float mu,
                                            float mv) {
  if (!(mu == mu) || !(mv == mv)) return 0;  // cvRound(NaN) = INT_MIN on the CPU: far outside
  const int fxq = (int)__builtin_rintf(mu * 32.0f), fyq = (int)__builtin_rintf(mv * 32.0f);  // saturating cvt
  const int sx = fxq >> 5, sy = fyq >> 5, ax = fxq & 31, ay = fyq & 31;
  if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) return 0;
  const bool x0 = sx >= 0, x1 = sx + 1 < w, y0 = sy >= 0, y1 = sy + 1 < h;
  const uint8_t *p = src + (ptrdiff_t)sy * sstride + sx;
  const int s00 = (x0 && y0) ? p[0] : 0, s01 = (x1 && y0) ? p[1] : 0;
  const int s10 = (x0 && y1) ? p[sstride] : 0, s11 = (x1 && y1) ? p[sstride + 1] : 0;
  const int sum = s00 * ((32 - ay) * (32 - ax)) + s01 * ((32 - ay) * ax) + s10 * (ay * (32 - ax)) + s11 * (ay * ax);
  return (sum + 511 + ((sum >> 10) & 1)) >> 10;  // round half to even of sum / 1024 (<= 255)
}
// One lane per padded pixel: the two map loads of a wavefront are 256 contiguous bytes each (the maps are
// 80 % of this kernel's bytes); four neighbouring lanes then pack their bytes so that the store is a dword.
__global__ __launch_bounds__(256) void remap_level0_kernel(RemapArgs a) {
  const int pxp = blockIdx.x * blockDim.x + threadIdx.x;  // padded column (the padded row is a multiple of 4 wide)
  const int py = blockIdx.y;
  const int pw = (a.w + 2 * VO_PAD + 3) & ~3;
  const int z = blockIdx.z;
  int val = 0;
  if (pxp < pw) {
    const int y = reflect101_dev(py - VO_PAD, a.h);
    const int x = reflect101_dev(pxp - VO_PAD, a.w);
    const size_t o = (size_t)y * a.w + x;
    val = remap_sample(a.src[z], a.w, a.h, a.sstride, a.mu[z][o], a.mv[z][o]);
  }
  uint32_t v = (uint32_t)val;
  v |= (uint32_t)__shfl_down(val, 1) << 8;
  v |= (uint32_t)__shfl_down(val, 2) << 16;
  v |= (uint32_t)__shfl_down(val, 3) << 24;
  if ((threadIdx.x & 3) == 0 && pxp < pw) *(uint32_t *)(a.dst[z] + (size_t)py * a.dstride + pxp) = v;
}

// effective maxLevel of buildOpticalFlowPyramid
int vo_pyr_levels_host(int w, int h, int win, int max_level) {
  for (int level = 0; level <= max_level; ++level) {
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (w <= win || h <= win) return level;
  }
  return max_level;
}

// lay out the levels of a slot for an image of size (w,h)
static void layout_slot(vo_ctx *c, vo_pyramid *P, int w, int h) {
  size_t off = 0;
  P->w = w;
  P->h = h;
  for (int l = 0; l <= c->cfg.max_level && l < VO_MAX_LEVELS; ++l) {
    const int stride = ((w + 2 * VO_PAD) + 63) & ~63;
    P->lv[l].w = w;
    P->lv[l].h = h;
    P->lv[l].stride = stride;
    P->lv[l].base = P->mem + off;
    off += (size_t)stride * (size_t)(h + 2 * VO_PAD);
    off = (off + 255) & ~(size_t)255;
    w = (w + 1) / 2;
    h = (h + 1) / 2;
  }
}

// Build the pyramids of one image (d_r == nullptr) or of a stereo pair.
static int build(vo_ctx *c, int slot_l, const uint8_t *d_l, int slot_r, const uint8_t *d_r, int w, int h,
                 int stride, const int *cams = nullptr) {
  const int nimg = d_r ? 2 : 1;
  if (slot_l < 0 || slot_l >= c->cfg.n_slots || (d_r && (slot_r < 0 || slot_r >= c->cfg.n_slots || slot_r == slot_l)))
    VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  if (w <= 0 || h <= 0 || w > c->cfg.max_width || h > c->cfg.max_height)
    VO_FAIL(c, VO_ERR_CAPACITY, "image %dx%d exceeds vo_config %dx%d", w, h, c->cfg.max_width, c->cfg.max_height);
  vo_pyramid *P[2] = {&c->slots[slot_l], d_r ? &c->slots[slot_r] : nullptr};
  // A slot may be rebuilt only when everything that reads it has been collected: with ingestion on the side stream
  // nothing on the device orders the rebuild behind a frame that is still in flight.
  if (c->frame_slots_busy)
    for (int i = 0; i < nimg; ++i)
      for (int k = 0; k < 3; ++k)
        if (c->frame_slot[k] == (i ? slot_r : slot_l))
          VO_FAIL(c, VO_ERR_INVALID, "slot %d is read by the frame in flight: collect its result first", c->frame_slot[k]);
  for (int i = 0; i < nimg; ++i) layout_slot(c, P[i], w, h);
  vo_ingest_scope ingest(c);  // launchers enqueue on c->stream: the ingest stream for the duration of this chain
  int top = c->cfg.max_level;
  if (c->pyr_win_hint > 0) top = vo_pyr_levels_host(w, h, c->pyr_win_hint, c->cfg.max_level);
  if (cams)
    for (int i = 0; i < nimg; ++i) {
      const int k = cams[i];
      if (k < 0 || k > 1 || !c->rect_u[k]) VO_FAIL(c, VO_ERR_INVALID, "no rectification map for camera %d", k);
      if (c->rect_w[k] != w || c->rect_h[k] != h)  // camera.cpp:169, :307, :324
        VO_FAIL(c, VO_ERR_SIZE, "provided image has not the same size as the camera model (%dx%d vs map %dx%d)", w, h,
                c->rect_w[k], c->rect_h[k]);
    }
  vo_prof_begin(c, VO_K_PYRAMID);
  if (cams) {
    RemapArgs a;
    a.src[0] = d_l;
    a.src[1] = d_r ? d_r : d_l;
    for (int i = 0; i < 2; ++i) {
      const int k = cams[i < nimg ? i : 0];
      a.mu[i] = c->rect_u[k];
      a.mv[i] = c->rect_v[k];
    }
    a.dst[0] = P[0]->lv[0].base;
    a.dst[1] = d_r ? P[1]->lv[0].base : P[0]->lv[0].base;
    a.w = w;
    a.h = h;
    a.sstride = stride;
    a.dstride = P[0]->lv[0].stride;
    dim3 grid((w + 2 * VO_PAD + 3 + 255) / 256, h + 2 * VO_PAD, nimg);
    hipLaunchKernelGGL(remap_level0_kernel, grid, dim3(256), 0, c->stream, a);
  }
  // every level (and, unless the remap wrote it, level 0 with its border) by ONE launch per PYR_NL_MAX levels
  vo_level L[2][VO_MAX_LEVELS];
  for (int i = 0; i < nimg; ++i) memcpy(L[i], P[i]->lv, sizeof(L[i]));
  const uint8_t *src[2] = {d_l, d_r ? d_r : d_l};
  hipStream_t st = c->stream;
  const int nl = pyr_plan_and_launch(L, nimg, src, stride, top, cams != nullptr, [&](const PyrTileArgs &a, int groups) {
    hipLaunchKernelGGL(pyr_build_kernel, dim3(groups, 1, nimg), dim3(PYR_NT), 0, st, a);
  });
  vo_prof_end(c);
  for (int i = 0; i < nimg; ++i) P[i]->n_levels = nl;
  VO_CHECK_HIP(c, hipGetLastError());
  VO_CHECK_HIP(c, hipEventRecord(c->ev_pyr, c->stream));
  const int k = c->stream == c->stream2 ? 1 : 0;
  for (int i = 0; i < nimg; ++i) {
    VO_CHECK_HIP(c, hipEventRecord(P[i]->ready, c->stream));
    P[i]->seen[k] = 1;
    P[i]->seen[1 - k] = 0;
  }
  return VO_OK;
}

int vo_pyramid_build(vo_ctx *c, int slot, const uint8_t *d_img, int w, int h, int stride) {
  return build(c, slot, d_img, -1, nullptr, w, h, stride);
}
int vo_pyramid_build_pair(vo_ctx *c, int slot_l, const uint8_t *d_l, int slot_r, const uint8_t *d_r, int w, int h,
                          int stride) {
  return build(c, slot_l, d_l, slot_r, d_r, w, h, stride);
}
// the same with the undistortion / rectification remap of camera `cam` (cam_l, cam_r) fused into level 0
int vo_pyramid_build_rectified(vo_ctx *c, int slot, const uint8_t *d_img, int w, int h, int stride, int cam) {
  return build(c, slot, d_img, -1, nullptr, w, h, stride, &cam);
}
int vo_pyramid_build_pair_rectified(vo_ctx *c, int slot_l, const uint8_t *d_l, int slot_r, const uint8_t *d_r, int w,
                                    int h, int stride) {
  const int cams[2] = {0, 1};
  return build(c, slot_l, d_l, slot_r, d_r, w, h, stride, cams);
}
