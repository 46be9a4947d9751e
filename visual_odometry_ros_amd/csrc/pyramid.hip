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
// Both kernels are HBM/L2 streaming kernels: one thread produces 4 horizontally
// adjacent bytes and stores one dword; consecutive lanes store consecutive
// dwords (256 B per wave-instruction).
#include "vo_internal.hpp"
#include "vo_kernels.hpp"

struct PadArgs {
  const uint8_t *src[2];
  uint8_t *dst[2];
  int w, h, sstride, dstride;
};

// level 0: copy the source image into the padded plane, REFLECT_101 border.
__global__ __launch_bounds__(256) void pad_level0_kernel(PadArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;  // dword index within a padded row
  const int py = blockIdx.y;                            // padded row
  const int pw = a.w + 2 * VO_PAD;
  if (q * 4 >= pw) return;
  const uint8_t *__restrict__ src = a.src[blockIdx.z];
  uint8_t *__restrict__ dst = a.dst[blockIdx.z];
  const int y = reflect101_dev(py - VO_PAD, a.h);
  const uint8_t *srow = src + (size_t)y * a.sstride;
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = reflect101_dev(q * 4 + k - VO_PAD, a.w);
    v |= (uint32_t)srow[x] << (8 * k);
  }
  *(uint32_t *)(dst + (size_t)py * a.dstride + q * 4) = v;
}

// cv::pyrDown (5-tap [1 4 6 4 1]/16 both ways, +128 >> 8) of the padded level
// l-1 into the whole padded level l. The source border already holds the
// REFLECT_101 extension, so interior outputs read straight through it; border
// outputs are the pyrDown value at the reflected coordinate (the same bytes
// copyMakeBorder would copy), recomputed instead of waiting for the interior.
struct DownArgs {
  vo_level S[2], D[2];
};
__device__ __forceinline__ int pyr_tap5(const uint8_t *p) {
  return (int)p[0] + 4 * (int)p[1] + 6 * (int)p[2] + 4 * (int)p[3] + (int)p[4];
}
__global__ __launch_bounds__(256) void pyr_down_kernel(DownArgs a) {
  const vo_level S = a.S[blockIdx.z], D = a.D[blockIdx.z];
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int py = blockIdx.y;
  const int pw = D.w + 2 * VO_PAD;
  if (q * 4 >= pw) return;
  const uint8_t *so = S.origin();
  const int y = reflect101_dev(py - VO_PAD, D.h);
  uint32_t v = 0;
  const int px0 = q * 4 - VO_PAD;
  if (px0 >= 0 && px0 + 3 < D.w) {
    // interior fast path: 4 outputs share source columns 2*px0-2 .. 2*px0+8
    int col[11];
#pragma unroll
    for (int c = 0; c < 11; ++c) col[c] = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int wgt = (r == 0 || r == 4) ? 1 : ((r == 1 || r == 3) ? 4 : 6);
      const uint8_t *row = so + (ptrdiff_t)(2 * y + r - 2) * S.stride + (2 * px0 - 2);
#pragma unroll
      for (int c = 0; c < 11; ++c) col[c] += wgt * (int)row[c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int s = col[2 * k] + 4 * col[2 * k + 1] + 6 * col[2 * k + 2] + 4 * col[2 * k + 3] + col[2 * k + 4];
      v |= (uint32_t)((s + 128) >> 8) << (8 * k);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = reflect101_dev(px0 + k, D.w);
      int s = 0;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int wgt = (r == 0 || r == 4) ? 1 : ((r == 1 || r == 3) ? 4 : 6);
        s += wgt * pyr_tap5(so + (ptrdiff_t)(2 * y + r - 2) * S.stride + (2 * x - 2));
      }
      v |= (uint32_t)((s + 128) >> 8) << (8 * k);
    }
  }
  *(uint32_t *)(D.base + (size_t)py * D.stride + q * 4) = v;
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
                 int stride) {
  const int nimg = d_r ? 2 : 1;
  if (slot_l < 0 || slot_l >= c->cfg.n_slots || (d_r && (slot_r < 0 || slot_r >= c->cfg.n_slots || slot_r == slot_l)))
    VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  if (w <= 0 || h <= 0 || w > c->cfg.max_width || h > c->cfg.max_height)
    VO_FAIL(c, VO_ERR_CAPACITY, "image %dx%d exceeds vo_config %dx%d", w, h, c->cfg.max_width, c->cfg.max_height);
  vo_pyramid *P[2] = {&c->slots[slot_l], d_r ? &c->slots[slot_r] : nullptr};
  for (int i = 0; i < nimg; ++i) layout_slot(c, P[i], w, h);
  int top = c->cfg.max_level;
  if (c->pyr_win_hint > 0) top = vo_pyr_levels_host(w, h, c->pyr_win_hint, c->cfg.max_level);
  vo_prof_begin(c, VO_K_PYRAMID);
  {
    PadArgs a;
    a.src[0] = d_l;
    a.src[1] = d_r ? d_r : d_l;
    a.dst[0] = P[0]->lv[0].base;
    a.dst[1] = d_r ? P[1]->lv[0].base : P[0]->lv[0].base;
    a.w = w;
    a.h = h;
    a.sstride = stride;
    a.dstride = P[0]->lv[0].stride;
    dim3 grid(((w + 2 * VO_PAD + 3) / 4 + 255) / 256, h + 2 * VO_PAD, nimg);
    hipLaunchKernelGGL(pad_level0_kernel, grid, dim3(256), 0, c->stream, a);
  }
  int nl = 1;
  for (int l = 1; l <= top && l < VO_MAX_LEVELS; ++l) {
    DownArgs a;
    a.S[0] = P[0]->lv[l - 1];
    a.D[0] = P[0]->lv[l];
    a.S[1] = d_r ? P[1]->lv[l - 1] : a.S[0];
    a.D[1] = d_r ? P[1]->lv[l] : a.D[0];
    if (a.S[0].w < 2 || a.S[0].h < 2) break;
    dim3 grid(((a.D[0].w + 2 * VO_PAD + 3) / 4 + 255) / 256, a.D[0].h + 2 * VO_PAD, nimg);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, c->stream, a);
    ++nl;
  }
  vo_prof_end(c);
  for (int i = 0; i < nimg; ++i) P[i]->n_levels = nl;
  VO_CHECK_HIP(c, hipGetLastError());
  return VO_OK;
}

int vo_pyramid_build(vo_ctx *c, int slot, const uint8_t *d_img, int w, int h, int stride) {
  return build(c, slot, d_img, -1, nullptr, w, h, stride);
}
int vo_pyramid_build_pair(vo_ctx *c, int slot_l, const uint8_t *d_l, int slot_r, const uint8_t *d_r, int w, int h,
                          int stride) {
  return build(c, slot_l, d_l, slot_r, d_r, w, h, stride);
}
