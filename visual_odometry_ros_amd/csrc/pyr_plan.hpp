// pyr_plan.hpp — host side of pyr_build_kernel (pyr_tile.hpp): which launches build levels 0..top of laid-out planes.
// One launch produces up to PYR_NL_MAX levels above its base; a deeper pyramid (max_level up to 9 without a window hint)
// chains a second launch whose base is the last level of the first. Plain C++: pyramid.hip and the CPU emulation
// harness (tests/emu/) both call it, so the harness exercises the launch plan the product uses.
#pragma once
#include "pyr_tile.hpp"

// L[i][l]: the planes of image i (laid out, base pointers set); src[i] / sstride: the image that becomes level 0 — or,
// with base_written, ignored: level 0 is already in its plane (the rectifying remap wrote it, border included).
// Levels stop early where cv::buildOpticalFlowPyramid's source would be narrower than 2 pixels (as pyramid.hip always did).
// launch(args, workgroups_per_image) enqueues one pyr_build_kernel. Returns the number of levels built (>= 1).
template <class Launch>
static int pyr_plan_and_launch(vo_level (*L)[VO_MAX_LEVELS], int nimg, const uint8_t *const src[2], int sstride, int top,
                               bool base_written, Launch launch) {
  int n_levels = 1;
  for (int l = 1; l <= top && l < VO_MAX_LEVELS; ++l) {
    if (L[0][l - 1].w < 2 || L[0][l - 1].h < 2) break;
    ++n_levels;
  }
  int base = 0;
  bool first = true;
  while (first || base < n_levels - 1) {
    PyrTileArgs a;
    const int nl = (n_levels - 1 - base) < PYR_NL_MAX ? (n_levels - 1 - base) : PYR_NL_MAX;
    a.nl = nl;
    a.T = PYR_T0 >> nl;
    a.write_base = (first && !base_written) ? 1 : 0;
    for (int i = 0; i < 2; ++i) {
      const int q = i < nimg ? i : 0;
      if (first && !base_written) {
        a.src[i] = src[q];
        a.sstride[i] = sstride;
      } else {
        a.src[i] = L[q][base].origin();
        a.sstride[i] = L[q][base].stride;
      }
      for (int k = 0; k <= PYR_NL_MAX; ++k) a.L[i][k] = L[q][base + (k <= nl ? k : nl)];
    }
    const vo_level &T = L[0][base + nl];
    a.tiles_x = (T.w + a.T - 1) / a.T;
    a.tiles_y = (T.h + a.T - 1) / a.T;
    if (nl > 0 || a.write_base) launch(a, a.tiles_x * a.tiles_y);
    base += nl;
    first = false;
    if (nl == 0) break;
  }
  return n_levels;
}
