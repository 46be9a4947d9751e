// orb_plan.hpp — host side of the detector's tile kernels (orb_tile.hpp): the coefficient tables of cv::resize
// INTER_LINEAR_EXACT and, per tile column / row and pyramid level, the pixels a workgroup OWNS and the image region it has
// to hold in LDS to produce them. Plain C++: orb_detect.hip and the CPU emulation harness (tests/emu/) share it.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

#ifndef ORB_MAX_LEVELS
#define ORB_MAX_LEVELS 12
#endif

// resize.cpp interpolationLinear<uchar>::getCoeffs on softdouble (= IEEE double, one rounding per operation)
static inline void orb_linear_exact_coeffs(int src_size, int dst_size, std::vector<int> &ofs, std::vector<int> &c1) {
  // interpolationLinear(inv_scale, ..): scale = softdouble::one() / softdouble(inv_scale), inv_scale = dst / src
  const double scale = 1.0 / ((double)dst_size / (double)src_size);
  ofs.assign(dst_size, 0);
  c1.assign(dst_size, 0);
  for (int v = 0; v < dst_size; ++v) {
    const double fval = scale * ((double)v + 0.5) - 0.5;
    const int ival = (int)floor(fval);
    if (ival >= 0 && src_size > 1) {
      if (ival < src_size - 1) {
        ofs[v] = ival;
        c1[v] = (int)lrint((fval - (double)ival) * 256.0);
      } else {  // the last source sample with full weight, written so that the kernel never reads past the row
        ofs[v] = src_size - 2;
        c1[v] = 256;
      }
    }  // else: the first source sample with full weight (ofs 0, c1 0)
  }
}

// ORB_Impl::detectAndCompute level sizes / scales and computeKeyPoints' per-level quotas (orb.cpp), as oracle_orb.c
static inline void orb_level_layout(int w, int h, int n_levels, double scale_factor, int nfeatures, int *lw, int *lh, float *lscale,
                                    int *quota) {
  for (int l = 0; l < n_levels; ++l) {
    const float s = (float)pow(scale_factor, (double)l);
    lscale[l] = s;
    lw[l] = (int)lrint((double)((float)w / s));
    lh[l] = (int)lrint((double)((float)h / s));
  }
  const float factor = (float)(1.0 / scale_factor);
  float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)n_levels));
  int sum = 0;
  for (int l = 0; l < n_levels - 1; ++l) {
    quota[l] = (int)lrint((double)nd);
    sum += quota[l];
    nd *= factor;
  }
  quota[n_levels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

// one axis of one tile at one level: owned pixels [own0, own1) (inside runByImageBorder's rectangle) and the image
// region [reg0, reg1) the workgroup stages in LDS (own +- 4 for FAST / the non-max ring / Harris, and the source
// footprint of the next level's region); empty intervals have reg1 <= reg0 / own1 <= own0
struct OrbSpan {
  int own0, own1, reg0, reg1;
};

struct OrbTilePlan {
  bool ok = false;          // false: the configuration does not fit the tile kernels (the general path runs)
  int n_levels = 0, nx = 0, ny = 0;
  int lds_off[ORB_MAX_LEVELS], lds_stride[ORB_MAX_LEVELS];  // image regions
  int sc_off[ORB_MAX_LEVELS], sc_stride[ORB_MAX_LEVELS];    // score tiles (owned + 1 ring)
  int tx_off[ORB_MAX_LEVELS], ty_off[ORB_MAX_LEVELS];       // level >= 1: the region's slices of the coefficient tables (ints)
  int stash_off = 0, stash_cap = 0;                         // candidates of one workgroup (8 bytes each)
  int lds_bytes = 0;
  std::vector<OrbSpan> gx, gy;                              // [level * nx + i], [level * ny + j]
  std::vector<int> tabx[ORB_MAX_LEVELS], taby[ORB_MAX_LEVELS];  // level >= 1: (source offset << 16) | weight of the next sample
};

static inline void orb_plan_axis(const int *dim, int n_levels, int edge, int n_tiles, const std::vector<int> *ofs /* [level] */,
                                 std::vector<OrbSpan> &out) {
  out.assign((size_t)n_levels * n_tiles, OrbSpan{0, 0, 0, 0});
  for (int i = 0; i < n_tiles; ++i) {
    int need0 = 0, need1 = 0;  // what level l + 1's region reads of level l
    for (int l = n_levels - 1; l >= 0; --l) {
      OrbSpan s{0, 0, 0, 0};
      // tiles are cut at the same RELATIVE positions on every level, so that a tile's pieces lie above each other
      int a = (int)(((long long)i * dim[l]) / n_tiles), b = (int)(((long long)(i + 1) * dim[l]) / n_tiles);
      if (a < edge) a = edge;
      if (b > dim[l] - edge) b = dim[l] - edge;
      if (b > a) {
        s.own0 = a;
        s.own1 = b;
        s.reg0 = a - 4;
        s.reg1 = b + 4;
      }
      if (need1 > need0) {
        if (s.reg1 <= s.reg0) {
          s.reg0 = need0;
          s.reg1 = need1;
        } else {
          if (need0 < s.reg0) s.reg0 = need0;
          if (need1 > s.reg1) s.reg1 = need1;
        }
      }
      out[(size_t)l * n_tiles + i] = s;
      if (l > 0 && s.reg1 > s.reg0) {
        need0 = ofs[l][s.reg0];
        need1 = ofs[l][s.reg1 - 1] + 2;
      } else {
        need0 = need1 = 0;
      }
    }
  }
}

// lw / lh: the level sizes of ORB_Impl::detectAndCompute; edge >= 4. tile_w x tile_h: the level-0 tile of a workgroup.
static inline void orb_tile_plan(const int *lw, const int *lh, int n_levels, int edge, int tile_w, int tile_h, int lds_limit,
                                 OrbTilePlan *P) {
  P->ok = false;
  P->n_levels = n_levels;
  // (the arg-max key of orb_finish_kernel packs level | y | x into 4 + 14 + 14 bits)
  if (n_levels < 1 || n_levels > ORB_MAX_LEVELS || edge < 4 || lw[0] > 16383 || lh[0] > 16383) return;
  P->nx = (lw[0] + tile_w - 1) / tile_w;
  P->ny = (lh[0] + tile_h - 1) / tile_h;
  std::vector<int> ox[ORB_MAX_LEVELS], oy[ORB_MAX_LEVELS], cx, cy;
  for (int l = 1; l < n_levels; ++l) {
    orb_linear_exact_coeffs(lw[l - 1], lw[l], ox[l], cx);
    P->tabx[l].resize(ox[l].size());
    for (size_t k = 0; k < ox[l].size(); ++k) P->tabx[l][k] = (ox[l][k] << 16) | cx[k];
    orb_linear_exact_coeffs(lh[l - 1], lh[l], oy[l], cy);
    P->taby[l].resize(oy[l].size());
    for (size_t k = 0; k < oy[l].size(); ++k) P->taby[l][k] = (oy[l][k] << 16) | cy[k];
  }
  orb_plan_axis(lw, n_levels, edge, P->nx, ox, P->gx);
  orb_plan_axis(lh, n_levels, edge, P->ny, oy, P->gy);
  int off = 0, stash = 0;
  for (int l = 0; l < n_levels; ++l) {
    int rw = 0, rh = 0, ow = 0, oh = 0;
    for (int i = 0; i < P->nx; ++i) {
      const OrbSpan &s = P->gx[(size_t)l * P->nx + i];
      if (s.reg1 - s.reg0 > rw) rw = s.reg1 - s.reg0;
      if (s.own1 - s.own0 > ow) ow = s.own1 - s.own0;
    }
    for (int j = 0; j < P->ny; ++j) {
      const OrbSpan &s = P->gy[(size_t)l * P->ny + j];
      if (s.reg1 - s.reg0 > rh) rh = s.reg1 - s.reg0;
      if (s.own1 - s.own0 > oh) oh = s.own1 - s.own0;
    }
    P->lds_off[l] = off;
    P->lds_stride[l] = l ? (rw + 3) & ~3 : (rw + 15) & ~15;  // (level 0 is staged in 16-byte pieces)
    off += P->lds_stride[l] * rh;
    off = (off + 15) & ~15;
    P->tx_off[l] = P->ty_off[l] = 0;
    if (l) {
      P->tx_off[l] = off;
      off += 4 * rw;
      P->ty_off[l] = off;
      off += 4 * rh;
      off = (off + 15) & ~15;
    }
    P->sc_stride[l] = (ow + 2 + 3) & ~3;
    P->sc_off[l] = 0;
    stash += ((ow + 1) / 2) * ((oh + 1) / 2);  // strict 3x3 maxima in an ow x oh rectangle: at most one per 2x2 cell
    if (rw >= 32768 || rh >= 32768) return;
  }
  for (int l = 0; l < n_levels; ++l) {
    int oh = 0;
    for (int j = 0; j < P->ny; ++j) {
      const OrbSpan &s = P->gy[(size_t)l * P->ny + j];
      if (s.own1 - s.own0 > oh) oh = s.own1 - s.own0;
    }
    P->sc_off[l] = off;
    off += P->sc_stride[l] * (oh + 2);
    off = (off + 15) & ~15;
  }
  P->stash_off = off;
  P->stash_cap = stash + 16;
  // the same place first holds the list of pixels that pass FAST's compass-point test (4 bytes each, at most every pixel of
  // the score rectangles), then the candidates (8 bytes each)
  {
    int sc_px = 0;
    for (int l = 0; l < n_levels; ++l) {
      int ow = 0, oh = 0;
      for (int i = 0; i < P->nx; ++i) {
        const OrbSpan &s = P->gx[(size_t)l * P->nx + i];
        if (s.own1 - s.own0 > ow) ow = s.own1 - s.own0;
      }
      for (int j = 0; j < P->ny; ++j) {
        const OrbSpan &s = P->gy[(size_t)l * P->ny + j];
        if (s.own1 - s.own0 > oh) oh = s.own1 - s.own0;
      }
      if (ow > 0 && oh > 0) sc_px += (ow + 2) * (oh + 2);
    }
    const int need = 4 * sc_px > 8 * P->stash_cap ? 4 * sc_px : 8 * P->stash_cap;
    off += (need + 15) & ~15;
  }
  P->lds_bytes = off;
  P->ok = off <= lds_limit && stash < 65536;
}
