// emu_orb.cpp — TEST INFRASTRUCTURE: csrc/orb_tile.hpp's two kernels on CPU threads, set up as orb_detect.hip's
// orb_tile_enqueue sets them up (same planner, same argument blocks). Writes the per-bin table, flags, the keypoint count
// and the per-level candidate lists for tests/test_tile_kernels_emu.py to compare with the oracle.
//   emu_orb <w> <h> <n_levels> <scale_factor> <nfeatures> <edge> <fast_thr> <n_bins_u> <n_bins_v> <inv_u hex> <inv_v hex> <in.raw> <out.bin> [tile_w tile_h [parts cidx_cap [hist_copies]]]
#include "hip_emu.h"

#include <stdio.h>
#include <stdlib.h>

#include "../../visual_odometry_ros_amd/csrc/orb_tile.hpp"

int main(int argc, char **argv) {
  if (argc < 14) return 2;
  const int w = atoi(argv[1]), h = atoi(argv[2]), nl = atoi(argv[3]);
  const double sf = atof(argv[4]);
  const int nfeatures = atoi(argv[5]), edge = atoi(argv[6]), thr = atoi(argv[7]), nbu = atoi(argv[8]), nbv = atoi(argv[9]);
  const unsigned iu_bits = (unsigned)strtoul(argv[10], nullptr, 16), iv_bits = (unsigned)strtoul(argv[11], nullptr, 16);
  const int tw = argc > 14 ? atoi(argv[14]) : 48, th = argc > 15 ? atoi(argv[15]) : 32;
  const int parts = argc > 16 ? atoi(argv[16]) : 1, cidx_cap = argc > 17 ? atoi(argv[17]) : ORB_RC * ORB_ST,
            copies = argc > 18 ? atoi(argv[18]) : 1;
  std::vector<uint8_t> img((size_t)w * h);
  FILE *f = fopen(argv[12], "rb");
  if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
  fclose(f);
  int lw[ORB_MAX_LEVELS], lh[ORB_MAX_LEVELS], quota[ORB_MAX_LEVELS];
  float lscale[ORB_MAX_LEVELS];
  orb_level_layout(w, h, nl, sf, nfeatures, lw, lh, lscale, quota);
  OrbTilePlan P;
  orb_tile_plan(lw, lh, nl, edge, tw, th, (tw > 48 || th > 32 ? 96 : 64) * 1024, &P);  // (as orb_prepare: larger tiles may take more)
  FILE *o = fopen(argv[13], "wb");
  const int ok = P.ok ? 1 : 0;
  fwrite(&ok, sizeof(int), 1, o);
  if (!P.ok) {
    fclose(o);
    return 0;
  }
  const int cand_cap = ((w * (size_t)h / 16 > 65536) ? (int)(w * (size_t)h / 16) : 65536) + 15 & ~15;
  const int max_out = nfeatures + 4096, nbins = nbu * nbv;
  std::vector<int> lvl_total(nl, 0), surv(ORB_MAX_LEVELS, 0), done(4, 0), devflags(4, 0), hostflags(16, 0);
  std::vector<int> hist(256 * (size_t)nl * copies, 0), cidx((size_t)cidx_cap * nl, -7), lvl_cnt(nl, 0), lvl_done(nl, 0);
  std::vector<short> cx((size_t)cand_cap * nl), cy((size_t)cand_cap * nl);
  std::vector<uint8_t> cs((size_t)cand_cap * nl), has(nbins, 0xEE);
  std::vector<float> cr((size_t)cand_cap * nl), xy(2 * (size_t)nbins, -1.f);
  std::vector<unsigned long long> key(nbins + 1, 0ull);
  OrbTileArgs a;
  memset(&a, 0, sizeof(a));
  a.img = img.data();
  a.img_end = img.data() + img.size();
  a.stride = w;
  a.n_levels = nl;
  a.nx = P.nx;
  a.ny = P.ny;
  a.fast_thr = thr;
  a.cand_cap = cand_cap;
  a.stash_off = P.stash_off;
  a.stash_cap = P.stash_cap;
  a.gx = P.gx.data();
  a.gy = P.gy.data();
  for (int l = 0; l < nl; ++l) {
    OrbTileLevel &L = a.L[l];
    L.w = lw[l];
    L.h = lh[l];
    L.lds_off = P.lds_off[l];
    L.lds_stride = P.lds_stride[l];
    L.sc_off = P.sc_off[l];
    L.sc_stride = P.sc_stride[l];
    L.cand_base = l * cand_cap;
    L.tx_off = P.tx_off[l];
    L.ty_off = P.ty_off[l];
    L.tabx = l ? P.tabx[l].data() : nullptr;
    L.taby = l ? P.taby[l].data() : nullptr;
  }
  a.lvl_total = lvl_total.data();
  a.hist = hist.data();
  a.hist_copies = copies;
  a.cx = cx.data();
  a.cy = cy.data();
  a.cs = cs.data();
  a.cr = cr.data();
  if (P.lds_bytes > (int)sizeof(emu_dyn_lds)) return 5;
  emu_launch(orb_tile_kernel, dim3(a.nx * a.ny), dim3(ORB_TILE_NT), a);
  const std::vector<int> totals = lvl_total;
  OrbFinishArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.n_levels = nl;
  fa.cand_cap = cand_cap;
  fa.max_out = max_out;
  for (int l = 0; l < nl; ++l) {
    fa.cand_base[l] = l * cand_cap;
    fa.quota[l] = quota[l];
    fa.scale[l] = lscale[l];
  }
  fa.lvl_total = lvl_total.data();
  fa.parts = parts;
  fa.cidx_cap = cidx_cap;
  fa.hist = hist.data();
  fa.hist_copies = copies;
  fa.cidx = cidx.data();
  fa.lvl_cnt = lvl_cnt.data();
  fa.lvl_done = lvl_done.data();
  fa.cx = cx.data();
  fa.cy = cy.data();
  fa.cs = cs.data();
  fa.cr = cr.data();
  fa.surv = surv.data();
  fa.done = done.data();
  fa.key = key.data();
  fa.n_bins_u = nbu;
  fa.n_bins_v = nbv;
  memcpy(&fa.inv_u, &iu_bits, 4);
  memcpy(&fa.inv_v, &iv_bits, 4);
  fa.tab_xy = xy.data();
  fa.tab_has = has.data();
  fa.host_flags = hostflags.data();
  fa.dev_flags = devflags.data();
  emu_launch(orb_finish_kernel, dim3(nl * parts), dim3(ORB_ST), fa);
  // what must be left behind for the next image: zeroed counters and keys
  int dirty = done[0] != 0;
  for (int l = 0; l < nl; ++l) dirty |= lvl_total[l] != 0 || lvl_cnt[l] != 0 || lvl_done[l] != 0;
  for (size_t k = 0; k < hist.size(); ++k) dirty |= hist[k] != 0;
  for (int j = 0; j < nbins; ++j) dirty |= key[j] != 0ull;
  const int hdr[8] = {hostflags[0], hostflags[1], dirty, P.nx, P.ny, P.lds_bytes, nbins, cand_cap};
  fwrite(hdr, sizeof(int), 8, o);
  fwrite(xy.data(), sizeof(float), xy.size(), o);
  fwrite(has.data(), 1, has.size(), o);
  fwrite(totals.data(), sizeof(int), nl, o);
  for (int l = 0; l < nl; ++l) {
    const int n = totals[l] < cand_cap ? totals[l] : cand_cap;
    fwrite(&cx[(size_t)l * cand_cap], sizeof(short), n, o);
    fwrite(&cy[(size_t)l * cand_cap], sizeof(short), n, o);
    fwrite(&cs[(size_t)l * cand_cap], 1, n, o);
    fwrite(&cr[(size_t)l * cand_cap], sizeof(float), n, o);
  }
  fclose(o);
  return 0;
}
