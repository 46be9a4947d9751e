// emu_pyr.cpp — TEST INFRASTRUCTURE: csrc/pyr_tile.hpp's kernel on CPU threads. Reads a raw u8 image, builds the padded
// pyramid planes the way pyramid.hip lays them out and launches them (same chaining of launches beyond four levels),
// writes every padded plane to a file for tests/test_tile_kernels_emu.py to compare with the oracle.
//   emu_pyr <w> <h> <top> <in.raw> <out.bin> [nimg]
#include "hip_emu.h"

#include <stdio.h>
#include <stdlib.h>

#include "../../visual_odometry_ros_amd/csrc/pyr_tile.hpp"
#include "../../visual_odometry_ros_amd/csrc/pyr_plan.hpp"

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  const int w = atoi(argv[1]), h = atoi(argv[2]), top = atoi(argv[3]);
  const int nimg = argc > 6 ? atoi(argv[6]) : 1;
  std::vector<uint8_t> img[2];
  FILE *f = fopen(argv[4], "rb");
  if (!f) return 3;
  for (int i = 0; i < nimg; ++i) {
    img[i].resize((size_t)w * h);
    if (fread(img[i].data(), 1, img[i].size(), f) != img[i].size()) return 4;
  }
  fclose(f);
  // layout as pyramid.hip: layout_slot
  vo_level L[2][VO_MAX_LEVELS];
  std::vector<uint8_t> mem[2];
  int nlv = 0;
  for (int i = 0; i < nimg; ++i) {
    size_t off = 0;
    int lw = w, lh = h;
    std::vector<size_t> offs;
    for (int l = 0; l <= top && l < VO_MAX_LEVELS; ++l) {
      const int stride = ((lw + 2 * VO_PAD) + 63) & ~63;
      L[i][l].w = lw;
      L[i][l].h = lh;
      L[i][l].stride = stride;
      offs.push_back(off);
      off += (size_t)stride * (size_t)(lh + 2 * VO_PAD);
      off = (off + 255) & ~(size_t)255;
      lw = (lw + 1) / 2;
      lh = (lh + 1) / 2;
    }
    mem[i].assign(off + 256, 0xCD);
    uint8_t *base = (uint8_t *)(((uintptr_t)mem[i].data() + 255) & ~(uintptr_t)255);
    for (size_t l = 0; l < offs.size(); ++l) L[i][l].base = base + offs[l];
  }
  const uint8_t *src[2] = {img[0].data(), nimg > 1 ? img[1].data() : img[0].data()};
  nlv = pyr_plan_and_launch(L, nimg, src, w, top, /*base_written=*/false,
                            [&](const PyrTileArgs &a, int groups) { emu_launch(pyr_build_kernel, dim3(groups, 1, nimg), dim3(PYR_NT), a); });
  FILE *o = fopen(argv[5], "wb");
  fwrite(&nlv, sizeof(int), 1, o);
  for (int i = 0; i < nimg; ++i)
    for (int l = 0; l < nlv; ++l) {
      const int hdr[3] = {L[i][l].w, L[i][l].h, L[i][l].stride};
      fwrite(hdr, sizeof(int), 3, o);
      fwrite(L[i][l].base, 1, (size_t)L[i][l].stride * (L[i][l].h + 2 * VO_PAD), o);
    }
  fclose(o);
  return 0;
}
