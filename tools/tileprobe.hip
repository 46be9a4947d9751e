// tileprobe.hip — measurement only: the tile kernels of the ingestion chain (pyr_tile.hpp, orb_tile.hpp) ALONE on the GPU,
// launch time by HIP events and, built with -DTILE_STAMP, the phases of one workgroup (s_memrealtime, 10 ns ticks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DTILE_STAMP] -I visual_odometry_ros_amd/csrc tools/tileprobe.hip -o /tmp/tileprobe
//   /tmp/tileprobe [w h [frame.raw]]      (frame.raw: python -c "import bench; ..." — see tools/run_tileprobe.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#ifdef TILE_STAMP
__device__ unsigned long long g_stamp[64];
__device__ int g_stamp_block;
#define TILE_STAMP_AT(k)                                                                      \
  do {                                                                                        \
    if (threadIdx.x == 0 && (int)blockIdx.x == g_stamp_block && blockIdx.z == 0) g_stamp[k] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#endif

#define VO_ALIGNBYTE(hi, lo, n) __builtin_amdgcn_alignbyte((hi), (lo), (n))
#include "vo_layout.hpp"
__device__ __forceinline__ int orb_wave_count(bool p) { return __popcll(__ballot(p)); }
__device__ __forceinline__ int orb_wave_rank(bool p, int *n) {
  const unsigned long long m = __ballot(p);
  *n = __popcll(m);
  return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}
__device__ __forceinline__ int orb_wave_first(int v, bool p) {
  const unsigned long long m = __ballot(p);
  return m ? __shfl(v, __ffsll((long long)m) - 1) : 0;
}
#define ORB_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
#define ORB_SET_PRIO() __builtin_amdgcn_s_setprio(3)
#define ORB_LD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_ST_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_ATOMIC_INC_AGENT(p) __hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_ATOMIC_ADD_AGENT(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORB_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define ORB_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define PYR_SET_PRIO() __builtin_amdgcn_s_setprio(3)
#include "pyr_plan.hpp"
#include "orb_tile.hpp"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// what the shader clock is while this probe runs: a chain of dependent adds timed by the constant 100 MHz counter
// (s_memrealtime) and by the shader clock counter (s_memtime)
__global__ void clock_kernel(unsigned long long *out, int n) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  int v = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(v));
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = r1 - r0;
    out[1] = c1 - c0;
    out[2] = (unsigned long long)v;
  }
}
static void stamp_block(int b) {
#ifdef TILE_STAMP
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_block), &b, sizeof(int)));
#else
  (void)b;
#endif
}
static void print_stamps(const char *what, int n) {
#ifdef TILE_STAMP
  unsigned long long h[64];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h)));
  printf("%s phases of one workgroup (us):", what);
  for (int k = 1; k < n; ++k) printf(" %.2f", (double)(h[k] - h[k - 1]) * 0.01);
  printf("  | total %.2f", (double)(h[n - 1] - h[0]) * 0.01);
  if (n == 10) printf("  | FAST compass pass %.2f, full test %.2f", (double)(h[10] - h[5]) * 0.01, (double)(h[6] - h[10]) * 0.01);
  printf("\n");
#else
  (void)what;
  (void)n;
#endif
}

int main(int argc, char **argv) {
  const int w = argc > 2 ? atoi(argv[1]) : 1241, h = argc > 2 ? atoi(argv[2]) : 376, top = 4, nimg = 2;
  std::vector<uint8_t> img((size_t)w * h);
  srand(1);
  for (auto &p : img) p = (uint8_t)(rand() & 255);
  if (argc > 3) {  // a raw u8 image of that size (e.g. a frame of the bench's renderer: python tools/tileprobe_frame.py)
    FILE *fi = fopen(argv[3], "rb");
    if (!fi || fread(img.data(), 1, img.size(), fi) != img.size()) {
      fprintf(stderr, "cannot read %s\n", argv[3]);
      return 1;
    }
    fclose(fi);
  }
  uint8_t *d_img[2];
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&d_img[i], img.size()));
    CK(hipMemcpy(d_img[i], img.data(), img.size(), hipMemcpyHostToDevice));
  }
  vo_level L[2][VO_MAX_LEVELS];
  for (int i = 0; i < nimg; ++i) {
    size_t off = 0;
    int lw = w, lh = h;
    std::vector<size_t> offs;
    for (int l = 0; l <= top; ++l) {
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
    uint8_t *mem;
    CK(hipMalloc(&mem, off));
    for (size_t l = 0; l < offs.size(); ++l) L[i][l].base = mem + offs[l];
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  {
    unsigned long long *d_clk, h_clk[3];
    CK(hipMalloc(&d_clk, 64));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, st, d_clk, 25000);
      CK(hipMemcpy(h_clk, d_clk, sizeof(h_clk), hipMemcpyDeviceToHost));
      printf("clock probe: 100000 dependent v_add in %.1f us = %.2f ns each; shader-clock counter advanced %llu (%.2f per add; %.0f MHz if it counts shader cycles)\n",
             h_clk[0] * 0.01, h_clk[0] * 10.0 / 100000.0, h_clk[1], (double)h_clk[1] / 100000.0, (double)h_clk[1] / (h_clk[0] * 0.01));
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint8_t *src[2] = {d_img[0], d_img[1]};
  int groups = 0;
  auto pyr = [&]() {
    pyr_plan_and_launch(L, nimg, src, w, top, false, [&](const PyrTileArgs &a, int g) {
      groups = g;
      hipLaunchKernelGGL(pyr_build_kernel, dim3(g, 1, nimg), dim3(PYR_NT), 0, st, a);
    });
  };
  stamp_block(groups > 0 ? groups / 2 : 60);
  for (int k = 0; k < 5; ++k) pyr();
  CK(hipStreamSynchronize(st));
  stamp_block(groups / 2 + 3);
  CK(hipEventRecord(e0, st));
  const int reps = 200;
  for (int k = 0; k < reps; ++k) pyr();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("pyr_build_kernel %dx%d x%d images, %d workgroups per image: %.2f us per launch (back to back)\n", w, h, nimg, groups, 1e3 * ms / reps);
  print_stamps("pyr_build_kernel", 8);

  // ---- detector ----
  const int nl = 8, edge = 31, thr = 15, nfeatures = 10000, nbu = 60, nbv = 25;
  int lw[ORB_MAX_LEVELS], lh[ORB_MAX_LEVELS], quota[ORB_MAX_LEVELS];
  float lscale[ORB_MAX_LEVELS];
  orb_level_layout(w, h, nl, 1.2, nfeatures, lw, lh, lscale, quota);
  OrbTilePlan P;
  const int tile_w = getenv("TILE_W") ? atoi(getenv("TILE_W")) : 48, tile_h = getenv("TILE_H") ? atoi(getenv("TILE_H")) : 32;
  orb_tile_plan(lw, lh, nl, edge, tile_w, tile_h, 160 * 1024, &P);
  if (!P.ok) {
    printf("tile %d x %d: the plan does not fit\n", tile_w, tile_h);
    return 1;
  }
  if (P.lds_bytes > 64 * 1024) CK(hipFuncSetAttribute((const void *)orb_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes));
  if (!P.ok) {
    printf("plan does not fit\n");
    return 0;
  }
  const int cand_cap = (((size_t)w * h / 16 > 65536 ? (int)((size_t)w * h / 16) : 65536) + 15) & ~15, nbins = nbu * nbv;
  auto dalloc = [&](size_t bytes, const void *init) {
    void *p;
    CK(hipMalloc(&p, bytes));
    if (init)
      CK(hipMemcpy(p, init, bytes, hipMemcpyHostToDevice));
    else
      CK(hipMemset(p, 0, bytes));
    return p;
  };
  OrbTileArgs a;
  memset(&a, 0, sizeof(a));
  a.img = L[0][0].origin();
  a.img_end = L[0][0].base + (size_t)L[0][0].stride * (L[0][0].h + 2 * VO_PAD);
  a.stride = L[0][0].stride;
  a.n_levels = nl;
  a.nx = P.nx;
  a.ny = P.ny;
  a.fast_thr = thr;
  a.cand_cap = cand_cap;
  a.stash_off = P.stash_off;
  a.stash_cap = P.stash_cap;
  a.gx = (const OrbSpan *)dalloc(sizeof(OrbSpan) * P.gx.size(), P.gx.data());
  a.gy = (const OrbSpan *)dalloc(sizeof(OrbSpan) * P.gy.size(), P.gy.data());
  for (int l = 0; l < nl; ++l) {
    OrbTileLevel &T = a.L[l];
    T.w = lw[l];
    T.h = lh[l];
    T.lds_off = P.lds_off[l];
    T.lds_stride = P.lds_stride[l];
    T.sc_off = P.sc_off[l];
    T.sc_stride = P.sc_stride[l];
    T.cand_base = l * cand_cap;
    T.tx_off = P.tx_off[l];
    T.ty_off = P.ty_off[l];
    T.tabx = l ? (const int *)dalloc(sizeof(int) * P.tabx[l].size(), P.tabx[l].data()) : nullptr;
    T.taby = l ? (const int *)dalloc(sizeof(int) * P.taby[l].size(), P.taby[l].data()) : nullptr;
  }
  a.lvl_total = (int *)dalloc(sizeof(int) * nl, nullptr);
  a.hist_copies = getenv("HIST_COPIES") ? atoi(getenv("HIST_COPIES")) : (P.nx * P.ny / 64 < 1 ? 1 : (P.nx * P.ny / 64 > 64 ? 64 : P.nx * P.ny / 64));
  const size_t hist_bytes = sizeof(int) * 256 * ORB_MAX_LEVELS * (size_t)a.hist_copies;
  a.hist = (int *)dalloc(hist_bytes, nullptr);
  a.cx = (short *)dalloc(sizeof(short) * (size_t)cand_cap * nl, nullptr);
  a.cy = (short *)dalloc(sizeof(short) * (size_t)cand_cap * nl, nullptr);
  a.cs = (uint8_t *)dalloc((size_t)cand_cap * nl, nullptr);
  a.cr = (float *)dalloc(sizeof(float) * (size_t)cand_cap * nl, nullptr);
  OrbFinishArgs f;
  memset(&f, 0, sizeof(f));
  f.n_levels = nl;
  f.cand_cap = cand_cap;
  f.max_out = nfeatures + 4096;
  for (int l = 0; l < nl; ++l) {
    f.cand_base[l] = l * cand_cap;
    f.quota[l] = quota[l];
    f.scale[l] = lscale[l];
  }
  f.lvl_total = a.lvl_total;
  f.parts = getenv("FINISH_PARTS") ? atoi(getenv("FINISH_PARTS")) : (int)(((size_t)w * h / 40 + 8191) / 8192);
  if (f.parts < 1) f.parts = 1;
  if (f.parts > 32) f.parts = 32;
  f.cidx_cap = ORB_RC * ORB_ST;
  f.hist = a.hist;
  f.hist_copies = a.hist_copies;
  f.cidx = (int *)dalloc(sizeof(int) * (size_t)f.cidx_cap * ORB_MAX_LEVELS, nullptr);
  f.lvl_cnt = (int *)dalloc(sizeof(int) * 2 * ORB_MAX_LEVELS, nullptr);
  f.lvl_done = f.lvl_cnt + ORB_MAX_LEVELS;
  f.cx = a.cx;
  f.cy = a.cy;
  f.cs = a.cs;
  f.cr = a.cr;
  f.surv = (int *)dalloc(sizeof(int) * ORB_MAX_LEVELS, nullptr);
  f.done = (int *)dalloc(16, nullptr);
  f.key = (unsigned long long *)dalloc(8 * (size_t)(nbins + 1), nullptr);
  f.n_bins_u = nbu;
  f.n_bins_v = nbv;
  f.inv_u = 1.0f / (float)(w / nbu);
  f.inv_v = 1.0f / (float)(h / nbv);
  f.tab_xy = (float *)dalloc(8 * (size_t)nbins, nullptr);
  f.tab_has = (uint8_t *)dalloc(nbins, nullptr);
  int *hf;
  CK(hipHostMalloc(&hf, 64, hipHostMallocDefault));
  f.host_flags = hf;
  f.dev_flags = (int *)dalloc(16, nullptr);
  auto det = [&](int which) {
    if (which & 1) hipLaunchKernelGGL(orb_tile_kernel, dim3(a.nx * a.ny), dim3(ORB_TILE_NT), (size_t)P.lds_bytes, st, a);
    if (which & 2) hipLaunchKernelGGL(orb_finish_kernel, dim3(nl * f.parts), dim3(ORB_ST), 0, st, f);
  };
  for (int k = 0; k < 5; ++k) det(3);
  CK(hipStreamSynchronize(st));
  printf("detector: %d x %d tiles, %d bytes of LDS, keypoints %d, flags %d\n", P.nx, P.ny, P.lds_bytes, hf[1], hf[0]);
  CK(hipEventRecord(e0, st));
  for (int k = 0; k < reps; ++k) det(3);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("orb_tile_kernel + orb_finish_kernel: %.2f us per pair of launches (back to back)\n", 1e3 * ms / reps);
  // the tile kernel alone, back to back (its level counters keep growing: candidates past the lists' capacity are dropped, the
  // phases in front of the last one do the same work): is its code still in the instruction cache when nothing else runs between?
  CK(hipEventRecord(e0, st));
  for (int k = 0; k < reps; ++k) det(1);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("orb_tile_kernel alone, back to back: %.2f us per launch\n", 1e3 * ms / reps);
  CK(hipMemsetAsync(a.lvl_total, 0, sizeof(int) * nl, st));
  CK(hipMemsetAsync(a.hist, 0, hist_bytes, st));
  // each alone: the tile kernel needs its counters zeroed by the finish kernel, so time (tile + finish) - finish via a third run
  std::vector<int> totals(nl);
  stamp_block(a.nx * (a.ny / 2) + a.nx / 2);
  det(1);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(totals.data(), a.lvl_total, sizeof(int) * nl, hipMemcpyDeviceToHost));
  printf("candidates per level:");
  for (int l = 0; l < nl; ++l) printf(" %d", totals[l]);
  printf("\n");
  print_stamps("orb_tile_kernel", 10);
  stamp_block(0);
  det(2);
  CK(hipStreamSynchronize(st));
  print_stamps("orb_finish_kernel (level 0)", 4);
  // finish alone, repeatedly (the lists stay what they are: restore the totals in front of every launch)
  int *d_tot = (int *)dalloc(sizeof(int) * nl, totals.data());
  int *d_hist = (int *)dalloc(hist_bytes, nullptr);
  det(1);  // (the finish launch above zeroed the counters and the histogram: fill them again and keep a copy)
  CK(hipMemcpyAsync(d_hist, a.hist, hist_bytes, hipMemcpyDeviceToDevice, st));
  det(2);
  CK(hipEventRecord(e0, st));
  for (int k = 0; k < reps; ++k) {
    CK(hipMemcpyAsync(a.lvl_total, d_tot, sizeof(int) * nl, hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(a.hist, d_hist, hist_bytes, hipMemcpyDeviceToDevice, st));
    det(2);
  }
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("orb_finish_kernel x %d workgroups per level (+ two small copies): %.2f us per launch\n", f.parts, 1e3 * ms / reps);
  return 0;
}
