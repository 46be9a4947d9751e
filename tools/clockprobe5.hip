// clockprobe5.hip — diagnostic: the IC block reduction (4 sums over 5 wavefronts) in isolation.
#include "../visual_odometry_ros_amd/csrc/vo_internal.hpp"
#include <stdio.h>
struct Sh { float4 red[2][5]; float sred[2][5][4]; };
// MODE 0: as in ic_refine.hip   1: no barrier (timing only)   2: DPP only, no exchange
// MODE 3: exchange through ds_read_b32 broadcast of 20 floats by lanes 0..19 + readlane
template <int MODE>
__global__ void probe(unsigned long long *out, int iters, float seed) {
  __shared__ Sh sh;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  float x = seed + t * 1e-3f;
  int buf = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    float4 w = make_float4(x, x * 0.5f, x * x, 1.0f);
    wave_sum4_f32(w.x, w.y, w.z, w.w);
    float r;
    if (MODE == 2) {
      r = w.x + w.y + w.z + w.w;
    } else if (MODE == 3) {
      if (lane < 4) sh.sred[buf][wave][lane] = lane == 0 ? w.x : lane == 1 ? w.y : lane == 2 ? w.z : w.w;
      __syncthreads();
      const float *f = &sh.sred[buf][0][0];
      const float v = lane < 20 ? f[lane] : 0.f;   // lane = wave*4 + k
      // tree over waves for each k: lanes k, 4+k, 8+k, 12+k, 16+k
      const float a01 = __shfl(v, (lane & 3)) + __shfl(v, 4 + (lane & 3));
      const float a23 = __shfl(v, 8 + (lane & 3)) + __shfl(v, 12 + (lane & 3));
      const float a4 = __shfl(v, 16 + (lane & 3));
      const float s = (a01 + a23) + ((a4 + 0.f) + 0.f);
      r = __shfl(s, 0) + __shfl(s, 1) + __shfl(s, 2) + __shfl(s, 3);
      buf ^= 1;
    } else {
      if (lane == 0) sh.red[buf][wave] = w;
      if (MODE == 0) __syncthreads();
      const float4 r0 = sh.red[buf][0], r1 = sh.red[buf][1], r2 = sh.red[buf][2], r3 = sh.red[buf][3], r4 = sh.red[buf][4];
      const float v0 = ((r0.x + r1.x) + (r2.x + r3.x)) + ((r4.x + 0.0f) + 0.0f);
      const float v1 = ((r0.y + r1.y) + (r2.y + r3.y)) + ((r4.y + 0.0f) + 0.0f);
      const float v2 = ((r0.z + r1.z) + (r2.z + r3.z)) + ((r4.z + 0.0f) + 0.0f);
      const float v3 = ((r0.w + r1.w) + (r2.w + r3.w)) + ((r4.w + 0.0f) + 0.0f);
      r = v0 + v1 + v2 + v3;
      buf ^= 1;
    }
    x = r * 1e-6f + seed;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) out[0] = t1 - t0;
  if (x == 12345.f) out[1] = 1;
}
template <int MODE> void run(unsigned long long *d, const char *name, int threads) {
  unsigned long long h; const int iters = 5000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(threads), 0, 0, d, iters, 1.0f);
  hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-58s %4d threads: %.1f cycles / iteration\n", name, threads, (double)h / iters);
}
int main() {
  unsigned long long *d; hipMalloc(&d, 64);
  run<2>(d, "4-way DPP sums only", 64);
  run<2>(d, "4-way DPP sums only", 320);
  run<0>(d, "DPP + float4 LDS exchange + barrier (as in ic_refine)", 320);
  run<1>(d, "same without the barrier (timing only)", 320);
  run<0>(d, "DPP + float4 LDS exchange + barrier", 256);
  run<0>(d, "DPP + float4 LDS exchange + barrier", 64);
  run<3>(d, "exchange via b32 writes + shuffles", 320);
  return 0;
}
