// clockprobe2.hip — diagnostic: issue cost of dependent vs independent VALU chains for one lone wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCH, int UNROLL>
__global__ void probe(unsigned long long *out, int iters) {
  float v[NCH];
  for (int c = 0; c < NCH; ++c) v[c] = threadIdx.x + c;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[c] = v[c] * 1.0001f + 0.5f;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int c = 0; c < NCH; ++c) s += v[c];
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.f) out[1] = 0;
}
template <int NCH, int UNROLL> void run(unsigned long long *d, const char *name) {
  unsigned long long h;
  const int iters = 20000;
  hipLaunchKernelGGL((probe<NCH, UNROLL>), dim3(1), dim3(64), 0, 0, d, iters);
  hipLaunchKernelGGL((probe<NCH, UNROLL>), dim3(1), dim3(64), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%s: %.2f cycles per VALU instr (%d chains, unroll %d)\n", name, (double)h / iters / (2.0 * NCH * UNROLL), NCH, UNROLL);
}
int main() {
  unsigned long long *d; hipMalloc(&d, 64);
  run<1, 1>(d, "dep x1 (loop overhead each 2 instr)");
  run<1, 16>(d, "dep chain, unrolled 16");
  run<2, 16>(d, "2 chains");
  run<4, 16>(d, "4 chains");
  run<8, 8>(d, "8 chains");
  return 0;
}
