// clockprobe.hip — diagnostic: effective shader clock seen by short, latency-bound kernels.
// clock = delta(s_memtime) / delta(s_memrealtime) * 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ void probe(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;  // dependent chain
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (v == 12345.f) out[0] = 0;
}
int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 1024 * 16);
  for (int rep = 0; rep < 3; ++rep) {
    for (int mode = 0; mode < 3; ++mode) {
      // mode 0: one short kernel after 50 ms idle; 1: after 2000 back-to-back short kernels; 2: long kernel
      if (mode == 0) usleep(50000);
      if (mode == 1) for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(probe, dim3(1500), dim3(64), 0, 0, d, 2000);
      int iters = mode == 2 ? 2000000 : 20000;
      hipLaunchKernelGGL(probe, dim3(mode == 2 ? 1 : 1500), dim3(64), 0, 0, d, iters);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("mode %d: cycles %llu realticks %llu -> %.0f MHz, %.2f cycles/iter (dependent mul+add pair)\n", mode, h[0], h[1],
             100.0 * h[0] / (double)h[1], (double)h[0] / iters);
    }
  }
  return 0;
}
