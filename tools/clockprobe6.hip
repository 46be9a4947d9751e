// clockprobe6.hip — diagnostic: 1500 workgroups in which ONE wavefront runs a long dependent chain
// while the other four wait at a barrier; how are those busy wavefronts spread over the SIMDs?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>  // 0: wave 0 works   1: wave (blockIdx % 4) works   2: the wave on the least used SIMD? (n/a)
__global__ void probe(float *out, int iters, int *simd_hist) {
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int worker = MODE == 1 ? (blockIdx.x % (nw < 4 ? nw : 4)) : 0;
  float x = 1.0f + threadIdx.x * 1e-3f;
  if (wave == worker) {
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) {
      unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all bits
      const int simd = (hw >> 4) & 3;
      atomicAdd(&simd_hist[simd], 1);
    }
  }
  __syncthreads();
  if (x == 12345.f) out[0] = x;
}
template <int MODE> void run(float *d, int *h, const char *name, int blocks, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipMemset(h, 0, 16);
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, d, iters, h);
  hipDeviceSynchronize();
  hipMemset(h, 0, 16);
  hipEventRecord(e0); hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, d, iters, h); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int hh[4]; hipMemcpy(hh, h, 16, hipMemcpyDeviceToHost);
  printf("%-44s %5d x %3d: %8.1f us   busy waves per SIMD id: %d %d %d %d\n", name, blocks, threads, ms * 1000, hh[0], hh[1], hh[2], hh[3]);
}
int main() {
  float *d; int *h; hipMalloc(&d, 64); hipMalloc(&h, 64);
  run<0>(d, h, "one wave per WG (reference)", 1500, 64);
  run<0>(d, h, "one wave per WG", 256, 64);
  run<0>(d, h, "5 waves per WG, wave 0 works", 1500, 320);
  run<1>(d, h, "5 waves per WG, wave blockIdx%4 works", 1500, 320);
  run<0>(d, h, "5 waves per WG, wave 0 works", 256, 320);
  run<0>(d, h, "5 waves per WG, wave 0 works", 1024, 320);
  return 0;
}
