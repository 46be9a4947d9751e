// clockprobe3.hip — diagnostic: cost of one s_barrier + LDS exchange round for a 5-wave workgroup,
// and of LDS byte reads, as seen by a single resident workgroup (latency, not throughput).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void probe(unsigned long long *out, int iters) {
  __shared__ float4 red[2][8];
  __shared__ unsigned char tile[4096];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 4096; i += blockDim.x) tile[i] = (unsigned char)i;
  __syncthreads();
  float v = t * 0.001f;
  int buf = 0, off = t;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // barrier + exchange
      if (lane == 0) red[buf][wave] = make_float4(v, v, v, v);
      __syncthreads();
      float4 a = red[buf][0], b = red[buf][1], c = red[buf][2], d = red[buf][3], e = red[buf][4];
      v = ((a.x + b.x) + (c.x + d.x)) + e.x * 1e-9f;
      buf ^= 1;
    } else if (MODE == 1) {  // barrier only
      __syncthreads();
      v = v * 1.0001f;
    } else if (MODE == 2) {  // 4 dependent-address LDS byte reads
      const unsigned char *q = tile + (off & 2047);
      int s = q[0] + q[1] + q[44] + q[45];
      off = off + s + 1;
      v += s;
    } else {  // 20 dependent float ops (a "solve")
#pragma unroll
      for (int k = 0; k < 20; ++k) v = v * 1.0001f + 0.5f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) out[0] = t1 - t0;
  if (v == 12345.f) out[1] = off;
}
template <int MODE> void run(unsigned long long *d, const char *name, int threads) {
  unsigned long long h; const int iters = 5000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(threads), 0, 0, d, iters);
  hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-44s %4d threads: %.1f cycles / iteration\n", name, threads, (double)h / iters);
}
int main() {
  unsigned long long *d; hipMalloc(&d, 64);
  run<0>(d, "LDS exchange + s_barrier", 320);
  run<0>(d, "LDS exchange + s_barrier", 256);
  run<1>(d, "s_barrier only", 320);
  run<1>(d, "s_barrier only", 64);
  run<2>(d, "4 LDS byte reads, dependent address", 64);
  run<3>(d, "40 dependent VALU", 64);
  return 0;
}
