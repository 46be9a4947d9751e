// clockprobe4.hip — diagnostic: pieces of the IC iteration in isolation, one resident workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float bilin(float I1, float I2, float I3, float I4, float ax, float ay, float axay) {
  return ((axay * (((I1 - I2) - I3) + I4) + ax * (-I1 + I2)) + ay * (-I1 + I3)) + I1;
}
template <int MODE>
__global__ void probe(unsigned long long *out, int iters, const unsigned char *img, int stride, int W, int H) {
  __shared__ unsigned int tile[440];
  const int t = threadIdx.x;
  for (int i = t; i < 440; i += blockDim.x) tile[i] = i * 2654435761u;
  __syncthreads();
  const unsigned char *sb = (const unsigned char *)tile;
  float pux = 600.3f + (t & 7), puy = 200.7f, scale = 1.1f, px = (float)((t % 23) - 11), py = (float)((t / 23) - 5);
  float acc = 0.f;
  const int x0 = 580, y0 = 181;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    float ax = pux - floorf(pux), ay = puy - floorf(puy), axay = ax * ay;
    if (MODE >= 1) { if (ax < 0 || ax > 1 || ay < 0 || ay > 1) break; if (isnan(ax + ay)) break; }
    const float uc = pux + px * scale, vc = puy + py * scale;
    const bool valid = !(uc < 1 || uc >= (float)(W - 2) || vc < 1 || vc >= (float)(H - 2));
    const int u0 = (int)uc, v0 = (int)vc;
    const int lx = u0 - x0, ly = v0 - y0;
    const bool inside = valid && (unsigned)lx < 43u && (unsigned)ly < 39u;
    const unsigned char *q = sb + (inside ? ly * 44 + lx : 0);
    float val = bilin((float)q[0], (float)q[1], (float)q[44], (float)q[45], ax, ay, axay);
    if (MODE >= 2) {
      if (valid && !inside) {
        const unsigned char *p = img + (ptrdiff_t)v0 * stride + u0;
        val = bilin((float)p[0], (float)p[1], (float)p[stride], (float)p[stride + 1], ax, ay, axay);
      }
    }
    acc += valid ? val : 0.f;
    pux += (val - 100.f) * 1e-6f;  // dependency to the next iteration
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) out[0] = t1 - t0;
  if (acc == 12345.f) out[1] = 1;
}
template <int MODE> void run(unsigned long long *d, const unsigned char *img, const char *name, int threads) {
  unsigned long long h; const int iters = 5000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(threads), 0, 0, d, iters, img, 1344, 1241, 376);
  hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-50s %4d threads: %.1f cycles / iteration\n", name, threads, (double)h / iters);
}
int main() {
  unsigned long long *d; hipMalloc(&d, 64);
  unsigned char *img; hipMalloc(&img, 1344 * 500); hipMemset(img, 7, 1344 * 500);
  run<0>(d, img, "frac + tap (LDS) + bilinear", 64);
  run<1>(d, img, " + range / NaN breaks", 64);
  run<2>(d, img, " + global fallback branch (not taken)", 64);
  run<2>(d, img, " + global fallback branch (not taken)", 320);
  return 0;
}
