// h2dprobe — what handing a PAGEABLE 1241 x 376 stereo pair to the device costs on this stack, four ways:
//   a  hipMemcpyAsync from the pageable arrays (what vo_set_stereo_pair_host_async does), then a kernel that reads the planes
//   b  memcpy into pinned memory only (the host's share of any staged path)
//   c  memcpy into pinned memory, then a kernel that reads the PINNED planes over PCIe (zero-copy)
//   d  hipMemcpyAsync from pinned memory, then the kernel
// build: hipcc --offload-arch=gfx950 -O2 tools/h2dprobe.hip -o /tmp/h2dprobe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void read_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, int n16, uint4 *__restrict__ out) {
  uint4 acc = {0, 0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
    const uint4 x = a[i], y = b[i];
    acc.x ^= x.x ^ y.x; acc.y ^= x.y ^ y.y; acc.z ^= x.z ^ y.z; acc.w ^= x.w ^ y.w;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t B = 1241 * 376, B16 = (B + 15) / 16;
  uint8_t *pl = (uint8_t *)malloc(B16 * 16), *pr = (uint8_t *)malloc(B16 * 16), *hl, *hr, *dl, *dr;
  uint4 *out;
  memset(pl, 1, B16 * 16); memset(pr, 2, B16 * 16);
  CK(hipHostMalloc((void **)&hl, B16 * 16, hipHostMallocDefault)); CK(hipHostMalloc((void **)&hr, B16 * 16, hipHostMallocDefault));
  CK(hipMalloc((void **)&dl, B16 * 16)); CK(hipMalloc((void **)&dr, B16 * 16)); CK(hipMalloc((void **)&out, 128 * 512 * 16));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int R = 300;
  for (int mode = 0; mode < 4; ++mode) {
    double t_issue = 0, t_all = 0;
    for (int r = -20; r < R; ++r) {
      pl[r & 1023] ^= 1;
      const double t0 = now();
      const uint4 *a = (const uint4 *)dl, *b = (const uint4 *)dr;
      if (mode == 0) { CK(hipMemcpyAsync(dl, pl, B, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(dr, pr, B, hipMemcpyHostToDevice, s)); }
      if (mode == 1 || mode == 2 || mode == 3) { memcpy(hl, pl, B); memcpy(hr, pr, B); }
      if (mode == 2) { a = (const uint4 *)hl; b = (const uint4 *)hr; }
      if (mode == 3) { CK(hipMemcpyAsync(dl, hl, B, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(dr, hr, B, hipMemcpyHostToDevice, s)); }
      if (mode != 1) read_kernel<<<128, 512, 0, s>>>(a, b, (int)B16, out);
      const double t1 = now();
      CK(hipStreamSynchronize(s));
      const double t2 = now();
      if (r >= 0) { t_issue += t1 - t0; t_all += t2 - t0; }
    }
    printf("mode %c: issue %.1f us, until the kernel has finished %.1f us\n", "abcd"[mode], t_issue / R * 1e6, t_all / R * 1e6);
  }
  return 0;
}
