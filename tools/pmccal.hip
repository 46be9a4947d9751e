// pmccal.hip — calibration for the FETCH_SIZE / WRITE_SIZE counters on the access pattern the
// tracker kernels use: one dword per lane, 64 lanes reading one 256-byte run (tile rows are
// row-contiguous dword loads). Reads and writes a known number of bytes exactly once.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void cal_read_dword(const unsigned *src, unsigned *sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void cal_write_dword(unsigned *dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (unsigned)i;
}
int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: well past the 256 MiB Infinity Cache
  unsigned *a, *s;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&s, 64) != hipSuccess) return 1;
  hipMemset(a, 1, bytes);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_read_dword, dim3(4096), dim3(256), 0, 0, a, s, bytes / 4);
  hipLaunchKernelGGL(cal_write_dword, dim3(4096), dim3(256), 0, 0, a, bytes / 4);
  hipDeviceSynchronize();
  printf("calibration: each kernel moves %zu bytes\n", bytes);
  return 0;
}
