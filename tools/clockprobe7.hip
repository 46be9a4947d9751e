// clockprobe7.hip — the cross-lane reduction of the KLT iteration (two exact int32 sums over 64 lanes), DPP butterfly vs
// an MFMA "ones-matrix" reduction. An MFMA sums along K, and K holds the data of only a few lanes: v_mfma_i32_16x16x32_i8
// takes 8 bytes per lane with K = 32, i.e. one instruction adds the bytes of 4 lanes (lanes l, l+16, l+32, l+48 feed the same
// column). 64 -> 1 therefore needs three dependent stages (4 x 4 x 4) with a re-layout of the int32 partial sums into bytes of
// other lanes in between (limbs of 7 bits so that int8 products cannot overflow). This probe times ONE such stage — split
// two int32 partials into 7-bit limbs, one MFMA against a limb-selecting 0/1 matrix, recombine — against the WHOLE
// two-sum DPP butterfly the kernel uses (wave_sum2_i32_to_f32's int path).
//   hipcc --offload-arch=gfx950 -O3 tools/clockprobe7.hip -o tools/clockprobe7 && tools/clockprobe7
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

__device__ __forceinline__ void dpp_sum2(int &p, int &q) {
#define ST(E1, E2) { const int a = E1, b = E2; p += a; q += b; }
  ST(dpp<0xB1>(p), dpp<0xB1>(q))
  ST(dpp<0x4E>(p), dpp<0x4E>(q))
  ST(dpp<0x141>(p), dpp<0x141>(q))
  ST(dpp<0x140>(p), dpp<0x140>(q))
  ST(__builtin_amdgcn_update_dpp(0, p, 0x142, 0xA, 0xF, false), __builtin_amdgcn_update_dpp(0, q, 0x142, 0xA, 0xF, false))
  ST(__builtin_amdgcn_update_dpp(0, p, 0x143, 0xC, 0xF, false), __builtin_amdgcn_update_dpp(0, q, 0x143, 0xC, 0xF, false))
#undef ST
  p = __builtin_amdgcn_readlane(p, 63);
  q = __builtin_amdgcn_readlane(q, 63);
}

// one 4:1 stage: limbs of p (4 x 7 bit, biased to unsigned) and q in the 8 bytes of a lane; A = limb selector
__device__ __forceinline__ void mfma_stage(int &p, int &q, long sel) {
  const unsigned up = (unsigned)p + (1u << 27), uq = (unsigned)q + (1u << 27);  // |partial| < 2^27: bias to non-negative
  const unsigned bp = (up & 0x7F) | ((up >> 7 & 0x7F) << 8) | ((up >> 14 & 0x7F) << 16) | ((up >> 21 & 0x7F) << 24);
  const unsigned bq = (uq & 0x7F) | ((uq >> 7 & 0x7F) << 8) | ((uq >> 14 & 0x7F) << 16) | ((uq >> 21 & 0x7F) << 24);
  const long b = (long)bp | ((long)bq << 32);
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(sel, b, acc, 0, 0, 0);
  // rows 0..3 of the lane's accumulator hold four limb sums of its column: recombine (what the next stage would re-limb)
  p = acc[0] + (acc[1] << 7) + (acc[2] << 14) + (acc[3] << 21);
  q = p ^ acc[3];
}

__global__ void probe(int *out, long long *cyc, int reps) {
  int p = threadIdx.x * 977 - 31000, q = 12345 - threadIdx.x * 413;
  const long sel = 0x0101010101010101L >> (threadIdx.x & 7);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    int a = p + r, b = q - r;
    dpp_sum2(a, b);
    p += a & 3;
    q += b & 3;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    int a = p + r, b = q - r;
    mfma_stage(a, b, sel);
    p += a & 3;
    q += b & 3;
  }
  long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    cyc[0] = t1 - t0;
    cyc[1] = t2 - t1;
  }
  out[threadIdx.x] = p + q;
}

int main() {
  int *o;
  long long *c, h[2];
  hipMalloc(&o, 256);
  hipMalloc(&c, 16);
  const int reps = 4096;
  for (int k = 0; k < 3; ++k) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o, c, reps);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
  // s_memtime counts shader clocks on gfx9-family parts
  printf("two-sum DPP butterfly (64 -> 1, both sums): %.0f shader cycles per reduction (loop overhead included)\n", (double)h[0] / reps);
  printf("ONE MFMA 4:1 stage incl. limb split / recombine: %.0f cycles; 64 -> 1 needs three such stages plus two re-layouts\n",
         (double)h[1] / reps);
  return 0;
}
