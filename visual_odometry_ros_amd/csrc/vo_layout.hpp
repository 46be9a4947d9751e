// vo_layout.hpp — the padded pyramid-level layout and the REFLECT_101 index map, shared by every kernel file and — as
// plain C++ — by the CPU emulation harness of the tile kernels (tests/emu/). No HIP header is needed to read this file;
// `__host__` / `__device__` are the includer's (hip_runtime.h, or empty in the harness).
#pragma once
#include <stddef.h>
#include <stdint.h>

#define VO_PAD 40          // border (pixels) around every pyramid level: >= winSize + tile halo + 4-byte alignment slack (winSize <= 31)
#define VO_MAX_LEVELS 10

struct vo_level {
  uint8_t *base;   // start of the padded allocation
  int w, h;        // image size at this level
  int stride;      // bytes per padded row (multiple of 64)
  __host__ __device__ const uint8_t *origin() const { return base + (size_t)VO_PAD * stride + VO_PAD; }
};

// cv::BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), any distance
__host__ __device__ inline int vo_reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

// i / d and i % d for 0 <= i * d < 2^32 by a multiplication (m = 2^32 / d rounded up): the tile kernels run over
// rectangles of a few thousand elements whose width is only known at run time (a software division costs ~40 instructions)
__host__ __device__ inline unsigned vo_magic(int d) { return d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u; }
__device__ inline void vo_divmod(int i, int d, unsigned m, int &q, int &r) {
  q = d > 1 ? (int)__umulhi((unsigned)i, m) : i;
  r = i - q * d;
}

// 16 bytes from an address of any alignment (global memory takes unaligned dword accesses on gfx950; the compiler is told
// so by the packed type)
struct __attribute__((packed, aligned(1))) vo_u128_unaligned {
  uint32_t v[4];
};
struct __attribute__((aligned(16))) vo_u128 {
  uint32_t v[4];
};

// Four consecutive bytes at any byte offset of a 4-byte-aligned array, by two ALIGNED dword reads and a byte shift. For LDS:
// a multi-byte ds_read at a misaligned address (what the compiler makes of two adjacent byte loads) is several times slower
// on gfx950 than aligned reads — measured in tools/tileprobe.hip: the detector's resize phase 20 us with merged misaligned
// 16-bit reads, 11 us with separate byte reads. VO_ALIGNBYTE(hi, lo, n) = bytes n .. n + 3 of the 8 bytes (hi : lo): the
// includer's (v_alignbyte_b32 on the device).
__device__ inline uint32_t vo_bytes4(const uint8_t *base4, int off) {
  const uint32_t *w = (const uint32_t *)(base4 + (off & ~3));
  return VO_ALIGNBYTE(w[1], w[0], off & 3);
}
