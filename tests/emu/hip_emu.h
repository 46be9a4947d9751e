// hip_emu.h — TEST INFRASTRUCTURE ONLY. Runs the code of a __global__ kernel on the CPU: one OS thread per HIP thread of
// ONE workgroup at a time, __syncthreads() = a pthread barrier, __shared__ = a function-local static (one workgroup runs
// at a time, so all of its threads see the same storage). It exists so that the index arithmetic of the tile kernels
// (csrc/pyr_tile.hpp, csrc/orb_tile.hpp) — regions, halos, REFLECT_101 mirror lists, ownership — can be checked against
// the oracle in the CPU suite, where no GPU is available. Nothing in the product includes this file, and nothing here is
// a fallback: the kernels are compiled for gfx950 by hipcc and run there; this header only lets g++ read the same text.
#pragma once
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define __forceinline__ inline

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 threadIdx;
static dim3 blockIdx, blockDim, gridDim;
static pthread_barrier_t emu_block_barrier;
static pthread_barrier_t emu_wave_barrier[64];  // one per wavefront of the workgroup (emu_wave_sum)
static int emu_wave_scratch[64 * 64];

static inline void __syncthreads() { pthread_barrier_wait(&emu_block_barrier); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
#define VO_ALIGNBYTE(hi, lo, n) ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint32_t)(lo)) >> (8 * ((n) & 3))))

template <class T>
static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T>
static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T>
static inline T atomicMax(T *p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return old;
}
static inline unsigned __float_as_uint(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float __uint_as_float(unsigned u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// sum over the 64 threads of the caller's wavefront (every thread of the wavefront must call it)
static inline int emu_wave_sum_i32(int v) {
  const int tid = (int)threadIdx.x, wave = tid >> 6;
  emu_wave_scratch[tid] = v;
  pthread_barrier_wait(&emu_wave_barrier[wave]);
  int s = 0;
  for (int l = 0; l < 64 && (wave << 6) + l < (int)blockDim.x; ++l) s += emu_wave_scratch[(wave << 6) + l];
  pthread_barrier_wait(&emu_wave_barrier[wave]);
  return s;
}

// launch: every workgroup of the grid in turn, its threads concurrently
template <class Kernel, class... Args>
static void emu_launch(Kernel k, dim3 grid, dim3 block, Args... args) {
  gridDim = grid;
  blockDim = block;
  const unsigned nt = block.x * block.y * block.z;
  pthread_barrier_init(&emu_block_barrier, nullptr, nt);
  for (unsigned w = 0; w < (nt + 63) / 64; ++w) pthread_barrier_init(&emu_wave_barrier[w], nullptr, (nt - 64 * w) < 64 ? nt - 64 * w : 64);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        std::vector<std::thread> th;
        th.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
          th.emplace_back([=]() {
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            k(args...);
          });
        for (auto &x : th) x.join();
      }
  pthread_barrier_destroy(&emu_block_barrier);
  for (unsigned w = 0; w < (nt + 63) / 64; ++w) pthread_barrier_destroy(&emu_wave_barrier[w]);
}

// ---- what csrc/orb_device.hpp / orb_tile.hpp ask their includer for -----------------------------------------------------
#include <math.h>
static inline int orb_wave_count(bool p) { return emu_wave_sum_i32(p ? 1 : 0); }
static int emu_wave_scratch2[64 * 64];
static inline int orb_wave_rank(bool p, int *n) {
  const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
  emu_wave_scratch[tid] = p ? 1 : 0;
  pthread_barrier_wait(&emu_wave_barrier[wave]);
  int r = 0, tot = 0;
  for (int l = 0; l < 64 && (wave << 6) + l < (int)blockDim.x; ++l) {
    if (l < lane) r += emu_wave_scratch[(wave << 6) + l];
    tot += emu_wave_scratch[(wave << 6) + l];
  }
  pthread_barrier_wait(&emu_wave_barrier[wave]);
  *n = tot;
  return r;
}
static inline int orb_wave_first(int v, bool p) {
  const int tid = (int)threadIdx.x, wave = tid >> 6;
  emu_wave_scratch[tid] = p ? 1 : 0;
  emu_wave_scratch2[tid] = v;
  pthread_barrier_wait(&emu_wave_barrier[wave]);
  int out = 0;
  for (int l = 0; l < 64 && (wave << 6) + l < (int)blockDim.x; ++l)
    if (emu_wave_scratch[(wave << 6) + l]) {
      out = emu_wave_scratch2[(wave << 6) + l];
      break;
    }
  pthread_barrier_wait(&emu_wave_barrier[wave]);
  return out;
}
static uint8_t emu_dyn_lds[160 * 1024] __attribute__((aligned(16)));
#define ORB_DYN_LDS(name) uint8_t *name = emu_dyn_lds
#define ORB_SET_PRIO()
static inline int __mul24(int a, int b) { return (int)((long long)((a << 8) >> 8) * ((b << 8) >> 8)); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
#define ORB_LD_AGENT(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define ORB_ST_AGENT(p, v) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define ORB_ATOMIC_INC_AGENT(p) __atomic_fetch_add((p), 1, __ATOMIC_SEQ_CST)
#define ORB_ATOMIC_ADD_AGENT(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define ORB_FENCE_RELEASE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define ORB_FENCE_ACQUIRE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
