// TEST INFRASTRUCTURE ONLY -- a CPU lane emulator for the kernels in torcheasyrec_amd/csrc.
//
// There is no GPU in the authoring container, so the logic of the HIP kernels (index math, scans,
// stable ranking, MFMA fragment maps) is exercised on CPU by compiling the *unchanged* .hip sources
// with the host clang against this header, which shadows <hip/hip_runtime.h>.  One OS thread per
// lane, one workgroup at a time; __syncthreads() is a real barrier, wave ops exchange through a
// per-wave buffer.  Nothing in the product (`torcheasyrec_amd/`) can reach this file: the library it
// produces reports tzr_backend() == "emu" and is loaded only by tests/ through an explicit path.
// It is not a fallback and is never timed.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cassert>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define TZR_EMU 1
using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
static constexpr hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

namespace emu {
static constexpr int kWave = 64;
static constexpr int kMaxThreads = 1024;

struct Wave {
  std::barrier<> bar{kWave};
  alignas(64) unsigned char buf[kWave][64];
  unsigned long long mask;
};

struct Pool {
  std::vector<std::thread> threads;
  std::unique_ptr<std::barrier<>> start, done, block;
  std::vector<std::unique_ptr<Wave>> waves;
  int nthreads = 0;
  std::function<void()> fn;
  dim3 grid, blk;
  unsigned cur_block = 0;
  bool quit = false;
};
inline Pool& pool() { static Pool p; return p; }

inline thread_local uint3_emu t_threadIdx, t_blockIdx;
inline thread_local int t_tid = 0;
inline dim3 g_blockDim, g_gridDim;

inline void worker(int tid) {
  Pool& p = pool();
  t_tid = tid;
  for (;;) {
    p.start->arrive_and_wait();
    if (p.quit) return;
    unsigned nblocks = p.grid.x * p.grid.y * p.grid.z;
    for (unsigned b = 0; b < nblocks; ++b) {
      t_blockIdx.x = b % p.grid.x;
      t_blockIdx.y = (b / p.grid.x) % p.grid.y;
      t_blockIdx.z = b / (p.grid.x * p.grid.y);
      t_threadIdx.x = tid % p.blk.x;
      t_threadIdx.y = (tid / p.blk.x) % p.blk.y;
      t_threadIdx.z = tid / (p.blk.x * p.blk.y);
      p.fn();
      p.block->arrive_and_wait();
    }
    p.done->arrive_and_wait();
  }
}

inline void shutdown() {
  Pool& p = pool();
  if (p.nthreads == 0) return;
  p.quit = true;
  p.start->arrive_and_wait();
  for (auto& t : p.threads) t.join();
  p.threads.clear();
  p.nthreads = 0;
  p.quit = false;
}

inline void ensure(int n) {
  Pool& p = pool();
  if (p.nthreads == n) return;
  shutdown();
  p.nthreads = n;
  p.start = std::make_unique<std::barrier<>>(n + 1);
  p.done = std::make_unique<std::barrier<>>(n + 1);
  p.block = std::make_unique<std::barrier<>>(n);
  p.waves.clear();
  for (int w = 0; w < (n + kWave - 1) / kWave; ++w) p.waves.emplace_back(std::make_unique<Wave>());
  for (int i = 0; i < n; ++i) p.threads.emplace_back(worker, i);
  static bool reg = false;
  if (!reg) { reg = true; atexit(shutdown); }
}

template <class F>
inline void launch(dim3 grid, dim3 blk, F&& f) {
  int n = blk.x * blk.y * blk.z;
  if (n % kWave != 0 || n > kMaxThreads) {
    fprintf(stderr, "emu: block size %d must be a multiple of 64 and <= 1024\n", n);
    abort();
  }
  if (grid.x * grid.y * grid.z == 0) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  ensure(n);
  Pool& p = pool();
  p.grid = grid; p.blk = blk; g_blockDim = blk; g_gridDim = grid;
  p.fn = std::function<void()>(f);
  p.start->arrive_and_wait();
  p.done->arrive_and_wait();
}

inline Wave& wave() { return *pool().waves[t_tid / kWave]; }
inline int lane() { return t_tid % kWave; }

template <class T>
inline T exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 64, "emu exchange payload too large");
  Wave& w = wave();
  memcpy(w.buf[lane()], &v, sizeof(T));
  w.bar.arrive_and_wait();
  T r;
  memcpy(&r, w.buf[src_lane & (kWave - 1)], sizeof(T));
  w.bar.arrive_and_wait();
  return r;
}
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }

inline void __syncthreads() { emu::pool().block->arrive_and_wait(); }
// lanes are OS threads here: a wave-level barrier has to be a real one
inline void __builtin_amdgcn_wave_barrier() { emu::wave().bar.arrive_and_wait(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }

// ---- wave-level (64 lanes) --------------------------------------------------------------------
template <class T> inline T __shfl(T v, int src, int width = 64) {
  int l = emu::lane();
  return emu::exchange(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) {
  int l = emu::lane();
  int s = l ^ m;
  if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return emu::exchange(v, s);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = emu::lane();
  int s = l + (int)d;
  if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return emu::exchange(v, s);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = emu::lane();
  int s = l - (int)d;
  if (s < 0 || (s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return emu::exchange(v, s);
}
inline unsigned long long __ballot(int pred) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  unsigned char b = pred ? 1 : 0;
  w.buf[l][0] = b;
  w.bar.arrive_and_wait();
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) m |= (unsigned long long)w.buf[i][0] << i;
  w.bar.arrive_and_wait();
  return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return __ballot(pred) == ~0ull; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }

// ---- atomics -----------------------------------------------------------------------------------
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  unsigned* up = reinterpret_cast<unsigned*>(p);
  unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED);
  for (;;) {
    float f; memcpy(&f, &old, 4); f += v;
    unsigned nw; memcpy(&nw, &f, 4);
    if (__atomic_compare_exchange_n(up, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      float r; memcpy(&r, &old, 4); return r;
    }
  }
}
template <class T> inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T> inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicCAS(T* p, T cmp, T v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}

// ---- vector types --------------------------------------------------------------------------------
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

inline float __fsqrt_rn(float x) { return sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }

// ---- gfx950 builtins used by the kernels ------------------------------------------------------
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// C/D: reg r of lane l is D[row=(l>>4)*4+r][col=l&15]; exact f32, k-ordered fmaf chain
// (cdna_hip_programming.md section 3).
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  emu::Wave& w = emu::wave();
  int l = emu::lane();
  float ab[2] = {a, b};
  memcpy(w.buf[l], ab, 8);
  w.bar.arrive_and_wait();
  emu_f32x4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, w.buf[k * 16 + row], 4);
      memcpy(&bv, w.buf[k * 16 + col] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  w.bar.arrive_and_wait();
  return d;
}
inline int __builtin_amdgcn_readfirstlane(int v) { return emu::exchange(v, 0); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
