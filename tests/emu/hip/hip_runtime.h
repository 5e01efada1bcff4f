// TEST INFRASTRUCTURE ONLY -- a CPU lane emulator for the kernels in torcheasyrec_amd/csrc.
//
// There is no GPU in the authoring container, so the logic of the HIP kernels (index math, scans,
// stable ranking, MFMA fragment maps) is exercised on CPU by compiling the *unchanged* .hip sources
// with the host clang against this header, which shadows <hip/hip_runtime.h>.  One fiber per lane
// on one OS thread, one workgroup at a time; __syncthreads() and the wave ops are cooperative
// barriers, wave ops exchange through a per-wave buffer.  Nothing in the product (`torcheasyrec_amd/`) can reach this file: the library it
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

// Scheduling: every lane of a workgroup is a FIBER (own stack, hand-rolled x86-64 context switch,
// tests/emu/abi_emu.cpp) and the whole workgroup runs on the calling OS thread.  A lane runs until it
// reaches a wave-level op or __syncthreads(), then yields round-robin; the last lane to arrive releases
// the barrier.  (The first version ran one OS thread per lane with std::barrier: every ballot was two
// 64-thread futex rounds, every __syncthreads() a 256-thread one -- the CPU suite spent 25 of its 30
// CPU-minutes in the kernel's futex code.)
extern "C" void tzr_emu_switch(void** save_sp, void* load_sp);

namespace emu {
static constexpr int kWave = 64;
static constexpr int kMaxThreads = 1024;
static constexpr size_t kStack = 256 * 1024;

struct Wave {
  int arrived = 0;
  unsigned gen = 0;
  int finished = 0;
  alignas(64) unsigned char buf[kWave][64];
};

struct Lane {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = true;
  uint3_emu tidx{0, 0, 0};
};

struct Sched {
  std::vector<Lane> lanes;
  std::vector<Wave> waves;
  int n = 0, cur = 0, finished = 0;
  int block_arrived = 0;
  unsigned block_gen = 0;
  void* main_sp = nullptr;
  std::function<void()> fn;
  uint3_emu bidx{0, 0, 0};
};
inline Sched& sched() { static Sched s; return s; }
inline dim3 g_blockDim, g_gridDim;

inline int lane() { return sched().cur % kWave; }
inline Wave& wave() { return sched().waves[sched().cur / kWave]; }

inline void switch_to(int next) {
  Sched& S = sched();
  const int prev = S.cur;
  S.cur = next;
  tzr_emu_switch(&S.lanes[prev].sp, S.lanes[next].sp);
}

// next unfinished lane after `from` inside [lo, hi), wrapping; -1 if `from` is the only one left
inline int next_lane(int from, int lo, int hi) {
  Sched& S = sched();
  for (int k = 1; k < hi - lo; ++k) {
    const int c = lo + (from - lo + k) % (hi - lo);
    if (!S.lanes[c].done) return c;
  }
  return -1;
}

inline void yield_in(int lo, int hi) {
  const int nx = next_lane(sched().cur, lo, hi);
  if (nx < 0) {
    fprintf(stderr, "emu: lane %d waits on a barrier no other lane can reach (divergent barrier)\n", sched().cur);
    abort();
  }
  switch_to(nx);
}

inline void wave_barrier() {
  Sched& S = sched();
  const int w = S.cur / kWave;
  Wave& W = S.waves[w];
  const unsigned g = W.gen;
  if (++W.arrived == kWave - W.finished) {
    W.arrived = 0;
    ++W.gen;
    return;
  }
  while (W.gen == g) yield_in(w * kWave, (w + 1) * kWave);
}

inline void block_barrier() {
  Sched& S = sched();
  const unsigned g = S.block_gen;
  if (++S.block_arrived == S.n - S.finished) {
    S.block_arrived = 0;
    ++S.block_gen;
    return;
  }
  while (S.block_gen == g) yield_in(0, S.n);
}

// first frame of every fiber
inline void lane_entry() {
  Sched& S = sched();
  S.fn();
  // this lane has left the kernel: barriers stop waiting for it
  Lane& L = S.lanes[S.cur];
  L.done = true;
  ++S.finished;
  Wave& W = S.waves[S.cur / kWave];
  ++W.finished;
  if (W.arrived > 0 && W.arrived == kWave - W.finished) { W.arrived = 0; ++W.gen; }
  if (S.block_arrived > 0 && S.block_arrived == S.n - S.finished) { S.block_arrived = 0; ++S.block_gen; }
  const int nx = next_lane(S.cur, 0, S.n);
  if (nx >= 0) {
    switch_to(nx);
  } else {
    void* dummy;
    tzr_emu_switch(&dummy, S.main_sp);
  }
  abort();  // a finished fiber is never resumed
}
extern "C" inline void tzr_emu_lane_entry_thunk() { lane_entry(); }

inline void prepare(int n) {
  Sched& S = sched();
  if ((int)S.lanes.size() < n) {
    const size_t old = S.lanes.size();
    S.lanes.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) S.lanes[i].stack = static_cast<char*>(aligned_alloc(64, kStack));
  }
  S.waves.assign((n + kWave - 1) / kWave, Wave{});
  S.n = n;
  S.finished = 0;
  S.block_arrived = 0;
  S.block_gen = 0;
  for (int i = 0; i < n; ++i) {
    Lane& L = S.lanes[i];
    L.done = false;
    // [r15 r14 r13 r12 rbx rbp][return address = entry thunk]; the slot of the return address is 16-byte
    // aligned so that the thunk starts with rsp = 8 (mod 16), like after a call
    uintptr_t top = (reinterpret_cast<uintptr_t>(L.stack) + kStack) & ~uintptr_t(15);
    void** ret = reinterpret_cast<void**>(top - 16);
    *ret = reinterpret_cast<void*>(&tzr_emu_lane_entry_thunk);
    void** regs = ret - 6;
    for (int k = 0; k < 6; ++k) regs[k] = nullptr;
    L.sp = regs;
  }
}

template <class F>
inline void launch(dim3 grid, dim3 blk, F&& f) {
  int n = blk.x * blk.y * blk.z;
  if (n % kWave != 0 || n > kMaxThreads) {
    fprintf(stderr, "emu: block size %d must be a multiple of 64 and <= 1024\n", n);
    abort();
  }
  const unsigned nblocks = grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  Sched& S = sched();
  g_blockDim = blk;
  g_gridDim = grid;
  S.fn = std::function<void()>(f);
  for (unsigned b = 0; b < nblocks; ++b) {
    prepare(n);
    S.bidx.x = b % grid.x;
    S.bidx.y = (b / grid.x) % grid.y;
    S.bidx.z = b / (grid.x * grid.y);
    for (int tid = 0; tid < n; ++tid) {
      S.lanes[tid].tidx.x = tid % blk.x;
      S.lanes[tid].tidx.y = (tid / blk.x) % blk.y;
      S.lanes[tid].tidx.z = tid / (blk.x * blk.y);
    }
    S.cur = 0;
    tzr_emu_switch(&S.main_sp, S.lanes[0].sp);  // returns when the last lane of the workgroup has finished
  }
}

template <class T>
inline T exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 64, "emu exchange payload too large");
  memcpy(wave().buf[lane()], &v, sizeof(T));
  wave_barrier();
  T r;
  memcpy(&r, wave().buf[src_lane & (kWave - 1)], sizeof(T));
  wave_barrier();
  return r;
}
}  // namespace emu

#define threadIdx (emu::sched().lanes[emu::sched().cur].tidx)
#define blockIdx (emu::sched().bidx)
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

// ---- host API of csrc/step_driver.hip (graphs, events, streams).  The emulator runs everything synchronously on the calling
// thread, so streams and events carry nothing; a "graph" is a recorded host function (tzr_emu_graph_create, abi_emu.cpp) that a
// launch calls.
typedef void* hipEvent_t;
struct hipGraphExecEmu {
  void (*fn)(void*);
  void* arg;
};
typedef hipGraphExecEmu* hipGraphExec_t;
static constexpr unsigned hipStreamNonBlocking = 1u, hipEventDisableTiming = 2u;
inline hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t) {
  if (!g || !g->fn) return 1;
  g->fn(g->arg);
  return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(1); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }

inline void __syncthreads() { emu::block_barrier(); }
// lanes are OS threads here: a wave-level barrier has to be a real one
inline void __builtin_amdgcn_wave_barrier() { emu::wave_barrier(); }
inline void __builtin_amdgcn_sched_barrier(int) {}  // a compiler scheduling fence: nothing to emulate
inline void __builtin_amdgcn_s_sleep(int) {}        // (workgroups run one after the other here: nobody to wait for)
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
  int l = emu::lane();
  unsigned char b = pred ? 1 : 0;
  emu::wave().buf[l][0] = b;
  emu::wave_barrier();
  unsigned long long m = 0;
  emu::Wave& w = emu::wave();
  // lanes that have left the kernel do not vote
  for (int i = 0; i < 64; ++i)
    if (!emu::sched().lanes[(emu::sched().cur / 64) * 64 + i].done) m |= (unsigned long long)w.buf[i][0] << i;
  emu::wave_barrier();
  return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return __ballot(pred) == ~0ull; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
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
  int l = emu::lane();
  float ab[2] = {a, b};
  memcpy(emu::wave().buf[l], ab, 8);
  emu::wave_barrier();
  emu::Wave& w = emu::wave();
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
  emu::wave_barrier();
  return d;
}
// v_mfma_f32_16x16x1_4b_f32: four independent 16 x 16 x 1 outer products.  Lane l supplies A[i=l&15] and B[j=l&15] of
// block l>>4; C/D: register 4 b + e of lane l is D_b[row=(l>>4)*4+e][col=l&15] (scripts/r04/probe_mfma_4b.hip reads the
// map off the device).
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
inline emu_f32x16 __builtin_amdgcn_mfma_f32_16x16x1f32(float a, float b, emu_f32x16 c, int, int, int) {
  int l = emu::lane();
  float ab[2] = {a, b};
  memcpy(emu::wave().buf[l], ab, 8);
  emu::wave_barrier();
  emu::Wave& w = emu::wave();
  emu_f32x16 d = c;
  int col = l & 15;
  for (int blk = 0; blk < 4; ++blk)
    for (int e = 0; e < 4; ++e) {
      int row = (l >> 4) * 4 + e;
      float av, bv;
      memcpy(&av, w.buf[blk * 16 + row], 4);
      memcpy(&bv, w.buf[blk * 16 + col] + 4, 4);
      d[4 * blk + e] = fmaf(av, bv, c[4 * blk + e]);
    }
  emu::wave_barrier();
  return d;
}
inline void __builtin_amdgcn_s_setprio(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return emu::exchange(v, 0); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
