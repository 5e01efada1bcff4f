// Shared device helpers for the gfx950 kernels of libtzrec_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/tzrec_hip.h"
#include <tzr_gfx950.h>

#define TZR_WAVE 64

#define TZR_CHECK_LAUNCH()                         \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return TZR_ERR_LAUNCH; \
  } while (0)

static inline size_t tzr_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Carve a 256-byte aligned region out of a caller workspace.
struct TzrCarver {
  char* base;
  size_t off;
  explicit TzrCarver(void* p) : base(static_cast<char*>(p)), off(0) {}
  template <class T>
  T* take(size_t n) {
    T* r = reinterpret_cast<T*>(base + off);
    off = tzr_align_up(off + n * sizeof(T));
    return r;
  }
};

// Four consecutive table weights starting at element `off` of a row-major table whose elements
// are fp32 or fp16 (TzrTable.w_dtype); fp16 is widened exactly, stores round to nearest even.
__device__ __forceinline__ float4 tzr_ldw4(const void* w, int dtype, int64_t off) {
  if (dtype == TZR_DT_F16) {
    struct alignas(8) H4 { _Float16 a, b, c, d; };
    union { uint2 u; H4 h; } c;
    c.u = tzr_ldg8(static_cast<const char*>(w) + off * 2);
    return make_float4((float)c.h.a, (float)c.h.b, (float)c.h.c, (float)c.h.d);
  }
  return tzr_ldg4(static_cast<const float*>(w) + off);  // (table rows are device memory: global, not FLAT, instructions)
}
__device__ __forceinline__ void tzr_stw4(void* w, int dtype, int64_t off, float4 v) {
  if (dtype == TZR_DT_F16) {
    struct alignas(8) H4 { _Float16 a, b, c, d; };
    union { uint2 u; H4 h; } c;
    c.h.a = (_Float16)v.x; c.h.b = (_Float16)v.y; c.h.c = (_Float16)v.z; c.h.d = (_Float16)v.w;
    tzr_stg8(static_cast<char*>(w) + off * 2, c.u);
    return;
  }
  tzr_stg4(static_cast<float*>(w) + off, v);
}

__device__ __forceinline__ float4 tzr_ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void tzr_st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ float4 tzr_fma4(float s, float4 a, float4 acc) {
  acc.x = fmaf(s, a.x, acc.x);
  acc.y = fmaf(s, a.y, acc.y);
  acc.z = fmaf(s, a.z, acc.z);
  acc.w = fmaf(s, a.w, acc.w);
  return acc;
}
__device__ __forceinline__ float4 tzr_add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 tzr_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// upper_bound(a[0..n), v) - 1 for a non-decreasing int64 array with a[0] <= v: index of the last
// element <= v.
__device__ __forceinline__ int64_t tzr_last_le(const int64_t* a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;  // invariant: a[lo] <= v, (hi == n or a[hi] > v)
  while (hi - lo > 1) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}
