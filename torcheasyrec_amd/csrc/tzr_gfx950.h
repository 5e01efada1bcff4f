// Hand-off of a few words between workgroups INSIDE one launch on gfx950 (8 XCDs, private L2s, a
// CU's vector L1 never refreshed by other CUs' stores).  A release fence (`buffer_wbl2 sc1`) would
// write back every dirty line of the XCD's L2 -- inside the reduce kernel that is the row updates
// of every workgroup on the XCD: folding the stitch step with __threadfence() made the kernel 2.3x
// slower (profiles/r02c).  Instead, the protocol of cdna_hip_programming.md guideline 16 (R1):
//   producer: payload with write-through stores -> every storing wave drains its vector memory
//             counter -> ONE lane's agent-scope counter increment;
//   consumer: the arrival that completes the count reads the payload with loads that bypass its
//             caches: no fence on either side.
// Included as <tzr_gfx950.h>: the CPU lane emulator of tests/ shadows this one file (everything in
// here is a gfx950 instruction or attribute with no meaning on a host CPU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(1))) uint32_t tzr_gu32;
typedef __attribute__((address_space(1))) uint64_t tzr_gu64;

// Agent scope (sc1) on both sides (guideline 16: sc1 loads may replace the consumer's acquire when the
// producer stored sc1).
__device__ __forceinline__ void tzr_publish_u32(uint32_t* p, uint32_t v) {
  __hip_atomic_store((tzr_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tzr_publish_u64(uint64_t* p, uint64_t v) {
  __hip_atomic_store((tzr_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t tzr_consume_u32(const uint32_t* p) {
  return __hip_atomic_load((tzr_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t tzr_consume_u64(const uint64_t* p) {
  return __hip_atomic_load((tzr_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wave that published calls this before the arrival is counted
__device__ __forceinline__ void tzr_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// returns the number of arrivals before this one
__device__ __forceinline__ uint32_t tzr_arrive(uint32_t* counter) {
  return __hip_atomic_fetch_add((tzr_gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Kernel attribute: compile for exactly `n` waves per SIMD (caps the VGPR budget at 512 / n).
#define TZR_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))

// Workgroup barrier for hand-offs through LDS that leaves the vector-memory counter alone.  __syncthreads() is a
// release fence as well: with a global store in flight hipcc emits s_waitcnt vmcnt(0) in front of s_barrier, and on
// gfx9 loads share that counter -- in a persistent loop every prefetch and every output store would land on the next
// barrier.  Only for kernels whose waves exchange nothing through global memory.
__device__ __forceinline__ void tzr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Make a lane value opaque to the optimiser at this point.  Used inside persistent loops on cheap, loop-invariant lane
// arithmetic (an LDS offset): hipcc otherwise hoists it out of the loop, runs out of registers and SPILLS it -- and the
// reload inside the loop comes with s_waitcnt vmcnt(0), which also waits for every prefetch in flight.
#define TZR_OPAQUE(x) asm volatile("" : "+v"(x))

// Wave priority of a workgroup by the residency slot it will take: with a grid of `per_round` workgroups resident at once per
// round-robin pass of the dispatcher (256 CUs), workgroup i shares its CU with i +- 256 k.  Co-resident workgroups that start
// together and run the same phases otherwise stay in lock step -- every one of them waiting on memory, then every one sorting in
// LDS; different priorities let them drift apart (measured on the cells apply: pooled_bwd_cells.hip).
__device__ __forceinline__ void tzr_prio_by_slot(unsigned block) {
  switch ((block >> 8) & 3u) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
}

// max(x, 0) as ONE instruction (fmaxf is two: it quiets a signalling NaN first; a NaN comes out as 0 here)
__device__ __forceinline__ float tzr_relu(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}


// Loads / stores through a pointer that is KNOWN to be device memory.  hipcc only knows that of a kernel's own pointer
// arguments; a pointer read out of a descriptor (TzrTable.w, a destination list, an LDS copy of either) is generic, and
// its accesses become FLAT instructions -- which count in the LDS counter too: every `s_waitcnt lgkmcnt(0)` in front of an
// LDS read then also waits for all the row gathers in flight.
#define TZR_GLOBAL_AS __attribute__((address_space(1)))
// (through builtin vector types: a HIP float4 is a struct whose copy operators take a generic `this` -- FLAT again)
typedef float tzr_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tzr_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tzr_ldg(const float* p) { return *(const TZR_GLOBAL_AS float*)p; }
__device__ __forceinline__ void tzr_stg(float* p, float v) { *(TZR_GLOBAL_AS float*)p = v; }
__device__ __forceinline__ float4 tzr_ldg4(const float* p) {
  const tzr_f32x4 v = *(const TZR_GLOBAL_AS tzr_f32x4*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void tzr_stg4(float* p, float4 v) { *(TZR_GLOBAL_AS tzr_f32x4*)p = tzr_f32x4{v.x, v.y, v.z, v.w}; }
// streaming form: output that nobody on the chip reads again soon (nontemporal: no L2 allocation kept for it)
__device__ __forceinline__ void tzr_stg4_nt(float* p, float4 v) { __builtin_nontemporal_store(tzr_f32x4{v.x, v.y, v.z, v.w}, (TZR_GLOBAL_AS tzr_f32x4*)p); }
// 8 bytes (four fp16 of a half-precision table row, as two dwords)
__device__ __forceinline__ uint2 tzr_ldg8(const void* p) {
  const tzr_u32x2 v = *(const TZR_GLOBAL_AS tzr_u32x2*)p;
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void tzr_stg8(void* p, uint2 v) { *(TZR_GLOBAL_AS tzr_u32x2*)p = tzr_u32x2{v.x, v.y}; }

// An LDS pointer that keeps its address space through `volatile` (a volatile generic pointer into LDS is accessed with FLAT
// instructions: slower than ds_read / ds_write, and in the memory counter as well)
#define TZR_LDS_AS __attribute__((address_space(3)))
