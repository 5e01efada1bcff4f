// TEST INFRASTRUCTURE ONLY -- lane-emulator stand-in for torcheasyrec_amd/csrc/tzr_gfx950.h (the
// agent-scope publish / consume helpers are gfx950 instructions).  Workgroups run one after the
// other here, so plain atomics are enough.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

inline void tzr_publish_u32(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline void tzr_publish_u64(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t tzr_consume_u32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline uint64_t tzr_consume_u64(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void tzr_drain_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline uint32_t tzr_arrive(uint32_t* counter) { return __atomic_fetch_add(counter, 1u, __ATOMIC_SEQ_CST); }
#define TZR_WAVES_PER_EU(n)
inline void tzr_lds_barrier() { __syncthreads(); }
#define TZR_OPAQUE(x) ((void)(x))
inline void tzr_prio_by_slot(unsigned) {}  // (scheduling only)
inline float tzr_relu(float x) { return x > 0.f ? x : 0.f; }  // (a register-allocation hint: nothing to emulate)
inline float tzr_ldg(const float* p) { return *p; }
inline void tzr_stg(float* p, float v) { *p = v; }
inline float4 tzr_ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
inline void tzr_stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
inline void tzr_stg4_nt(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
inline uint2 tzr_ldg8(const void* p) { uint2 v; memcpy(&v, p, 8); return v; }
inline void tzr_stg8(void* p, uint2 v) { memcpy(p, &v, 8); }
#define TZR_LDS_AS
