// K13: zero-collision-hash (managed collision) id remap for gfx950.
//
// Replaces torchrec MCHManagedCollisionModule.remap / .profile [upstream 1.7.0], which tzrec builds in
// BaseFeature.mc_module (/root/reference/tzrec/features/feature.py:693-736) and wires in front of the
// embedding lookup with ManagedCollisionEmbeddingBagCollection
// (/root/reference/tzrec/modules/embedding.py:856-864).  torchrec keeps a SORTED raw-id array and
// binary-searches it: ~log2(zch_size) dependent HBM probes per id (28 for a 200 M-slot table).  Here
// the raw-id -> row map is an open-addressing table (linear probing, load factor <= 1/2, 12 bytes
// per cell): the expected probe sequence is < 1.5 cells, i.e. one or two 64-byte lines per id, and
// the kernel is a pure HBM gather.  The sorted view torchrec exposes (zch_util.py:29,
// `_mch_sorted_raw_ids`) is derived from the per-row id array on the host side when exporting.
//
// remap:   id present  -> its row            (+ profile: count[row] += 1, last_iter[row] = iter)
//          id absent   -> row zch_size-1     (+ profile: candidates[i] = id, else TZR_ZCH_EMPTY)
// The candidate array is positional (one cell per input id): no cursor, no atomics on a shared
// address -- a single append cursor serialises ~27 k wave-level atomics per batch (measured 290 us).
// The only atomics left are the integer count bumps on random rows: order independent.
#include "tzr_common.h"

#define ZCH_THREADS 256

__device__ __forceinline__ uint64_t zch_mix(int64_t id) {  // splitmix64 finaliser
  uint64_t x = (uint64_t)id + 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Ring mode (tzr_zch_remap_ring: the step replayed from a hipGraph): the iteration number comes from DEVICE memory
// (the graph's own counter, bumped by the caller inside the graph), and the candidates of a step go to slot
// `iter % ring_slots` of a ring that holds a [n_cand_keys][per_key] block per slot -- only the keys that HAVE a module
// (key_cand[f] = their index among those, -1 otherwise), per_key = B * uniform ids each.  Nothing on the host names a step.
__global__ __launch_bounds__(ZCH_THREADS) void tzr_zch_remap_kernel(
    const TzrZchModule* __restrict__ mods, const int32_t* __restrict__ key_module, int n_keys,
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets, int64_t B, int uniform,
    int64_t iter, int profile, int64_t* __restrict__ out, int64_t* __restrict__ candidates,
    const int64_t* __restrict__ d_iter, const int32_t* __restrict__ key_cand, int n_cand_keys, int64_t ring_slots) {
  const int f = blockIdx.y;
  const int m = key_module[f];
  const int64_t s = uniform ? (int64_t)f * B * uniform : offsets[(int64_t)f * B];
  const int64_t e = uniform ? (int64_t)(f + 1) * B * uniform : offsets[(int64_t)(f + 1) * B];
  if (d_iter) {
    iter = *d_iter;
    if (profile) {
      const int zk = key_cand[f];
      const int64_t per_key = B * uniform;
      // candidates[i] below = the ring cell of id i: base of (slot, key) - s
      candidates = zk < 0 ? nullptr : candidates + ((iter % ring_slots) * n_cand_keys + zk) * per_key - s;
      if (!candidates) profile = 0;
    }
  }
  if (m < 0) {  // key without a ZCH module: ids pass through
    for (int64_t i = s + (int64_t)blockIdx.x * ZCH_THREADS + threadIdx.x; i < e;
         i += (int64_t)gridDim.x * ZCH_THREADS) {
      if (out != values) out[i] = values[i];
      if (profile) candidates[i] = TZR_ZCH_EMPTY;
    }
    return;
  }
  const TzrZchModule M = mods[m];
  const uint64_t mask = (uint64_t)M.capacity - 1;
  for (int64_t i = s + (int64_t)blockIdx.x * ZCH_THREADS + threadIdx.x; i < e;
       i += (int64_t)gridDim.x * ZCH_THREADS) {
    const int64_t id = values[i];
    int64_t row = M.zch_size - 1;
    bool hit = false;
    uint64_t h = zch_mix(id) & mask;
    // (the two sentinel values are never ids of the map: they are served from the shared row and never admitted)
    for (int64_t probe = 0; probe < M.capacity && id != TZR_ZCH_TOMB && id != TZR_ZCH_EMPTY; ++probe) {
      const int64_t k = M.keys[h];
      if (k == id) {
        row = M.rows[h];
        hit = true;
        break;
      }
      if (k == TZR_ZCH_EMPTY) break;
      h = (h + 1) & mask;
    }
    out[i] = row;
    if (profile) {
      if (hit) {
        atomicAdd(reinterpret_cast<unsigned long long*>(M.counts + row), 1ull);
        M.last_iter[row] = iter;  // every writer stores the same value
      }
      candidates[i] = (hit || id == TZR_ZCH_TOMB) ? TZR_ZCH_EMPTY : id;
    }
  }
}

__global__ __launch_bounds__(ZCH_THREADS) void tzr_zch_clear_kernel(TzrZchModule M) {
  for (int64_t i = (int64_t)blockIdx.x * ZCH_THREADS + threadIdx.x; i < M.capacity;
       i += (int64_t)gridDim.x * ZCH_THREADS)
    M.keys[i] = TZR_ZCH_EMPTY;
}

// Insert (id, row) pairs; ids are distinct (the caller passes one pair per occupied row).  Which
// cell an id lands in depends on the insertion race, what a lookup returns does not.
__global__ __launch_bounds__(ZCH_THREADS) void tzr_zch_insert_kernel(
    TzrZchModule M, const int64_t* __restrict__ ids, const int32_t* __restrict__ rows, int64_t n) {
  const uint64_t mask = (uint64_t)M.capacity - 1;
  for (int64_t i = (int64_t)blockIdx.x * ZCH_THREADS + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * ZCH_THREADS) {
    const int64_t id = ids[i];
    if (id == TZR_ZCH_EMPTY || id == TZR_ZCH_TOMB) continue;
    uint64_t h = zch_mix(id) & mask;
    for (int64_t probe = 0; probe < M.capacity; ++probe) {
      const unsigned long long prev =
          atomicCAS(reinterpret_cast<unsigned long long*>(M.keys + h),
                    (unsigned long long)TZR_ZCH_EMPTY, (unsigned long long)id);
      if (prev == (unsigned long long)TZR_ZCH_EMPTY || prev == (unsigned long long)id) {
        M.rows[h] = rows[i];
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

// Incremental form of the rebuild after an admission / eviction round: the ids that lost their row leave a
// tombstone (lookups walk over it, it never equals an id), the admitted ids take the first free or tombstone
// cell of their probe sequence.  Two launches (all deletions before any insertion).
__global__ __launch_bounds__(ZCH_THREADS) void tzr_zch_delete_kernel(TzrZchModule M, const int64_t* __restrict__ ids,
                                                                     int64_t n) {
  const uint64_t mask = (uint64_t)M.capacity - 1;
  for (int64_t i = (int64_t)blockIdx.x * ZCH_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * ZCH_THREADS) {
    const int64_t id = ids[i];
    if (id == TZR_ZCH_EMPTY || id == TZR_ZCH_TOMB) continue;
    uint64_t h = zch_mix(id) & mask;
    for (int64_t probe = 0; probe < M.capacity; ++probe) {
      const int64_t k = M.keys[h];
      if (k == id) {
        M.keys[h] = TZR_ZCH_TOMB;
        break;
      }
      if (k == TZR_ZCH_EMPTY) break;
      h = (h + 1) & mask;
    }
  }
}

__global__ __launch_bounds__(ZCH_THREADS) void tzr_zch_reinsert_kernel(TzrZchModule M, const int64_t* __restrict__ ids,
                                                                       const int32_t* __restrict__ rows, int64_t n) {
  const uint64_t mask = (uint64_t)M.capacity - 1;
  for (int64_t i = (int64_t)blockIdx.x * ZCH_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * ZCH_THREADS) {
    const int64_t id = ids[i];
    if (id == TZR_ZCH_EMPTY || id == TZR_ZCH_TOMB) continue;
    uint64_t h = zch_mix(id) & mask;
    bool done = false;
    for (int64_t probe = 0; probe < M.capacity && !done; ++probe) {
      unsigned long long* cell = reinterpret_cast<unsigned long long*>(M.keys + h);
      unsigned long long k = *reinterpret_cast<volatile unsigned long long*>(cell);
      while (k == (unsigned long long)TZR_ZCH_EMPTY || k == (unsigned long long)TZR_ZCH_TOMB) {
        const unsigned long long prev = atomicCAS(cell, k, (unsigned long long)id);
        if (prev == k) {
          M.rows[h] = rows[i];
          done = true;
          break;
        }
        k = prev;  // somebody else took the cell (or it changed kind): look again
      }
      h = (h + 1) & mask;
    }
  }
}

extern "C" int tzr_zch_update(const TzrZchModule* h_module, const int64_t* d_old_ids, const int64_t* d_new_ids,
                              const int32_t* d_rows, int64_t n, void* stream) {
  if (!h_module || !h_module->keys || !h_module->rows || h_module->capacity <= 0 ||
      (h_module->capacity & (h_module->capacity - 1)) || n < 0)
    return TZR_ERR_INVALID;
  if (n == 0) return TZR_OK;
  if (!d_old_ids || !d_new_ids || !d_rows) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned g = (unsigned)std::min<int64_t>(4096, (n + ZCH_THREADS - 1) / ZCH_THREADS);
  hipLaunchKernelGGL(tzr_zch_delete_kernel, dim3(g), dim3(ZCH_THREADS), 0, s, *h_module, d_old_ids, n);
  hipLaunchKernelGGL(tzr_zch_reinsert_kernel, dim3(g), dim3(ZCH_THREADS), 0, s, *h_module, d_new_ids, d_rows, n);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_zch_remap(const TzrZchModule* d_modules, const int32_t* d_key_module, int n_keys,
                             const int64_t* d_values, const int64_t* d_offsets, int64_t B,
                             int uniform_bag_len, int64_t n_values, int64_t iter, int profile,
                             int64_t* d_out_values, int64_t* d_candidates, void* stream) {
  if (!d_modules || !d_key_module || n_keys <= 0 || B < 0 || n_values < 0 || uniform_bag_len < 0)
    return TZR_ERR_INVALID;
  if (!uniform_bag_len && !d_offsets) return TZR_ERR_INVALID;
  if (n_values == 0 || B == 0) return TZR_OK;
  if (!d_values || !d_out_values) return TZR_ERR_INVALID;
  if (profile && !d_candidates) return TZR_ERR_INVALID;
  const int64_t per_key = (n_values + n_keys - 1) / n_keys;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (per_key + ZCH_THREADS - 1) / ZCH_THREADS));
  hipLaunchKernelGGL(tzr_zch_remap_kernel, dim3(gx, (unsigned)n_keys), dim3(ZCH_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_modules, d_key_module, n_keys, d_values,
                     d_offsets, B, uniform_bag_len, iter, profile, d_out_values, d_candidates,
                     static_cast<const int64_t*>(nullptr), static_cast<const int32_t*>(nullptr), 0, (int64_t)1);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_zch_remap_ring(const TzrZchModule* d_modules, const int32_t* d_key_module, int n_keys,
                                  const int64_t* d_values, int64_t B, int uniform_bag_len, int64_t n_values,
                                  const int64_t* d_iter, int profile, int64_t* d_out_values, const int32_t* d_key_cand,
                                  int n_cand_keys, int64_t* d_cand_ring, int64_t ring_slots, void* stream) {
  if (!d_modules || !d_key_module || n_keys <= 0 || B < 0 || n_values < 0 || uniform_bag_len <= 0 || !d_iter)
    return TZR_ERR_INVALID;
  if (n_values != (int64_t)n_keys * B * uniform_bag_len) return TZR_ERR_INVALID;  // (uniform bags: a step's ring block has one shape)
  if (n_values == 0) return TZR_OK;
  if (!d_values || !d_out_values) return TZR_ERR_INVALID;
  if (profile && (!d_key_cand || n_cand_keys <= 0 || !d_cand_ring || ring_slots <= 0)) return TZR_ERR_INVALID;
  const int64_t per_key = B * uniform_bag_len;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (per_key + ZCH_THREADS - 1) / ZCH_THREADS));
  hipLaunchKernelGGL(tzr_zch_remap_kernel, dim3(gx, (unsigned)n_keys), dim3(ZCH_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_modules, d_key_module, n_keys, d_values,
                     static_cast<const int64_t*>(nullptr), B, uniform_bag_len, (int64_t)0, profile, d_out_values, d_cand_ring,
                     d_iter, d_key_cand, n_cand_keys, ring_slots);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_zch_build(const TzrZchModule* h_module, const int64_t* d_ids,
                             const int32_t* d_rows, int64_t n, void* stream) {
  if (!h_module || !h_module->keys || !h_module->rows || h_module->capacity <= 0 ||
      (h_module->capacity & (h_module->capacity - 1)) || n < 0 || n * 2 > h_module->capacity)
    return TZR_ERR_INVALID;
  if (n > 0 && (!d_ids || !d_rows)) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned gc = (unsigned)std::min<int64_t>(4096, (h_module->capacity + ZCH_THREADS - 1) / ZCH_THREADS);
  hipLaunchKernelGGL(tzr_zch_clear_kernel, dim3(gc), dim3(ZCH_THREADS), 0, s, *h_module);
  if (n > 0) {
    const unsigned gi = (unsigned)std::min<int64_t>(4096, (n + ZCH_THREADS - 1) / ZCH_THREADS);
    hipLaunchKernelGGL(tzr_zch_insert_kernel, dim3(gi), dim3(ZCH_THREADS), 0, s, *h_module, d_ids, d_rows, n);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
