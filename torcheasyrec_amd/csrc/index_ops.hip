// K1-K4: integer index stage of the sharded embedding path (bit-exact vs the oracle).
//
//   K3 tzr_lengths_to_offsets  <- fbgemm asynchronous_complete_cumsum (KJT.offsets())
//   K4 tzr_bounds_check        <- fbgemm bounds_check_indices
//   K1 tzr_kjt_permute         <- fbgemm permute_2D_sparse_data (sharded input_dist)
//   K2 tzr_block_bucketize     <- fbgemm block_bucketize_sparse_features (row-wise tables)
// all reached from self.ebc(kjt), /root/reference/tzrec/modules/embedding.py:930, through
// torchrec's sharded EmbeddingBagCollection [upstream 1.7.0].  Pure HBM-streaming integer work:
// coalesced 8-byte loads, LDS only for the workgroup scans.
#include "tzr_common.h"

#define IDX_THREADS 256
#define IDX_ITEMS 8
#define IDX_TILE (IDX_THREADS * IDX_ITEMS)

__device__ __forceinline__ int64_t idx_load_len(const void* p, int itemsize, int64_t i) {
  return itemsize == 4 ? (int64_t) reinterpret_cast<const int32_t*>(p)[i]
                       : reinterpret_cast<const int64_t*>(p)[i];
}
__device__ __forceinline__ void idx_store_len(void* p, int itemsize, int64_t i, int64_t v) {
  if (itemsize == 4) reinterpret_cast<int32_t*>(p)[i] = (int32_t)v;
  else reinterpret_cast<int64_t*>(p)[i] = v;
}

// Inclusive scan of one value per thread across the workgroup (wave shuffles + LDS).
__device__ __forceinline__ int64_t idx_block_inclusive(int64_t v, int64_t* wave_tot /*[4] LDS*/) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  for (int d = 1; d < TZR_WAVE; d <<= 1) {
    const int64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  if (lane == TZR_WAVE - 1) wave_tot[wv] = v;
  __syncthreads();
  int64_t pre = 0;
  for (int w = 0; w < wv; ++w) pre += wave_tot[w];
  __syncthreads();
  return v + pre;
}

// pass 1: per-tile sums
__global__ __launch_bounds__(IDX_THREADS) void tzr_scan_tile_sums_kernel(
    const void* __restrict__ in, int itemsize, int64_t n, int64_t* __restrict__ tile_sums) {
  __shared__ int64_t wt[IDX_THREADS / TZR_WAVE];
  const int64_t base = (int64_t)blockIdx.x * IDX_TILE;
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < IDX_ITEMS; ++j) {
    const int64_t i = base + (int64_t)j * IDX_THREADS + threadIdx.x;
    if (i < n) s += idx_load_len(in, itemsize, i);
  }
  const int64_t inc = idx_block_inclusive(s, wt);
  if (threadIdx.x == IDX_THREADS - 1) tile_sums[blockIdx.x] = inc;
}

// pass 2: one workgroup turns tile sums into exclusive tile prefixes (loop with carry)
__global__ __launch_bounds__(IDX_THREADS) void tzr_scan_tile_prefix_kernel(
    int64_t* __restrict__ tile_sums, int64_t n_tiles, int64_t* __restrict__ total_out) {
  __shared__ int64_t wt[IDX_THREADS / TZR_WAVE];
  __shared__ int64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t b = 0; b < n_tiles; b += IDX_THREADS) {
    const int64_t i = b + threadIdx.x;
    const int64_t v = i < n_tiles ? tile_sums[i] : 0;
    const int64_t inc = idx_block_inclusive(v, wt);
    const int64_t carry = carry_s;
    if (i < n_tiles) tile_sums[i] = carry + inc - v;
    __syncthreads();
    if (threadIdx.x == IDX_THREADS - 1) carry_s = carry + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

// pass 3: exclusive scan inside each tile + tile prefix; thread t owns IDX_ITEMS consecutive items
__global__ __launch_bounds__(IDX_THREADS) void tzr_scan_final_kernel(
    const void* __restrict__ in, int itemsize, int64_t n, const int64_t* __restrict__ tile_prefix,
    int64_t* __restrict__ out /*[n+1]*/) {
  __shared__ int64_t wt[IDX_THREADS / TZR_WAVE];
  const int64_t base = (int64_t)blockIdx.x * IDX_TILE + (int64_t)threadIdx.x * IDX_ITEMS;
  int64_t v[IDX_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < IDX_ITEMS; ++j) {
    v[j] = (base + j < n) ? idx_load_len(in, itemsize, base + j) : 0;
    s += v[j];
  }
  const int64_t inc = idx_block_inclusive(s, wt);
  int64_t run = tile_prefix[blockIdx.x] + inc - s;
#pragma unroll
  for (int j = 0; j < IDX_ITEMS; ++j) {
    if (base + j < n) out[base + j] = run;
    run += v[j];
    if (base + j == n - 1) out[n] = run;
  }
  if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

static int idx_scan(const void* in, int itemsize, int64_t n, int64_t* out, void* ws,
                    size_t ws_bytes, hipStream_t s) {
  const int64_t tiles = n > 0 ? (n + IDX_TILE - 1) / IDX_TILE : 1;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < (size_t)tiles * 8)
    return TZR_ERR_WORKSPACE;
  int64_t* tile_sums = static_cast<int64_t*>(ws);
  hipLaunchKernelGGL(tzr_scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(IDX_THREADS), 0, s, in,
                     itemsize, n, tile_sums);
  hipLaunchKernelGGL(tzr_scan_tile_prefix_kernel, dim3(1), dim3(IDX_THREADS), 0, s, tile_sums,
                     tiles, (int64_t*)nullptr);
  hipLaunchKernelGGL(tzr_scan_final_kernel, dim3((unsigned)tiles), dim3(IDX_THREADS), 0, s, in,
                     itemsize, n, tile_sums, out);
  return TZR_OK;
}

extern "C" size_t tzr_lengths_to_offsets_workspace(int64_t n) {
  const int64_t tiles = n > 0 ? (n + IDX_TILE - 1) / IDX_TILE : 1;
  return tzr_align_up((size_t)tiles * 8) + 256;
}

extern "C" int tzr_lengths_to_offsets(const void* d_lengths, int lengths_itemsize, int64_t n,
                                      int64_t* d_offsets, void* ws, size_t ws_bytes,
                                      void* stream) {
  if (!d_offsets || n < 0 || (lengths_itemsize != 4 && lengths_itemsize != 8)) return TZR_ERR_INVALID;
  if (n > 0 && !d_lengths) return TZR_ERR_INVALID;
  const int rc = idx_scan(d_lengths, lengths_itemsize, n, d_offsets, ws, ws_bytes,
                          static_cast<hipStream_t>(stream));
  if (rc != TZR_OK) return rc;
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- K4 -------------------------------------------------------------------------------------

__global__ __launch_bounds__(IDX_THREADS) void tzr_bounds_check_kernel(
    const TzrTable* __restrict__ tables, const TzrFeature* __restrict__ feats, int F,
    int64_t* __restrict__ values, const int64_t* __restrict__ offsets, int64_t B, int mode,
    unsigned long long* __restrict__ oob) {
  // one (key, sample-tile) per workgroup: blockIdx.y = key, blockIdx.x tiles the key's ids
  const int f = blockIdx.y;
  if (feats[f].table < 0) return;  // key not owned by this module
  const int64_t rows = tables[feats[f].table].rows;
  const int64_t key = feats[f].key;
  const int64_t s = offsets[key * B], e = offsets[(key + 1) * B];
  unsigned bad = 0;
  for (int64_t i = s + (int64_t)blockIdx.x * IDX_THREADS + threadIdx.x; i < e;
       i += (int64_t)gridDim.x * IDX_THREADS) {
    const int64_t id = values[i];
    if ((uint64_t)id >= (uint64_t)rows) {
      ++bad;
      if (mode != TZR_BOUNDS_FATAL) values[i] = 0;
    }
  }
  if (bad && mode != TZR_BOUNDS_IGNORE) atomicAdd(oob, (unsigned long long)bad);
}

extern "C" int tzr_bounds_check(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                                int64_t* d_values, const int64_t* d_offsets, int64_t B, int mode,
                                int64_t* d_oob_count, void* stream) {
  if (!d_tables || !d_feats || n_feats <= 0 || !d_offsets || B < 0 || !d_oob_count || mode < 0 ||
      mode > 2)
    return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  if (!d_values) return TZR_ERR_INVALID;
  const unsigned gx = (unsigned)std::min<int64_t>(256, (B + IDX_THREADS - 1) / IDX_THREADS);
  hipLaunchKernelGGL(tzr_bounds_check_kernel, dim3(gx, (unsigned)n_feats), dim3(IDX_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_tables, d_feats, n_feats, d_values,
                     d_offsets, B, mode, reinterpret_cast<unsigned long long*>(d_oob_count));
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- K1 -------------------------------------------------------------------------------------

__global__ __launch_bounds__(IDX_THREADS) void tzr_permute_lengths_kernel(
    const int32_t* __restrict__ permute, int64_t B, const void* __restrict__ in_lengths,
    int itemsize, void* __restrict__ out_lengths) {
  const int t = blockIdx.y;
  const int64_t p = permute[t];
  for (int64_t b = (int64_t)blockIdx.x * IDX_THREADS + threadIdx.x; b < B;
       b += (int64_t)gridDim.x * IDX_THREADS)
    idx_store_len(out_lengths, itemsize, (int64_t)t * B + b,
                  idx_load_len(in_lengths, itemsize, p * B + b));
}

// Key t's values are one contiguous segment in both layouts: a segmented memcpy.
__global__ __launch_bounds__(IDX_THREADS) void tzr_permute_values_kernel(
    const int32_t* __restrict__ permute, int64_t B, const int64_t* __restrict__ in_offsets,
    const int64_t* __restrict__ out_offsets, const int64_t* __restrict__ in_values,
    const float* __restrict__ in_weights, int64_t* __restrict__ out_values,
    float* __restrict__ out_weights, int64_t n_out_max) {
  const int t = blockIdx.y;
  const int64_t p = permute[t];
  const int64_t src = in_offsets[p * B];
  const int64_t dst = out_offsets[(int64_t)t * B];
  int64_t n = out_offsets[(int64_t)(t + 1) * B] - dst;
  if (dst + n > n_out_max) n = n_out_max > dst ? n_out_max - dst : 0;
  for (int64_t i = (int64_t)blockIdx.x * IDX_THREADS + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * IDX_THREADS) {
    out_values[dst + i] = in_values[src + i];
    if (in_weights) out_weights[dst + i] = in_weights[src + i];
  }
}

extern "C" size_t tzr_kjt_permute_workspace(int64_t T, int64_t B) {
  return tzr_lengths_to_offsets_workspace(T * B);
}

extern "C" int tzr_kjt_permute(const int32_t* d_permute, int T, int F, int64_t B,
                               const void* d_in_lengths, int lengths_itemsize,
                               const int64_t* d_in_offsets, const int64_t* d_in_values,
                               const float* d_in_weights, void* d_out_lengths,
                               int64_t* d_out_offsets, int64_t* d_out_values,
                               float* d_out_weights, int64_t n_out_max, void* ws, size_t ws_bytes,
                               void* stream) {
  if (!d_permute || T <= 0 || F <= 0 || B < 0 || !d_in_offsets || !d_out_offsets ||
      (lengths_itemsize != 4 && lengths_itemsize != 8) || n_out_max < 0)
    return TZR_ERR_INVALID;
  if (d_in_weights && !d_out_weights) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    if (hipMemsetAsync(d_out_offsets, 0, 8, s) != hipSuccess) return TZR_ERR_LAUNCH;
    return TZR_OK;
  }
  if (!d_in_lengths || !d_out_lengths) return TZR_ERR_INVALID;
  const unsigned gx = (unsigned)std::min<int64_t>(64, (B + IDX_THREADS - 1) / IDX_THREADS);
  hipLaunchKernelGGL(tzr_permute_lengths_kernel, dim3(gx, (unsigned)T), dim3(IDX_THREADS), 0, s,
                     d_permute, B, d_in_lengths, lengths_itemsize, d_out_lengths);
  const int rc = idx_scan(d_out_lengths, lengths_itemsize, (int64_t)T * B, d_out_offsets, ws,
                          ws_bytes, s);
  if (rc != TZR_OK) return rc;
  if (n_out_max > 0) {
    if (!d_in_values || !d_out_values) return TZR_ERR_INVALID;
    hipLaunchKernelGGL(tzr_permute_values_kernel, dim3(128, (unsigned)T), dim3(IDX_THREADS), 0, s,
                       d_permute, B, d_in_offsets, d_out_offsets, d_in_values, d_in_weights,
                       d_out_values, d_out_weights, n_out_max);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- K2 -------------------------------------------------------------------------------------
// Owner of id x of key f: x / block_size[f] (row-wise blocks), or -- block_size[f] == 0 -- a hash of
// the raw id (zero-collision-hash tables: raw ids are arbitrary 64-bit values, the owner maps them
// to rows itself; the id travels unchanged).
__device__ __forceinline__ uint64_t idx_mix(int64_t id) {  // splitmix64 finaliser (as in zch.hip)
  uint64_t x = (uint64_t)id + 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ int64_t idx_owner(int64_t id, int64_t bs, int W) {
  if (bs == 0) return (int64_t)(idx_mix(id) % (uint64_t)W);
  const int64_t r = id / bs;
  return r < 0 ? 0 : (r > W - 1 ? W - 1 : r);
}
// One thread per bag (a bag is owned by one thread, so its W counters / cursors are private plain
// read-modify-writes: deterministic, ids keep their order inside every (rank, key, sample) bag).

__global__ __launch_bounds__(IDX_THREADS) void tzr_bucketize_count_kernel(
    const int64_t* __restrict__ block_sizes, const int32_t* __restrict__ rank_offsets, int F,
    int64_t B, int W, const int64_t* __restrict__ offsets, const int64_t* __restrict__ values,
    void* __restrict__ new_lengths, int itemsize) {
  const int64_t bag = (int64_t)blockIdx.x * IDX_THREADS + threadIdx.x;
  if (bag >= (int64_t)F * B) return;
  const int64_t bs = block_sizes[bag / B];
  const int64_t ro = rank_offsets ? rank_offsets[bag / B] : 0;
  const int64_t FB = (int64_t)F * B;
  for (int64_t i = offsets[bag]; i < offsets[bag + 1]; ++i) {
    const int64_t r = idx_owner(values[i], bs, W);
    const int64_t o = ((r + ro) % W) * FB + bag;
    idx_store_len(new_lengths, itemsize, o, idx_load_len(new_lengths, itemsize, o) + 1);
  }
}

__global__ __launch_bounds__(IDX_THREADS) void tzr_bucketize_scatter_kernel(
    const int64_t* __restrict__ block_sizes, const int32_t* __restrict__ rank_offsets, int F,
    int64_t B, int W, const int64_t* __restrict__ offsets, const int64_t* __restrict__ values,
    const float* __restrict__ weights, int64_t* __restrict__ cursor /*[W*F*B] = new_offsets copy*/,
    int64_t* __restrict__ new_values, float* __restrict__ new_weights,
    int64_t* __restrict__ unbucketize) {
  const int64_t bag = (int64_t)blockIdx.x * IDX_THREADS + threadIdx.x;
  if (bag >= (int64_t)F * B) return;
  const int64_t bs = block_sizes[bag / B];
  const int64_t ro = rank_offsets ? rank_offsets[bag / B] : 0;
  const int64_t FB = (int64_t)F * B;
  for (int64_t i = offsets[bag]; i < offsets[bag + 1]; ++i) {
    const int64_t id = values[i];
    const int64_t r = idx_owner(id, bs, W);
    const int64_t o = ((r + ro) % W) * FB + bag;
    const int64_t pos = cursor[o];
    cursor[o] = pos + 1;
    new_values[pos] = id - r * bs;
    if (weights) new_weights[pos] = weights[i];
    if (unbucketize) unbucketize[i] = pos;
  }
}

extern "C" size_t tzr_block_bucketize_workspace(int64_t F, int64_t B, int W) {
  const int64_t n = (int64_t)W * F * B;
  return tzr_lengths_to_offsets_workspace(n) + tzr_align_up((size_t)n * 8) + 256;
}

extern "C" int tzr_block_bucketize(const int64_t* d_block_sizes, const int32_t* d_rank_offsets,
                                   int F, int64_t B, int W,
                                   const int64_t* d_offsets, const int64_t* d_values,
                                   const float* d_weights, int64_t n_values, void* d_new_lengths,
                                   int lengths_itemsize, int64_t* d_new_offsets,
                                   int64_t* d_new_values, float* d_new_weights,
                                   int64_t* d_unbucketize_permute, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!d_block_sizes || F <= 0 || B < 0 || W <= 0 || !d_offsets || n_values < 0 ||
      !d_new_offsets || (lengths_itemsize != 4 && lengths_itemsize != 8))
    return TZR_ERR_INVALID;
  if (d_weights && !d_new_weights) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)W * F * B;
  if (n == 0) {
    if (hipMemsetAsync(d_new_offsets, 0, 8, s) != hipSuccess) return TZR_ERR_LAUNCH;
    return TZR_OK;
  }
  if (!d_new_lengths || (n_values > 0 && (!d_values || !d_new_values))) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) ||
      ws_bytes < tzr_block_bucketize_workspace(F, B, W) - 256)
    return TZR_ERR_WORKSPACE;
  const size_t scan_ws = tzr_lengths_to_offsets_workspace(n) - 256;
  int64_t* cursor = reinterpret_cast<int64_t*>(static_cast<char*>(ws) + tzr_align_up(scan_ws));
  if (hipMemsetAsync(d_new_lengths, 0, (size_t)n * lengths_itemsize, s) != hipSuccess)
    return TZR_ERR_LAUNCH;
  const unsigned gb = (unsigned)(((int64_t)F * B + IDX_THREADS - 1) / IDX_THREADS);
  hipLaunchKernelGGL(tzr_bucketize_count_kernel, dim3(gb), dim3(IDX_THREADS), 0, s, d_block_sizes,
                     d_rank_offsets, F, B, W, d_offsets, d_values, d_new_lengths, lengths_itemsize);
  const int rc = idx_scan(d_new_lengths, lengths_itemsize, n, d_new_offsets, ws, scan_ws, s);
  if (rc != TZR_OK) return rc;
  if (hipMemcpyAsync(cursor, d_new_offsets, (size_t)n * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return TZR_ERR_LAUNCH;
  hipLaunchKernelGGL(tzr_bucketize_scatter_kernel, dim3(gb), dim3(IDX_THREADS), 0, s,
                     d_block_sizes, d_rank_offsets, F, B, W, d_offsets, d_values, d_weights, cursor, d_new_values,
                     d_new_weights, d_unbucketize_permute);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- exchange bucketize (uniform bags) --------------------------------------------------------------
// The requester side of the sharded exchange only needs, for the SELECTED keys of a KJT whose bags
// all hold L ids: the ids grouped by (owner rank, key) in lookup order, where every lookup went, and
// how many ids each (rank, key) pair holds.  The general K2 above also materialises W*F*B bag
// lengths and their scan (fbgemm's contract) after a K1 permute: 13 launches.  This is a stable
// counting sort with W*F' buckets in 3: per-tile rank counts, one-workgroup scan, scatter.
#define XB_THREADS 256
#define XB_TILE 1024  // ids per workgroup

__global__ __launch_bounds__(XB_THREADS) void tzr_xb_count_kernel(
    const int32_t* __restrict__ sel, const int64_t* __restrict__ block_sizes,
    const int32_t* __restrict__ rank_offsets, int W, int64_t n_per_key, const int64_t* __restrict__ values,
    int32_t* __restrict__ tile_cnt /*[F'][tiles][W]*/) {
  __shared__ int cnt[64];
  const int f = blockIdx.y;
  const int tiles = gridDim.x;
  const int64_t base = (int64_t)sel[f] * n_per_key + (int64_t)blockIdx.x * XB_TILE;
  const int64_t end = min((int64_t)sel[f] * n_per_key + n_per_key, base + XB_TILE);
  const int64_t bs = block_sizes[f];
  const int ro = rank_offsets ? rank_offsets[f] : 0;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  // one LDS atomic per distinct destination of a wave (match-any by ballots): at W = 8 every counter is hit
  // by 1/8 of the tile, at W = 1 by all of it -- per-lane atomics on so few addresses serialise (45 us at
  // B = 65536 on the 1-rank proxy, profiles/r03r)
  int bits = 0;
  while ((1 << bits) < W) ++bits;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  for (int64_t i0 = base; i0 < end; i0 += XB_THREADS) {  // workgroup-uniform trip count
    const int64_t i = i0 + threadIdx.x;
    const bool valid = i < end;
    const int d = valid ? (int)((idx_owner(values[i], bs, W) + ro) % W) : 0;
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const int on = (d >> b) & 1;
      const unsigned long long bm = __ballot(on);
      peers &= on ? bm : ~bm;
    }
    if (valid && (peers & ((1ull << lane) - 1ull)) == 0) atomicAdd(&cnt[d], (int)__popcll(peers));  // integer counts: order independent
  }
  __syncthreads();
  if ((int)threadIdx.x < W) tile_cnt[((size_t)f * tiles + blockIdx.x) * W + threadIdx.x] = cnt[threadIdx.x];
}

// exclusive prefix in (dest, key, tile) order, in place; cnt_out[dest*F + f] = ids of key f for dest.
// S > 0 (capacity-bounded exchange): dest d owns the fixed slice [d*S, (d+1)*S) of the message -- F counts,
// one overflow word, then at most S - F - 1 ids.  Counts are clamped to what fits (the scatter drops the
// rest) and every destination is told whether this rank dropped anything for anybody.
__global__ __launch_bounds__(XB_THREADS) void tzr_xb_scan_kernel(int32_t* __restrict__ tile_cnt, int F, int tiles, int W,
                                                                 int64_t* __restrict__ cnt_out, int64_t S) {
  __shared__ int64_t seg_tot[XB_THREADS];
  __shared__ int64_t seg_base[XB_THREADS];
  __shared__ int64_t seg_keep[XB_THREADS];
  __shared__ int dropped;
  // one (dest, key) segment per thread (W * F <= XB_THREADS, checked by the launcher)
  const int s = threadIdx.x;
  const int nseg = W * F;
  int64_t tot = 0;
  if (s < nseg) {
    const int d = s / F, f = s % F;
    // (16 tiles of independent loads at a time: tile by tile this was a chain of `tiles` round trips)
    for (int t0 = 0; t0 < tiles; t0 += 16) {
      int32_t c16[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) c16[j] = t0 + j < tiles ? tile_cnt[((size_t)f * tiles + t0 + j) * W + d] : 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) tot += c16[j];
    }
    if (S == 0) cnt_out[s] = tot;
  }
  seg_tot[s] = s < nseg ? tot : 0;
  __syncthreads();
  if (s == 0) {
    int64_t run = 0;
    int over = 0;
    for (int k = 0; k < nseg; ++k) {
      if (S > 0 && k % F == 0) run = (int64_t)(k / F) * S + F + 1;
      seg_base[k] = run;
      int64_t keep = seg_tot[k];
      if (S > 0) {
        const int64_t room = (int64_t)(k / F + 1) * S - run;
        if (keep > room) {
          keep = room;
          over = 1;
        }
      }
      seg_keep[k] = keep;
      run += keep;
    }
    dropped = over;
  }
  __syncthreads();
  if (s < nseg) {
    const int d = s / F, f = s % F;
    if (S > 0) {
      cnt_out[(int64_t)d * S + f] = seg_keep[s];
      if (f == 0) cnt_out[(int64_t)d * S + F] = dropped;
    }
    int64_t run = seg_base[s];
    for (int t0 = 0; t0 < tiles; t0 += 16) {
      int32_t c16[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) c16[j] = t0 + j < tiles ? tile_cnt[((size_t)f * tiles + t0 + j) * W + d] : 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (t0 + j < tiles) tile_cnt[((size_t)f * tiles + t0 + j) * W + d] = (int32_t)run;  // start of this tile's ids for (dest, key)
        run += c16[j];
      }
    }
  }
}

__global__ __launch_bounds__(XB_THREADS) void tzr_xb_scatter_kernel(
    const int32_t* __restrict__ sel, const int64_t* __restrict__ block_sizes,
    const int32_t* __restrict__ rank_offsets, int W, int64_t n_per_key, const int64_t* __restrict__ values,
    const int32_t* __restrict__ tile_base, int64_t* __restrict__ out_ids, int64_t* __restrict__ unbucketize,
    int64_t S, int F) {
  __shared__ int run[64];                              // ids of each dest placed by earlier rounds / waves
  __shared__ int wcnt[XB_THREADS / TZR_WAVE][64];
  const int f = blockIdx.y;
  const int tiles = gridDim.x;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int64_t key_base = (int64_t)sel[f] * n_per_key;
  const int64_t base = key_base + (int64_t)blockIdx.x * XB_TILE;
  const int64_t end = min(key_base + n_per_key, base + XB_TILE);
  const int64_t bs = block_sizes[f];
  const int ro = rank_offsets ? rank_offsets[f] : 0;
  const int32_t* tb = tile_base + ((size_t)f * tiles + blockIdx.x) * W;
  if (threadIdx.x < 64) run[threadIdx.x] = 0;
  for (int w = 0; w < XB_THREADS / TZR_WAVE; ++w)
    if (threadIdx.x < 64) wcnt[w][threadIdx.x] = 0;
  __syncthreads();
  int bits = 0;
  while ((1 << bits) < W) ++bits;
  for (int64_t i0 = base; i0 < end; i0 += XB_THREADS) {
    const int64_t i = i0 + threadIdx.x;
    const bool valid = i < end;
    int64_t id = 0, r = 0;
    int d = 0;
    if (valid) {
      id = values[i];
      r = idx_owner(id, bs, W);
      d = (int)((r + ro) % W);
    }
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const int on = (d >> b) & 1;
      const unsigned long long bm = __ballot(on);
      peers &= on ? bm : ~bm;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcnt[wv][d] = (int)__popcll(peers);
    __syncthreads();
    if (valid) {
      int pre = run[d];
      for (int w = 0; w < wv; ++w) pre += wcnt[w][d];
      int64_t pos = (int64_t)tb[d] + pre + rank;
      if (S > 0 && pos >= (int64_t)(d + 1) * S) pos = (int64_t)d * S + F;  // over capacity: dropped (flagged by the scan)
      else out_ids[pos] = id - r * bs;
      unbucketize[(int64_t)f * n_per_key + (i - key_base)] = pos;
    }
    __syncthreads();
    if ((int)threadIdx.x < W) {
      int t = 0;
      for (int w = 0; w < XB_THREADS / TZR_WAVE; ++w) {
        t += wcnt[w][threadIdx.x];
        wcnt[w][threadIdx.x] = 0;
      }
      run[threadIdx.x] += t;
    }
    __syncthreads();
  }
}

extern "C" size_t tzr_exchange_bucketize_workspace(int n_sel, int64_t n_per_key, int W) {
  const int64_t tiles = (n_per_key + XB_TILE - 1) / XB_TILE;
  return tzr_align_up((size_t)std::max<int64_t>(1, (int64_t)n_sel * tiles * W) * sizeof(int32_t)) + 256;
}

static int xb_launch(const int32_t* d_sel, int n_sel, const int64_t* d_block_sizes, const int32_t* d_rank_offsets,
                     int64_t B, int bag_len, int W, const int64_t* d_values, int64_t* d_out_ids,
                     int64_t* d_unbucketize, int64_t* d_counts, int64_t S, void* ws, size_t ws_bytes, void* stream) {
  if (!d_sel || n_sel <= 0 || !d_block_sizes || B < 0 || bag_len <= 0 || W <= 0 || !d_counts)
    return TZR_ERR_INVALID;
  if (W > 64 || (int64_t)W * n_sel > XB_THREADS) return TZR_ERR_UNSUPPORTED;
  const int64_t n_per_key = B * bag_len;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_per_key == 0) {
    if (S == 0) {
      if (hipMemsetAsync(d_counts, 0, (size_t)W * n_sel * 8, s) != hipSuccess) return TZR_ERR_LAUNCH;
    } else {
      for (int d = 0; d < W; ++d)
        if (hipMemsetAsync(d_counts + (int64_t)d * S, 0, (size_t)(n_sel + 1) * 8, s) != hipSuccess) return TZR_ERR_LAUNCH;
    }
    return TZR_OK;
  }
  if (!d_values || !d_out_ids || !d_unbucketize) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) ||
      ws_bytes < tzr_exchange_bucketize_workspace(n_sel, n_per_key, W) - 256)
    return TZR_ERR_WORKSPACE;
  const int64_t tiles = (n_per_key + XB_TILE - 1) / XB_TILE;
  if (tiles > 0x7fffffffLL) return TZR_ERR_UNSUPPORTED;
  if (S > 0 && (int64_t)W * S > 0x7fffffffLL) return TZR_ERR_UNSUPPORTED;  // tile starts are 32-bit positions
  int32_t* tile_cnt = static_cast<int32_t*>(ws);
  hipLaunchKernelGGL(tzr_xb_count_kernel, dim3((unsigned)tiles, (unsigned)n_sel), dim3(XB_THREADS), 0, s, d_sel,
                     d_block_sizes, d_rank_offsets, W, n_per_key, d_values, tile_cnt);
  hipLaunchKernelGGL(tzr_xb_scan_kernel, dim3(1), dim3(XB_THREADS), 0, s, tile_cnt, n_sel, (int)tiles, W, d_counts, S);
  hipLaunchKernelGGL(tzr_xb_scatter_kernel, dim3((unsigned)tiles, (unsigned)n_sel), dim3(XB_THREADS), 0, s, d_sel,
                     d_block_sizes, d_rank_offsets, W, n_per_key, d_values, tile_cnt, d_out_ids, d_unbucketize, S,
                     n_sel);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_exchange_bucketize(const int32_t* d_sel, int n_sel, const int64_t* d_block_sizes,
                                      const int32_t* d_rank_offsets, int64_t B, int bag_len, int W,
                                      const int64_t* d_values, int64_t* d_out_ids, int64_t* d_unbucketize,
                                      int64_t* d_counts, void* ws, size_t ws_bytes, void* stream) {
  return xb_launch(d_sel, n_sel, d_block_sizes, d_rank_offsets, B, bag_len, W, d_values, d_out_ids, d_unbucketize,
                   d_counts, 0, ws, ws_bytes, stream);
}

extern "C" int64_t tzr_exchange_message_stride(int n_sel, int64_t capacity) {
  if (n_sel <= 0 || capacity < 0) return 0;
  return (int64_t)n_sel + 1 + capacity;
}

extern "C" int tzr_exchange_bucketize_capped(const int32_t* d_sel, int n_sel, const int64_t* d_block_sizes,
                                             const int32_t* d_rank_offsets, int64_t B, int bag_len, int W,
                                             const int64_t* d_values, int64_t capacity, int64_t* d_message,
                                             int64_t* d_unbucketize, void* ws, size_t ws_bytes, void* stream) {
  if (capacity <= 0 || !d_message) return TZR_ERR_INVALID;
  const int64_t S = tzr_exchange_message_stride(n_sel, capacity);
  return xb_launch(d_sel, n_sel, d_block_sizes, d_rank_offsets, B, bag_len, W, d_values, d_message, d_unbucketize,
                   d_message, S, ws, ws_bytes, stream);
}

// The capacity-bounded layout from a DENSE bucketize result (the general path: ragged / weighted bags go through
// tzr_block_bucketize, whose output is rank-major without gaps): ids re-laid into the fixed slices, every lookup's
// position remapped, headers written.  One launch; every workgroup derives the W slice bases from the W*F counts.
__global__ __launch_bounds__(XB_THREADS) void tzr_xb_pad_kernel(const int64_t* __restrict__ counts /*[W][F]*/, int W, int F,
                                                                int64_t S, const int64_t* __restrict__ ids,
                                                                const int64_t* __restrict__ unb, int64_t N,
                                                                int64_t* __restrict__ msg, int64_t* __restrict__ unb_out) {
  __shared__ int64_t s_tot[64];
  __shared__ int64_t s_start[65];
  __shared__ int s_over;
  const int64_t cap = S - F - 1;
  if ((int)threadIdx.x < W) {
    int64_t t = 0;
    for (int f = 0; f < F; ++f) t += counts[(int64_t)threadIdx.x * F + f];
    s_tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    int over = 0;
    for (int d = 0; d < W; ++d) {
      s_start[d] = run;
      run += s_tot[d];
      over |= s_tot[d] > cap;
    }
    s_start[W] = run;
    s_over = over;
  }
  __syncthreads();
  if (blockIdx.x == 0 && (int)threadIdx.x < W) {  // headers: clamped counts (the first `cap` ids of a slice stay) + flag
    const int d = threadIdx.x;
    int64_t run = 0;
    for (int f = 0; f < F; ++f) {
      int64_t c = counts[(int64_t)d * F + f];
      if (c > cap - run) c = cap - run;
      msg[(int64_t)d * S + f] = c;
      run += c;
    }
    msg[(int64_t)d * S + F] = s_over;
  }
  for (int64_t i = (int64_t)blockIdx.x * XB_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * XB_THREADS) {
    const int64_t p = unb[i];
    int d = 0;
    while (d + 1 < W && s_start[d + 1] <= p) ++d;
    const int64_t off = p - s_start[d];
    int64_t pos = (int64_t)d * S + F;  // over capacity: dropped
    if (off < cap) {
      pos = (int64_t)d * S + F + 1 + off;
      msg[pos] = ids[p];
    }
    unb_out[i] = pos;
  }
}

extern "C" int tzr_exchange_pad(const int64_t* d_counts, int W, int n_sel, int64_t capacity, const int64_t* d_ids,
                                const int64_t* d_unbucketize, int64_t n_ids, int64_t* d_message,
                                int64_t* d_unbucketize_out, void* stream) {
  if (!d_counts || W <= 0 || n_sel <= 0 || capacity <= 0 || n_ids < 0 || !d_message) return TZR_ERR_INVALID;
  if (W > 64) return TZR_ERR_UNSUPPORTED;
  if (n_ids > 0 && (!d_ids || !d_unbucketize || !d_unbucketize_out)) return TZR_ERR_INVALID;
  const int64_t S = tzr_exchange_message_stride(n_sel, capacity);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n_ids + XB_THREADS - 1) / XB_THREADS));
  hipLaunchKernelGGL(tzr_xb_pad_kernel, dim3(grid), dim3(XB_THREADS), 0, static_cast<hipStream_t>(stream), d_counts, W,
                     n_sel, S, d_ids, d_unbucketize, n_ids, d_message, d_unbucketize_out);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// Owner side of the capacity-bounded exchange: the W received message slices -> key segments over the
// positions [0, W*S) of the received buffer.  Per source rank s: one dead key (the gap before its ids: the
// previous rank's unused capacity + this slice's header), then its n_sel keys; a last dead key closes the
// buffer.  *d_overflow = 1 if any rank reported dropped ids (or a header is not a valid count list).
__global__ __launch_bounds__(64) void tzr_xb_owner_segments_kernel(const int64_t* __restrict__ msg, int W, int F,
                                                                   int64_t S, int64_t* __restrict__ key_start,
                                                                   int64_t* __restrict__ overflow) {
  __shared__ int64_t s_end[64];
  __shared__ int s_over;
  const int s = threadIdx.x;
  if (s == 0) s_over = 0;
  __syncthreads();
  int64_t run = (int64_t)s * S + F + 1;
  if (s < W) {
    const int64_t* h = msg + (int64_t)s * S;
    int bad = h[F] != 0;
    for (int f = 0; f < F; ++f) {
      int64_t c = h[f];
      const int64_t room = (int64_t)(s + 1) * S - run;
      if (c < 0 || c > room) {
        c = c < 0 ? 0 : room;
        bad = 1;
      }
      key_start[(int64_t)s * (F + 1) + 1 + f] = run;
      run += c;
    }
    s_end[s] = run;
    if (bad) atomicOr(&s_over, 1);
  }
  __syncthreads();
  if (s < W) key_start[(int64_t)s * (F + 1)] = s == 0 ? 0 : s_end[s - 1];
  if (s == 0) {
    key_start[(int64_t)W * (F + 1)] = s_end[W - 1];
    key_start[(int64_t)W * (F + 1) + 1] = (int64_t)W * S;
    *overflow = s_over;
  }
}

extern "C" int tzr_exchange_owner_segments(const int64_t* d_message, int W, int n_sel, int64_t capacity,
                                           int64_t* d_key_start, int64_t* d_overflow, void* stream) {
  if (!d_message || W <= 0 || n_sel <= 0 || capacity <= 0 || !d_key_start || !d_overflow) return TZR_ERR_INVALID;
  if (W > 64) return TZR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(tzr_xb_owner_segments_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_message,
                     W, n_sel, tzr_exchange_message_stride(n_sel, capacity), d_key_start, d_overflow);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
