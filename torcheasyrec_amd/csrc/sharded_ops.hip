// Owner / requester kernels of the row-wise sharded exchange (one process per GPU, ids and rows
// travel by RCCL all-to-all over xGMI; see torcheasyrec_amd/sharding.py).
//
// The reference reaches this through torchrec's sharded EmbeddingBagCollection [upstream 1.7.0]:
// KJTAllToAll -> per-shard TBE lookup -> PooledEmbeddingsReduceScatter, i.e. every rank returns a
// dense [B_local, F*D] partial to every peer.  On a point-to-point xGMI mesh that costs 7x the
// bytes of what is needed when bags are short, so this build exchanges at id granularity:
//   forward   requester: bucketize ids by owner (K2)            -> all-to-all ids
//             owner:     tzr_rows_gather: one row per id        -> all-to-all rows
//             requester: K5 pooled gather over the received rows (ids = unbucketize positions)
//   backward  requester: tzr_lookup_grads: one gradient row per id -> all-to-all
//             owner:     K6 plan + K7 apply with per-id gradient rows (grad_mode 1)
// Both kernels are HBM-bound gathers/scatters of 64-byte rows; all per-key / per-lookup metadata
// is resolved once per workgroup into LDS so a lane's dependent chain is id -> row.
#include "tzr_common.h"

#define SH_THREADS 256
#define SH_MAXKEYS 256  // keys (owner: W*F segments; requester: lookups) staged in LDS per pass
#define SH_GB 4         // rows-gather items in flight per thread

struct ShKey {  // resolved key segment on the owner
  const void* w;
  int64_t rows;
  int32_t w_stride;
  int32_t dim;
  int32_t w_dtype;
  int32_t pad;
};

__device__ __forceinline__ void sh_resolve_key(const TzrTable* __restrict__ tables, int t, ShKey* e) {
  if (t < 0) {  // dead key
    e->w = nullptr;
    e->w_dtype = 0;
    e->rows = 0;
    e->w_stride = 0;
    e->dim = -1;
    e->pad = 0;
    return;
  }
  const TzrTable tb = tables[t];
  e->w = reinterpret_cast<const void*>(tb.w);
  e->w_dtype = tb.w_dtype;
  e->rows = tb.rows;
  e->w_stride = tb.w_stride;
  e->dim = tb.dim;
  e->pad = 0;
}

// out[j, :] = W_{table(key(j))}[ids[j], :] for j in [0, n): key(j) = segment of key_start holding j
// (key_table[k] < 0: a dead key, its positions are left untouched).
__global__ __launch_bounds__(SH_THREADS) void tzr_rows_gather_kernel(
    const TzrTable* __restrict__ tables, const int32_t* __restrict__ key_table,
    const int64_t* __restrict__ key_start, int n_keys, const int64_t* __restrict__ ids, int64_t n,
    float* __restrict__ out, int64_t out_stride, int lg, int rows_per_block) {
  __shared__ int64_t s_start[SH_MAXKEYS + 1];
  __shared__ ShKey s_key[SH_MAXKEYS];
  const int64_t j0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t j1 = min(n, j0 + rows_per_block);
  // the keys overlapping this block's rows: [k0, k1]
  int k0 = (int)tzr_last_le(key_start, n_keys, j0);
  int k1 = (int)tzr_last_le(key_start, n_keys, j1 - 1);
  const bool staged = k1 - k0 < SH_MAXKEYS;
  if (staged) {
    for (int k = k0 + threadIdx.x; k <= k1; k += SH_THREADS) {
      ShKey e;
      sh_resolve_key(tables, key_table[k], &e);
      s_key[k - k0] = e;
      s_start[k - k0] = key_start[k];
    }
    if (threadIdx.x == 0) s_start[k1 - k0 + 1] = key_start[k1 + 1];
  }
  __syncthreads();
  // SH_GB (row, column group) items per thread at a time: ids, then rows, then stores -- one memory round
  // trip per stage for the batch instead of a dependent id -> row chain per item
  const int total = (int)(j1 - j0) * lg;
  for (int kb = threadIdx.x; kb < total; kb += SH_GB * SH_THREADS) {
    ShKey e[SH_GB];
    int64_t id[SH_GB];
    float4 v[SH_GB];
#pragma unroll
    for (int u = 0; u < SH_GB; ++u) {
      const int k = kb + u * SH_THREADS;
      e[u].dim = -1;
      id[u] = 0;
      if (k < total) {
        const int64_t j = j0 + k / lg;
        if (staged) {
          int a = 0, b = k1 - k0 + 1;  // s_start[a] <= j < s_start[b]
          while (b - a > 1) {
            const int m = (a + b) >> 1;
            if (s_start[m] <= j) a = m; else b = m;
          }
          e[u] = s_key[a];
        } else {
          sh_resolve_key(tables, key_table[tzr_last_le(key_start, n_keys, j)], &e[u]);
        }
        if (e[u].dim >= 0) id[u] = ids[j];  // (dead key: positions nobody reads -- capacity-bounded exchange)
      }
    }
#pragma unroll
    for (int u = 0; u < SH_GB; ++u) {
      const int c = (kb + u * SH_THREADS) % lg;
      if ((uint64_t)id[u] >= (uint64_t)e[u].rows) id[u] = 0;
      v[u] = tzr_zero4();
      if (e[u].dim >= 0 && 4 * c < e[u].dim) v[u] = tzr_ldw4(e[u].w, e[u].w_dtype, id[u] * (int64_t)e[u].w_stride + 4 * c);
    }
#pragma unroll
    for (int u = 0; u < SH_GB; ++u) {
      const int k = kb + u * SH_THREADS;
      if (e[u].dim >= 0) tzr_st4(out + (j0 + k / lg) * out_stride + 4 * (k % lg), v[u]);
    }
  }
}

extern "C" int tzr_rows_gather(const TzrTable* d_tables, const int32_t* d_key_table,
                               const int64_t* d_key_start, int n_keys, const int64_t* d_ids,
                               int64_t n_ids, float* d_out, int64_t out_stride, int dim,
                               void* stream) {
  if (!d_tables || !d_key_table || !d_key_start || n_keys <= 0 || n_ids < 0 || dim <= 0 ||
      (dim & 3) || (out_stride & 3) || out_stride < dim)
    return TZR_ERR_INVALID;
  if (n_ids == 0) return TZR_OK;
  if (!d_ids || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 15)) return TZR_ERR_INVALID;
  const int lg = dim >> 2;
  // one batch of SH_GB items per thread (256 rows per workgroup at dim 16): 8 192 x 21 received ids are 840
  // workgroups, not 210 that leave a fifth of the CUs empty and walk 16 dependent chains per thread
  // (profiles/r03az: 26 us for 13.7 MB)
  int rows_per_block = SH_GB * SH_THREADS / lg;
  if (rows_per_block < 16) rows_per_block = 16;
  const unsigned grid = (unsigned)((n_ids + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(tzr_rows_gather_kernel, dim3(grid), dim3(SH_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_tables, d_key_table, d_key_start, n_keys,
                     d_ids, n_ids, d_out, out_stride, lg, rows_per_block);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

struct ShGrads {
  TzrDst d[TZR_MAX_DST];
};

struct ShLookup {  // resolved requester lookup
  const float* gp[TZR_MAX_FEAT_DST];  // group gradient buffer + first column (statically indexed)
  int64_t gs[TZR_MAX_FEAT_DST];
  int32_t key;
  int32_t n_dst;
  int32_t mean;
  int32_t pad;
};

// out[pos[i], :] = scale_i * sum_{groups g of key f} grad_g[b, col_g(f) : +dim] for every id i of
// bag (f, b); scale_i = weight_i (/ len(bag) for mean pooling).  workgroup = tile of samples x all
// lookups; thread = (sample, lookup, chunk) with chunk fastest, so the D/4 lanes of a lookup read
// one 64-byte gradient slice and write one 64-byte row.
__global__ __launch_bounds__(SH_THREADS) void tzr_lookup_grads_kernel(
    const TzrFeature* __restrict__ feats, int n_feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ weights, int64_t B, int uniform,
    const int64_t* __restrict__ positions, ShGrads G, float* __restrict__ out, int64_t out_stride,
    int lg, int tile_b) {
  __shared__ ShLookup s_lk[SH_MAXKEYS];
  __shared__ TzrDst s_g[TZR_MAX_DST];
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < TZR_MAX_DST; ++i) s_g[i] = G.d[i];
  }
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * tile_b;
  const int nb = (int)min((int64_t)tile_b, B - b0);
  for (int f0 = 0; f0 < n_feats; f0 += SH_MAXKEYS) {
    const int nf = min(SH_MAXKEYS, n_feats - f0);
    __syncthreads();
    for (int f = threadIdx.x; f < nf; f += SH_THREADS) {
      const TzrFeature* ft = feats + f0 + f;
      ShLookup e;
      e.key = ft->key;
      e.n_dst = ft->n_dst;
      e.mean = ft->pooling == TZR_POOL_MEAN;
      e.pad = 0;
#pragma unroll
      for (int d = 0; d < TZR_MAX_FEAT_DST; ++d) {
        const int di = d < ft->n_dst ? ft->dst[d] : 0;
        e.gp[d] = reinterpret_cast<const float*>(s_g[di].ptr) + ft->col[d];
        e.gs[d] = s_g[di].stride;
      }
      s_lk[f] = e;
    }
    __syncthreads();
    const int total = nb * nf * lg;
    for (int k = threadIdx.x; k < total; k += SH_THREADS) {
      const int c = k % lg;
      const int r = k / lg;
      const int f = r % nf;
      const int64_t b = b0 + r / nf;
      const ShLookup* e = &s_lk[f];
      const int64_t bag = (int64_t)e->key * B + b;
      const int64_t st = uniform ? bag : offsets[bag];
      const int64_t en = uniform ? bag + 1 : offsets[bag + 1];
      if (st >= en) continue;
      float4 g = tzr_ld4(e->gp[0] + b * e->gs[0] + 4 * c);
      if (e->n_dst > 1) g = tzr_add4(g, tzr_ld4(e->gp[1] + b * e->gs[1] + 4 * c));
      if (e->n_dst > 2) g = tzr_add4(g, tzr_ld4(e->gp[2] + b * e->gs[2] + 4 * c));
      if (e->n_dst > 3) g = tzr_add4(g, tzr_ld4(e->gp[3] + b * e->gs[3] + 4 * c));
      const float inv = (e->mean && en - st > 1) ? 1.0f / (float)(en - st) : 1.0f;
      for (int64_t i = st; i < en; ++i) {
        const float sc = (weights ? weights[i] : 1.0f) * inv;
        const int64_t p = positions ? positions[i] : i;
        tzr_st4(out + p * out_stride + 4 * c, make_float4(g.x * sc, g.y * sc, g.z * sc, g.w * sc));
      }
    }
  }
}

extern "C" int tzr_lookup_grads(const TzrFeature* d_feats, int n_feats, const int64_t* d_offsets,
                                const float* d_weights, int64_t B, int uniform_bag_len,
                                const int64_t* d_positions, const TzrDst* h_grads, int n_dst,
                                float* d_out, int64_t out_stride, int dim, void* stream) {
  if (!d_feats || n_feats <= 0 || B < 0 || !h_grads || n_dst <= 0 || n_dst > TZR_MAX_DST ||
      dim <= 0 || (dim & 3) || (out_stride & 3) || out_stride < dim)
    return TZR_ERR_INVALID;
  const bool uniform = uniform_bag_len == 1;
  if (!uniform && !d_offsets) return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 15)) return TZR_ERR_INVALID;
  ShGrads G;
  for (int i = 0; i < TZR_MAX_DST; ++i) {
    G.d[i].ptr = 0;
    G.d[i].stride = 0;
  }
  for (int i = 0; i < n_dst; ++i) {
    if (!h_grads[i].ptr || (h_grads[i].stride & 3) || (h_grads[i].ptr & 15)) return TZR_ERR_INVALID;
    G.d[i] = h_grads[i];
  }
  const int lg = dim >> 2;
  const int tile_b = B <= 16384 ? 8 : 32;
  const unsigned grid = (unsigned)((B + tile_b - 1) / tile_b);
  hipLaunchKernelGGL(tzr_lookup_grads_kernel, dim3(grid), dim3(SH_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_feats, n_feats, d_offsets, d_weights, B,
                     (int)uniform, d_positions, G, d_out, out_stride, lg, tile_b);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
