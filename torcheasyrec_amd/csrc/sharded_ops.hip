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
#include "tzr_common.h"

#define SH_THREADS 256

// out[j, :] = W_{table(key(j))}[ids[j], :] for j in [0, n): key(j) = segment of key_start holding j.
__global__ __launch_bounds__(SH_THREADS) void tzr_rows_gather_kernel(
    const TzrTable* __restrict__ tables, const int32_t* __restrict__ key_table,
    const int64_t* __restrict__ key_start, int n_keys, const int64_t* __restrict__ ids, int64_t n,
    float* __restrict__ out, int64_t out_stride, int lg) {
  const int64_t total = n * lg;
  for (int64_t k = (int64_t)blockIdx.x * SH_THREADS + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * SH_THREADS) {
    const int64_t j = k / lg;
    const int c = (int)(k - j * lg);
    const int64_t key = tzr_last_le(key_start, n_keys, j);
    const TzrTable tb = tables[key_table[key]];
    int64_t id = ids[j];
    if ((uint64_t)id >= (uint64_t)tb.rows) id = 0;
    float4 v = tzr_zero4();
    if (4 * c < tb.dim)
      v = tzr_ld4(reinterpret_cast<const float*>(tb.w) + id * (int64_t)tb.w_stride + 4 * c);
    tzr_st4(out + j * out_stride + 4 * c, v);
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
  const int64_t total = n_ids * lg;
  const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + SH_THREADS - 1) / SH_THREADS);
  hipLaunchKernelGGL(tzr_rows_gather_kernel, dim3(grid), dim3(SH_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_tables, d_key_table, d_key_start, n_keys,
                     d_ids, n_ids, d_out, out_stride, lg);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

struct ShGrads {
  TzrDst d[TZR_MAX_DST];
};

// out[pos[i], :] = scale_i * sum_{groups g of key f} grad_g[b, col_g(f) : +dim] for every id i of
// bag (f, b); scale_i = weight_i (/ len(bag) for mean pooling).  thread = (sample, lookup, chunk).
__global__ __launch_bounds__(SH_THREADS) void tzr_lookup_grads_kernel(
    const TzrFeature* __restrict__ feats, int n_feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ weights, int64_t B, int uniform,
    const int64_t* __restrict__ positions, ShGrads G, float* __restrict__ out, int64_t out_stride,
    int lg) {
  const int64_t total = B * n_feats * lg;
  for (int64_t k = (int64_t)blockIdx.x * SH_THREADS + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * SH_THREADS) {
    const int c = (int)(k % lg);
    const int64_t r = k / lg;
    const int f = (int)(r % n_feats);
    const int64_t b = r / n_feats;
    const TzrFeature ft = feats[f];
    const int64_t bag = (int64_t)ft.key * B + b;
    const int64_t st = uniform ? bag : offsets[bag];
    const int64_t en = uniform ? bag + 1 : offsets[bag + 1];
    if (st >= en) continue;
    float4 g = tzr_zero4();
    for (int d = 0; d < ft.n_dst; ++d)
      g = tzr_add4(g, tzr_ld4(reinterpret_cast<const float*>(G.d[ft.dst[d]].ptr) +
                              b * G.d[ft.dst[d]].stride + ft.col[d] + 4 * c));
    const float inv = (ft.pooling == TZR_POOL_MEAN && en - st > 1) ? 1.0f / (float)(en - st) : 1.0f;
    for (int64_t i = st; i < en; ++i) {
      const float sc = (weights ? weights[i] : 1.0f) * inv;
      const int64_t p = positions ? positions[i] : i;
      tzr_st4(out + p * out_stride + 4 * c, make_float4(g.x * sc, g.y * sc, g.z * sc, g.w * sc));
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
  const int64_t total = B * n_feats * lg;
  const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + SH_THREADS - 1) / SH_THREADS);
  hipLaunchKernelGGL(tzr_lookup_grads_kernel, dim3(grid), dim3(SH_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_feats, n_feats, d_offsets, d_weights, B,
                     (int)uniform, d_positions, G, d_out, out_stride, lg);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
