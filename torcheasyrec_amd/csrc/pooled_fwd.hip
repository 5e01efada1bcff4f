// K5 (+K8 fused): pooled embedding gather forward for gfx950.
//
// Replaces torchrec EmbeddingBagCollection.forward -> fbgemm
// split_embedding_codegen_forward_{un,}weighted_kernel (self.ebc(kjt),
// /root/reference/tzrec/modules/embedding.py:930) fused with the feature-group regroup copy
// (KeyedTensor.regroup_as_dict, embedding.py:972-976).
//
// Mapping (HBM-bound, no reuse => no MFMA, no LDS tile of rows):
//   * a workgroup owns `tile_b` consecutive samples x up to FWD_MAX_SLOTS float4 "slots" of their
//     destination rows; thread k of the tile handles (sample k / ns, slot k % ns), so consecutive
//     lanes write consecutive float4s of one destination row (full 128-B lines, 1 KiB per wave
//     store) and the D/4 lanes of a feature read one embedding row as contiguous 16-B pieces;
//   * the per-slot metadata (row base + chunk, destination base) is resolved once per workgroup
//     into LDS, so the id -> row dependent chain is two loads (id, row);
//   * every thread keeps FWD_UNROLL independent gathers in flight (random 64-B row reads are
//     latency-bound: ~1 us under load x 8 TB/s needs ~16 rows in flight per wave);
//   * ids of a tile are `tile_b` consecutive int64 per feature (key-major KJT), re-used from L1 by
//     the lanes of the tile.
#include "tzr_common.h"

#ifdef IT_PROF  // scripts/build_prof_lib.sh: wall-clock stamps (100 MHz) of every workgroup's phases, read back by tzr_fwd_prof_dump
#define FWD_PROF_WGS 4096
__device__ uint64_t g_fwd_prof[FWD_PROF_WGS * 4];
#define FWD_PROF_MARK(i) do { if (blockIdx.y == 0 && blockIdx.x < FWD_PROF_WGS && threadIdx.x == 0) g_fwd_prof[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
extern "C" int tzr_fwd_prof_dump(uint64_t* h_out, int n_wg) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_fwd_prof), (size_t)std::min(n_wg, FWD_PROF_WGS) * 4 * sizeof(uint64_t)) == hipSuccess ? 0 : -1;
}
#else
#define FWD_PROF_MARK(i)
#endif

#include "pooled_fwd_u1.h"


struct FwdRSlot {  // resolved slot, 40 bytes
  const void* w;   // table row 0 + the slot's float4 column group (elements fp32 or fp16, see w_dtype)
  float* dst;      // destination buffer + col
  int64_t rows;    // ids outside [0, rows) read row 0 (memory safety; K4 counts / reports them)
  int32_t dst_stride;
  int32_t w_stride;
  int32_t feat;  // KJT key index
  int16_t pooling;
  int16_t w_dtype;
};

int g_tzr_fwd_tile_b = 0;  // 0 = auto; set through tzr_tune("fwd_tile_b", v)

template <bool UNIFORM1, bool WEIGHTED, bool MIXED>
__global__ __launch_bounds__(FWD_THREADS) void tzr_pooled_fwd_kernel(
    const TzrTable* __restrict__ tables, const TzrFeature* __restrict__ feats,
    const TzrSlot* __restrict__ slots, int n_slots, const int64_t* __restrict__ values,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int tile_b,
    FwdDsts dsts) {
  __shared__ FwdRSlot rs[FWD_MAX_SLOTS];
  const int s0 = blockIdx.y * FWD_MAX_SLOTS;
  const int ns = min(FWD_MAX_SLOTS, n_slots - s0);
  for (int s = threadIdx.x; s < ns; s += FWD_THREADS) {
    TzrSlot sl = slots[s0 + s];
    TzrFeature ft = feats[sl.feature];
    TzrTable tb = tables[ft.table];
    FwdRSlot r;
    r.w = reinterpret_cast<const char*>(tb.w) + (size_t)sl.chunk * 4 * (tb.w_dtype == TZR_DT_F16 ? 2 : 4);
    r.w_dtype = (int16_t)tb.w_dtype;
    r.dst = reinterpret_cast<float*>(dsts.d[sl.dst].ptr) + sl.col;
    r.dst_stride = (int32_t)dsts.d[sl.dst].stride;
    r.w_stride = tb.w_stride;
    r.rows = tb.rows;
    r.feat = ft.key;
    r.pooling = ft.pooling;
    rs[s] = r;
  }
  __syncthreads();

  const int64_t b0 = (int64_t)blockIdx.x * tile_b;
  const int nb = (int)min((int64_t)tile_b, B - b0);
  const int total = nb * ns;

  for (int k0 = threadIdx.x; k0 < total; k0 += FWD_THREADS * FWD_UNROLL) {
    const void* wp[FWD_UNROLL];
    int wdt[FWD_UNROLL];
    float* dp[FWD_UNROLL];
    int64_t st[FWD_UNROLL], en[FWD_UNROLL];
    int32_t wstride[FWD_UNROLL];
    int64_t rows[FWD_UNROLL];
    int32_t pool[FWD_UNROLL];
    float4 acc[FWD_UNROLL];
#pragma unroll
    for (int u = 0; u < FWD_UNROLL; ++u) {
      int k = k0 + u * FWD_THREADS;
      const bool ok = k < total;
      if (UNIFORM1) k = ok ? k : total - 1;  // (one id per bag: an element behind the tile repeats the last one and is not stored -- no condition around its loads)
      const int bl = (ok || UNIFORM1) ? k / ns : 0;
      const int s = (ok || UNIFORM1) ? k - bl * ns : 0;
      const FwdRSlot r = rs[s];
      const int64_t b = b0 + bl;
      const int64_t bag = (int64_t)r.feat * B + b;
      wp[u] = r.w;
      wdt[u] = MIXED ? r.w_dtype : TZR_DT_F32;  // fp32-only launches compile the branch away
      wstride[u] = r.w_stride;
      rows[u] = r.rows;
      pool[u] = r.pooling;
      dp[u] = r.dst + b * (int64_t)r.dst_stride;
      if (UNIFORM1) {
        st[u] = bag;
        en[u] = bag + 1;
      } else {
        st[u] = ok ? offsets[bag] : 0;
        en[u] = ok ? offsets[bag + 1] : 0;
      }
      acc[u] = tzr_zero4();
    }
    if (UNIFORM1) {
      // one id per bag: FWD_UNROLL independent id loads, then FWD_UNROLL independent row loads
      int64_t id[FWD_UNROLL];
      float sc[FWD_UNROLL];
#pragma unroll
      // (no lane-dependent condition around a load: to hipcc that is a branch, and a branch between two loads a wait)
      for (int u = 0; u < FWD_UNROLL; ++u) {
        id[u] = values[st[u]];
        sc[u] = WEIGHTED ? weights[st[u]] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < FWD_UNROLL; ++u)
        if ((uint64_t)id[u] >= (uint64_t)rows[u]) id[u] = 0;
#pragma unroll
      for (int u = 0; u < FWD_UNROLL; ++u) {
        float4 v = tzr_ldw4(wp[u], wdt[u], id[u] * (int64_t)wstride[u]);
        if (WEIGHTED) {
          v.x *= sc[u]; v.y *= sc[u]; v.z *= sc[u]; v.w *= sc[u];
        }
        acc[u] = v;
      }
    } else {
      int64_t maxlen = 0;
#pragma unroll
      for (int u = 0; u < FWD_UNROLL; ++u) maxlen = max(maxlen, en[u] - st[u]);
      for (int64_t j = 0; j < maxlen; ++j) {
        int64_t id[FWD_UNROLL];
        float sc[FWD_UNROLL];
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u) {
          const bool on = st[u] + j < en[u];
          id[u] = on ? values[st[u] + j] : 0;
          if ((uint64_t)id[u] >= (uint64_t)rows[u]) id[u] = 0;
          sc[u] = (WEIGHTED && on) ? weights[st[u] + j] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < FWD_UNROLL; ++u) {
          if (st[u] + j < en[u]) {
            const float4 v = tzr_ldw4(wp[u], wdt[u], id[u] * (int64_t)wstride[u]);
            // accumulation order = bag order, one fmaf per element (matches the oracle's
            // sequential fp32 sum up to fma contraction of the per-sample weight)
            acc[u] = WEIGHTED ? tzr_fma4(sc[u], v, acc[u]) : tzr_add4(acc[u], v);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < FWD_UNROLL; ++u) {
        const int64_t len = en[u] - st[u];
        if (pool[u] == TZR_POOL_MEAN && len > 1) {
          const float inv = 1.0f / (float)len;
          acc[u].x *= inv; acc[u].y *= inv; acc[u].z *= inv; acc[u].w *= inv;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < FWD_UNROLL; ++u) {
      if (k0 + u * FWD_THREADS < total) tzr_stg4(dp[u], acc[u]);
    }
  }
}

int g_tzr_fwd_variant = 0;  // tzr_tune("fwd_variant"): 0 = by shape, 1 = general kernel only, 2 = LDS-ids kernel whenever eligible

#ifndef FWD1_WAVES  // (timing experiments: -DFWD1_WAVES=n compiles the kernel for n waves per SIMD, -DFWD1_UNROLL=m gathers in flight per thread)
#define FWD1_WAVES_ATTR
#else
#define FWD1_WAVES_ATTR TZR_WAVES_PER_EU(FWD1_WAVES)
#endif
__global__ __launch_bounds__(FWD_THREADS) FWD1_WAVES_ATTR void tzr_pooled_fwd_u1_kernel(
    const TzrTable* __restrict__ tables, const TzrFeature* __restrict__ feats,
    const TzrSlot* __restrict__ slots, int n_slots, const int64_t* __restrict__ values, int64_t B,
    int tile_b, FwdDsts dsts) {
  __shared__ Fwd1Lds S;
  fwd_u1_body<FWD1_UNROLL>(tables, feats, slots, n_slots, values, B, tile_b, dsts, blockIdx.x, blockIdx.y, S);
}

// flags: TZR_FWD_MIXED_DTYPE = some table holds fp16 rows (TzrTable.w_dtype is honoured); without it
// every table is read as fp32 and the kernel carries no per-row dtype test (measured: the test costs
// 4-5 us of the 49 us DLRM-Criteo forward).
extern "C" int tzr_pooled_fwd_ex(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                                 const TzrSlot* d_slots, int n_slots, const int64_t* d_values,
                                 const int64_t* d_offsets, const float* d_weights, int64_t B,
                                 const TzrDst* h_dsts, int n_dst, int uniform_bag_len, int flags,
                                 void* stream) {
  if (!d_tables || !d_feats || !d_slots || !h_dsts || n_feats <= 0 || n_slots <= 0 || B < 0 ||
      n_dst <= 0 || n_dst > TZR_MAX_DST || (flags & ~TZR_FWD_MIXED_DTYPE))
    return TZR_ERR_INVALID;
  if (uniform_bag_len != 1 && !d_offsets) return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  if (!d_values) return TZR_ERR_INVALID;
  FwdDsts dsts;
  for (int i = 0; i < TZR_MAX_DST; ++i) {
    dsts.d[i].ptr = 0;
    dsts.d[i].stride = 0;
  }
  for (int i = 0; i < n_dst; ++i) {
    if (!h_dsts[i].ptr || (h_dsts[i].stride & 3) || (h_dsts[i].ptr & 15)) return TZR_ERR_INVALID;
    if (h_dsts[i].stride > 0x7fffffffLL) return TZR_ERR_UNSUPPORTED;
    dsts.d[i] = h_dsts[i];
  }
  const bool u1 = uniform_bag_len == 1;
  const bool wt = d_weights != nullptr;
  const bool mx = (flags & TZR_FWD_MIXED_DTYPE) != 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // measured on MI355X (profiles/r03a/emb_ab.txt): equal on uniform ids at B = 65536 (48-49 us both), 42.6 vs
  // 45.2 us on Zipf ids; at B = 8192 the general kernel is faster (9.9 vs 10.9-12.4 us: the prologue is paid by
  // more, smaller workgroups) -- so by batch size unless forced (fwd_variant 2)
  if (u1 && !wt && !mx && (g_tzr_fwd_variant == 2 || (g_tzr_fwd_variant == 0 && B >= 32768))) {
    // tiles of 32 samples at Criteo batch sizes: 26 id groups x 32 = 832 ids per workgroup in LDS
    const int tb1 = g_tzr_fwd_tile_b > 0 ? g_tzr_fwd_tile_b : (B >= 32768 ? 32 : (B >= 8192 ? 16 : 8));
    dim3 grid1((unsigned)((B + tb1 - 1) / tb1), (unsigned)((n_slots + FWD1_SLOTS - 1) / FWD1_SLOTS));
    hipLaunchKernelGGL(tzr_pooled_fwd_u1_kernel, grid1, dim3(FWD_THREADS), 0, s, d_tables, d_feats, d_slots, n_slots,
                       d_values, B, tb1, dsts);
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  int tile_b = g_tzr_fwd_tile_b;
  // measured on MI355X (scripts/fwd_sweep.py): B=65536 runs ~3 us faster with 128-sample tiles
  if (tile_b <= 0) tile_b = B <= 16384 ? 8 : (B <= 32768 ? 16 : 128);
  dim3 grid((unsigned)((B + tile_b - 1) / tile_b), (unsigned)((n_slots + FWD_MAX_SLOTS - 1) / FWD_MAX_SLOTS));
#define TZR_FWD_LAUNCH(U, W, M)                                                                  \
  hipLaunchKernelGGL((tzr_pooled_fwd_kernel<U, W, M>), grid, dim3(FWD_THREADS), 0, s, d_tables,  \
                     d_feats, d_slots, n_slots, d_values, d_offsets, d_weights, B, tile_b, dsts)
#define TZR_FWD_PICK(M)                      \
  do {                                       \
    if (u1 && wt) TZR_FWD_LAUNCH(true, true, M);        \
    else if (u1) TZR_FWD_LAUNCH(true, false, M);        \
    else if (wt) TZR_FWD_LAUNCH(false, true, M);        \
    else TZR_FWD_LAUNCH(false, false, M);               \
  } while (0)
  if (mx) TZR_FWD_PICK(true);
  else TZR_FWD_PICK(false);
#undef TZR_FWD_PICK
#undef TZR_FWD_LAUNCH
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_pooled_fwd(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                              const TzrSlot* d_slots, int n_slots, const int64_t* d_values,
                              const int64_t* d_offsets, const float* d_weights, int64_t B,
                              const TzrDst* h_dsts, int n_dst, int uniform_bag_len, void* stream) {
  return tzr_pooled_fwd_ex(d_tables, d_feats, n_feats, d_slots, n_slots, d_values, d_offsets, d_weights, B,
                           h_dsts, n_dst, uniform_bag_len, 0, stream);
}
