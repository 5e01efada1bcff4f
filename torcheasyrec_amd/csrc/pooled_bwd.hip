// K6 + K7: backward index plan and fused sparse-optimizer update for gfx950.
//
// Replaces fbgemm transpose_embedding_input (linearize + cub radix sort + run-length) and
// split_embedding_backward_codegen_{sgd,adagrad,rowwise_adagrad}_*_exact_{warp,cta}_per_row_1,
// reached from autograd of self.ebc(kjt) (/root/reference/tzrec/modules/embedding.py:930) with the
// optimizer fused by apply_optimizer_in_backward (/root/reference/tzrec/main.py:774-781).
//
// Plan (K6).  Lookups are regrouped table-major (keys of one table adjacent, key order kept), then
// each table segment is sorted by local row id with a stable LSD radix sort whose digit width and
// pass count are per table: bits = ceil(log2 rows), passes = ceil(bits / 9).  A 3-row table costs
// one pass, a 40M-row table three.  Stability + the fixed table-major start order make the final
// order (row, original lookup position): every summation order below is a function of the ids
// alone => bit-reproducible updates, no float atomics anywhere.
//
// Apply (K7) lives in pooled_bwd_apply.hip; the plan reaches it through the workspace (pooled_bwd.h).
#include "pooled_bwd.h"

extern "C" size_t tzr_pooled_bwd_workspace(int64_t n_values, int64_t n_positions, int n_feats,
                                           int n_tables, int64_t B, int max_dim) {
  (void)B;
  if (n_values < 0 || n_positions < 0 || n_feats <= 0 || n_tables <= 0 || max_dim <= 0) return 0;
  return bwd_layout(nullptr, nullptr, n_values, n_positions, n_feats, n_tables, max_dim) + 256;
}

// ------------------------------------------------------------------------------------------
// plan kernels
// ------------------------------------------------------------------------------------------

// One workgroup: table-major segment starts, chunk map, per-table digit geometry.
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_prep_kernel(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats, int F,
    const int64_t* __restrict__ offsets, int64_t B, int uniform, BwdPlan P) {
  for (int f = threadIdx.x; f < F; f += BWD_THREADS) {
    const int o = feats[f].order;
    // keys of the KJT this module does not own (table < 0) are ordered last and contribute nothing
    const int64_t key = feats[f].key;
    const int64_t n =
        feats[f].table < 0 ? 0 : (uniform ? B : offsets[(key + 1) * B] - offsets[key * B]);
    P.feat_start[o] = n;
    P.feat_by_order[o] = f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int o = 0; o < F; ++o) {
      const int64_t n = P.feat_start[o];
      P.feat_start[o] = run;
      run += n;
    }
    P.feat_start[F] = run;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += BWD_THREADS) {
    const TzrTable tb = tables[t];
    const int64_t s = tb.n_feats > 0 ? P.feat_start[tb.first_order] : 0;
    const int64_t e = tb.n_feats > 0 ? P.feat_start[tb.first_order + tb.n_feats] : 0;
    P.tab_start[t] = s;
    P.tab_chunk[t] = (int32_t)((e - s + BWD_CH - 1) / BWD_CH);
    const int bits = tb.rows <= 1 ? 0 : 64 - __clzll((long long)(tb.rows - 1));
    // at least one pass: pass 0 is also the table-major regroup of the lookups
    const int npass = bits == 0 ? 1 : (bits + BWD_RB - 1) / BWD_RB;
    P.tab_npass[t] = npass;
    P.tab_width[t] = (bits + npass - 1) / npass;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // tables are visited in first_order order == table-major position order only if table ids
    // follow it; starts are absolute, so the chunk map just needs a prefix in table-id order.
    int32_t run = 0;
    for (int t = 0; t < T; ++t) {
      const int32_t n = P.tab_chunk[t];
      P.tab_chunk[t] = run;
      run += n;
    }
    P.tab_chunk[T] = run;
    P.tab_start[T] = P.feat_start[F];
  }
}

// Table-major position p of table tb -> (local row, original lookup position): the lookups of a
// table are the concatenation, in key order, of the id segments of the keys that read it.
struct BwdSrcArgs {
  const TzrFeature* feats;
  const int64_t* values;
  const int64_t* offsets;
  int64_t B;
  int uniform;
};

__device__ __forceinline__ void bwd_elem0(const BwdPlan& P, const TzrTable& tb, const BwdSrcArgs& A,
                                          int64_t p, uint32_t* key_out, uint32_t* src_out,
                                          int64_t* kjt_key_out) {
  int o = tb.first_order;
  while (o + 1 < tb.first_order + tb.n_feats && P.feat_start[o + 1] <= p) ++o;
  const int64_t key = A.feats[P.feat_by_order[o]].key;
  const int64_t fbase = A.uniform ? key * A.B : A.offsets[key * A.B];
  const int64_t i = fbase + (p - P.feat_start[o]);
  int64_t id = A.values[i];
  if ((uint64_t)id >= (uint64_t)tb.rows) id = 0;  // memory safety; K4 reports/clamps
  *key_out = (uint32_t)id;
  *src_out = (uint32_t)i;
  *kjt_key_out = key;
}

__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_hist_kernel(
    const TzrTable* __restrict__ tables, int T, int pass, BwdSrcArgs A, BwdPlan P) {
  __shared__ unsigned h[BWD_NB];
  BwdChunkDesc cd;
  if (pass == 0) {
    // the first kernel after prep resolves chunk -> table once and leaves the descriptor for every
    // later kernel of the plan and the apply (see BwdChunkDesc)
    const int c = blockIdx.x;
    cd.t = -1;
    cd.width = cd.npass = cd.last_chunk = 0;
    cd.s = cd.e = cd.ts = cd.te = 0;
    if (c < P.tab_chunk[T]) {
      int lo = 0, hi = T;  // last t with tab_chunk[t] <= c (the non-empty table holding it)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (P.tab_chunk[mid] <= c) lo = mid; else hi = mid;
      }
      const TzrTable tb = tables[lo];
      cd.t = lo;
      cd.width = P.tab_width[lo];
      cd.npass = P.tab_npass[lo];
      cd.last_chunk = P.tab_chunk[lo + 1];
      cd.ts = P.tab_start[lo];
      cd.te = tb.n_feats > 0 ? P.feat_start[tb.first_order + tb.n_feats] : cd.ts;
      cd.s = cd.ts + (int64_t)(c - P.tab_chunk[lo]) * BWD_CH;
      cd.e = min(cd.te, cd.s + (int64_t)BWD_CH);
    }
    if (threadIdx.x == 0) P.cdesc[c] = cd;
    if (cd.t < 0) return;
  } else if (!bwd_chunk(P, blockIdx.x, &cd)) {
    return;
  }
  if (cd.npass <= pass) return;
  const int t = cd.t;
  const int64_t s = cd.s, e = cd.e;
  const int width = cd.width;
  const int shift = pass * width;
  const unsigned mask = (1u << width) - 1u;
  const uint2* __restrict__ kin = (pass & 1) ? P.ks[1] : P.ks[0];
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) h[i] = 0;
  // all of the chunk's keys are loaded before any is counted: independent loads in flight, one
  // memory latency per workgroup instead of one per element
  constexpr int kRounds = BWD_CH / BWD_THREADS;
  uint32_t kreg[kRounds];
  if (pass == 0) {
    const TzrTable tb = tables[t];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int64_t p = s + (int64_t)r * BWD_THREADS + threadIdx.x;
      uint32_t sv;
      int64_t kk;
      kreg[r] = 0u;
      if (p < e) bwd_elem0(P, tb, A, p, &kreg[r], &sv, &kk);
    }
  } else {
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int64_t p = s + (int64_t)r * BWD_THREADS + threadIdx.x;
      kreg[r] = p < e ? kin[p].x : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t p = s + (int64_t)r * BWD_THREADS + threadIdx.x;
    if (p < e) atomicAdd(&h[(kreg[r] >> shift) & mask], 1u);
  }
  __syncthreads();
  uint32_t* out = P.hist + (size_t)blockIdx.x * BWD_NB;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) out[i] = h[i];
}

// One workgroup per table, one thread per digit: exclusive scan over the table's chunks (in
// place) and over digits -> binbase.  Chunk columns are read in batches of independent loads.
#define BWD_SCAN_BATCH 64
__global__ __launch_bounds__(BWD_NB) void tzr_bwd_scan_kernel(int T, int pass, BwdPlan P) {
  __shared__ unsigned tot[BWD_NB];
  const int t = blockIdx.x;
  if (P.tab_npass[t] <= pass) return;
  const int c0 = P.tab_chunk[t];
  const int C = P.tab_chunk[t + 1] - c0;
  const int bin = threadIdx.x;
  unsigned run = 0;
  for (int cb = 0; cb < C; cb += BWD_SCAN_BATCH) {
    unsigned v[BWD_SCAN_BATCH];
#pragma unroll
    for (int j = 0; j < BWD_SCAN_BATCH; ++j)
      v[j] = (cb + j < C) ? P.hist[(size_t)(c0 + cb + j) * BWD_NB + bin] : 0u;
#pragma unroll
    for (int j = 0; j < BWD_SCAN_BATCH; ++j) {
      if (cb + j < C) P.hist[(size_t)(c0 + cb + j) * BWD_NB + bin] = run;
      run += v[j];
    }
  }
  tot[bin] = run;
  __syncthreads();
  // Hillis-Steele inclusive scan over 512 digits
  for (int d = 1; d < BWD_NB; d <<= 1) {
    const unsigned add = bin >= d ? tot[bin - d] : 0u;
    __syncthreads();
    tot[bin] += add;
    __syncthreads();
  }
  P.binbase[(size_t)t * BWD_NB + bin] = (unsigned)P.tab_start[t] + tot[bin] - run;
}

// Stable scatter of one chunk.  Element order inside a chunk is position order; per round of 256
// positions each wave ranks its lanes by digit with ballots (match-any), waves are ordered through
// per-wave digit counts in LDS, rounds through the running per-digit count.  That gives every
// element its index among the chunk's elements of the same digit; the elements are then laid out
// digit-major in LDS and written back in THAT order, so lanes that are neighbours in a wave store to
// neighbouring addresses whenever they share a digit (one 8-byte {key, src} store per element;
// element-order stores are 2 x 4 bytes to unrelated lines, ~3x write amplification measured).
static_assert(BWD_NB == 2 * BWD_THREADS, "the local digit scan gives two digits to every thread");
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_scatter_kernel(
    const TzrTable* __restrict__ tables, int T, int pass, BwdSrcArgs A, BwdPlan P) {
  __shared__ unsigned base0[BWD_NB];  // global position of the chunk's first element of each digit
  __shared__ unsigned cnt[BWD_NB];    // running count of the digit over the rounds done so far
  __shared__ unsigned lstart[BWD_NB]; // chunk-local start of each digit (digit-major order)
  __shared__ unsigned wcnt[BWD_WAVES][BWD_NB];
  __shared__ unsigned wtot[BWD_WAVES];
  __shared__ uint2 stage[BWD_CH];
  BwdChunkDesc cd;
  if (!bwd_chunk(P, blockIdx.x, &cd)) return;
  if (cd.npass <= pass) return;
  const int t = cd.t;
  const int64_t s = cd.s, e = cd.e;
  const int width = cd.width;
  const int shift = pass * width;
  const unsigned mask = (1u << width) - 1u;
  const uint2* __restrict__ kin = (pass & 1) ? P.ks[1] : P.ks[0];
  uint2* __restrict__ kout = (pass & 1) ? P.ks[0] : P.ks[1];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const unsigned* hrow = P.hist + (size_t)blockIdx.x * BWD_NB;
  const unsigned* bb = P.binbase + (size_t)t * BWD_NB;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) {
    base0[i] = bb[i] + hrow[i];
    cnt[i] = 0;
#pragma unroll
    for (int w = 0; w < BWD_WAVES; ++w) wcnt[w][i] = 0;
  }
  // all of the chunk's elements are loaded up front (independent coalesced loads): the ranking
  // rounds below then run out of registers and pay one memory latency per workgroup
  constexpr int kRounds = BWD_CH / BWD_THREADS;
  uint32_t kreg[kRounds], sreg[kRounds], loc[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t p = s + (int64_t)r * BWD_THREADS + threadIdx.x;
    kreg[r] = 0u;
    sreg[r] = 0u;
    loc[r] = 0u;
    if (p < e) {
      if (pass == 0) {
        int64_t kk;
        bwd_elem0(P, tables[t], A, p, &kreg[r], &sreg[r], &kk);
        if (!A.uniform) {  // bag of every lookup, once (same value from every table a key feeds)
          const int64_t b = tzr_last_le(A.offsets + kk * A.B, A.B, (int64_t)sreg[r]);
          P.bag_of[sreg[r]] = (uint32_t)(kk * A.B + b);
        }
      } else {
        const uint2 v = kin[p];
        kreg[r] = v.x;
        sreg[r] = v.y;
      }
    }
  }
  __syncthreads();
  const int n = (int)(e - s);
  const int rounds = (n + BWD_THREADS - 1) / BWD_THREADS;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    if (r >= rounds) break;  // uniform across the workgroup
    const bool valid = r * BWD_THREADS + (int)threadIdx.x < n;
    const unsigned d = (kreg[r] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
    for (int bit = 0; bit < width; ++bit) {
      const int on = (d >> bit) & 1;
      const unsigned long long bm = __ballot(on);
      peers &= on ? bm : ~bm;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcnt[wv][d] = (unsigned)__popcll(peers);
    __syncthreads();
    if (valid) {
      unsigned pre = cnt[d];
      for (int w = 0; w < wv; ++w) pre += wcnt[w][d];
      loc[r] = pre + (unsigned)rank;  // index among the chunk's elements with digit d
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) {
      unsigned tsum = 0;
#pragma unroll
      for (int w = 0; w < BWD_WAVES; ++w) {
        tsum += wcnt[w][i];
        wcnt[w][i] = 0;
      }
      cnt[i] += tsum;
    }
    __syncthreads();
  }
  // lstart = exclusive scan of the digit totals: thread i owns digits 2i, 2i+1
  {
    const unsigned c0 = cnt[2 * threadIdx.x], c1 = cnt[2 * threadIdx.x + 1];
    unsigned v = c0 + c1;
    for (int dd = 1; dd < TZR_WAVE; dd <<= 1) {
      const unsigned o = __shfl_up(v, dd, TZR_WAVE);
      if (lane >= dd) v += o;
    }
    if (lane == TZR_WAVE - 1) wtot[wv] = v;
    __syncthreads();
    unsigned pre = 0;
    for (int w = 0; w < wv; ++w) pre += wtot[w];
    const unsigned excl = pre + v - (c0 + c1);
    lstart[2 * threadIdx.x] = excl;
    lstart[2 * threadIdx.x + 1] = excl + c0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    if (r * BWD_THREADS + (int)threadIdx.x < n) {
      const unsigned d = (kreg[r] >> shift) & mask;
      stage[lstart[d] + loc[r]] = make_uint2(kreg[r], sreg[r]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += BWD_THREADS) {
    const uint2 v = stage[i];
    const unsigned d = (v.x >> shift) & mask;
    kout[base0[d] + ((unsigned)i - lstart[d])] = v;
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" int tzr_pooled_bwd_plan(const TzrTable* d_tables, int n_tables,
                                   const TzrFeature* d_feats, int n_feats, int n_keys,
                                   int64_t max_rows, int max_dim, const int64_t* d_values, const int64_t* d_offsets,
                                   int64_t n_values, int64_t n_positions, int64_t B,
                                   int uniform_bag_len, void* ws,
                                   size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B < 0 ||
      max_dim <= 0 || max_rows < 0)
    return TZR_ERR_INVALID;
  const bool uniform = uniform_bag_len == 1;
  if (!uniform && !d_offsets) return TZR_ERR_INVALID;
  if (n_keys <= 0) return TZR_ERR_INVALID;
  if (n_values >= (1LL << 32) || (int64_t)n_keys * B >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  BwdPlan P;
  if (n_positions < 0 || n_positions >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  if (bwd_layout(&P, ws, n_values, n_positions, n_feats, n_tables, max_dim) > ws_bytes)
    return TZR_ERR_WORKSPACE;
  if (n_values == 0 || B == 0) return TZR_OK;
  if (!d_values) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int bits = max_rows <= 1 ? 0 : 64 - __builtin_clzll((unsigned long long)(max_rows - 1));
  const int max_pass = bits == 0 ? 1 : (bits + BWD_RB - 1) / BWD_RB;
  const unsigned chunks = (unsigned)P.max_chunks;
  hipLaunchKernelGGL(tzr_bwd_prep_kernel, dim3(1), dim3(BWD_THREADS), 0, s, d_tables, n_tables,
                     d_feats, n_feats, d_offsets, B, (int)uniform, P);
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = d_offsets;
  A.B = B;
  A.uniform = (int)uniform;
  for (int pass = 0; pass < max_pass; ++pass) {
    hipLaunchKernelGGL(tzr_bwd_hist_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, pass, A, P);
    hipLaunchKernelGGL(tzr_bwd_scan_kernel, dim3(n_tables), dim3(BWD_NB), 0, s, n_tables, pass, P);
    hipLaunchKernelGGL(tzr_bwd_scatter_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, pass, A, P);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
