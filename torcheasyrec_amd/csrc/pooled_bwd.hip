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
// Apply (K7).  Sorted runs of equal rows are reduced by the D/4-lane group that owns the run head
// (wave-level ownership, no two workgroups ever touch the same row => no atomics and no
// cross-XCD L2 coherence hazard on weights), then ONE read-modify-write of the row's weights and
// optimizer state.  Runs longer than BWD_LONG (tiny tables: 65536 lookups into 3 rows) are cut
// into fixed pieces reduced by whole workgroups with a fixed tree, then finished by one wave.
#include "tzr_common.h"

#define BWD_THREADS 256
#define BWD_CH 2048      // sorted positions per chunk (= per workgroup in hist/scatter/apply)
#define BWD_RB 9         // max radix digit width
#define BWD_NB 512       // bins per chunk histogram row (1 << BWD_RB)
#define BWD_LONG 32      // runs longer than this take the piece path
#define BWD_PIECE 2048   // elements per long-run piece
#define BWD_MAXLG 64     // lanes per row group: dim <= 256

struct BwdHdr {  // device-side counters, zeroed by tzr_pooled_bwd_apply
  unsigned n_long;
  unsigned n_pieces;
  unsigned pad[62];
};

struct BwdPlan {  // pointers into the caller workspace
  BwdHdr* hdr;
  int64_t* feat_start;     // [F+1] start of each key (by order) in table-major position space
  int32_t* feat_by_order;  // [F]
  int64_t* tab_start;      // [T+1]
  int32_t* tab_chunk;      // [T+1] first chunk of each table
  int32_t* tab_width;      // [T] digit width (0 = nothing to sort)
  int32_t* tab_npass;      // [T]
  uint32_t* key[2];        // [N] local row id
  uint32_t* src[2];        // [N] original lookup position
  uint32_t* bag_of;        // [N] bag index f*B+b of every lookup (only when bags are jagged)
  uint32_t* hist;          // [max_chunks * BWD_NB] chunk-exclusive digit counts
  uint32_t* binbase;       // [T * BWD_NB] global start of every (table, digit)
  uint32_t* long_pos;      // [max_long]
  uint32_t* long_len;
  int32_t* long_tab;
  uint32_t* long_pbase;
  uint32_t* piece_run;     // [max_pieces]
  uint32_t* piece_idx;
  float* partial;          // [max_pieces * max_dim]
  int64_t max_chunks, max_long, max_pieces;
};

static int64_t bwd_max_chunks(int64_t N, int T) { return N / BWD_CH + T + 1; }

static size_t bwd_layout(BwdPlan* p, void* ws, int64_t NV, int64_t N, int F, int T, int max_dim) {
  // NV = ids in the KJT values array; N = capacity of the table-major position space (sum over
  // lookups of their key length: a key read through two tables is sorted twice).  N >= 1.
  TzrCarver c(ws);
  BwdPlan q;
  q.max_chunks = bwd_max_chunks(N, T);
  q.max_long = N / (BWD_LONG + 1) + 1;
  q.max_pieces = N / BWD_PIECE + q.max_long + 1;
  q.hdr = c.take<BwdHdr>(1);
  q.feat_start = c.take<int64_t>(F + 1);
  q.feat_by_order = c.take<int32_t>(F);
  q.tab_start = c.take<int64_t>(T + 1);
  q.tab_chunk = c.take<int32_t>(T + 1);
  q.tab_width = c.take<int32_t>(T);
  q.tab_npass = c.take<int32_t>(T);
  for (int i = 0; i < 2; ++i) {
    q.key[i] = c.take<uint32_t>(N);
    q.src[i] = c.take<uint32_t>(N);
  }
  q.bag_of = c.take<uint32_t>(NV);
  q.hist = c.take<uint32_t>(q.max_chunks * BWD_NB);
  q.binbase = c.take<uint32_t>((size_t)T * BWD_NB);
  q.long_pos = c.take<uint32_t>(q.max_long);
  q.long_len = c.take<uint32_t>(q.max_long);
  q.long_tab = c.take<int32_t>(q.max_long);
  q.long_pbase = c.take<uint32_t>(q.max_long);
  q.piece_run = c.take<uint32_t>(q.max_pieces);
  q.piece_idx = c.take<uint32_t>(q.max_pieces);
  q.partial = c.take<float>((size_t)q.max_pieces * max_dim);
  if (p) *p = q;
  return c.off;
}

extern "C" size_t tzr_pooled_bwd_workspace(int64_t n_values, int64_t n_positions, int n_feats,
                                           int n_tables, int64_t B,
                                           int max_dim) {
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
    const int npass = (bits + BWD_RB - 1) / BWD_RB;
    P.tab_npass[t] = npass;
    P.tab_width[t] = npass ? (bits + npass - 1) / npass : 0;
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

// chunk id -> (table, first position, end position); returns false for surplus workgroups.
__device__ __forceinline__ bool bwd_chunk(const BwdPlan& P, const TzrTable* tables, int T,
                                          int chunk, int* t_out, int64_t* s_out, int64_t* e_out,
                                          int64_t* tab_s_out, int64_t* tab_e_out) {
  if (chunk >= P.tab_chunk[T]) return false;
  int lo = 0, hi = T;  // last t with tab_chunk[t] <= chunk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (P.tab_chunk[mid] <= chunk) lo = mid; else hi = mid;
  }
  // skip empty tables that share the same chunk start
  int t = lo;
  const TzrTable tb = tables[t];
  const int64_t ts = P.tab_start[t];
  const int64_t te = tb.n_feats > 0 ? P.feat_start[tb.first_order + tb.n_feats] : ts;
  const int64_t s = ts + (int64_t)(chunk - P.tab_chunk[t]) * BWD_CH;
  *t_out = t;
  *s_out = s;
  *e_out = min(te, s + BWD_CH);
  *tab_s_out = ts;
  *tab_e_out = te;
  return true;
}

// Regroup lookups table-major: key[0][p] = local row, src[0][p] = original lookup position.
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_gather_kernel(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets, int64_t B,
    int uniform, BwdPlan P) {
  int t;
  int64_t s, e, ts, te;
  if (!bwd_chunk(P, tables, T, blockIdx.x, &t, &s, &e, &ts, &te)) return;
  const TzrTable tb = tables[t];
  for (int64_t p = s + threadIdx.x; p < e; p += BWD_THREADS) {
    int o = tb.first_order;
    while (o + 1 < tb.first_order + tb.n_feats && P.feat_start[o + 1] <= p) ++o;
    const int64_t key = feats[P.feat_by_order[o]].key;
    const int64_t fbase = uniform ? key * B : offsets[key * B];
    const int64_t i = fbase + (p - P.feat_start[o]);
    int64_t id = values[i];
    if ((uint64_t)id >= (uint64_t)tb.rows) id = 0;  // memory safety; K4 reports/clamps
    P.key[0][p] = (uint32_t)id;
    P.src[0][p] = (uint32_t)i;
    if (!uniform) {
      const int64_t b = tzr_last_le(offsets + key * B, B, i);
      P.bag_of[i] = (uint32_t)(key * B + b);  // same value from every table this key feeds
    }
  }
}

__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_hist_kernel(
    const TzrTable* __restrict__ tables, int T, int pass, BwdPlan P) {
  __shared__ unsigned h[BWD_NB];
  int t;
  int64_t s, e, ts, te;
  if (!bwd_chunk(P, tables, T, blockIdx.x, &t, &s, &e, &ts, &te)) return;
  if (P.tab_npass[t] <= pass) return;
  const int width = P.tab_width[t];
  const int shift = pass * width;
  const unsigned mask = (1u << width) - 1u;
  const uint32_t* __restrict__ kin = P.key[pass & 1];
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) h[i] = 0;
  __syncthreads();
  for (int64_t p = s + threadIdx.x; p < e; p += BWD_THREADS)
    atomicAdd(&h[(kin[p] >> shift) & mask], 1u);
  __syncthreads();
  uint32_t* out = P.hist + (size_t)blockIdx.x * BWD_NB;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) out[i] = h[i];
}

// One workgroup per table, one thread per digit: exclusive scan over the table's chunks (in
// place) and over digits -> binbase.  Chunk columns are read in batches of independent loads.
#define BWD_SCAN_BATCH 16
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
// per-wave digit counts in LDS, rounds through the running per-digit base.
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_scatter_kernel(
    const TzrTable* __restrict__ tables, int T, int pass, BwdPlan P) {
  __shared__ unsigned base[BWD_NB];
  __shared__ unsigned wcnt[BWD_THREADS / TZR_WAVE][BWD_NB];
  int t;
  int64_t s, e, ts, te;
  if (!bwd_chunk(P, tables, T, blockIdx.x, &t, &s, &e, &ts, &te)) return;
  if (P.tab_npass[t] <= pass) return;
  const int width = P.tab_width[t];
  const int shift = pass * width;
  const unsigned mask = (1u << width) - 1u;
  const uint32_t* __restrict__ kin = P.key[pass & 1];
  const uint32_t* __restrict__ sin = P.src[pass & 1];
  uint32_t* __restrict__ kout = P.key[(pass + 1) & 1];
  uint32_t* __restrict__ sout = P.src[(pass + 1) & 1];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const unsigned* hrow = P.hist + (size_t)blockIdx.x * BWD_NB;
  const unsigned* bb = P.binbase + (size_t)t * BWD_NB;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) {
    base[i] = bb[i] + hrow[i];
#pragma unroll
    for (int w = 0; w < BWD_THREADS / TZR_WAVE; ++w) wcnt[w][i] = 0;
  }
  __syncthreads();
  const int rounds = (int)((e - s + BWD_THREADS - 1) / BWD_THREADS);
  for (int r = 0; r < rounds; ++r) {
    const int64_t p = s + (int64_t)r * BWD_THREADS + threadIdx.x;
    const bool valid = p < e;
    const uint32_t k = valid ? kin[p] : 0u;
    const uint32_t sv = valid ? sin[p] : 0u;
    const unsigned d = (k >> shift) & mask;
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
      unsigned pre = 0;
      for (int w = 0; w < wv; ++w) pre += wcnt[w][d];
      const unsigned pos = base[d] + pre + (unsigned)rank;
      kout[pos] = k;
      sout[pos] = sv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) {
      unsigned tsum = 0;
#pragma unroll
      for (int w = 0; w < BWD_THREADS / TZR_WAVE; ++w) {
        tsum += wcnt[w][i];
        wcnt[w][i] = 0;
      }
      base[i] += tsum;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// apply kernels
// ------------------------------------------------------------------------------------------

struct BwdGrads {
  TzrDst d[TZR_MAX_DST];
};

struct BwdOpt {
  int kind, wd_mode, clip;
  const float* lr;
  float eps, wd, max_grad;
};

// dL/d(row contribution) of lookup position i, float4 chunk c of its row.
__device__ __forceinline__ float4 bwd_lookup_grad(const TzrFeature* __restrict__ feats,
                                                  const TzrTable& tb,
                                                  const int32_t* __restrict__ feat_by_order,
                                                  const BwdGrads& G,
                                                  const int64_t* __restrict__ offsets,
                                                  const float* __restrict__ weights,
                                                  const uint32_t* __restrict__ bag_of, int64_t B,
                                                  int uniform, uint32_t i, int c) {
  const uint32_t bag = uniform ? i : bag_of[i];
  const uint32_t key = bag / (uint32_t)B;
  const int64_t b = bag - key * (uint32_t)B;
  // the lookup of this table that reads `key` (a table is read at most once per key)
  int o = tb.first_order;
  if (tb.n_feats > 1)
    while (o + 1 < tb.first_order + tb.n_feats && feats[feat_by_order[o]].key != (int32_t)key) ++o;
  const TzrFeature ft = feats[feat_by_order[o]];
  float4 g = tzr_zero4();
  for (int d = 0; d < ft.n_dst; ++d) {
    const float* gp = reinterpret_cast<const float*>(G.d[ft.dst[d]].ptr) +
                      b * G.d[ft.dst[d]].stride + ft.col[d] + 4 * c;
    g = tzr_add4(g, tzr_ld4(gp));
  }
  float sc = weights ? weights[i] : 1.0f;
  if (!uniform && ft.pooling == TZR_POOL_MEAN) {
    const int64_t len = offsets[(int64_t)bag + 1] - offsets[bag];
    if (len > 1) sc = sc / (float)len;
  }
  if (weights || (!uniform && ft.pooling == TZR_POOL_MEAN)) {
    g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc;
  }
  return g;
}

// Sum of v over the `lg` lanes of a row group (all 64 lanes call it).
__device__ __forceinline__ float bwd_group_sum(float v, int lg, int lane_in_group, int lane) {
  if ((lg & (lg - 1)) == 0) {
    for (int m = lg >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  }
  float s = 0.f;
  const int g0 = lane - lane_in_group;
  for (int l = 0; l < lg; ++l) s += __shfl(v, g0 + l, 64);
  return s;
}

// ONE update of row `row`, chunk c; `active` lanes hold the summed gradient g.  All 64 lanes of
// the wave must call (row-wise adagrad reduces across the group).
__device__ __forceinline__ void bwd_apply_row(const TzrTable& tb, const BwdOpt& opt, float lr,
                                              int64_t row, int c, float4 g, float4 w4, bool active,
                                              int lg, int lane_in_group, int lane) {
  if (opt.clip) {
    g.x = fminf(fmaxf(g.x, -opt.max_grad), opt.max_grad);
    g.y = fminf(fmaxf(g.y, -opt.max_grad), opt.max_grad);
    g.z = fminf(fmaxf(g.z, -opt.max_grad), opt.max_grad);
    g.w = fminf(fmaxf(g.w, -opt.max_grad), opt.max_grad);
  }
  float* wp = reinterpret_cast<float*>(tb.w) + row * (int64_t)tb.w_stride + 4 * c;
  if (opt.kind == TZR_OPT_ADAGRAD) {
    if (active) {
      float* mp = reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c;
      float4 m4 = tzr_ld4(mp);
      m4.x += g.x * g.x; m4.y += g.y * g.y; m4.z += g.z * g.z; m4.w += g.w * g.w;
      tzr_st4(mp, m4);
      w4.x -= lr * g.x / (sqrtf(m4.x) + opt.eps);
      w4.y -= lr * g.y / (sqrtf(m4.y) + opt.eps);
      w4.z -= lr * g.z / (sqrtf(m4.z) + opt.eps);
      w4.w -= lr * g.w / (sqrtf(m4.w) + opt.eps);
      tzr_st4(wp, w4);
    }
  } else if (opt.kind == TZR_OPT_ROWWISE_ADAGRAD) {
    float4 gl = g;
    if (opt.wd_mode == TZR_WD_L2) gl = tzr_fma4(opt.wd, w4, g);
    float ss = active ? (gl.x * gl.x + gl.y * gl.y + gl.z * gl.z + gl.w * gl.w) : 0.f;
    ss = bwd_group_sum(ss, lg, lane_in_group, lane);
    // the row's scalar state is read by the group's first lane only and broadcast, so no lane
    // can observe the store below
    float* mp = reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride;
    float mold = (active && lane_in_group == 0) ? *mp : 0.f;
    mold = __shfl(mold, lane - lane_in_group, 64);
    if (active) {
      const float mnew = mold + ss / (float)tb.dim;
      const float mult = lr / (sqrtf(mnew) + opt.eps);
      float corr = 1.0f;
      if (opt.wd_mode == TZR_WD_L2) corr = 1.0f - mult * opt.wd;
      else if (opt.wd_mode == TZR_WD_DECOUPLE) corr = 1.0f - lr * opt.wd;
      w4.x = corr * w4.x - mult * g.x;
      w4.y = corr * w4.y - mult * g.y;
      w4.z = corr * w4.z - mult * g.z;
      w4.w = corr * w4.w - mult * g.w;
      tzr_st4(wp, w4);
      if (lane_in_group == 0) *mp = mnew;
    }
  } else {  // SGD
    if (active) {
      w4.x -= lr * g.x; w4.y -= lr * g.y; w4.z -= lr * g.z; w4.w -= lr * g.w;
      tzr_st4(wp, w4);
    }
  }
}

// Runs of <= BWD_LONG equal rows: the group sitting on the run head sums the duplicates in sorted
// (= original lookup) order and updates the row.  Longer runs are queued for the piece path.
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_update_short_kernel(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    BwdGrads G, BwdOpt opt, BwdPlan P) {
  int t;
  int64_t s, e, ts, te;
  if (!bwd_chunk(P, tables, T, blockIdx.x, &t, &s, &e, &ts, &te)) return;
  const TzrTable tb = tables[t];
  const int par = P.tab_npass[t] & 1;
  const uint32_t* __restrict__ K = P.key[par];
  const uint32_t* __restrict__ S = P.src[par];
  const int lg = tb.dim >> 2;                 // lanes per row
  const int gw = TZR_WAVE / lg;               // row groups per wave
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int gi = lane / lg;                   // group in wave
  const int c = lane - gi * lg;               // float4 chunk of the row
  const bool lane_on = gi < gw;
  const int gpb = gw * (BWD_THREADS / TZR_WAVE);  // groups per workgroup
  const float lr = *opt.lr;
  const int iters = (int)((e - s + gpb - 1) / gpb);
  for (int it = 0; it < iters; ++it) {
    const int64_t p = s + (int64_t)it * gpb + wv * gw + gi;
    bool head = false;
    uint32_t key = 0;
    int64_t len = 0;
    if (lane_on && p < e) {
      key = K[p];
      head = (p == ts) || (K[p - 1] != key);
      if (head) {
        len = 1;
        while (len <= BWD_LONG && p + len < te && K[p + len] == key) ++len;
      }
    }
    const bool is_short = head && len <= BWD_LONG;
    float4 g = tzr_zero4();
    float4 w4 = tzr_zero4();
    if (is_short) {
      // row load first: it depends only on the key and overlaps the gradient gathers
      w4 = tzr_ld4(reinterpret_cast<const float*>(tb.w) + (int64_t)key * tb.w_stride + 4 * c);
      for (int64_t j = 0; j < len; ++j)
        g = tzr_add4(g, bwd_lookup_grad(feats, tb, P.feat_by_order, G, offsets, weights,
                                        P.bag_of, B, uniform, S[p + j], c));
    } else if (head && c == 0) {
      // long run: full length by binary search for the first position with a larger key
      int64_t lo = p + BWD_LONG, hi = te;  // K[lo] == key, K[hi] > key (or hi == te)
      while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (K[mid] == key) lo = mid; else hi = mid;
      }
      const uint32_t full = (uint32_t)(hi - p);
      const uint32_t np = (full + BWD_PIECE - 1) / BWD_PIECE;
      const uint32_t r = atomicAdd(&P.hdr->n_long, 1u);
      const uint32_t pb = atomicAdd(&P.hdr->n_pieces, np);
      P.long_pos[r] = (uint32_t)p;
      P.long_len[r] = full;
      P.long_tab[r] = t;
      P.long_pbase[r] = pb;
      for (uint32_t j = 0; j < np; ++j) {
        P.piece_run[pb + j] = r;
        P.piece_idx[pb + j] = j;
      }
    }
    bwd_apply_row(tb, opt, lr, (int64_t)key, c, g, w4, is_short, lg, c, lane);
  }
}

// One workgroup per long-run piece: group g sums elements g, g+G, ... of the piece, group 0 then
// adds the G partials in group order (fixed tree => deterministic) -> partial[piece][dim].
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_long_partial_kernel(
    const TzrTable* __restrict__ tables, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    BwdGrads G, int max_dim, BwdPlan P) {
  __shared__ float4 red[BWD_THREADS];
  const unsigned n_pieces = P.hdr->n_pieces;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  for (unsigned piece = blockIdx.x; piece < n_pieces; piece += gridDim.x) {
    const unsigned r = P.piece_run[piece];
    const unsigned pi = P.piece_idx[piece];
    const int t = P.long_tab[r];
    const TzrTable tb = tables[t];
    const uint32_t* __restrict__ S = P.src[P.tab_npass[t] & 1];
    const int lg = tb.dim >> 2;
    const int gw = TZR_WAVE / lg;
    const int gi = lane / lg;
    const int c = lane - gi * lg;
    const int gpb = gw * (BWD_THREADS / TZR_WAVE);
    const int g_id = wv * gw + gi;
    const int64_t s = (int64_t)P.long_pos[r] + (int64_t)pi * BWD_PIECE;
    const int64_t e = min((int64_t)P.long_pos[r] + P.long_len[r], s + BWD_PIECE);
    float4 acc = tzr_zero4();
    if (gi < gw) {
      for (int64_t p = s + g_id; p < e; p += gpb)
        acc = tzr_add4(acc, bwd_lookup_grad(feats, tb, P.feat_by_order, G, offsets, weights,
                                            P.bag_of, B, uniform, S[p], c));
      red[g_id * lg + c] = acc;
    }
    __syncthreads();
    if (threadIdx.x < lg) {
      float4 tot = red[threadIdx.x];
      for (int g2 = 1; g2 < gpb; ++g2) tot = tzr_add4(tot, red[g2 * lg + threadIdx.x]);
      tzr_st4(P.partial + (size_t)piece * max_dim + 4 * threadIdx.x, tot);
    }
    __syncthreads();
  }
}

// One wave per long run: add its pieces in piece order, update the row once.
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_long_final_kernel(
    const TzrTable* __restrict__ tables, BwdOpt opt, int max_dim, BwdPlan P) {
  const unsigned n_long = P.hdr->n_long;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const float lr = *opt.lr;
  const unsigned stride = gridDim.x * (BWD_THREADS / TZR_WAVE);
  // uniform trip count per wave: every lane of a wave sees the same r
  for (unsigned r = blockIdx.x * (BWD_THREADS / TZR_WAVE) + wv; r < n_long; r += stride) {
    const int t = P.long_tab[r];
    const TzrTable tb = tables[t];
    const int lg = tb.dim >> 2;
    const bool on = lane < lg;
    const uint32_t key = P.key[P.tab_npass[t] & 1][P.long_pos[r]];
    const unsigned np = (P.long_len[r] + BWD_PIECE - 1) / BWD_PIECE;
    const unsigned pb = P.long_pbase[r];
    float4 g = tzr_zero4();
    float4 w4 = tzr_zero4();
    if (on) {
      w4 = tzr_ld4(reinterpret_cast<const float*>(tb.w) + (int64_t)key * tb.w_stride + 4 * lane);
      for (unsigned j = 0; j < np; ++j)
        g = tzr_add4(g, tzr_ld4(P.partial + (size_t)(pb + j) * max_dim + 4 * lane));
    }
    // group = lanes [0, lg) of the wave; lanes beyond contribute zeros to the row-wise reduction
    bwd_apply_row(tb, opt, lr, (int64_t)key, lane, g, w4, on, TZR_WAVE, lane, lane);
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
  const int max_pass = (bits + BWD_RB - 1) / BWD_RB;
  const unsigned chunks = (unsigned)P.max_chunks;
  hipLaunchKernelGGL(tzr_bwd_prep_kernel, dim3(1), dim3(BWD_THREADS), 0, s, d_tables, n_tables,
                     d_feats, n_feats, d_offsets, B, (int)uniform, P);
  hipLaunchKernelGGL(tzr_bwd_gather_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                     n_tables, d_feats, d_values, d_offsets, B, (int)uniform, P);
  for (int pass = 0; pass < max_pass; ++pass) {
    hipLaunchKernelGGL(tzr_bwd_hist_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, pass, P);
    hipLaunchKernelGGL(tzr_bwd_scan_kernel, dim3(n_tables), dim3(BWD_NB), 0, s, n_tables, pass, P);
    hipLaunchKernelGGL(tzr_bwd_scatter_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, pass, P);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_pooled_bwd_apply(const TzrTable* d_tables, const TzrFeature* d_feats,
                                    int n_feats, int n_tables, int max_dim,
                                    const int64_t* d_offsets, const float* d_weights,
                                    int64_t n_values, int64_t n_positions, int64_t B,
                                    int uniform_bag_len,
                                    const TzrDst* h_grads, int n_dst,
                                    const TzrSparseOptim* h_optim, void* ws, size_t ws_bytes,
                                    void* stream) {
  if (!d_tables || !d_feats || !h_grads || !h_optim || n_tables <= 0 || n_feats <= 0 ||
      n_values < 0 || B < 0 || n_dst <= 0 || n_dst > TZR_MAX_DST || max_dim <= 0 ||
      max_dim > 4 * BWD_MAXLG || (max_dim & 3))
    return TZR_ERR_INVALID;
  const bool uniform = uniform_bag_len == 1;
  if (!uniform && !d_offsets) return TZR_ERR_INVALID;
  if (!h_optim->d_lr) return TZR_ERR_INVALID;
  if (h_optim->kind != TZR_OPT_SGD && h_optim->kind != TZR_OPT_ADAGRAD &&
      h_optim->kind != TZR_OPT_ROWWISE_ADAGRAD)
    return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  BwdPlan P;
  if (n_positions < 0 || n_positions >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  if (bwd_layout(&P, ws, n_values, n_positions, n_feats, n_tables, max_dim) > ws_bytes)
    return TZR_ERR_WORKSPACE;
  if (n_values == 0 || B == 0) return TZR_OK;
  BwdGrads G;
  for (int i = 0; i < TZR_MAX_DST; ++i) {
    G.d[i].ptr = 0;
    G.d[i].stride = 0;
  }
  for (int i = 0; i < n_dst; ++i) {
    if (!h_grads[i].ptr || (h_grads[i].stride & 3) || (h_grads[i].ptr & 15)) return TZR_ERR_INVALID;
    G.d[i] = h_grads[i];
  }
  BwdOpt opt;
  opt.kind = h_optim->kind;
  opt.wd_mode = h_optim->weight_decay_mode;
  opt.clip = h_optim->gradient_clipping;
  opt.lr = reinterpret_cast<const float*>(h_optim->d_lr);
  opt.eps = h_optim->eps;
  opt.wd = h_optim->weight_decay;
  opt.max_grad = h_optim->max_gradient;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(P.hdr, 0, sizeof(BwdHdr), s) != hipSuccess) return TZR_ERR_LAUNCH;
  const unsigned chunks = (unsigned)P.max_chunks;
  hipLaunchKernelGGL(tzr_bwd_update_short_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                     n_tables, d_feats, d_offsets, d_weights, B, (int)uniform, G, opt, P);
  const unsigned pgrid = (unsigned)std::min<int64_t>(1024, P.max_pieces);
  hipLaunchKernelGGL(tzr_bwd_long_partial_kernel, dim3(pgrid), dim3(BWD_THREADS), 0, s, d_tables,
                     d_feats, d_offsets, d_weights, B, (int)uniform, G, max_dim, P);
  hipLaunchKernelGGL(tzr_bwd_long_final_kernel, dim3(64), dim3(BWD_THREADS), 0, s, d_tables, opt,
                     max_dim, P);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
