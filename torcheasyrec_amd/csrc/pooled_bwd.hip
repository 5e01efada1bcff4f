// K6: backward index plan for gfx950.
//
// Replaces fbgemm transpose_embedding_input (linearize + cub radix sort + run-length), reached from
// autograd of self.ebc(kjt) (/root/reference/tzrec/modules/embedding.py:930) with the optimizer
// fused by apply_optimizer_in_backward (/root/reference/tzrec/main.py:774-781).
//
// The apply (K7, pooled_bwd_apply.hip) wants, per table, the lookups ordered by (row, original
// lookup position): every summation order is then a function of the ids alone => bit-reproducible
// updates, no float atomics anywhere.  Round 1 got there with a 3-pass LSD radix sort = 10 dependent
// launches of latency-bound kernels (95 us at B = 65536 for ~20 us of traffic).  Now:
//
//   hist     one launch: every workgroup derives the table-major geometry itself (no prep launch),
//            counts its chunk's lookups per BUCKET (<= 512 per table, ~rows/512 consecutive row ids
//            each, evenly filled by uniform ids -- bwd_bucket_params);
//   scan     per table: bucket starts, the unit grid of the apply, the list of heavy buckets;
//   scatter  ONE stable partition pass into buckets (lookups regrouped table-major on the way);
//   heavy    buckets with more than BWD_TH lookups (hot rows: Zipf heads, default ids) are sorted
//            here by one workgroup each, tile by tile; nothing to do for uniform ids;
//   (apply)  the reduce kernel orders the light buckets of its unit in LDS (<= 1280 lookups, 2-3
//            counting passes on the row-id bits left inside a bucket) before it reduces them.
//
// 4 launches (3 + one that usually finds an empty list) instead of 10, ~1/3 of the ks traffic.
#include "pooled_bwd.h"

extern "C" size_t tzr_pooled_bwd_workspace(int64_t n_values, int64_t n_positions, int n_feats,
                                           int n_tables, int64_t B, int max_dim) {
  (void)B;
  if (n_values < 0 || n_positions < 0 || n_feats <= 0 || n_tables <= 0 || max_dim <= 0) return 0;
  return bwd_layout(nullptr, nullptr, n_values, n_positions, n_feats, n_tables, max_dim) + 256;
}

// ------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------

struct BwdSrcArgs {
  const TzrFeature* feats;
  const int64_t* values;
  const int64_t* offsets;
  int64_t B;
  int uniform;
};

struct BwdGeo {  // table-major geometry, in LDS (fused) or in the workspace
  const uint32_t* fstart;  // [F+1]
  const int32_t* fkey;     // [F]
  const int32_t* tchunk;   // [T+1]
};

// In-place exclusive scan of a[0..n) by the whole workgroup; a[n] = total.
__device__ __forceinline__ void bwd_block_scan(uint32_t* a, int n, uint32_t* wtot) {
  const int tid = threadIdx.x;
  const int lane = tid & (TZR_WAVE - 1);
  const int wv = tid / TZR_WAVE;
  uint32_t carry = 0;
  for (int base = 0; base < n; base += BWD_THREADS) {
    const int i = base + tid;
    const uint32_t v = i < n ? a[i] : 0u;
    uint32_t incl = v;
    for (int dd = 1; dd < TZR_WAVE; dd <<= 1) {
      const uint32_t o = __shfl_up(incl, dd, TZR_WAVE);
      if (lane >= dd) incl += o;
    }
    if (lane == TZR_WAVE - 1) wtot[wv] = incl;
    __syncthreads();
    uint32_t pre = carry, tot = 0;
#pragma unroll
    for (int w = 0; w < BWD_WAVES; ++w) {
      if (w < wv) pre += wtot[w];
      tot += wtot[w];
    }
    if (i < n) a[i] = pre + incl - v;
    carry += tot;
    __syncthreads();
  }
  if (tid == 0) a[n] = carry;
  __syncthreads();
}

struct BwdGeoLds {
  uint32_t fstart[BWD_GEO + 1];
  int32_t fkey[BWD_GEO];
  uint32_t tchunk[BWD_GEO + 1];
  uint32_t wtot[BWD_WAVES];
};

// Table-major segment starts and the chunk map, derived by every hist workgroup on its own
// (F + T small loads and two block scans) so that the plan needs no single-workgroup launch ahead
// of it.  Keys of the KJT this module does not own (table < 0) are ordered last and contribute
// nothing.
__device__ __forceinline__ void bwd_geometry(const TzrTable* __restrict__ tables, int T,
                                             const BwdSrcArgs& A, int F, BwdGeoLds& G) {
  for (int f = threadIdx.x; f < F; f += BWD_THREADS) {
    const TzrFeature ft = A.feats[f];
    const int64_t key = ft.key;
    const int64_t n =
        ft.table < 0 ? 0 : (A.uniform ? A.B : A.offsets[(key + 1) * A.B] - A.offsets[key * A.B]);
    G.fstart[ft.order] = (uint32_t)n;
    G.fkey[ft.order] = ft.key;
  }
  __syncthreads();
  bwd_block_scan(G.fstart, F, G.wtot);
  for (int t = threadIdx.x; t < T; t += BWD_THREADS) {
    const TzrTable tb = tables[t];
    const uint32_t s = tb.n_feats > 0 ? G.fstart[tb.first_order] : 0u;
    const uint32_t e = tb.n_feats > 0 ? G.fstart[tb.first_order + tb.n_feats] : 0u;
    G.tchunk[t] = (e - s + BWD_CH - 1) / BWD_CH;
  }
  __syncthreads();
  // tables are visited in first_order order == table-major position order only if table ids
  // follow it; starts are absolute, so the chunk map just needs a prefix in table-id order
  bwd_block_scan(G.tchunk, T, G.wtot);
}

// Fallback for more than BWD_GEO lookups or tables: one workgroup writes the geometry to the
// workspace (serial prefixes), the hist workgroups read it from there.
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_prep_kernel(
    const TzrTable* __restrict__ tables, int T, BwdSrcArgs A, int F, BwdPlan P) {
  for (int f = threadIdx.x; f < F; f += BWD_THREADS) {
    const TzrFeature ft = A.feats[f];
    const int64_t key = ft.key;
    const int64_t n =
        ft.table < 0 ? 0 : (A.uniform ? A.B : A.offsets[(key + 1) * A.B] - A.offsets[key * A.B]);
    P.feat_start[ft.order] = (uint32_t)n;
    P.feat_key[ft.order] = ft.key;
    P.feat_by_order[ft.order] = f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int o = 0; o < F; ++o) {
      const uint32_t n = P.feat_start[o];
      P.feat_start[o] = run;
      run += n;
    }
    P.feat_start[F] = run;
    P.hcount[0] = 0;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += BWD_THREADS) {
    const TzrTable tb = tables[t];
    const uint32_t s = tb.n_feats > 0 ? P.feat_start[tb.first_order] : 0u;
    const uint32_t e = tb.n_feats > 0 ? P.feat_start[tb.first_order + tb.n_feats] : 0u;
    P.tab_chunk[t] = (int32_t)((e - s + BWD_CH - 1) / BWD_CH);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int t = 0; t < T; ++t) {
      const int32_t n = P.tab_chunk[t];
      P.tab_chunk[t] = run;
      run += n;
    }
    P.tab_chunk[T] = run;
  }
}

// Table-major position p of table tb -> (local row, original lookup position): the lookups of a
// table are the concatenation, in key order, of the id segments of the keys that read it.
__device__ __forceinline__ void bwd_elem0(const BwdGeo& G, const TzrTable& tb, const BwdSrcArgs& A,
                                          int64_t p, uint32_t* key_out, uint32_t* src_out,
                                          int64_t* kjt_key_out) {
  int o = tb.first_order;
  while (o + 1 < tb.first_order + tb.n_feats && (int64_t)G.fstart[o + 1] <= p) ++o;
  const int64_t key = G.fkey[o];
  const int64_t fbase = A.uniform ? key * A.B : A.offsets[key * A.B];
  const int64_t i = fbase + (p - (int64_t)G.fstart[o]);
  int64_t id = A.values[i];
  if ((uint64_t)id >= (uint64_t)tb.rows) id = 0;  // memory safety; K4 reports/clamps
  *key_out = (uint32_t)id;
  *src_out = (uint32_t)i;
  *kjt_key_out = key;
}

// ------------------------------------------------------------------------------------------
// hist: chunk descriptors + bucket counts of every chunk
// ------------------------------------------------------------------------------------------
template <bool FUSED>
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_hist_kernel(
    const TzrTable* __restrict__ tables, int T, int F, BwdSrcArgs A, BwdPlan P) {
  __shared__ unsigned h[BWD_NB];
  __shared__ BwdGeoLds GL;
  BwdGeo G;
  if (FUSED) {
    bwd_geometry(tables, T, A, F, GL);
    G.fstart = GL.fstart;
    G.fkey = GL.fkey;
    G.tchunk = reinterpret_cast<const int32_t*>(GL.tchunk);
    if (blockIdx.x == 0) {  // the later kernels of the plan and the apply read it from the workspace
      for (int o = threadIdx.x; o <= F; o += BWD_THREADS) P.feat_start[o] = GL.fstart[o];
      for (int o = threadIdx.x; o < F; o += BWD_THREADS) P.feat_key[o] = GL.fkey[o];
      for (int f = threadIdx.x; f < F; f += BWD_THREADS) P.feat_by_order[A.feats[f].order] = f;
      for (int t = threadIdx.x; t <= T; t += BWD_THREADS) P.tab_chunk[t] = (int32_t)GL.tchunk[t];
      if (threadIdx.x == 0) P.hcount[0] = 0;
    }
  } else {
    G.fstart = P.feat_start;
    G.fkey = P.feat_key;
    G.tchunk = P.tab_chunk;
  }
  const int c = blockIdx.x;
  BwdChunkDesc cd;
  cd.t = -1;
  cd.nb = cd.exact = cd.last_chunk = 0;
  cd.s = cd.e = cd.ts = cd.te = 0;
  cd.mult = 0;
  TzrTable tb;
  if (c < G.tchunk[T]) {
    int lo = 0, hi = T;  // last t with tchunk[t] <= c (the non-empty table holding it)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (G.tchunk[mid] <= c) lo = mid; else hi = mid;
    }
    tb = tables[lo];
    cd.t = lo;
    bwd_bucket_params(tb.rows, &cd.nb, &cd.mult);
    cd.exact = tb.rows <= BWD_NB;
    cd.last_chunk = G.tchunk[lo + 1];
    cd.ts = tb.n_feats > 0 ? (int64_t)G.fstart[tb.first_order] : 0;
    cd.te = tb.n_feats > 0 ? (int64_t)G.fstart[tb.first_order + tb.n_feats] : cd.ts;
    cd.s = cd.ts + (int64_t)(c - G.tchunk[lo]) * BWD_CH;
    cd.e = min(cd.te, cd.s + (int64_t)BWD_CH);
  }
  if (threadIdx.x == 0) P.cdesc[c] = cd;
  if (cd.t < 0) return;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) h[i] = 0;
  // all of the chunk's keys are loaded before any is counted: independent loads in flight, one
  // memory latency per workgroup instead of one per element
  constexpr int kRounds = BWD_CH / BWD_THREADS;
  uint32_t kreg[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t p = cd.s + (int64_t)r * BWD_THREADS + threadIdx.x;
    uint32_t sv;
    int64_t kk;
    kreg[r] = 0u;
    if (p < cd.e) bwd_elem0(G, tb, A, p, &kreg[r], &sv, &kk);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t p = cd.s + (int64_t)r * BWD_THREADS + threadIdx.x;
    if (p < cd.e) atomicAdd(&h[bwd_bucket(kreg[r], cd.mult)], 1u);
  }
  __syncthreads();
  uint32_t* out = P.hist + (size_t)blockIdx.x * BWD_NB;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) out[i] = h[i];
}

// ------------------------------------------------------------------------------------------
// scan: bucket starts, the unit grid, heavy buckets
// ------------------------------------------------------------------------------------------
// One workgroup per table, one thread per bucket: exclusive scan over the table's chunks (in
// place) and over buckets -> binbase.  Chunk columns are read in batches of independent loads.
//
// Units of the apply.  Cut points = bucket boundaries + the BWD_CH-block boundaries that fall
// INSIDE a heavy bucket; unit j of a table starts at the first cut point at or after block j
// (ts + j * BWD_CH).  So a unit is a whole number of light buckets (which the reduce kernel sorts
// in LDS) and/or block-sized slices of heavy buckets (sorted by the heavy kernel), it holds fewer
// than BWD_CH + BWD_TH lookups, and a run of one row can only cross a unit boundary inside a heavy
// bucket.
#define BWD_SCAN_BATCH 64
__global__ __launch_bounds__(BWD_NB) void tzr_bwd_scan_kernel(const TzrTable* __restrict__ tables,
                                                              int T, BwdPlan P) {
  __shared__ unsigned tot[BWD_NB];
  const int t = blockIdx.x;
  const int c0 = P.tab_chunk[t];
  const int C = P.tab_chunk[t + 1] - c0;
  if (C <= 0) return;
  const TzrTable tb = tables[t];
  const uint32_t ts = P.feat_start[tb.first_order];
  const uint32_t te = P.feat_start[tb.first_order + tb.n_feats];
  const bool exact = tb.rows <= BWD_NB;
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
  // Hillis-Steele inclusive scan over the buckets
  for (int d = 1; d < BWD_NB; d <<= 1) {
    const unsigned add = bin >= d ? tot[bin - d] : 0u;
    __syncthreads();
    tot[bin] += add;
    __syncthreads();
  }
  const uint32_t end = ts + tot[bin];
  const uint32_t start = end - run;
  uint32_t* bb = P.binbase + (size_t)t * (BWD_NB + 1);
  bb[bin] = start;
  if (bin == BWD_NB - 1) bb[BWD_NB] = te;
  if (run == 0) return;
  // an exact table's buckets are single rows: in final order after the partition pass whatever
  // their size, cut at blocks like heavy buckets but never listed for the heavy kernel
  const bool heavy = !exact && run > BWD_TH;
  const bool sorted = exact || heavy;
  // blocks whose first position lies in this bucket
  const uint32_t j0 = (start - ts + BWD_CH - 1) / BWD_CH;
  const uint32_t j1 = (end - 1 - ts) / BWD_CH;
  for (uint32_t j = j0; j <= j1; ++j) {
    const uint32_t bs = ts + j * BWD_CH;
    P.ucut[c0 + j] = (bs == start || sorted) ? bs : end;
    const uint32_t bnext = min(bs + (uint32_t)BWD_CH, te);
    P.uflag[c0 + j] = (exact || (heavy && bnext <= end)) ? 1u : 0u;
  }
  if (heavy) {
    const uint32_t slot = atomicAdd(P.hcount, 1u);
    BwdHeavy hv;
    hv.t = t;
    hv.bin = (uint32_t)bin;
    hv.start = start;
    hv.end = end;
    P.hlist[slot] = hv;
  }
}

// ------------------------------------------------------------------------------------------
// scatter: the one global pass -- stable partition of every chunk into its table's buckets
// ------------------------------------------------------------------------------------------
// Element order inside a chunk is position order; bwd_rank_tile gives every element its index in
// the chunk's stable bucket-sorted order; the elements are laid out in THAT order in LDS and
// written back from there, so lanes that are neighbours in a wave store to neighbouring addresses
// whenever they share a bucket (one 8-byte {key, src} store per element; element-order stores are
// 2 x 4 bytes to unrelated lines, ~3x write amplification measured).
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_scatter_kernel(
    const TzrTable* __restrict__ tables, int T, BwdSrcArgs A, BwdPlan P) {
  __shared__ BwdRankLds<BWD_NB> L;
  __shared__ unsigned base0[BWD_NB];  // global position of the chunk's first element of each bucket
  __shared__ uint2 stage[BWD_CH];
  BwdChunkDesc cd;
  if (!bwd_chunk(P, blockIdx.x, &cd)) return;
  const int t = cd.t;
  const TzrTable tb = tables[t];
  const int n = (int)(cd.e - cd.s);
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const unsigned* hrow = P.hist + (size_t)blockIdx.x * BWD_NB;
  const unsigned* bb = P.binbase + (size_t)t * (BWD_NB + 1);
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) base0[i] = bb[i] + hrow[i];
  BwdGeo G;
  G.fstart = P.feat_start;
  G.fkey = P.feat_key;
  G.tchunk = P.tab_chunk;
  // all of the chunk's elements are loaded up front (independent coalesced loads): the ranking
  // then runs out of registers and pays one memory latency per workgroup
  constexpr int kRounds = BWD_CH / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds], dig[kRounds], dest[kRounds];
  uint32_t vmask = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    kreg[r] = sreg[r] = dig[r] = 0u;
    if (r < rounds && lp < n) {
      vmask |= 1u << r;
      int64_t kk;
      bwd_elem0(G, tb, A, cd.s + lp, &kreg[r], &sreg[r], &kk);
      dig[r] = bwd_bucket(kreg[r], cd.mult);
      if (!A.uniform) {  // bag of every lookup, once (same value from every table a key feeds)
        const int64_t b = tzr_last_le(A.offsets + kk * A.B, A.B, (int64_t)sreg[r]);
        P.bag_of[sreg[r]] = (uint32_t)(kk * A.B + b);
      }
    }
  }
  bwd_rank_tile<BWD_NB, kRounds>(dig, vmask, rounds, bwd_bits((uint32_t)cd.nb - 1u), L, dest);
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if ((vmask >> r) & 1u) stage[dest[r]] = make_uint2(kreg[r], sreg[r]);
  __syncthreads();
  uint2* __restrict__ kout = P.ks[1];
  for (int i = threadIdx.x; i < n; i += BWD_THREADS) {
    const uint2 v = stage[i];
    const unsigned d = bwd_bucket(v.x, cd.mult);
    kout[base0[d] + ((unsigned)i - (unsigned)L.lstart[d])] = v;
  }
}

// ------------------------------------------------------------------------------------------
// heavy buckets: stable LSD sort on the row-id bits left inside the bucket, one workgroup per
// bucket, tile by tile through ks[0] and back (an even number of passes ends in place).  A single
// workgroup walks the tiles in order, so the running per-digit offsets ARE the cross-tile prefix:
// no per-tile histogram storage, no inter-workgroup traffic.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_heavy_kernel(
    const TzrTable* __restrict__ tables, BwdPlan P) {
  __shared__ BwdRankLds<BWD_NB> L;
  __shared__ unsigned gstart[BWD_NB + 1];
  __shared__ uint32_t wtot[BWD_WAVES];
  const unsigned nh = P.hcount[0];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  constexpr int kRounds = BWD_HT / BWD_THREADS;
  for (unsigned hi = blockIdx.x; hi < nh; hi += gridDim.x) {
    const BwdHeavy H = P.hlist[hi];
    const int64_t rows = tables[H.t].rows;
    int nb;
    uint64_t mult;
    bwd_bucket_params(rows, &nb, &mult);
    // row ids of the bucket: [klo, khi)
    const uint64_t klo = (((uint64_t)H.bin << 32) + mult - 1) / mult;
    uint64_t khi = (((uint64_t)(H.bin + 1) << 32) + mult - 1) / mult;
    if (khi > (uint64_t)rows) khi = (uint64_t)rows;
    const int bits = max(1, bwd_bits((uint32_t)(khi - klo - 1)));
    const int npass = bits <= 2 * BWD_RB ? 2 : 4;
    const int width = (bits + npass - 1) / npass;
    const unsigned mask = (1u << width) - 1u;
    const int n = (int)(H.end - H.start);
    for (int pass = 0; pass < npass; ++pass) {
      const uint2* __restrict__ src = ((pass & 1) ? P.ks[0] : P.ks[1]) + H.start;
      uint2* __restrict__ dst = ((pass & 1) ? P.ks[1] : P.ks[0]) + H.start;
      const int shift = pass * width;
      for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) gstart[i] = 0;
      __syncthreads();
      for (int i0 = 0; i0 < n; i0 += 4 * BWD_THREADS) {
        uint32_t k4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = i0 + j * BWD_THREADS + (int)threadIdx.x;
          k4[j] = i < n ? src[i].x : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = i0 + j * BWD_THREADS + (int)threadIdx.x;
          if (i < n) atomicAdd(&gstart[((k4[j] - (uint32_t)klo) >> shift) & mask], 1u);
        }
      }
      __syncthreads();
      bwd_block_scan(gstart, BWD_NB, wtot);
      for (int t0 = 0; t0 < n; t0 += BWD_HT) {
        const int nt = min(BWD_HT, n - t0);
        const int pw = bwd_wave_span(nt);
        const int rounds = pw / TZR_WAVE;
        uint32_t kreg[kRounds], sreg[kRounds], dig[kRounds], dest[kRounds];
        uint32_t vmask = 0;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
          const int lp = wv * pw + r * TZR_WAVE + lane;
          kreg[r] = sreg[r] = dig[r] = 0u;
          if (r < rounds && lp < nt) {
            vmask |= 1u << r;
            const uint2 v = src[t0 + lp];
            kreg[r] = v.x;
            sreg[r] = v.y;
            dig[r] = ((v.x - (uint32_t)klo) >> shift) & mask;
          }
        }
        bwd_rank_tile<BWD_NB, kRounds>(dig, vmask, rounds, width, L, dest);
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
          if ((vmask >> r) & 1u)
            dst[gstart[dig[r]] + (dest[r] - (uint32_t)L.lstart[dig[r]])] = make_uint2(kreg[r], sreg[r]);
        __syncthreads();
        for (int d = threadIdx.x; d < BWD_NB; d += BWD_THREADS)
          gstart[d] += (unsigned)L.lstart[d + 1] - (unsigned)L.lstart[d];
        __syncthreads();
      }
      __threadfence();  // this workgroup reads the pass's output back in the next pass
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

int g_tzr_bwd_force_prep = 0;  // tzr_tune("bwd_force_prep"): take the > BWD_GEO geometry path

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
  if (max_rows > (1LL << 32)) return TZR_ERR_UNSUPPORTED;  // row ids travel as 32-bit keys
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  BwdPlan P;
  if (n_positions < 0 || n_positions >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  if (bwd_layout(&P, ws, n_values, n_positions, n_feats, n_tables, max_dim) > ws_bytes)
    return TZR_ERR_WORKSPACE;
  if (n_values == 0 || B == 0) return TZR_OK;
  if (!d_values) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned chunks = (unsigned)P.max_chunks;
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = d_offsets;
  A.B = B;
  A.uniform = (int)uniform;
  if (n_feats > BWD_GEO || n_tables > BWD_GEO || g_tzr_bwd_force_prep) {
    hipLaunchKernelGGL(tzr_bwd_prep_kernel, dim3(1), dim3(BWD_THREADS), 0, s, d_tables, n_tables, A,
                       n_feats, P);
    hipLaunchKernelGGL(tzr_bwd_hist_kernel<false>, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, n_feats, A, P);
  } else {
    hipLaunchKernelGGL(tzr_bwd_hist_kernel<true>, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, n_feats, A, P);
  }
  hipLaunchKernelGGL(tzr_bwd_scan_kernel, dim3(n_tables), dim3(BWD_NB), 0, s, d_tables, n_tables, P);
  hipLaunchKernelGGL(tzr_bwd_scatter_kernel, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                     n_tables, A, P);
  const unsigned hgrid = (unsigned)std::min<int64_t>(P.max_heavy, 512);
  hipLaunchKernelGGL(tzr_bwd_heavy_kernel, dim3(hgrid), dim3(BWD_THREADS), 0, s, d_tables, P);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
