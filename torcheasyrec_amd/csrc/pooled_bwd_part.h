// The partition pass of the backward plan (K6, see pooled_bwd.hip) as device functions: included by
// pooled_bwd.hip (its own launch) and by pooled_fwd.hip (the same workgroups inside the forward's
// launch: the pass depends on the ids only and its ~15 us of latency-bound work disappear under the
// forward's memory traffic).
#pragma once
#include <tzr_gfx950.h>

#include "pooled_bwd.h"


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
  const int32_t* tpchunk;  // [T+1]
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
  uint32_t tpchunk[BWD_GEO + 1];  // partition chunks
  int64_t trows[BWD_GEO];     // the fields of TzrTable the plan reads: the descriptor is loaded ONCE,
  int32_t tfirst[BWD_GEO];    // together with the feature descriptors (one round trip for the geometry)
  int32_t tnfeats[BWD_GEO];
  uint32_t wtot[BWD_WAVES];
};

// Table-major segment starts and the chunk map, derived by every partition workgroup on its own
// (F + T small loads -- issued together, one round trip -- and two block scans) so that the plan needs
// no single-workgroup launch ahead of it.  Keys of the KJT this module does not own (table < 0) are
// ordered last and contribute nothing.
__device__ __forceinline__ void bwd_geometry(const TzrTable* __restrict__ tables, int T,
                                             const BwdSrcArgs& A, int F, uint32_t ch, uint32_t pch, BwdGeoLds& G) {
  static_assert(BWD_GEO <= BWD_THREADS, "one feature and one table descriptor per thread");
  const int i = threadIdx.x;
  TzrFeature ft;
  TzrTable tb;
  ft.table = -1; ft.key = 0; ft.order = 0;
  tb.rows = 0; tb.first_order = 0; tb.n_feats = 0;
  if (i < F) ft = A.feats[i];
  if (i < T) tb = tables[i];
  if (i < F) {
    const int64_t key = ft.key;
    const int64_t n =
        ft.table < 0 ? 0 : (A.uniform ? A.B : A.offsets[(key + 1) * A.B] - A.offsets[key * A.B]);
    G.fstart[ft.order] = (uint32_t)n;
    G.fkey[ft.order] = ft.key;
  }
  if (i < T) {
    G.trows[i] = tb.rows;
    G.tfirst[i] = tb.first_order;
    G.tnfeats[i] = tb.n_feats;
  }
  __syncthreads();
  bwd_block_scan(G.fstart, F, G.wtot);
  if (i < T) {
    const uint32_t s = tb.n_feats > 0 ? G.fstart[tb.first_order] : 0u;
    const uint32_t e = tb.n_feats > 0 ? G.fstart[tb.first_order + tb.n_feats] : 0u;
    G.tchunk[i] = (e - s + ch - 1) / ch;
    G.tpchunk[i] = (e - s + pch - 1) / pch;
  }
  __syncthreads();
  // tables are visited in first_order order == table-major position order only if table ids
  // follow it; starts are absolute, so the chunk maps just need a prefix in table-id order
  bwd_block_scan(G.tchunk, T, G.wtot);
  bwd_block_scan(G.tpchunk, T, G.wtot);
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


// the per-table arrival counters and item counts of a plan start at zero (one tiny launch ahead of the
// partition pass, wherever that runs)
static __global__ void tzr_bwd_zero_kernel(uint32_t* tarr, uint32_t* tcount, int T) {
  for (int t = threadIdx.x; t < T; t += blockDim.x) tarr[t] = tcount[t] = 0;
}

// argument checks + workspace layout shared by the entry points that start a plan
static inline int bwd_plan_check(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats, int n_feats,
                                 int n_keys, int64_t max_rows, int max_dim, const int64_t* d_offsets,
                                 int64_t n_values, int64_t n_positions, int64_t B, int uniform_bag_len, void* ws,
                                 size_t ws_bytes, BwdPlan* P) {
  if (!d_tables || !d_feats || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B < 0 || max_dim <= 0 || max_rows < 0)
    return TZR_ERR_INVALID;
  if (uniform_bag_len != 1 && !d_offsets) return TZR_ERR_INVALID;
  if (n_keys <= 0) return TZR_ERR_INVALID;
  if (n_values >= (1LL << 32) || (int64_t)n_keys * B >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  if (max_rows > (1LL << 32)) return TZR_ERR_UNSUPPORTED;  // row ids travel as 32-bit keys
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  if (n_positions < 0 || n_positions >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  if (bwd_layout(P, ws, n_values, n_positions, n_feats, n_tables, max_dim) > ws_bytes) return TZR_ERR_WORKSPACE;
  return TZR_OK;
}

// ------------------------------------------------------------------------------------------
// table scan: by the last chunk of the table to arrive
// ------------------------------------------------------------------------------------------
struct BwdPartLds {
  // (L first: a generic pointer to the LDS object at offset 0 -- the geometry arrays, read through BwdGeo --
  // trips "Illegal instruction detected: Operand has incorrect register class" in this hipcc)
  BwdRankLds<BWD_NB> L;
  union {
    BwdGeoLds G;             // prologue: table-major geometry
    uint2 stage[BWD_SUB];    // then: one sub-tile in slab order
  };
  uint32_t tot[BWD_NB + 1];  // chunk: bucket counts -> slab-local bucket starts; table scan: bucket totals -> starts
  uint16_t run[BWD_NB];      // chunk: lookups of each bucket placed by the sub-tiles so far
  uint32_t wtot[BWD_WAVES];
  uint32_t flag, any_heavy, nitems;
};

// Units of the apply.  Cut points = bucket boundaries + the BWD_CH-block boundaries that fall
// INSIDE a heavy bucket; unit j of a table starts at the first cut point at or after block j
// (ts + j * ch).  So a unit is a whole number of light buckets (which the sort kernel orders in LDS)
// and/or block-sized slices of heavy buckets (ordered by the heavy workers), it holds fewer
// than BWD_CH + BWD_TH lookups, and a run of one row can only cross a unit boundary inside a heavy
// bucket.
__device__ __forceinline__ void bwd_table_scan(const TzrTable& tb, const BwdChunkDesc& cd,
                                               int one_wg_heavy, const BwdPlan& P, BwdPartLds& S) {
  const int tid = threadIdx.x;
  const int lane = tid & (TZR_WAVE - 1);
  const int t = cd.t;
  const int c0 = cd.first_chunk;    // unit blocks
  const int q0 = cd.first_pchunk;   // partition chunks: the rows read here
  const uint32_t ts = (uint32_t)cd.ts, te = (uint32_t)cd.te;
  const uint32_t ch = (uint32_t)P.ch;
  const int C = (int)((te - ts + (uint32_t)P.pch - 1) / (uint32_t)P.pch);
  // 1. bucket totals = column sums of the chunks' counts.  Thread q of a half owns buckets 4q .. 4q+3
  //    (one published 8-byte word of a row + the first entry of the next word), the two halves of the
  //    workgroup take even / odd chunks; 8 chunks of independent loads in flight per thread.
  {
    const int half = tid >> 7, q = tid & 127;
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    constexpr int kBatch = 8;
    for (int cb = half; cb < C; cb += 2 * kBatch) {
      uint64_t a[kBatch], b[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const int c = cb + 2 * j;
        a[j] = b[j] = 0;
        if (c < C) {
          const uint64_t* row = reinterpret_cast<const uint64_t*>(P.lst + (size_t)(q0 + c) * BWD_LROW);
          a[j] = tzr_consume_u64(row + q);
          b[j] = tzr_consume_u64(row + q + 1);
        }
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const uint32_t e0 = (uint32_t)(a[j] & 0xFFFFu), e1 = (uint32_t)((a[j] >> 16) & 0xFFFFu);
        const uint32_t e2 = (uint32_t)((a[j] >> 32) & 0xFFFFu), e3 = (uint32_t)(a[j] >> 48);
        const uint32_t e4 = (uint32_t)(b[j] & 0xFFFFu);
        acc[0] += e1 - e0; acc[1] += e2 - e1; acc[2] += e3 - e2; acc[3] += e4 - e3;
      }
    }
    if (half == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) S.tot[4 * q + j] = acc[j];
    }
    if (tid == 0) S.any_heavy = S.nitems = 0;
    __syncthreads();
    if (half == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) S.tot[4 * q + j] += acc[j];
    }
    __syncthreads();
  }
  bwd_block_scan(S.tot, BWD_NB, S.wtot);  // exclusive: S.tot[b] = lookups in buckets < b, S.tot[NB] = all
  uint32_t* bb = P.binbase + (size_t)t * (BWD_NB + 1);
  const int64_t rows = tb.rows;
  const bool exact = rows <= BWD_NB;
  for (int r = 0; r < BWD_NB / BWD_THREADS; ++r) {  // workgroup-uniform
    const int bin = r * BWD_THREADS + tid;
    const uint32_t start = ts + S.tot[bin];
    const uint32_t end = ts + S.tot[bin + 1];
    const uint32_t run = end - start;
    bb[bin] = start;
    if (bin == BWD_NB - 1) bb[BWD_NB] = te;
    const bool heavy = run > BWD_TH;
    {  // bitmap of the heavy buckets (read by the unit sort): one ballot per wave = two words
      const unsigned long long hb = __ballot(heavy);
      if (lane == 0) {
        uint32_t* hw = P.hbits + (size_t)t * (BWD_NB / 32) + (bin / TZR_WAVE) * 2;
        hw[0] = (uint32_t)hb;
        hw[1] = (uint32_t)(hb >> 32);
        if (hb != 0) atomicOr(&S.any_heavy, 1u);
      }
    }
    // Stitch groups of the apply: a run of one row can only cross a unit boundary inside a heavy
    // bucket; the units overlapping such a bucket meet at its counter (reduce kernel)
    {
      uint32_t expect = 0;
      if (heavy) {
        const uint32_t units = (end - 1 - ts) / ch - (start - ts) / ch + 1;
        expect = units > 1 ? units : 0u;
      }
      P.sexp[(size_t)t * BWD_NB + bin] = expect;
      P.sarr[(size_t)t * BWD_NB + bin] = 0;
    }
    if (run == 0) continue;
    // blocks whose first position lies in this bucket
    const uint32_t j0 = (start - ts + ch - 1) / ch;
    const uint32_t j1 = (end - 1 - ts) / ch;
    for (uint32_t j = j0; j <= j1; ++j) {
      const uint32_t bs = ts + j * ch;
      const bool here = bs == start || heavy;
      P.ucut[c0 + j] = here ? bs : end;
      P.ub0[c0 + j] = here ? (uint32_t)bin : (uint32_t)bin + 1u;
      const uint32_t bnext = min(bs + ch, te);
      P.uflag[c0 + j] = (heavy && bnext <= end) ? 1u : 0u;
    }
    if (!heavy) continue;
    // row ids of the bucket: [klo, khi).  At most BWD_NB of them (tables up to BWD_NB^2 rows): one
    // counting pass on (row id - klo) sorts the bucket, tile by tile, one workgroup per tile.  A wider
    // bucket is heavy because of ONE row nearly always (the clipped Zipf tail, a default id): tiles
    // again, splitting around that row; a wide bucket of a single tile is sorted in LDS by one workgroup.
    int kind;
    if (exact) {
      kind = BWD_HK_COPY;
    } else {
      int nb;
      uint64_t mult;
      bwd_bucket_params(rows, &nb, &mult);
      const uint64_t klo = (((uint64_t)bin << 32) + mult - 1) / mult;
      uint64_t khi = (((uint64_t)(bin + 1) << 32) + mult - 1) / mult;
      if (khi > (uint64_t)rows) khi = (uint64_t)rows;
      kind = khi - klo <= (uint64_t)BWD_NB ? BWD_HK_ONEPASS : (run > BWD_HT ? BWD_HK_HOT : BWD_HK_SERIAL);
      if (one_wg_heavy) kind = BWD_HK_SERIAL;  // tzr_tune("bwd_one_wg_heavy"): no tile parallelism
    }
    // tiles = runs of G whole chunks; the slots come from a workgroup-local counter (one table = one
    // workgroup here: no global atomics), the items go to the table's own region of the list
    const uint32_t G = kind == BWD_HK_SERIAL ? (uint32_t)C : bwd_tile_chunks(run, (uint32_t)C);
    const uint32_t ntiles = ((uint32_t)C + G - 1) / G;
    const uint32_t slot0 = atomicAdd(&S.nitems, ntiles);
    BwdHeavy hv;
    hv.t = t;
    hv.bin = (uint32_t)bin;
    hv.start = start;
    hv.kind = kind;
    hv.pad[0] = hv.pad[1] = 0;
    BwdHeavy* list = P.hlist + bwd_hbase(ts, (uint32_t)t);
    for (uint32_t k = 0; k < ntiles; ++k) {
      hv.c_begin = (int32_t)(k * G);
      hv.c_end = (int32_t)min((k + 1) * G, (uint32_t)C);
      list[slot0 + k] = hv;
    }
  }
  __syncthreads();
  if (tid == 0) {
    P.tab_stitch[t] = S.any_heavy;
    P.tcount[t] = S.nitems;
  }
}

// ------------------------------------------------------------------------------------------
// part: one partition chunk -> its slab
// ------------------------------------------------------------------------------------------
// A partition chunk is up to BWD_PK unit blocks of one table (pch positions).  All its lookups are loaded up
// front (independent coalesced loads: one memory latency per workgroup), their buckets counted with LDS
// atomics -> slab-local bucket starts (the row published for the table scan), then the lookups are ranked
// sub-tile by sub-tile (BWD_SUB at a time, stable: bwd_rank_tile + the running count of each bucket) and
// written to their slab position, staged through LDS so that neighbouring lanes store neighbouring
// elements whenever they share a bucket.
#define BWD_PROF(k)                                                                    \
  do {                                                                                 \
    if (prof_on && threadIdx.x == 0) P.prof[(size_t)q * 8 + (k)] = wall_clock64();     \
  } while (0)

template <bool FUSED>
__device__ __forceinline__ void bwd_part_body(const TzrTable* __restrict__ tables, int T, int F,
                                              const BwdSrcArgs& A, const BwdPlan& P, int one_wg_heavy,
                                              BwdPartLds& S, int q) {
  const bool prof_on = (one_wg_heavy & 2) != 0;  // bit 1 of the knob word: phase timestamps (tzr_tune("bwd_prof"))
  one_wg_heavy &= 1;
  BWD_PROF(0);
  BwdGeo G;
  if (FUSED) {
    bwd_geometry(tables, T, A, F, (uint32_t)P.ch, (uint32_t)P.pch, S.G);
    G.fstart = S.G.fstart;
    G.fkey = S.G.fkey;
    G.tchunk = reinterpret_cast<const int32_t*>(S.G.tchunk);
    G.tpchunk = reinterpret_cast<const int32_t*>(S.G.tpchunk);
    if (q == 0) {  // the later kernels of the plan and the apply read it from the workspace
      for (int o = threadIdx.x; o <= F; o += BWD_THREADS) P.feat_start[o] = S.G.fstart[o];
      for (int o = threadIdx.x; o < F; o += BWD_THREADS) P.feat_key[o] = S.G.fkey[o];
      for (int f = threadIdx.x; f < F; f += BWD_THREADS) P.feat_by_order[A.feats[f].order] = f;
      for (int t = threadIdx.x; t <= T; t += BWD_THREADS) {
        P.tab_chunk[t] = (int32_t)S.G.tchunk[t];
        P.tab_pchunk[t] = (int32_t)S.G.tpchunk[t];
      }
    }
  } else {
    G.fstart = P.feat_start;
    G.fkey = P.feat_key;
    G.tchunk = P.tab_chunk;
    G.tpchunk = P.tab_pchunk;
  }
  if (q == 0) {  // unit blocks behind the last table: nobody's
    BwdChunkDesc none;
    none.t = -1;
    none.nb = none.exact = none.last_chunk = none.first_chunk = none.first_pchunk = 0;
    none.s = none.e = none.ts = none.te = 0;
    none.mult = 0;
    for (int64_t c = (int64_t)G.tchunk[T] + threadIdx.x; c < P.max_chunks; c += BWD_THREADS) P.cdesc[c] = none;
  }
  if (q >= G.tpchunk[T]) return;
  BwdChunkDesc cd;
  TzrTable tb;
  {
    int lo = 0, hi = T;  // last t with tpchunk[t] <= q (the non-empty table holding it)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (G.tpchunk[mid] <= q) lo = mid; else hi = mid;
    }
    if (FUSED) {  // the descriptor fields the plan reads sit in LDS already
      tb.rows = S.G.trows[lo];
      tb.first_order = S.G.tfirst[lo];
      tb.n_feats = S.G.tnfeats[lo];
    } else {
      tb = tables[lo];
    }
    cd.t = lo;
    bwd_bucket_params(tb.rows, &cd.nb, &cd.mult);
    cd.exact = tb.rows <= BWD_NB;
    cd.first_chunk = G.tchunk[lo];
    cd.last_chunk = G.tchunk[lo + 1];
    cd.first_pchunk = G.tpchunk[lo];
    cd.ts = tb.n_feats > 0 ? (int64_t)G.fstart[tb.first_order] : 0;
    cd.te = tb.n_feats > 0 ? (int64_t)G.fstart[tb.first_order + tb.n_feats] : cd.ts;
  }
  const int64_t qs = cd.ts + (int64_t)(q - cd.first_pchunk) * P.pch;  // this workgroup's positions [qs, qe)
  const int64_t qe = min(cd.te, qs + (int64_t)P.pch);
  const int n = (int)(qe - qs);
  {  // descriptors of the unit blocks the chunk covers
    const int pk = P.pch / P.ch;
    if ((int)threadIdx.x < pk) {
      const int64_t s = qs + (int64_t)threadIdx.x * P.ch;
      if (s < cd.te) {
        BwdChunkDesc u = cd;
        u.s = s;
        u.e = min(cd.te, s + (int64_t)P.ch);
        P.cdesc[cd.first_chunk + (s - cd.ts) / P.ch] = u;
      }
    }
  }
  BWD_PROF(1);
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  // sub-tile j = positions [j * BWD_SUB, ...) of the chunk, dealt wave-contiguously inside the sub-tile
  constexpr int kSubR = BWD_SUB / BWD_THREADS;  // rounds per sub-tile
  constexpr int kMaxSub = BWD_PK * BWD_CH / BWD_SUB;
  const int nsub = (n + BWD_SUB - 1) / BWD_SUB;
  uint32_t kreg[kMaxSub * kSubR], sreg[kMaxSub * kSubR];
  uint32_t vmask = 0;  // bit j * kSubR + r
#pragma unroll
  for (int j = 0; j < kMaxSub; ++j) {
    const int nj = min(BWD_SUB, n - j * BWD_SUB);  // <= 0: past the chunk (workgroup-uniform)
    const int pw = nj > 0 ? bwd_wave_span(nj) : 0;
#pragma unroll
    for (int r = 0; r < kSubR; ++r) {
      const int lp = wv * pw + r * TZR_WAVE + lane;
      kreg[j * kSubR + r] = sreg[j * kSubR + r] = 0u;
      if (r * TZR_WAVE < pw && lp < nj) {
        vmask |= 1u << (j * kSubR + r);
        int64_t kk;
        bwd_elem0(G, tb, A, qs + j * BWD_SUB + lp, &kreg[j * kSubR + r], &sreg[j * kSubR + r], &kk);
        if (!A.uniform) {  // bag of every lookup, once (same value from every table a key feeds)
          const int64_t b = tzr_last_le(A.offsets + kk * A.B, A.B, (int64_t)sreg[j * kSubR + r]);
          P.bag_of[sreg[j * kSubR + r]] = (uint32_t)(kk * A.B + b);
        }
      }
    }
  }
  for (int i = threadIdx.x; i <= BWD_NB; i += BWD_THREADS) S.tot[i] = 0;
  for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) S.run[i] = 0;
  __syncthreads();  // (the geometry in LDS is dead from here on: its space becomes the stage)
  BWD_PROF(2);
#pragma unroll
  for (int i = 0; i < kMaxSub * kSubR; ++i)
    if ((vmask >> i) & 1u) atomicAdd(&S.tot[bwd_bucket(kreg[i], cd.mult)], 1u);
  __syncthreads();
  bwd_block_scan(S.tot, BWD_NB, S.wtot);  // slab-local start of every bucket, S.tot[NB] = n
  BWD_PROF(3);
  // the chunk's row of bucket starts: published for the table scan (another workgroup of this launch), and the
  // arrival counted, BEFORE the slab is written -- the scan reads rows only, and it is the tail of the launch's
  // longest dependent chain
  {
    uint64_t* row = reinterpret_cast<uint64_t*>(P.lst + (size_t)q * BWD_LROW);
    for (int i = threadIdx.x; i < BWD_LROW / 4; i += BWD_THREADS) {
      uint64_t w = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) w |= (uint64_t)(S.tot[min(4 * i + j, BWD_NB)] & 0xFFFFu) << (16 * j);
      tzr_publish_u64(row + i, w);
    }
  }
  tzr_drain_stores();
  __syncthreads();
  BWD_PROF(4);
  if (threadIdx.x == 0) {
    const uint32_t Cp = (uint32_t)((cd.te - cd.ts + P.pch - 1) / P.pch);
    S.flag = tzr_arrive(P.tarr + cd.t) == Cp - 1u ? 1u : 0u;
  }
  uint2* __restrict__ slab = P.ks[1] + qs;
  const int wbits = bwd_bits((uint32_t)cd.nb - 1u);
#pragma unroll
  for (int j = 0; j < kMaxSub; ++j) {
    if (j < nsub) {  // workgroup-uniform
      const int nj = min(BWD_SUB, n - j * BWD_SUB);
      const int rounds = bwd_wave_span(nj) / TZR_WAVE;
      uint32_t dig[kSubR], dest[kSubR];
      const uint32_t vm = (vmask >> (j * kSubR)) & ((1u << kSubR) - 1u);
#pragma unroll
      for (int r = 0; r < kSubR; ++r) dig[r] = ((vm >> r) & 1u) ? bwd_bucket(kreg[j * kSubR + r], cd.mult) : 0u;
      bwd_rank_tile<BWD_NB, kSubR>(dig, vm, rounds, wbits, S.L, dest);
      // sub-tile order first (LDS), then out: lanes that are neighbours in a wave store neighbouring elements
#pragma unroll
      for (int r = 0; r < kSubR; ++r)
        if ((vm >> r) & 1u) S.stage[dest[r]] = make_uint2(kreg[j * kSubR + r], sreg[j * kSubR + r]);
      __syncthreads();
      for (int i = threadIdx.x; i < nj; i += BWD_THREADS) {
        const uint2 v = S.stage[i];
        const uint32_t d = bwd_bucket(v.x, cd.mult);
        slab[S.tot[d] + (uint32_t)S.run[d] + ((uint32_t)i - (uint32_t)S.L.lstart[d])] = v;
      }
      __syncthreads();
      for (int d = threadIdx.x; d < BWD_NB; d += BWD_THREADS)
        S.run[d] = (uint16_t)(S.run[d] + (S.L.lstart[d + 1] - S.L.lstart[d]));
      __syncthreads();
    }
  }
  BWD_PROF(5);
  if (S.flag) {
    bwd_table_scan(tb, cd, one_wg_heavy, P, S);
    BWD_PROF(6);
  }
}
