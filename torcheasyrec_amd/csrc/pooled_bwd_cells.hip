// K6 / K7 in the "cells" form: a one-launch index plan and the apply that reads it (see pooled_bwd_cells.h for the design).
//
// Replaces, like pooled_bwd.hip / pooled_bwd_apply.hip, fbgemm's transpose_embedding_input + split_embedding_backward_codegen_*
// behind autograd of self.ebc(kjt) (/root/reference/tzrec/modules/embedding.py:930) with the optimizer fused by
// apply_optimizer_in_backward (/root/reference/tzrec/main.py:774-781; optim/optimizer_builder.py:30-97): per distinct (table, row)
// the gradient rows of its lookups are added in LOOKUP-POSITION order and the row is updated once -- the same sums in the
// same order as the four-launch plan's apply, bit for bit (tests/test_pooled_parity.py: bwd_path "cells").
#include <tzr_gfx950.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "pooled_bwd_apply.h"
#include "pooled_bwd_cells.h"
#include "pooled_bwd_sort.h"

// ------------------------------------------------------------------------------------------------------------------------
// geometry (host)
// ------------------------------------------------------------------------------------------------------------------------
namespace {

struct HostGeo {
  BwdCellsGeo g;
  std::vector<BwdCellChunk> chunks;
  std::vector<BwdCellUnit> units;
  std::vector<uint32_t> fstart;
  std::vector<int32_t> fkey, fbo;
};

inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

// bwd_bucket_params, host side
inline void bucket_params_host(int64_t rows, int* nb, uint64_t* mult) {
  if (rows <= BWD_NB) {
    *nb = rows < 1 ? 1 : (int)rows;
    *mult = 1ull << 32;
  } else {
    *nb = BWD_NB;
    *mult = ((uint64_t)BWD_NB << 32) / (uint64_t)rows;
  }
}

// TZR_OK, or TZR_ERR_UNSUPPORTED when this (tables, batch size) is not a case for the cells plan
int build_geometry(const TzrTable* tabs, int T, const TzrFeature* feats, int F, int64_t B, int max_dim, HostGeo& H) {
  if (T <= 0 || F <= 0 || B <= 0 || max_dim <= 0 || max_dim > BWD_MAXDIM || (max_dim & 3)) return TZR_ERR_INVALID;
  H.fstart.assign(F + 1, 0);
  H.fkey.assign(F, 0);
  H.fbo.assign(F, 0);
  std::vector<int> seen(F, 0);
  for (int f = 0; f < F; ++f) {
    const TzrFeature& ft = feats[f];
    if (ft.order < 0 || ft.order >= F || seen[ft.order]) return TZR_ERR_INVALID;
    seen[ft.order] = 1;
    H.fstart[ft.order] = ft.table < 0 ? 0u : (uint32_t)B;  // one id per bag
    H.fkey[ft.order] = ft.key;
    H.fbo[ft.order] = f;
  }
  {
    uint64_t run = 0;
    for (int o = 0; o < F; ++o) {
      const uint32_t n = H.fstart[o];
      H.fstart[o] = (uint32_t)run;
      run += n;
    }
    if (run >= (1ull << 32)) return TZR_ERR_UNSUPPORTED;
    H.fstart[F] = (uint32_t)run;
  }
  const int64_t N = H.fstart[F];
  if (N == 0) return TZR_ERR_UNSUPPORTED;
  const int ch = bwd_pick_ch(N);
  int64_t n_chunks = 0;
  std::vector<int64_t> cfirst(T + 1, 0);
  for (int t = 0; t < T; ++t) {
    const TzrTable& tb = tabs[t];
    if (tb.n_feats < 0 || (tb.n_feats > 0 && (tb.first_order < 0 || tb.first_order + tb.n_feats > F))) return TZR_ERR_INVALID;
    const int64_t n = tb.n_feats > 0 ? (int64_t)H.fstart[tb.first_order + tb.n_feats] - (int64_t)H.fstart[tb.first_order] : 0;
    const int64_t C = (n + ch - 1) / ch;
    if (C > BWD_CELLS_MAXC) return TZR_ERR_UNSUPPORTED;
    cfirst[t] = n_chunks;
    n_chunks += C;
  }
  cfirst[T] = n_chunks;
  if (n_chunks > bwd_max_chunks(N, T, ch)) return TZR_ERR_UNSUPPORTED;
  int64_t n_recs = 0, n_counters = 0;
  for (int t = 0; t < T; ++t) {
    const TzrTable& tb = tabs[t];
    const int64_t C = cfirst[t + 1] - cfirst[t];
    if (C == 0) continue;
    if (tb.rows <= 0 || tb.rows > (1LL << 32)) return TZR_ERR_UNSUPPORTED;
    const int64_t ts = H.fstart[tb.first_order], te = H.fstart[tb.first_order + tb.n_feats];
    const int64_t n = te - ts;
    int nb;
    uint64_t mult;
    bucket_params_host(tb.rows, &nb, &mult);
    const int64_t fbase = tb.n_feats == 1 ? (int64_t)H.fkey[tb.first_order] * B : -1;
    for (int64_t c = 0; c < C; ++c) {
      BwdCellChunk cd;
      std::memset(&cd, 0, sizeof(cd));
      cd.t = t;
      cd.nb = nb;
      cd.s = ts + c * ch;
      cd.e = std::min(te, cd.s + ch);
      cd.ts = ts;
      cd.mult = mult;
      cd.rows = tb.rows;
      cd.fbase = fbase;
      H.chunks.push_back(cd);
    }
    BwdCellUnit u;
    std::memset(&u, 0, sizeof(u));
    u.tb = tb;
    u.t = t;
    u.feat = H.fbo[tb.first_order];
    u.ts = ts;
    u.rec = u.rec0 = u.counter = -1;
    const int c_lo = (int)cfirst[t], c_hi = (int)cfirst[t + 1];
    if (tb.rows > BWD_NB) {
      // bucketed table: C units, each a range of the 512 buckets over all chunks (C <= BWD_CELLS_MAXC <= BWD_NB)
      const int64_t U = C;
      for (int64_t j = 0; j < U; ++j) {
        u.c0 = c_lo;
        u.c1 = c_hi;
        u.b0 = (int32_t)(j * nb / U);
        u.b1 = (int32_t)((j + 1) * nb / U);
        u.split = 0;
        if (u.b1 > u.b0) H.units.push_back(u);
      }
    } else if (n <= (int64_t)ch * tb.rows) {
      // a bucket is a row; expected lookups per row <= one chunk: whole rows grouped up to ~ch lookups, over all chunks
      const int64_t rpu = std::max<int64_t>(1, ((int64_t)ch * tb.rows) / n);
      for (int64_t r = 0; r < tb.rows; r += rpu) {
        u.c0 = c_lo;
        u.c1 = c_hi;
        u.b0 = (int32_t)r;
        u.b1 = (int32_t)std::min<int64_t>(tb.rows, r + rpu);
        u.split = 0;
        H.units.push_back(u);
      }
    } else {
      // fewer rows than chunks' worth of lookups: a row is SPLIT over K chunk ranges, partial sums combined by the last to arrive
      const int64_t K = std::min<int64_t>(C, (n + (int64_t)ch * tb.rows - 1) / ((int64_t)ch * tb.rows));
      for (int64_t r = 0; r < tb.rows; ++r) {
        for (int64_t k = 0; k < K; ++k) {
          u.c0 = c_lo + (int32_t)(k * C / K);
          u.c1 = c_lo + (int32_t)((k + 1) * C / K);
          u.b0 = (int32_t)r;
          u.b1 = (int32_t)r + 1;
          u.split = K > 1 ? (int32_t)K : 0;
          if (K > 1) {
            u.rec = (int32_t)(n_recs + k);
            u.rec0 = (int32_t)n_recs;
            u.counter = (int32_t)n_counters;
          }
          H.units.push_back(u);
        }
        if (K > 1) {
          n_recs += K;
          n_counters += 1;
        }
        u.rec = u.rec0 = u.counter = -1;
      }
    }
  }
  BwdCellsGeo& g = H.g;
  std::memset(&g, 0, sizeof(g));
  g.n_chunks = n_chunks;
  g.n_units = (int64_t)H.units.size();
  g.n_recs = n_recs;
  g.n_counters = n_counters;
  g.n_feats = F;
  g.max_dim = max_dim;
  g.ch = ch;
  g.n_positions = N;
  int64_t off = align256(sizeof(BwdCellsGeo));
  g.off_chunks = off;
  off = align256(off + n_chunks * (int64_t)sizeof(BwdCellChunk));
  g.off_units = off;
  off = align256(off + g.n_units * (int64_t)sizeof(BwdCellUnit));
  g.off_fstart = off;
  off = align256(off + (F + 1) * 4);
  g.off_fkey = off;
  off = align256(off + F * 4);
  g.off_fbo = off;
  off = align256(off + F * 4);
  g.off_recs = off;
  off = align256(off + std::max<int64_t>(1, n_recs) * max_dim * 4);
  g.off_rcount = off;
  off = align256(off + std::max<int64_t>(1, n_recs) * 4);
  g.off_counters = off;
  off = align256(off + std::max<int64_t>(1, n_counters) * 4);
  g.off_overflow = off;
  off = align256(off + 16);
  g.bytes = off;
  return TZR_OK;
}

}  // namespace

// Host image of the geometry buffer for (h_tables, h_feats, B): every bag holds exactly one id.  h_out == NULL: only the size.
// out_info[0] = bytes, [1] = chunks (= workgroups of the plan launch), [2] = units (= workgroups of the apply launch),
// [3] = table-major positions, [4] = positions per chunk, [5] = partial-sum records, [6] = byte offset of the overflow word.
// TZR_ERR_UNSUPPORTED: not a case for the cells plan (a table with more than BWD_CELLS_MAXC chunks, no lookups, ...).
extern "C" int tzr_bwd_cells_geometry(const TzrTable* h_tables, int n_tables, const TzrFeature* h_feats, int n_feats, int64_t B,
                                      int max_dim, void* h_out, size_t out_bytes, int64_t* out_info8) {
  if (!h_tables || !h_feats || !out_info8) return TZR_ERR_INVALID;
  HostGeo H;
  const int rc = build_geometry(h_tables, n_tables, h_feats, n_feats, B, max_dim, H);
  if (rc != TZR_OK) return rc;
  const BwdCellsGeo& g = H.g;
  out_info8[0] = g.bytes;
  out_info8[1] = g.n_chunks;
  out_info8[2] = g.n_units;
  out_info8[3] = g.n_positions;
  out_info8[4] = g.ch;
  out_info8[5] = g.n_recs;
  out_info8[6] = g.off_overflow;
  out_info8[7] = 0;
  if (!h_out) return TZR_OK;
  if ((int64_t)out_bytes < g.bytes) return TZR_ERR_WORKSPACE;
  char* b = static_cast<char*>(h_out);
  std::memset(b, 0, (size_t)g.bytes);
  std::memcpy(b, &g, sizeof(g));
  std::memcpy(b + g.off_chunks, H.chunks.data(), H.chunks.size() * sizeof(BwdCellChunk));
  std::memcpy(b + g.off_units, H.units.data(), H.units.size() * sizeof(BwdCellUnit));
  std::memcpy(b + g.off_fstart, H.fstart.data(), H.fstart.size() * 4);
  std::memcpy(b + g.off_fkey, H.fkey.data(), H.fkey.size() * 4);
  std::memcpy(b + g.off_fbo, H.fbo.data(), H.fbo.size() * 4);
  return TZR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// partition: every chunk ordered by bucket in place + its bucket starts.  No workgroup talks to another.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_cells_partition_kernel(
    BwdCellsView V, const TzrTable* __restrict__ tables, BwdSrcArgs A, uint2* __restrict__ slab, uint16_t* __restrict__ lstart) {
  __shared__ BwdRankLds<BWD_NB> L;
  __shared__ uint2 stage[BWD_CH];
  const BwdCellChunk cd = V.chunks[blockIdx.x];
  if (cd.t < 0) return;
  const int n = (int)(cd.e - cd.s);
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  constexpr int kRounds = BWD_CH / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds], dig[kRounds], dest[kRounds];
  uint32_t vmask = 0;
  if (cd.fbase >= 0) {  // (workgroup-uniform) one key reads the table: every id load unconditional, position clamped (bwd_elem_one)
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int lp = wv * pw + r * TZR_WAVE + lane;
      const int lc = lp < n ? lp : n - 1;
      const int64_t i = cd.fbase + (cd.s - cd.ts) + lc;
      int64_t id = A.values[i];
      if ((uint64_t)id >= (uint64_t)cd.rows) id = 0;  // memory safety; K4 reports / clamps
      kreg[r] = (uint32_t)id;
      sreg[r] = (uint32_t)i;
      dig[r] = bwd_bucket(kreg[r], cd.mult);
      vmask |= (r < rounds && lp < n) ? 1u << r : 0u;
    }
  } else {
    const TzrTable tb = tables[cd.t];
    BwdGeo G;
    G.fstart = V.fstart;
    G.fkey = V.fkey;
    G.tchunk = nullptr;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int lp = wv * pw + r * TZR_WAVE + lane;
      kreg[r] = sreg[r] = dig[r] = 0u;
      if (r < rounds && lp < n) {
        vmask |= 1u << r;
        int64_t kk;
        bwd_elem0(G, tb, A, cd.s + lp, &kreg[r], &sreg[r], &kk);
        dig[r] = bwd_bucket(kreg[r], cd.mult);
      }
    }
  }
  bwd_rank_tile<BWD_NB, kRounds>(dig, vmask, rounds, bwd_bits((uint32_t)cd.nb - 1u), L, dest);
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if ((vmask >> r) & 1u) stage[dest[r]] = make_uint2(kreg[r], sreg[r]);
  // the chunk's bucket starts (digits at or above nb hold nothing: their start is n)
  uint16_t* lrow = lstart + (size_t)blockIdx.x * BWD_CELLS_LROW;
  for (int d = threadIdx.x; d <= BWD_NB; d += BWD_THREADS) lrow[d] = L.lstart[d];
  __syncthreads();
  uint2* out = slab + cd.s;
  for (int i = threadIdx.x; i < n; i += BWD_THREADS) out[i] = stage[i];  // one coalesced 8-byte store per lookup
}

// ------------------------------------------------------------------------------------------------------------------------
// apply
// ------------------------------------------------------------------------------------------------------------------------
union BwdCellsLds {
  BwdSortLds S;
  BwdUnitLds U;
};

// lookup `i` (0 .. n) of a unit whose cells' exclusive counts / slab starts lie in LDS
__device__ __forceinline__ uint2 bwd_cells_elem(const uint2* __restrict__ slab, const uint32_t* cpre, const uint32_t* cbase,
                                                int ncell, uint32_t i) {
  int lo = 0;
#pragma unroll
  for (int step = BWD_CELLS_MAXC / 2; step > 0; step >>= 1) {  // largest cell with cpre[cell] <= i (empty cells are skipped)
    const int cand = lo + step;
    if (cand < ncell && cpre[cand] <= i) lo = cand;
  }
  return slab[cbase[lo] + (i - cpre[lo])];
}

// Partial sum of one unit of a split row -> its record; the last of the row's units to arrive adds the records in unit order
// and updates the row.  Wave 0; `sum` valid in the lanes < dim / 4.
template <bool ADAM>
__device__ __forceinline__ void bwd_cells_combine(const BwdCellUnit& u, const BwdCellsView& V, const BwdOpt& opt, float lr,
                                                  int max_dim, float4 sum, uint32_t count, int lane) {
  const int lg = u.tb.dim >> 2;
  if (lane < lg) bwd_publish4(V.recs + (size_t)u.rec * max_dim + 4 * lane, sum);
  if (lane == 0) tzr_publish_u32(V.rcount + u.rec, count);
  tzr_drain_stores();
  int last = 0;
  if (lane == 0) last = tzr_arrive(V.counters + u.counter) == (uint32_t)(u.split - 1) ? 1 : 0;
  last = __shfl(last, 0, TZR_WAVE);
  if (!last) return;
  if (lane == 0) tzr_publish_u32(V.counters + u.counter, 0u);  // the counter is zero again when the launch ends
  float4 tot = tzr_zero4();
  uint32_t cnt = 0;
  for (int k = 0; k < u.split; ++k) {
    if (lane < lg) tot = tzr_add4(tot, bwd_consume4(V.recs + (size_t)(u.rec0 + k) * max_dim + 4 * lane));
    cnt += tzr_consume_u32(V.rcount + u.rec0 + k);
  }
  if (cnt) bwd_apply_row_wave<ADAM>(u.tb, opt, lr, (uint32_t)u.b0, tot, lane);  // (a row nobody looked up is not touched)
}

// A unit with more lookups than the LDS unit holds (skewed ids: this geometry expects evenly filled buckets): its rows one
// after the other in ascending order, as many passes over the unit as it has distinct rows.  Pass 1 finds the smallest row id
// not done yet; pass 2 adds that row's gradient rows -- lane group q takes lookups q, q + G, q + 2 G, ... in order, the groups'
// sums are added in group order: a function of the ids alone.  Correct for any ids; slow on purpose-built ones -- the caller
// sees the overflow word move and sends this distribution to the exact plan.
template <bool ADAM>
__device__ __forceinline__ void bwd_cells_slow_unit(
    const BwdCellUnit& u, const BwdCellsView& V, const uint2* __restrict__ slab, const uint32_t* cpre, const uint32_t* cbase,
    int ncell, int n, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B, int grad_mode,
    const BwdOpt& opt, int max_dim, const TzrDst* sG, float* red, uint32_t* sm) {
  const TzrTable& tb = u.tb;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int lg = tb.dim >> 2, gw = TZR_WAVE / lg;
  const int gi = lane / lg, c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const int groups = gw * BWD_WAVES, q = wv * gw + gi;
  const float lr = *opt.lr;
  const bool single = tb.n_feats == 1;
  const BwdSrc one = bwd_resolve(feats + V.feat_by_order[tb.first_order], sG);
  uint32_t cur = 0;
  float4 rowsum = tzr_zero4();  // split unit: its one row
  uint32_t rowcnt = 0;
  for (;;) {
    uint32_t m = BWD_SENT;
    for (int i = threadIdx.x; i < n; i += BWD_THREADS) {
      const uint32_t k = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)i).x;
      if (k >= cur) m = min(m, k);
    }
    for (int d = TZR_WAVE >> 1; d > 0; d >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d, TZR_WAVE));
    __syncthreads();  // (sm / red of the previous row are read)
    if (lane == 0) sm[wv] = m;
    __syncthreads();
    m = sm[0];
#pragma unroll
    for (int w = 1; w < BWD_WAVES; ++w) m = min(m, sm[w]);
    if (m == BWD_SENT) break;  // workgroup-uniform
    float4 acc = tzr_zero4();
    uint32_t cnt = 0;
    const int trips = (n + groups - 1) / groups;
    for (int j = 0; j < trips; ++j) {
      const int i = j * groups + q;
      const bool valid = lane_on && i < n;
      const uint2 e = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)(i < n ? i : n - 1));
      if (valid && e.x == m) {
        acc = tzr_add4(acc, bwd_lookup_grad(feats, tb, V.feat_by_order, sG, one, single, grad_mode, nullptr, weights, nullptr, B, 1,
                                            e.y, c));
        cnt += 1;
      }
    }
    if (lane_on) {
      float* r = red + (size_t)(q * lg + c) * 4;
      r[0] = acc.x; r[1] = acc.y; r[2] = acc.z; r[3] = acc.w;
    }
    if (lane_on && c == 0) sm[BWD_WAVES + q] = cnt;
    __syncthreads();
    if (wv == 0) {
      float4 tot = tzr_zero4();
      uint32_t ct = 0;
      for (int g = 0; g < groups; ++g) {
        if (lane < lg) {
          const float* r = red + (size_t)(g * lg + lane) * 4;
          tot = tzr_add4(tot, make_float4(r[0], r[1], r[2], r[3]));
        }
        ct += sm[BWD_WAVES + g];
      }
      if (u.split > 0) {
        rowsum = tot;
        rowcnt = ct;
      } else {
        bwd_apply_row_wave<ADAM>(tb, opt, lr, m, tot, lane);
      }
    }
    if (m == 0xFFFFFFFEu) break;
    cur = m + 1u;
  }
  if (u.split > 0 && wv == 0) bwd_cells_combine<ADAM>(u, V, opt, lr, max_dim, rowsum, rowcnt, lane);
}

template <bool ADAM, int FK, int NT>
__device__ __forceinline__ void bwd_cells_apply_body(
    BwdCellsView V, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B, int grad_mode,
    const BwdGrads& G, const BwdOpt& opt, int max_dim, const uint2* __restrict__ slab, const uint16_t* __restrict__ lstart, int ch) {
  __shared__ BwdCellsLds L;
  __shared__ TzrDst sG[TZR_MAX_DST];
  const BwdCellUnit u = V.units[blockIdx.x];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int ncell = u.c1 - u.c0;
  // the table's first key: read together with the cells' bounds (no round trip of its own)
  const int ft_dst = feats[u.feat].n_dst;
  const int cfirst_rel = 0;
  (void)cfirst_rel;
  // ---- the cells' bounds: two 2-byte loads per cell, all independent ----
  uint32_t* const cpre = L.S.pk;          // [ncell + 1] exclusive counts (dead before the sort's exchange buffer is written)
  uint32_t* const cbase = L.S.ps;         // [ncell] slab position of the cell's first lookup
  if ((int)threadIdx.x < ncell) {
    const int cc = u.c0 + (int)threadIdx.x;
    const uint16_t* lrow = lstart + (size_t)cc * BWD_CELLS_LROW;
    const uint32_t a = lrow[u.b0], b = lrow[u.b1];
    const BwdCellChunk* cdp = V.chunks + cc;
    const int64_t cs = cdp->s;  // (independent of the two loads above)
    cpre[threadIdx.x] = b - a;
    cbase[threadIdx.x] = (uint32_t)cs + a;
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < TZR_MAX_DST; ++i) sG[i] = G.d[i];  // static indices: straight from kernarg
  }
  __syncthreads();
  bwd_block_scan(cpre, ncell, L.S.wtot);
  const int n = (int)cpre[ncell];
  const float lr = *opt.lr;
  const TzrTable& tb = u.tb;
  const int lg = tb.dim >> 2;
  if (n == 0) {
    if (u.split > 0 && wv == 0) bwd_cells_combine<ADAM>(u, V, opt, lr, max_dim, tzr_zero4(), 0u, lane);
    return;
  }
  if (n > BWD_UMAX) {  // skewed ids
    if (threadIdx.x == 0) atomicAdd(V.overflow, 1u);
    bwd_cells_slow_unit<ADAM>(u, V, slab, cpre, cbase, ncell, n, feats, weights, B, grad_mode, opt, max_dim, sG,
                              reinterpret_cast<float*>(&L.S.L), L.S.smm - 0 + 0 == nullptr ? nullptr : L.S.gstart);
    return;
  }
  // ---- gather + sort by (row id, position) in LDS: bwd_stage_unit's fused form with the cells as the source ----
  constexpr int kRounds = BWD_UMAX / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds], dest[kRounds];
  uint32_t vmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    const bool in = r < rounds && lp < n;
    const uint2 v = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)(lp < n ? lp : n - 1));  // (clamped, unconditional)
    kreg[r] = in ? v.x : 0u;
    sreg[r] = in ? v.y : 0u;
    if (in) {
      vmask |= 1u << r;
      kmin = min(kmin, v.x);
      kmax = max(kmax, v.x);
    }
  }
  for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, m, TZR_WAVE));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, m, TZR_WAVE));
  }
  if (lane == 0) {
    L.S.smm[wv] = kmin;
    L.S.smm[BWD_WAVES + wv] = kmax;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    kmin = min(kmin, L.S.smm[w]);
    kmax = max(kmax, L.S.smm[BWD_WAVES + w]);
  }
  __syncthreads();  // smm is reused by the core
  bwd_sort_core<kRounds>(kreg, sreg, vmask, pw, rounds, kmin, max(1, bwd_bits(kmax - kmin)), true, L.S, dest);
  __syncthreads();  // the sort's LDS is dead: the unit's arrays take its place
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if ((vmask >> r) & 1u) {
      L.U.sK[dest[r] + 1] = kreg[r];
      L.U.sS[dest[r]] = sreg[r];
    }
  if (threadIdx.x == 0) {
    // a unit owns its rows -- no run continues outside it -- except one slice of a split row: open on both sides, so that
    // the reduction leaves the slice's whole sum as its leading piece and updates nothing
    const uint32_t edge = u.split > 0 ? (uint32_t)u.b0 : BWD_SENT;
    L.U.sK[0] = edge;
    L.U.sK[n + 1] = edge;
  }
  __syncthreads();
  auto tail = [&](unsigned cf, uint32_t okey, const float4& clead, const float4& osum) {  // wave 0
    (void)cf;
    (void)okey;
    (void)osum;
    if (u.split > 0) bwd_cells_combine<ADAM>(u, V, opt, lr, max_dim, clead, (uint32_t)n, lane);
  };
  if constexpr (FK != 0) {
    const bool fast = tb.w_dtype == TZR_DT_F32 && (grad_mode == 1 || (tb.n_feats == 1 && ft_dst == 1));  // (workgroup-uniform)
    if (fast)
      bwd_reduce_unit<false, NT, FK>(tb, feats, V.feat_by_order, nullptr, nullptr, weights, B, 1, grad_mode, opt, L.U, sG, n, tail);
    else
      bwd_reduce_unit<false, 1, 0>(tb, feats, V.feat_by_order, nullptr, nullptr, weights, B, 1, grad_mode, opt, L.U, sG, n, tail);
  } else {
    bwd_reduce_unit<ADAM, 1, 0>(tb, feats, V.feat_by_order, nullptr, nullptr, weights, B, 1, grad_mode, opt, L.U, sG, n, tail);
  }
  (void)ch;
  (void)lg;
}

#define TZR_CELLS_APPLY_KERNEL(NAME, ADAM_, FK_, ATTR)                                                                         \
  __global__ __launch_bounds__(BWD_THREADS) ATTR void NAME(BwdCellsView V, const TzrFeature* __restrict__ feats,               \
                                                          const float* __restrict__ weights, int64_t B, int grad_mode,         \
                                                          BwdGrads G, BwdOpt opt, int max_dim, const uint2* __restrict__ slab, \
                                                          const uint16_t* __restrict__ lstart, int ch) {                       \
    bwd_cells_apply_body<ADAM_, FK_, 1>(V, feats, weights, B, grad_mode, G, opt, max_dim, slab, lstart, ch);                   \
  }
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_adagrad_kernel, false, TZR_OPT_ADAGRAD, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_rowwise_kernel, false, TZR_OPT_ROWWISE_ADAGRAD, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_sgd_kernel, false, TZR_OPT_SGD, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_general_kernel, false, 0, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_adam_kernel, true, 0, )

// ------------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------------
static int cells_common(const void* h_geo, void* d_geo, void* ws, size_t ws_bytes, int64_t n_values, int n_feats, int n_tables,
                        int max_dim, BwdCellsGeo* g, BwdCellsView* V, BwdPlan* P) {
  if (!h_geo || !d_geo || (reinterpret_cast<uintptr_t>(d_geo) & 255)) return TZR_ERR_INVALID;
  std::memcpy(g, h_geo, sizeof(BwdCellsGeo));
  if (g->n_feats != n_feats || g->max_dim != max_dim || g->n_chunks <= 0 || g->n_units <= 0) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  if (bwd_layout(P, ws, n_values, g->n_positions, n_feats, n_tables, max_dim) > ws_bytes) return TZR_ERR_WORKSPACE;
  if (P->ch != (int)g->ch || g->n_chunks > P->max_chunks) return TZR_ERR_INVALID;  // (a geometry built for another batch size)
  static_assert(BWD_CELLS_LROW * 2 <= BWD_NB * 4, "the lstart rows live in the plan's histogram area");
  *V = bwd_cells_view(d_geo, *g);
  return TZR_OK;
}

// The cells plan of one batch of ONE id per bag.  `h_geo` / `d_geo`: the geometry image tzr_bwd_cells_geometry made for these
// tables and this B, on the host and (a copy the caller uploaded, 256-byte aligned) on the device.  `ws`: the workspace of
// tzr_pooled_bwd_workspace (the same call sizes it for either plan).  One launch, asynchronous on `stream`.
extern "C" int tzr_pooled_bwd_cells_plan(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats, int n_feats,
                                         int max_dim, const int64_t* d_values, int64_t n_values, int64_t B, const void* h_geo,
                                         void* d_geo, void* ws, size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B <= 0) return TZR_ERR_INVALID;
  BwdCellsGeo g;
  BwdCellsView V;
  BwdPlan P;
  const int rc = cells_common(h_geo, d_geo, ws, ws_bytes, n_values, n_feats, n_tables, max_dim, &g, &V, &P);
  if (rc != TZR_OK) return rc;
  if (!d_values) return TZR_ERR_INVALID;
  if (n_values >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = nullptr;
  A.B = B;
  A.uniform = 1;
  hipLaunchKernelGGL(tzr_bwd_cells_partition_kernel, dim3((unsigned)g.n_chunks), dim3(BWD_THREADS), 0, static_cast<hipStream_t>(stream), V,
                     d_tables, A, P.ks[1], reinterpret_cast<uint16_t*>(P.hist));
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// The apply of a cells plan: same meaning of grad_mode / h_grads / h_optim as tzr_pooled_bwd_apply, same sums in the same order.
extern "C" int tzr_pooled_bwd_cells_apply(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats, int n_tables, int max_dim,
                                          const float* d_weights, int64_t n_values, int64_t B, int grad_mode, const TzrDst* h_grads,
                                          int n_dst, const TzrSparseOptim* h_optim, const void* h_geo, void* d_geo, void* ws,
                                          size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || !h_grads || !h_optim || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B <= 0 || n_dst <= 0 ||
      n_dst > TZR_MAX_DST || max_dim <= 0 || max_dim > BWD_MAXDIM || (max_dim & 3) || (grad_mode != 0 && grad_mode != 1))
    return TZR_ERR_INVALID;
  if (!h_optim->d_lr) return TZR_ERR_INVALID;
  if (h_optim->kind != TZR_OPT_SGD && h_optim->kind != TZR_OPT_ADAGRAD && h_optim->kind != TZR_OPT_ROWWISE_ADAGRAD &&
      h_optim->kind != TZR_OPT_ACCUMULATE && h_optim->kind != TZR_OPT_ADAM)
    return TZR_ERR_UNSUPPORTED;
  if (h_optim->kind == TZR_OPT_ADAM && !h_optim->d_adam) return TZR_ERR_INVALID;
  BwdCellsGeo g;
  BwdCellsView V;
  BwdPlan P;
  const int rc = cells_common(h_geo, d_geo, ws, ws_bytes, n_values, n_feats, n_tables, max_dim, &g, &V, &P);
  if (rc != TZR_OK) return rc;
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
  opt.beta1 = h_optim->beta1;
  opt.beta2 = h_optim->beta2;
  opt.adam = reinterpret_cast<const float*>(h_optim->d_adam);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define TZR_CELLS_LAUNCH(K)                                                                                               \
  hipLaunchKernelGGL(K, dim3((unsigned)g.n_units), dim3(BWD_THREADS), 0, s, V, d_feats, d_weights, B, grad_mode, G, opt, max_dim, \
                     (const uint2*)P.ks[1], (const uint16_t*)reinterpret_cast<uint16_t*>(P.hist), (int)g.ch)
  const bool fast_shape = !d_weights && (opt.kind == TZR_OPT_ADAGRAD || opt.kind == TZR_OPT_ROWWISE_ADAGRAD || opt.kind == TZR_OPT_SGD);
  if (opt.kind == TZR_OPT_ADAM) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_adam_kernel);
  } else if (!fast_shape) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_general_kernel);
  } else if (opt.kind == TZR_OPT_ADAGRAD) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_adagrad_kernel);
  } else if (opt.kind == TZR_OPT_ROWWISE_ADAGRAD) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_rowwise_kernel);
  } else {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_sgd_kernel);
  }
#undef TZR_CELLS_LAUNCH
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
