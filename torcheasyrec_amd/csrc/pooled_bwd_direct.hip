// K6 + K7 in ONE launch for small batches: the fused backward + sparse optimizer without an index plan.
//
// Replaces, like pooled_bwd.hip + pooled_bwd_apply.hip, fbgemm's transpose_embedding_input + split_embedding_backward_*_exact
// (autograd of self.ebc(kjt), /root/reference/tzrec/modules/embedding.py:930, optimizer fused by apply_optimizer_in_backward,
// /root/reference/tzrec/main.py:774-781) -- for the regime the planned path is worst at: a rank's share of the batch
// (8 192 samples per rank = `batch_size: 8192` of examples/dlrm_criteo.config, 213 k lookups).  There the plan is four
// dependent launches of >= 7 us each that move 1.7 MB, and the apply a fifth (33.8 + 27.8 us on MI355X for 8.5 us of HBM
// time, BENCH_r03 secondary.config2_batch8192).  At that size a table's ids are 64 KB: they fit an XCD's L2 many times
// over, so nothing has to be PARTITIONED through HBM at all --
//
//   workgroup (t, j) of table t's k = ceil(lookups / ch) workgroups OWNS the j-th of k equal ranges of the table's rows.
//   It reads ALL ids of the table (coalesced, L2-resident after the first reader), keeps the lookups of its own rows in
//   LDS in position order (ballot compaction), sorts them by row id in LDS (pooled_bwd_sort.h: one LDS atomic per lookup
//   + in-group ranking) and reduces + applies them exactly like a unit of the planned apply (pooled_bwd_apply.h:
//   wave-level segmented sums, ONE read-modify-write per row).
//
// Row ranges are disjoint, so no row is touched by two workgroups: no atomics on rows, no cross-workgroup records for them
// -- and the result is a function of the ids alone (bit-reproducible run to run).
// Redundant reading is the price: every workgroup of a table reads the table's whole id list, k * n_t ids per table,
// 54 MB of L2 reads at 8 192 per rank, 218 MB at 16 384 -- quadratic, which is why this is the SMALL-batch path
// (tzr_pooled_bwd_direct_supported; the launcher of torcheasyrec_amd/embedding.py takes it up to 16 384 lookups per
// table).
//
// Tiny tables (fewer rows than half their workgroups: the 3-row table of Criteo has 2 730 lookups per row at 8 192) are the
// one case where workgroups cooperate: every row is split over k / rows workgroups by POSITION, each sums the gradient
// rows of its slice, and the last of them to arrive (a self-resetting counter in the caller's workspace) adds the partial
// sums in slice order and applies the row.
//
// Skew.  A range that holds more lookups than fit the LDS unit (BWD_UMAX) -- a Zipf head, a default id -- is walked in
// pieces: a 512-bin histogram of the range in LDS tells the largest
// prefix of sub-ranges that fits; that prefix is gathered / sorted / applied, and so on; a sub-range that does not fit
// alone is narrowed the same way until it is ONE row, whose gradient rows are then summed by streaming (every wave a
// stripe of the matching lookups, fixed combination order) and applied once.  Every piece costs one or two more passes
// over the table's ids (L2 hits): correct for any distribution, fast for the ones that matter.
#include <tzr_gfx950.h>

#include "pooled_bwd_apply.h"
#include "pooled_bwd_sort.h"

// A value every lane of the wave holds (read from LDS, or derived from such): moved to a scalar register, so that the
// kernel's many workgroup-uniform quantities (range bounds, counts, table geometry) do not occupy vector registers.
__device__ __forceinline__ uint32_t tzr_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int64_t tzr_uni64(int64_t v) {
  const uint32_t lo = tzr_uni((uint32_t)v), hi = tzr_uni((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ TzrTable tzr_uni_table(const TzrTable& v) {  // a table descriptor read from LDS -> scalar registers
  TzrTable r;
  r.w = (uint64_t)tzr_uni64((int64_t)v.w);
  r.m = (uint64_t)tzr_uni64((int64_t)v.m);
  r.rows = tzr_uni64(v.rows);
  r.dim = (int32_t)tzr_uni((uint32_t)v.dim);
  r.w_stride = (int32_t)tzr_uni((uint32_t)v.w_stride);
  r.m_stride = (int32_t)tzr_uni((uint32_t)v.m_stride);
  r.first_order = (int32_t)tzr_uni((uint32_t)v.first_order);
  r.n_feats = (int32_t)tzr_uni((uint32_t)v.n_feats);
  r.w_dtype = (int32_t)tzr_uni((uint32_t)v.w_dtype);
  return r;
}

#define BWD_DGEO 256  // lookups / tables up to which the direct kernel runs (every workgroup derives the geometry itself)
#define BWD_DR 8      // ids per thread and block of the id walk: 2 048 positions per block, 512 per wave

struct BwdDirectGeo {
  uint32_t fstart[BWD_DGEO + 1];
  int32_t fkey[BWD_DGEO];
  uint32_t tchunk[BWD_DGEO + 1];
  uint32_t wtot[BWD_WAVES];
};

#define BWD_DTAB 64                       // tables whose descriptors every workgroup stashes in LDS with the geometry
#define BWD_WCAP (BWD_UMAX / BWD_WAVES)  // lookups of its range one WAVE's quarter of the table's positions may hold

struct BwdDirectLds {
  BwdDirectGeo G;
  int32_t fbo[BWD_DGEO];  // lookup (index into feats) by order
  TzrTable tabs[BWD_DTAB];
  TzrDst sG[TZR_MAX_DST];
  uint32_t wcnt[BWD_WAVES];
  uint32_t red[BWD_WAVES];
  uint32_t hcnt[BWD_WAVES];  // lookups of the table's hot-row candidate a wave met on its walk
  union {
    BwdSortLds S;  // gather target (S.pk / S.ps, position order), histogram (S.gstart), the sort
    BwdUnitLds U;  // the sorted unit and its reduction
  };
};

// A key segment of a table: table-major positions [s, e) are values[fbase + (p - s)] (bwd_elem0's addressing, resolved
// once per segment into scalars).  Every id walk below goes segment by segment, so that the loads inside are
// BRANCH-FREE: `if (p < end) id = values[...]` per element compiles to branch / load / s_waitcnt vmcnt(0) per element --
// the 32 id loads of a lane ran as 32 dependent L2 round trips, 25 of the kernel's first 55 us (profiles/r04i).  Here the
// position is clamped into the segment, the load is unconditional, and validity is a mask.
struct BwdDSeg {
  int64_t s, e, fbase;
};
__device__ __forceinline__ BwdDSeg bwd_direct_seg(const BwdGeo& G, const BwdSrcArgs& A, int o) {
  BwdDSeg g;
  g.s = (int64_t)tzr_uni(G.fstart[o]);
  g.e = (int64_t)tzr_uni(G.fstart[o + 1]);
  const int64_t key = (int64_t)(int32_t)tzr_uni((uint32_t)G.fkey[o]);
  g.fbase = A.uniform ? key * A.B : A.offsets[key * A.B];
  return g;
}
// row id of position p of segment g, p clamped below `end` (s < end <= e); ids outside the table read as row 0 (K4 reports them)
__device__ __forceinline__ uint32_t bwd_direct_id(const BwdSrcArgs& A, int64_t rows, const BwdDSeg& g, int64_t p, int64_t end) {
  const int64_t pc = p < end ? p : end - 1;
  int64_t id = A.values[g.fbase + (pc - g.s)];
  if ((uint64_t)id >= (uint64_t)rows) id = 0;
  return (uint32_t)id;
}
// for (seg, a, b) over the segments of table `tb` clipped to positions [P0, P1): body
#define BWD_DIRECT_FOR_SEGMENTS(G, A, tb, P0, P1, seg, a, b)                              \
  for (int o_ = (tb).first_order; o_ < (tb).first_order + (tb).n_feats; ++o_)            \
    if (BwdDSeg seg = bwd_direct_seg(G, A, o_); true)                                     \
      if (const int64_t a = max((int64_t)(P0), seg.s), b = min((int64_t)(P1), seg.e); a < b)

// One block of the id walk: R * 256 positions of ONE segment from `base` (up to `end`), wave w owning the w-th
// contiguous quarter.  Lookups whose row id satisfies `pred` are numbered in position order from `offset`;
// emit(index, row id, lookup position).  Returns the number of such lookups in the block (workgroup-uniform).  Two
// barriers.
template <int R, class Pred, class Emit>
__device__ __forceinline__ uint32_t bwd_direct_block(const BwdSrcArgs& A, int64_t rows, const BwdDSeg& g, int64_t base,
                                                     int64_t end, uint32_t* wcnt, uint32_t offset, Pred&& pred,
                                                     Emit&& emit) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = (int)tzr_uni(threadIdx.x / TZR_WAVE);
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t k[R], pos[R];
  uint32_t mm = 0, run = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)  // all loads of the block first: one L2 round trip per block
    k[r] = bwd_direct_id(A, rows, g, base + (int64_t)(wv * R + r) * TZR_WAVE + lane, end);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool m = base + (int64_t)(wv * R + r) * TZR_WAVE + lane < end && pred(k[r]);
    const unsigned long long bal = __ballot(m);
    pos[r] = run + (uint32_t)__popcll(bal & lt);
    run += (uint32_t)__popcll(bal);
    mm |= m ? 1u << r : 0u;
  }
  if (lane == 0) wcnt[wv] = run;
  __syncthreads();
  uint32_t ahead = offset, tot = 0;
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    const uint32_t c = tzr_uni(wcnt[w]);
    if (w < wv) ahead += c;
    tot += c;
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if ((mm >> r) & 1u)
      emit(ahead + pos[r], k[r], (uint32_t)(g.fbase + (base + (int64_t)(wv * R + r) * TZR_WAVE + lane - g.s)));
  __syncthreads();
  return tot;
}

// The lookups of rows [lo, hi) of the table, in position order, into S.pk / S.ps (the first BWD_UMAX of them); returns
// how many there are.
__device__ __forceinline__ uint32_t bwd_direct_gather(const BwdGeo& G, const TzrTable& tb, const BwdSrcArgs& A,
                                                      int64_t ts, int64_t te, uint32_t lo, uint32_t hi,
                                                      BwdDirectLds& L, bool has_x = false, uint32_t xrow = 0u) {
  uint32_t total = 0;
  BWD_DIRECT_FOR_SEGMENTS(G, A, tb, ts, te, seg, a, b)
    for (int64_t base = a; base < b; base += BWD_DR * BWD_THREADS)
      total += bwd_direct_block<BWD_DR>(
          A, tb.rows, seg, base, b, L.wcnt, total, [&](uint32_t key) { return key >= lo && key < hi && !(has_x && key == xrow); },
          [&](uint32_t at, uint32_t key, uint32_t src) {
            if (at < (uint32_t)BWD_UMAX) {
              L.S.pk[at] = key;
              L.S.ps[at] = src;
            }
          });
  return total;
}

// The same without a barrier inside the walk: every WAVE takes a contiguous quarter of the table's positions and keeps the
// lookups of rows [lo, hi) it finds there in its own region of S.pk / S.ps (BWD_WCAP entries), sixteen id loads in flight
// per lane.  Position order = (wave, index in the region), which is all the sort needs.  Returns the number of lookups
// found; *fits = every wave's share fitted its region (else the caller takes the ordered gather above, piece by piece).
// `detect`: the table's hot-row CANDIDATE (see the body) is determined on the way -- the mode of the table's first 16 ids, loaded
// together with the walk's first ids, so no round trip of its own -- and its lookups ANYWHERE in the table are counted
// (*has_c, *crow, *n_c).  `has_x`: row `xrow` is left out of the gather.
#define BWD_DGR 16  // id loads in flight per lane (64-bit each: 32 registers)
__device__ __forceinline__ uint32_t bwd_direct_gather_waves(const BwdGeo& G, const TzrTable& tb, const BwdSrcArgs& A,
                                                            int64_t ts, int64_t te, uint32_t lo, uint32_t hi,
                                                            BwdDirectLds& L, bool* fits, bool detect = false, bool* has_c_out = nullptr,
                                                            uint32_t* crow_out = nullptr, uint32_t* n_c = nullptr, bool has_x = false,
                                                            uint32_t xrow = 0u) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = (int)tzr_uni(threadIdx.x / TZR_WAVE);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t n_t = te - ts;
  const int64_t per = (((n_t + BWD_WAVES - 1) / BWD_WAVES) + TZR_WAVE - 1) & ~(int64_t)(TZR_WAVE - 1);
  const int64_t w0 = min(te, ts + (int64_t)wv * per), w1 = min(te, w0 + per);
  uint32_t* const rk = L.S.pk + wv * BWD_WCAP;
  uint32_t* const rs = L.S.ps + wv * BWD_WCAP;
  uint32_t cnt = 0, hc = 0;
  bool has_c = false, pending = false;
  uint32_t crow = 0u, sv = 0u;
  int slen = 0;
  if (detect) {  // (workgroup-uniform) the sample: issued here, looked at behind the first batch of id loads below
    const BwdDSeg s0 = bwd_direct_seg(G, A, tb.first_order);
    slen = (int)min((int64_t)16, s0.e - s0.s);
    if (slen > 0) {
      sv = bwd_direct_id(A, tb.rows, s0, s0.s + (lane & 15), s0.e);
      pending = true;
    }
  }
  BWD_DIRECT_FOR_SEGMENTS(G, A, tb, w0, w1, seg, sa, sb)
    for (int64_t base = sa; base < sb; base += (int64_t)BWD_DGR * TZR_WAVE) {
      uint32_t k[BWD_DGR];
#pragma unroll
      for (int r = 0; r < BWD_DGR; ++r) k[r] = bwd_direct_id(A, tb.rows, seg, base + (int64_t)r * TZR_WAVE + lane, sb);
      if (pending) {  // every wave of every workgroup of the table looks at the same 16 ids: the same candidate everywhere
        pending = false;
        uint32_t same = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) same += (i < slen && (uint32_t)__shfl((int)sv, i, TZR_WAVE) == sv) ? 1u : 0u;
        uint32_t best = (lane < slen) ? ((same << 8) | (uint32_t)(15 - lane)) : 0u;  // most frequent, then the earliest
        for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, m, TZR_WAVE));
        best = tzr_uni(best);
        if ((best >> 8) >= 3u) {
          has_c = true;
          crow = tzr_uni((uint32_t)__shfl((int)sv, 15 - (int)(best & 255u), TZR_WAVE));
        }
      }
#pragma unroll
      for (int r = 0; r < BWD_DGR; ++r) {
        const int64_t p = base + (int64_t)r * TZR_WAVE + lane;
        if (has_c) hc += (uint32_t)__popcll(__ballot(p < sb && k[r] == crow));  // (workgroup-uniform branch)
        const bool m = p < sb && k[r] >= lo && k[r] < hi && !(has_x && k[r] == xrow);
        const unsigned long long bal = __ballot(m);
        if (bal == 0ull) continue;  // wave-uniform: most rounds of a wave hold none of this range's lookups
        const uint32_t at = cnt + (uint32_t)__popcll(bal & lt);
        if (m && at < (uint32_t)BWD_WCAP) {
          rk[at] = k[r];
          rs[at] = (uint32_t)(seg.fbase + (p - seg.s));
        }
        cnt += (uint32_t)__popcll(bal);
      }
    }
  if (lane == 0) {
    L.wcnt[wv] = cnt;
    L.hcnt[wv] = hc;
    if (detect && wv == 0) {  // (a wave whose quarter of the positions is empty never looked: the candidate is wave 0's)
      L.red[0] = has_c ? 1u : 0u;
      L.red[1] = crow;
    }
  }
  __syncthreads();
  uint32_t total = 0, htot = 0;
  bool ok = true;
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    const uint32_t c = tzr_uni(L.wcnt[w]);
    total += c;
    htot += tzr_uni(L.hcnt[w]);
    ok = ok && c <= (uint32_t)BWD_WCAP;
  }
  *fits = ok;
  if (detect) {
    *has_c_out = tzr_uni(L.red[0]) != 0u;
    *crow_out = tzr_uni(L.red[1]);
    *n_c = htot;
  }
  return total;
}

// The unit's lookups in LDS in position order -> sorted by row id -> reduced and applied.  `regions`: they lie in the four
// wave regions of bwd_direct_gather_waves (L.wcnt[w] entries each); else in S.pk / S.ps[0 .. n), n <= BWD_UMAX.
template <bool ADAM, int NT, int FK = 0>
__device__ __forceinline__ void bwd_direct_unit(const TzrTable& tb, const TzrFeature* __restrict__ feats,
                                                const BwdSrcArgs& A, const float* __restrict__ weights, int grad_mode,
                                                const BwdOpt& opt, BwdDirectLds& L, int n, bool regions = false,
                                                bool sort_only = false) {
  if (n <= 0) return;  // workgroup-uniform
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  constexpr int kRounds = BWD_UMAX / BWD_THREADS;
  static_assert(BWD_WCAP == kRounds * TZR_WAVE, "a wave region is what one wave holds in registers");
  const int pw = regions ? BWD_WCAP : bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  const int mine = regions ? (int)tzr_uni(L.wcnt[wv]) : n - wv * pw;  // elements of this wave's span
  uint32_t kreg[kRounds], sreg[kRounds], dest[kRounds];
  uint32_t vmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    kreg[r] = sreg[r] = 0u;
    if (r < rounds && r * TZR_WAVE + lane < mine) {
      vmask |= 1u << r;
      kreg[r] = L.S.pk[lp];
      sreg[r] = L.S.ps[lp];
      kmin = min(kmin, kreg[r]);
      kmax = max(kmax, kreg[r]);
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
  kmin = tzr_uni(kmin);
  kmax = tzr_uni(kmax);
  __syncthreads();  // smm is reused by the core
  bwd_sort_core<kRounds>(kreg, sreg, vmask, pw, rounds, kmin, max(1, bwd_bits(kmax - kmin)), true, L.S, dest);
  __syncthreads();  // the sort's LDS is dead: the unit's arrays take its place
  if (sort_only) return;
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if ((vmask >> r) & 1u) {
      L.U.sK[dest[r] + 1] = kreg[r];
      L.U.sS[dest[r]] = sreg[r];
    }
  if (threadIdx.x == 0) {
    L.U.sK[0] = BWD_SENT;  // row ranges are disjoint: no run of this unit continues anywhere else
    L.U.sK[n + 1] = BWD_SENT;
  }
  __syncthreads();
  auto none = [](unsigned, uint32_t, const float4&, const float4&) {};
  if constexpr (FK != 0) {
    // fp32 table read by one key with one gradient buffer, pooled gradients, no per-sample weights: the fast memory side of
    // the tile loop (pooled_bwd_apply.h: bwd_apply_row_fast); every other unit of the launch the general one
    const TzrFeature* const ft = feats + L.fbo[tb.first_order];
    if (tb.w_dtype == TZR_DT_F32 && !weights && (grad_mode == 1 || (tb.n_feats == 1 && ft->n_dst == 1)))
      bwd_reduce_unit<ADAM, NT, FK>(tb, feats, L.fbo, nullptr, A.offsets, weights, A.B, A.uniform, grad_mode, opt, L.U, L.sG, n, none);
    else
      bwd_reduce_unit<ADAM, NT>(tb, feats, L.fbo, nullptr, A.offsets, weights, A.B, A.uniform, grad_mode, opt, L.U, L.sG, n, none);
  } else {
    bwd_reduce_unit<ADAM, NT>(tb, feats, L.fbo, nullptr, A.offsets, weights, A.B, A.uniform, grad_mode, opt, L.U, L.sG, n, none);
  }
  __syncthreads();  // wave 0's stitch reads U while the others would already refill S
}

// The gradient rows of ONE row's lookups among positions [s0, s1), summed by streaming.  The matching lookups of a
// 1 024-position block are listed in LDS, every wave takes a contiguous quarter of the list, every lane group a stripe of
// that quarter (4 gathers in flight per lane); stripes, waves and blocks are combined in a fixed order.  The sum is
// returned in wave 0 (lane l < D/4 holds floats 4l .. 4l+3); all threads of the workgroup call.
__device__ __forceinline__ float4 bwd_direct_row_sum(const BwdGeo& G, const TzrTable& tb,
                                                     const TzrFeature* __restrict__ feats, const BwdSrcArgs& A,
                                                     const float* __restrict__ weights, int grad_mode, int64_t s0,
                                                     int64_t s1, uint32_t row, BwdDirectLds& L) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = (int)tzr_uni(threadIdx.x / TZR_WAVE);
  const int lg = tb.dim >> 2;
  const int gw = TZR_WAVE / lg;
  const int gi = lane / lg;
  const int c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const bool single = tb.n_feats == 1;
  const BwdSrc one = bwd_resolve(feats + L.fbo[tb.first_order], L.sG);
  float4 acc = tzr_zero4();
  constexpr int SR = 4;  // 1 024 positions per block: at most 1 024 <= BWD_UMAX matches
  BWD_DIRECT_FOR_SEGMENTS(G, A, tb, s0, s1, seg, sa, sb)
  for (int64_t base = sa; base < sb; base += SR * BWD_THREADS) {
    const uint32_t m = bwd_direct_block<SR>(
        A, tb.rows, seg, base, sb, L.wcnt, 0u, [&](uint32_t key) { return key == row; },
        [&](uint32_t at, uint32_t, uint32_t src) { L.S.ps[at] = src; });
    const int q = ((int)m + BWD_WAVES - 1) / BWD_WAVES;
    const int r0 = min((int)m, wv * q), r1 = min((int)m, r0 + q);
    for (int t0 = r0; t0 < r1; t0 += 4 * gw) {
      float4 g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = t0 + u * gw + gi;
        g[u] = tzr_zero4();
        if (lane_on && idx < r1)
          g[u] = bwd_lookup_grad(feats, tb, L.fbo, L.sG, one, single, grad_mode, A.offsets, weights, nullptr, A.B,
                                 A.uniform, L.S.ps[idx], c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = tzr_add4(acc, g[u]);
    }
    __syncthreads();  // the list is rewritten by the next block
  }
  // stripes -> the wave's first lane group (fixed tree), waves in order
  for (int d = 1; d < gw; d <<= 1) {
    const float4 o = make_float4(__shfl_down(acc.x, d * lg, 64), __shfl_down(acc.y, d * lg, 64),
                                 __shfl_down(acc.z, d * lg, 64), __shfl_down(acc.w, d * lg, 64));
    if ((gi & (2 * d - 1)) == 0 && gi + d < gw) acc = tzr_add4(acc, o);
  }
  float* const part = &L.U.rlead[0][0];  // (S / U are idle here)
  if (lane < lg) {
    part[wv * BWD_MAXDIM + 4 * lane + 0] = acc.x;
    part[wv * BWD_MAXDIM + 4 * lane + 1] = acc.y;
    part[wv * BWD_MAXDIM + 4 * lane + 2] = acc.z;
    part[wv * BWD_MAXDIM + 4 * lane + 3] = acc.w;
  }
  __syncthreads();
  float4 sum = tzr_zero4();
  if (wv == 0 && lane < lg)
    for (int w = 0; w < BWD_WAVES; ++w)
      sum = tzr_add4(sum, make_float4(part[w * BWD_MAXDIM + 4 * lane], part[w * BWD_MAXDIM + 4 * lane + 1],
                                      part[w * BWD_MAXDIM + 4 * lane + 2], part[w * BWD_MAXDIM + 4 * lane + 3]));
  __syncthreads();
  return sum;
}

// ONE row with more lookups than an LDS unit holds, owned by this workgroup alone: summed over the whole table, applied.
template <bool ADAM>
__device__ __forceinline__ void bwd_direct_stream_row(const BwdGeo& G, const TzrTable& tb,
                                                      const TzrFeature* __restrict__ feats, const BwdSrcArgs& A,
                                                      const float* __restrict__ weights, int grad_mode,
                                                      const BwdOpt& opt, int64_t ts, int64_t te, uint32_t row,
                                                      BwdDirectLds& L) {
  const float4 sum = bwd_direct_row_sum(G, tb, feats, A, weights, grad_mode, ts, te, row, L);
  if (threadIdx.x < TZR_WAVE)  // (the row update is wave-collective: all 64 lanes of wave 0)
    bwd_apply_row_wave<ADAM>(tb, opt, *opt.lr, row, sum, (int)threadIdx.x);
  __syncthreads();
}

// first row id of sub-range p of [cur, cur + span) cut into sub-ranges by sub(k) = ((k - cur) * m2) >> 32
__device__ __forceinline__ uint32_t bwd_direct_sub_start(uint32_t cur, uint64_t m2, uint32_t p) {
  return cur + (uint32_t)((((uint64_t)p << 32) + m2 - 1) / m2);
}

template <bool ADAM, int NT, int FK = 0>
__device__ __forceinline__ void bwd_direct_body(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats, int F, const BwdSrcArgs& A,
    const float* __restrict__ weights, int grad_mode, const BwdGrads& Gr, const BwdOpt& opt, int ch,
    uint32_t* __restrict__ wcount, float* __restrict__ wpart, int max_dim) {
  __shared__ BwdDirectLds L;
  const int dbg = (ch >> 16) & 0xFF;  // tzr_tune("bwd_direct_debug"): stop behind 1 = geometry, 2 = id walk, 3 = sort (timing experiments)
  const bool no_hot = (ch >> 24) & 1;  // no hot-row candidate: every hot row is streamed by its range's workgroup (the launcher: TZR_GRAD_HOT_ROWS / tzr_tune "bwd_direct_hot")
  ch &= 0xFFFF;
  // geometry (bwd_geometry of pooled_bwd_sort.h, with every global load of it -- lookups, their key lengths, table
  // descriptors -- issued before the first barrier: one memory round trip instead of two + the table fetch behind them)
  {
    TzrFeature ft;
    int64_t n = 0;
    const bool hf = (int)threadIdx.x < F, ht = (int)threadIdx.x < T && T <= BWD_DTAB;
    TzrTable tt;
    if (hf) {
      ft = feats[threadIdx.x];
      const int64_t key = ft.key;
      n = ft.table < 0 ? 0 : (A.uniform ? A.B : A.offsets[(key + 1) * A.B] - A.offsets[key * A.B]);
    }
    if (ht) tt = tables[threadIdx.x];
    if (hf) {
      L.G.fstart[ft.order] = (uint32_t)n;
      L.G.fkey[ft.order] = ft.key;
      L.fbo[ft.order] = (int)threadIdx.x;
    }
    if (ht) L.tabs[threadIdx.x] = tt;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < TZR_MAX_DST; ++i) L.sG[i] = Gr.d[i];  // static indices: straight from kernarg
    }
    __syncthreads();
    bwd_block_scan(L.G.fstart, F, L.G.wtot);
    for (int t2 = threadIdx.x; t2 < T; t2 += BWD_THREADS) {
      const int fo = T <= BWD_DTAB ? L.tabs[t2].first_order : tables[t2].first_order;
      const int nf = T <= BWD_DTAB ? L.tabs[t2].n_feats : tables[t2].n_feats;
      const uint32_t s2 = nf > 0 ? L.G.fstart[fo] : 0u;
      const uint32_t e2 = nf > 0 ? L.G.fstart[fo + nf] : 0u;
      L.G.tchunk[t2] = (e2 - s2 + (uint32_t)ch - 1) / (uint32_t)ch;
    }
    __syncthreads();
    bwd_block_scan(L.G.tchunk, T, L.G.wtot);
  }
  const int cidx = blockIdx.x;
  if (cidx >= (int)tzr_uni(L.G.tchunk[T]) || dbg == 1) return;
  int t = 0;
  {
    int lo = 0, hi = T;  // last t with tchunk[t] <= cidx (the non-empty table holding it)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)tzr_uni(L.G.tchunk[mid]) <= cidx) lo = mid; else hi = mid;
    }
    t = lo;
  }
  const TzrTable tb = T <= BWD_DTAB ? tzr_uni_table(L.tabs[t]) : tables[t];
  if (tb.n_feats <= 0 || tb.rows <= 0) return;
  const int64_t ts = tzr_uni(L.G.fstart[tb.first_order]), te = tzr_uni(L.G.fstart[tb.first_order + tb.n_feats]);
  const uint32_t k = tzr_uni(L.G.tchunk[t + 1]) - tzr_uni(L.G.tchunk[t]);  // workgroups of this table = ranges of its rows
  const uint32_t j = (uint32_t)cidx - tzr_uni(L.G.tchunk[t]);
  BwdGeo G;
  G.fstart = L.G.fstart;
  G.fkey = L.G.fkey;
  G.tchunk = reinterpret_cast<const int32_t*>(L.G.tchunk);
  const uint64_t rows = (uint64_t)tb.rows;
  if (dbg == 5 && rows * 2 > (uint64_t)k) return;  // (timing experiments: 4 = no row-split workgroups, 5 = only them)
  if (rows * 2 <= (uint64_t)k) {
    if (dbg == 4) return;
    // ---- a tiny table (fewer rows than half its workgroups: 3 rows x 2 730 lookups each at 8 192 per rank): every row
    // is SPLIT over P = k / rows workgroups by position; each sums its slice's gradient rows, the last of a row's P
    // workgroups to arrive adds the P partial sums in slice order (write-through records, agent-scope counter:
    // tzr_gfx950.h) and applies the row once.  Whoever arrives last, the order of the additions is the same.
    const uint32_t R = (uint32_t)rows, P = k / R;
    if (j >= R * P) return;
    const uint32_t row = j % R, part = j / R;
    const int64_t n_t = te - ts;
    const int64_t s0 = ts + n_t * part / P, s1 = ts + n_t * (part + 1) / P;
    const float4 sum = bwd_direct_row_sum(G, tb, feats, A, weights, grad_mode, s0, s1, row, L);
    if (threadIdx.x >= TZR_WAVE) return;
    const int lane = (int)threadIdx.x;
    const int lg = tb.dim >> 2;
    const uint32_t c0 = tzr_uni(L.G.tchunk[t]);
    if (lane < lg) bwd_publish4(wpart + (size_t)cidx * max_dim + 4 * lane, sum);
    tzr_drain_stores();
    int last = 0;
    if (lane == 0) last = tzr_arrive(wcount + c0 + row) == P - 1 ? 1 : 0;
    last = __shfl(last, 0, TZR_WAVE);
    if (!last) return;
    if (lane == 0) tzr_publish_u32(wcount + c0 + row, 0u);  // the counters are zero again when the launch ends
    float4 tot = tzr_zero4();
    for (uint32_t q = 0; q < P; ++q)
      if (lane < lg) tot = tzr_add4(tot, bwd_consume4(wpart + (size_t)(c0 + q * R + row) * max_dim + 4 * lane));
    bwd_apply_row_wave<ADAM>(tb, opt, *opt.lr, row, tot, lane);
    return;
  }
  // range j of k: rows [(j << 32) / mult, ((j + 1) << 32) / mult), bucket(row) = (row * mult) >> 32 (bwd_bucket_params)
  const bool row_per_wg = rows <= (uint64_t)k;
  const uint64_t mult = row_per_wg ? (1ull << 32) : (((uint64_t)k << 32) / rows);
  // (a workgroup without rows of its own -- more workgroups than rows, or nothing left by rounding -- still walks the ids
  // when a hot row is looked for: ALL k workgroups of the table take a slice of the hot row's positions and arrive below)
  const bool detect = te - ts > (int64_t)BWD_UMAX && !no_hot;  // (workgroup-uniform; fewer lookups always fit)
  uint64_t lo64 = (((uint64_t)j << 32) + mult - 1) / mult;
  uint64_t hi64 = (((uint64_t)(j + 1) << 32) + mult - 1) / mult;
  if (hi64 > rows || j + 1 == k) hi64 = rows;  // the last range takes what rounding left
  if ((row_per_wg && (uint64_t)j >= rows) || lo64 >= hi64) {
    if (!detect) return;
    lo64 = hi64 = 0;  // an empty range: gathers nothing, counts the candidate's lookups like everyone else
  }
  const uint32_t lo = (uint32_t)lo64, hi = (uint32_t)hi64;

  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = (int)tzr_uni(threadIdx.x / TZR_WAVE);
  // ---- hot row.  One row that holds more lookups than an LDS unit -- the shared row of a zero-collision hash's unseen ids
  // (/root/reference/tzrec/features/feature.py:693-736: every id without a slot reads row zch_size - 1), a default id, a Zipf
  // head -- used to be summed by its range's workgroup alone, streaming: 140 us of a 45 us kernel at 8 192 lookups of which
  // 95 % read one row (profiles/r05j).  Every workgroup of the table reads all of the table's ids anyway, so all of them can
  // agree on it without talking: the CANDIDATE is the mode of the table's first 16 ids (a function of the ids alone, the same
  // in every workgroup; three of the sixteen must agree), its lookups are counted on the walk below, and when they exceed a unit
  // every workgroup sums the candidate's gradient rows over ITS slice of the table's positions; the last to arrive adds
  // the partial sums in slice order and applies the row (the tiny tables' mechanism: same counter, same records).  The
  // candidate's range workgroup leaves the row out of its own range.  A miss (no candidate, or a hot row that is not the
  // mode of the sample) costs nothing but the old path.  Looking costs ~2 us of a 43 us launch on ids without a hot row
  // (profiles/r05ac), so it is the CALLER's statement (grad_mode | TZR_GRAD_HOT_ROWS) that such rows are expected.
  bool fits, has_c = false;
  uint32_t n_c = 0, crow = 0u;
  uint32_t total = bwd_direct_gather_waves(G, tb, A, ts, te, lo, hi, L, &fits, detect, &has_c, &crow, &n_c);
  if (dbg == 2) return;
  const bool hot = has_c && n_c > (uint32_t)BWD_UMAX;  // the same in every workgroup of the table
  const bool has_x = hot && crow >= lo && crow < hi;   // this workgroup's range holds the hot row: gathered again without it
  if (has_x) {
    __syncthreads();
    total = bwd_direct_gather_waves(G, tb, A, ts, te, lo, hi, L, &fits, false, nullptr, nullptr, nullptr, true, crow);
  }
  if (total != 0 && fits) {
    bwd_direct_unit<ADAM, NT, FK>(tb, feats, A, weights, grad_mode, opt, L, (int)total, true, dbg == 3);
  } else if (total != 0) {
    __syncthreads();  // (S is reused by the walk below)
    // ---- more lookups than one LDS unit: piece by piece ----
    uint32_t cur = lo;
    while (cur < hi) {
      uint32_t lim = hi;
      for (;;) {
        const uint32_t span = lim - cur;
        const uint64_t m2 = span <= (uint32_t)BWD_NB ? (1ull << 32) : (((uint64_t)BWD_NB << 32) / span);
        const int sbits = bwd_bits(min(span, (uint32_t)BWD_NB) - 1u);
        for (int i = threadIdx.x; i <= BWD_NB; i += BWD_THREADS) L.S.gstart[i] = 0;
        __syncthreads();
        BWD_DIRECT_FOR_SEGMENTS(G, A, tb, ts, te, seg, sa, sb)
          for (int64_t base = sa; base < sb; base += 4 * BWD_THREADS) {
            uint32_t kk[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) kk[r] = bwd_direct_id(A, tb.rows, seg, base + r * BWD_THREADS + threadIdx.x, sb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (base + r * BWD_THREADS >= sb) break;  // wave-uniform
              const bool v = base + r * BWD_THREADS + (int64_t)threadIdx.x < sb && kk[r] >= cur && kk[r] < lim &&
                             !(has_x && kk[r] == crow);
              const uint32_t sub = v ? (uint32_t)(((uint64_t)(kk[r] - cur) * m2) >> 32) : 0u;
              bwd_wave_count(L.S.gstart, sub, v, max(sbits, 1), lane);
            }
          }
        __syncthreads();
        bwd_block_scan(L.S.gstart, BWD_NB, L.S.wtot);  // exclusive starts; gstart[BWD_NB] = lookups in [cur, lim)
        const uint32_t in_range = tzr_uni(L.S.gstart[BWD_NB]);
        // p = the number of leading sub-ranges that fit one unit together: gstart is monotone, so p = #{i in 1..NB: gstart[i] <= UMAX}
        uint32_t fit = 0;
        for (int i = 1 + (int)threadIdx.x; i <= BWD_NB; i += BWD_THREADS) fit += L.S.gstart[i] <= (uint32_t)BWD_UMAX ? 1u : 0u;
        for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) fit += (uint32_t)__shfl_xor((int)fit, m, TZR_WAVE);
        if (lane == 0) L.red[wv] = fit;
        __syncthreads();
        uint32_t p = 0;
#pragma unroll
        for (int w = 0; w < BWD_WAVES; ++w) p += tzr_uni(L.red[w]);
        __syncthreads();  // gstart / red are rewritten below
        if (in_range == 0) {
          cur = lim;
          break;
        }
        if (p >= 1) {
          const uint32_t end = p >= (uint32_t)BWD_NB ? lim : min(lim, bwd_direct_sub_start(cur, m2, p));
          const uint32_t n = bwd_direct_gather(G, tb, A, ts, te, cur, end, L, has_x, crow);
          bwd_direct_unit<ADAM, NT, FK>(tb, feats, A, weights, grad_mode, opt, L, (int)min(n, (uint32_t)BWD_UMAX));
          cur = end;
          break;
        }
        // the first sub-range alone does not fit
        const uint32_t end0 = min(lim, bwd_direct_sub_start(cur, m2, 1));
        if (end0 - cur <= 1u) {
          bwd_direct_stream_row<ADAM>(G, tb, feats, A, weights, grad_mode, opt, ts, te, cur, L);
          cur += 1;
          break;
        }
        lim = end0;  // narrow
      }
    }
  }
  if (!hot || dbg == 3) return;
  // ---- the hot row: this workgroup's slice of the table's positions, the last of the k arrivals applies ----
  {
    __syncthreads();
    const int64_t n_t = te - ts;
    const int64_t s0 = ts + n_t * (int64_t)j / (int64_t)k, s1 = ts + n_t * (int64_t)(j + 1) / (int64_t)k;
    const float4 sum = bwd_direct_row_sum(G, tb, feats, A, weights, grad_mode, s0, s1, crow, L);
    if (threadIdx.x >= TZR_WAVE) return;
    const int lg = tb.dim >> 2;
    const uint32_t c0 = tzr_uni(L.G.tchunk[t]);
    if (lane < lg) bwd_publish4(wpart + (size_t)cidx * max_dim + 4 * lane, sum);
    tzr_drain_stores();
    int last = 0;
    if (lane == 0) last = tzr_arrive(wcount + c0) == k - 1 ? 1 : 0;
    last = __shfl(last, 0, TZR_WAVE);
    if (!last) return;
    if (lane == 0) tzr_publish_u32(wcount + c0, 0u);  // the counter is zero again when the launch ends
    float4 tot = tzr_zero4();
    for (uint32_t q = 0; q < k; ++q)
      if (lane < lg) tot = tzr_add4(tot, bwd_consume4(wpart + (size_t)(c0 + q) * max_dim + 4 * lane));
    bwd_apply_row_wave<ADAM>(tb, opt, *opt.lr, crow, tot, lane);
  }
}

// (wave priorities by residency slot, tzr_gfx950.h: the four workgroups of a CU out of lock step: 36.0 -> 34.5 us at B = 8 192,
// profiles/r06av; the same in the exact plan's apply: 76.5 -> 79.2 us, in the cells partition: 14.9 -> 15.9 -- not there)
#define BWD_DIRECT_KERNEL(NAME, ADAM_, WAVES, NT_, FK_)                                                                      \
  __global__ __launch_bounds__(BWD_THREADS) TZR_WAVES_PER_EU(WAVES) void NAME(                                        \
      const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats, int F, BwdSrcArgs A,         \
      const float* __restrict__ weights, int grad_mode, BwdGrads Gr, BwdOpt opt, int ch,                             \
      uint32_t* __restrict__ wcount, float* __restrict__ wpart, int max_dim) {                                       \
    tzr_prio_by_slot(blockIdx.x);                                                                                    \
    bwd_direct_body<ADAM_, NT_, FK_>(tables, T, feats, F, A, weights, grad_mode, Gr, opt, ch, wcount, wpart, max_dim);    \
  }
// 4 waves per SIMD (128 VGPRs): 1 024 workgroups resident = the whole grid of an 8 192-per-rank step at once (at 3 waves,
// 768 slots for ~860 workgroups: 66 vs 39 us).  Two tiles of the reduction in flight per wave; four (at 3 waves) or one
// measured the same or worse: the kernel is not bound by the reduction's round trips (profiles/r04j, r04k).
BWD_DIRECT_KERNEL(tzr_bwd_direct_kernel, false, 4, 2, 0)
BWD_DIRECT_KERNEL(tzr_bwd_direct_adam_kernel, true, 3, 2, 0)
// the optimizer kind at compile time: units of fp32 single-key tables take the fast tile loop (tzr_tune "bwd_apply_fast" >= 0)
BWD_DIRECT_KERNEL(tzr_bwd_direct_adagrad_kernel, false, 4, 2, TZR_OPT_ADAGRAD)
BWD_DIRECT_KERNEL(tzr_bwd_direct_rowwise_kernel, false, 4, 2, TZR_OPT_ROWWISE_ADAGRAD)
BWD_DIRECT_KERNEL(tzr_bwd_direct_sgd_kernel, false, 4, 2, TZR_OPT_SGD)
extern int g_tzr_bwd_apply_fast;  // pooled_bwd_apply.hip
int g_tzr_bwd_direct_debug = 0;  // tzr_tune("bwd_direct_debug"): timing experiments, see bwd_direct_body
int g_tzr_bwd_direct_hot = 1;    // tzr_tune("bwd_direct_hot"): hot rows shared among a table's workgroups 1 = when the caller sets TZR_GRAD_HOT_ROWS, 0 = never, 2 = always

int g_tzr_bwd_direct_ch = 0;  // tzr_tune("bwd_direct_ch"): lookups per workgroup (0 = by problem size)
int g_tzr_bwd_direct = 0;     // tzr_tune("bwd_direct"): 0 = up to BWD_DIRECT_MAX lookups per table on average, 1 = whenever the
                              // shape is supported, -1 = never (callers then take the planned pair)
#define BWD_DIRECT_MAX 16384

static inline int bwd_direct_pick_ch(int64_t N) {
  if (g_tzr_bwd_direct_ch >= 64 && g_tzr_bwd_direct_ch <= BWD_CH) return g_tzr_bwd_direct_ch;  // (>= 64: the workspace's sizing)
  return N <= 256 * 1024 ? 256 : 512;
}

static inline int bwd_direct_shape_ok(int64_t n_positions, int n_feats, int n_tables, int uniform_bag_len, int grad_mode) {
  if (n_feats <= 0 || n_tables <= 0 || n_feats > BWD_DGEO || n_tables > BWD_DGEO) return 0;
  if (grad_mode == 0 && uniform_bag_len != 1) return 0;  // ragged pooled bags need the bag of every lookup: the planned path
  if (n_positions < 0 || n_positions >= (1LL << 31)) return 0;
  return 1;
}

// shape the kernel takes AND size the policy wants it for
extern "C" int tzr_pooled_bwd_direct_supported(int64_t n_positions, int n_feats, int n_tables,
                                               int uniform_bag_len, int grad_mode) {
  if (!bwd_direct_shape_ok(n_positions, n_feats, n_tables, uniform_bag_len, grad_mode & ~TZR_GRAD_HOT_ROWS)) return 0;
  if (g_tzr_bwd_direct < 0) return 0;
  if (g_tzr_bwd_direct == 0 && n_positions > (int64_t)BWD_DIRECT_MAX * n_tables) return 0;  // quadratic id reads beyond this
  return 1;
}

// workspace: [max_chunks] arrival counters (ZERO before the first launch; every launch leaves them zero) +
// [max_chunks * max_dim] partial row sums of the row-split workgroups.  Sized for the smallest chunk (64 lookups).
static inline size_t bwd_direct_layout(void* ws, int64_t n_positions, int n_tables, int max_dim, uint32_t** cnt, float** part) {
  TzrCarver c(ws);
  const int64_t mc = bwd_max_chunks(n_positions, n_tables, 64);
  uint32_t* a = c.take<uint32_t>(mc);
  float* b = c.take<float>((size_t)mc * max_dim);
  if (cnt) *cnt = a;
  if (part) *part = b;
  return c.off;
}

extern "C" size_t tzr_pooled_bwd_direct_workspace(int64_t n_positions, int n_tables, int max_dim) {
  if (n_positions < 0 || n_tables <= 0 || max_dim <= 0) return 0;
  return bwd_direct_layout(nullptr, n_positions, n_tables, max_dim, nullptr, nullptr) + 256;
}

extern "C" int tzr_pooled_bwd_direct(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats,
                                     int n_feats, int64_t max_rows, int max_dim, const int64_t* d_values,
                                     const int64_t* d_offsets, const float* d_weights, int64_t n_values,
                                     int64_t n_positions, int64_t B, int uniform_bag_len, int grad_mode,
                                     const TzrDst* h_grads, int n_dst, const TzrSparseOptim* h_optim,
                                     void* ws, size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || !h_grads || !h_optim || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B < 0 ||
      n_dst <= 0 || n_dst > TZR_MAX_DST || max_dim <= 0 || max_dim > BWD_MAXDIM || (max_dim & 3) ||
      max_rows < 0)
    return TZR_ERR_INVALID;
  const int hot_rows = (grad_mode & TZR_GRAD_HOT_ROWS) ? 1 : 0;  // the caller expects rows hotter than an LDS unit (see the body)
  grad_mode &= ~TZR_GRAD_HOT_ROWS;
  if (grad_mode != 0 && grad_mode != 1) return TZR_ERR_INVALID;
  const bool uniform = uniform_bag_len == 1;
  if (!uniform && !d_offsets) return TZR_ERR_INVALID;
  if (!bwd_direct_shape_ok(n_positions, n_feats, n_tables, uniform_bag_len, grad_mode)) return TZR_ERR_UNSUPPORTED;
  if (max_rows > (1LL << 32) || n_values >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;  // row ids and positions travel as 32-bit
  if (!h_optim->d_lr) return TZR_ERR_INVALID;
  if (h_optim->kind != TZR_OPT_SGD && h_optim->kind != TZR_OPT_ADAGRAD && h_optim->kind != TZR_OPT_ROWWISE_ADAGRAD &&
      h_optim->kind != TZR_OPT_ACCUMULATE && h_optim->kind != TZR_OPT_ADAM)
    return TZR_ERR_UNSUPPORTED;
  if (h_optim->kind == TZR_OPT_ADAM && !h_optim->d_adam) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  uint32_t* wcount;
  float* wpart;
  if (bwd_direct_layout(ws, n_positions, n_tables, max_dim, &wcount, &wpart) > ws_bytes) return TZR_ERR_WORKSPACE;
  if (n_values == 0 || n_positions == 0 || B == 0) return TZR_OK;
  if (!d_values) return TZR_ERR_INVALID;
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
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = d_offsets;
  A.B = B;
  A.uniform = (int)uniform;
  const int ch0 = bwd_direct_pick_ch(n_positions);
  const unsigned grid = (unsigned)bwd_max_chunks(n_positions, n_tables, ch0);
  // tzr_tune("bwd_direct_hot"): 1 = the caller's flag decides, 0 = never, 2 = always
  const bool look = g_tzr_bwd_direct_hot == 2 || (g_tzr_bwd_direct_hot == 1 && hot_rows);
  const int ch = ch0 | ((g_tzr_bwd_direct_debug & 0xFF) << 16) | (look ? 0 : 1 << 24);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define BWD_DIRECT_LAUNCH(K)                                                                                              \
  hipLaunchKernelGGL(K, dim3(grid), dim3(BWD_THREADS), 0, s, d_tables, n_tables, d_feats, n_feats, A, d_weights, grad_mode, G, opt, ch, \
                     wcount, wpart, max_dim)
  if (opt.kind == TZR_OPT_ADAM)
    hipLaunchKernelGGL(tzr_bwd_direct_adam_kernel, dim3(grid), dim3(BWD_THREADS), 0, s, d_tables, n_tables, d_feats,
                       n_feats, A, d_weights, grad_mode, G, opt, ch, wcount, wpart, max_dim);
  else if (g_tzr_bwd_apply_fast >= 0 && !d_weights && opt.kind == TZR_OPT_ADAGRAD)
    BWD_DIRECT_LAUNCH(tzr_bwd_direct_adagrad_kernel);
  else if (g_tzr_bwd_apply_fast >= 0 && !d_weights && opt.kind == TZR_OPT_ROWWISE_ADAGRAD)
    BWD_DIRECT_LAUNCH(tzr_bwd_direct_rowwise_kernel);
  else if (g_tzr_bwd_apply_fast >= 0 && !d_weights && opt.kind == TZR_OPT_SGD)
    BWD_DIRECT_LAUNCH(tzr_bwd_direct_sgd_kernel);
  else
    hipLaunchKernelGGL(tzr_bwd_direct_kernel, dim3(grid), dim3(BWD_THREADS), 0, s, d_tables, n_tables, d_feats,
                       n_feats, A, d_weights, grad_mode, G, opt, ch, wcount, wpart, max_dim);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
