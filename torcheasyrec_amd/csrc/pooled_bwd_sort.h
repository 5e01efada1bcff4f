// Device pieces of the backward index plan that more than one translation unit uses (pooled_bwd.hip: the four-launch
// plan; pooled_bwd_direct.hip: the one-launch backward of small batches): the table-major geometry of a KJT, the
// position -> (row id, lookup) map, and the in-LDS sort of one unit's lookups.
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
template <class GeoLds>
__device__ __forceinline__ void bwd_geometry(const TzrTable* __restrict__ tables, int T,
                                             const BwdSrcArgs& A, int F, uint32_t ch, GeoLds& G) {
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
    G.tchunk[t] = (e - s + ch - 1) / ch;
  }
  __syncthreads();
  // tables are visited in first_order order == table-major position order only if table ids
  // follow it; starts are absolute, so the chunk map just needs a prefix in table-id order
  bwd_block_scan(G.tchunk, T, G.wtot);
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

// The same for a table read by ONE key (nearly every table), BRANCH-FREE.  `if (p < end) bwd_elem0(...)` per element
// compiles to branch / load / s_waitcnt vmcnt(0) per element (and, with the geometry in global memory, to a chain of
// three dependent loads per element: key, bag offset, id): the "independent loads in flight" of the hist / scatter
// kernels ran as 4 .. 12 dependent L2 round trips (found with the one-launch backward, profiles/r04i).  Here the key and
// the segment base are resolved once per workgroup, the position is clamped below `end` (> the segment's start), the id
// load is unconditional; the caller masks positions >= end.
struct BwdOneSeg {
  int64_t ts, fbase;
};
__device__ __forceinline__ BwdOneSeg bwd_one_seg(const BwdGeo& G, const TzrTable& tb, const BwdSrcArgs& A) {
  BwdOneSeg g;
  const int64_t key = G.fkey[tb.first_order];
  g.ts = (int64_t)G.fstart[tb.first_order];
  g.fbase = A.uniform ? key * A.B : A.offsets[key * A.B];
  return g;
}
__device__ __forceinline__ void bwd_elem_one(const BwdOneSeg& g, const TzrTable& tb, const BwdSrcArgs& A, int64_t p,
                                             int64_t end, uint32_t* key_out, uint32_t* src_out) {
  const int64_t pc = p < end ? p : end - 1;
  const int64_t i = g.fbase + (pc - g.ts);
  int64_t id = A.values[i];
  if ((uint64_t)id >= (uint64_t)tb.rows) id = 0;  // memory safety; K4 reports/clamps
  *key_out = (uint32_t)id;
  *src_out = (uint32_t)i;
}

// counts[d] += number of valid lanes with digit d, one LDS atomic per distinct digit of the wave
// (a hot row id is every lane's digit: per-lane atomics on one address serialise)
__device__ __forceinline__ void bwd_wave_count(unsigned* counts, uint32_t d, bool v, int wbits,
                                               int lane) {
  unsigned long long peers = __ballot(v);
  for (int bit = 0; bit < wbits; ++bit) {
    const int on = (d >> bit) & 1;
    const unsigned long long bm = __ballot(on);
    peers &= on ? bm : ~bm;
  }
  if (v && (peers & ((1ull << lane) - 1ull)) == 0) atomicAdd(&counts[d], (unsigned)__popcll(peers));
}

#define BWD_GMAX 16  // largest group of equal low digits the in-group ranking takes on

struct BwdSortLds {  // < 20 KB: 8 workgroups per CU, the whole grid of a B = 65536 step resident at once
  BwdRankLds<BWD_NB> L;
  unsigned gstart[BWD_NB + 1];
  union {
    unsigned pre[BWD_NB];  // heavy tile: bucket counts ahead of the tile
    uint16_t hb[BWD_NB];   // unit: lookups of heavy buckets ahead of each bucket
  };
  uint32_t pk[BWD_UMAX], ps[BWD_UMAX];  // exchange buffer of the LDS-resident passes
  uint32_t wtot[BWD_WAVES];
  uint32_t smm[2 * BWD_WAVES];
};
static_assert(BWD_HT <= BWD_UMAX, "the exchange buffer holds a heavy tile");

// Orders the valid elements a workgroup holds wave-contiguously (element r of a lane sits at local
// position wv*pw + r*64 + lane) so that equal row ids end up adjacent and ordered by original lookup
// position.  On return the thread holds (kreg[r], sreg[r]) for the r of `vmask` and dest[r] = the
// element's index in the new order.
//   grouped (only with `grouped_ok`): the elements are counted into 512 groups by the LOW 9 bits
//     of (row id - kmin) with one LDS atomic each -- no match-any ballots, no per-round barriers --
//     and every element then ranks itself inside its group by (row id, lookup position) with a
//     handful of LDS reads (a unit averages 2.3 elements per group).  Order: (low digit, row id,
//     position): equal row ids adjacent, which is all the apply needs, at a fraction of the
//     instructions of three stable passes.  Given up when a group holds more than BWD_GMAX
//     elements (hot rows).  Not for a unit that shares its position range with heavy buckets:
//     that one needs ascending bucket order.
//   else: stable LSD counting passes over the key span: ascending row ids.
template <int MAXR>
__device__ __forceinline__ void bwd_sort_core(uint32_t (&kreg)[MAXR], uint32_t (&sreg)[MAXR],
                                              uint32_t& vmask, int pw, int rounds, uint32_t kmin,
                                              int bits, bool grouped_ok, BwdSortLds& S,
                                              uint32_t (&dest)[MAXR]) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  uint32_t dig[MAXR];
  if (grouped_ok) {
    const unsigned mask0 = BWD_NB - 1;
    for (int i = threadIdx.x; i < BWD_NB; i += BWD_THREADS) S.gstart[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      dig[r] = (kreg[r] - kmin) & mask0;
      dest[r] = 0;
      if ((vmask >> r) & 1u) dest[r] = atomicAdd(&S.gstart[dig[r]], 1u);  // slot inside the group
    }
    __syncthreads();
    static_assert(BWD_NB == 2 * BWD_THREADS, "two group counters per thread");
    // exclusive scan of the 512 group counts (two per thread: one wave scan, one barrier) + the largest group
    const uint32_t c0 = S.gstart[2 * threadIdx.x], c1 = S.gstart[2 * threadIdx.x + 1];
    uint32_t g = max(c0, c1);
    uint32_t incl = c0 + c1;
    for (int dd = 1; dd < TZR_WAVE; dd <<= 1) {
      const uint32_t o = __shfl_up(incl, dd, TZR_WAVE);
      if (lane >= dd) incl += o;
    }
    for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) g = max(g, (uint32_t)__shfl_xor((int)g, m, TZR_WAVE));
    if (lane == TZR_WAVE - 1) S.wtot[wv] = incl;
    if (lane == 0) S.smm[BWD_WAVES + wv] = g;
    __syncthreads();
    uint32_t excl = incl - (c0 + c1);
#pragma unroll
    for (int w = 0; w < BWD_WAVES; ++w) {
      if (w < wv) excl += S.wtot[w];
      g = max(g, S.smm[BWD_WAVES + w]);
    }
    S.gstart[2 * threadIdx.x] = excl;
    S.gstart[2 * threadIdx.x + 1] = excl + c0;
    if (threadIdx.x == BWD_THREADS - 1) S.gstart[BWD_NB] = excl + c0 + c1;
    __syncthreads();
    if (g <= BWD_GMAX) {
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
        if ((vmask >> r) & 1u) {
          const uint32_t at = S.gstart[dig[r]] + dest[r];
          S.pk[at] = kreg[r];
          S.ps[at] = (uint32_t)(wv * pw + r * TZR_WAVE + lane);  // arrival (= table-major) position
        }
      __syncthreads();
      // rank inside the group by (row id, arrival position): the groups of a thread's elements are walked
      // together, member j of every group per step, so the LDS reads of one step are independent
      uint32_t lo[MAXR], len[MAXR];
      uint32_t steps = 0;
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        lo[r] = len[r] = 0;
        dest[r] = 0;
        if ((vmask >> r) & 1u) {
          lo[r] = S.gstart[dig[r]];
          len[r] = S.gstart[dig[r] + 1] - lo[r];
          steps = max(steps, len[r]);
        }
      }
      for (uint32_t j = 0; j < steps; ++j) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
          if (j < len[r]) {
            const uint32_t kj = S.pk[lo[r] + j], sj = S.ps[lo[r] + j];
            dest[r] += (kj < kreg[r] || (kj == kreg[r] && sj < (uint32_t)(wv * pw + r * TZR_WAVE + lane))) ? 1u : 0u;
          }
      }
#pragma unroll
      for (int r = 0; r < MAXR; ++r) dest[r] += lo[r];
      return;
    }
    __syncthreads();  // smm / wtot are reused below
  }
  const int npass = (bits + BWD_RB - 1) / BWD_RB;
  const int width = (bits + npass - 1) / npass;
  const unsigned mask = (1u << width) - 1u;
  for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
    for (int r = 0; r < MAXR; ++r) dig[r] = ((kreg[r] - kmin) >> (pass * width)) & mask;
    bwd_rank_tile<BWD_NB, MAXR>(dig, vmask, rounds, width, S.L, dest);
    if (pass + 1 < npass) {
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
        if ((vmask >> r) & 1u) {
          S.pk[dest[r]] = kreg[r];
          S.ps[dest[r]] = sreg[r];
        }
      __syncthreads();
      // the elements are dense in [0, nv) now: re-deal them wave-contiguously
      const int nv = (int)S.L.lstart[BWD_NB];
      vmask = 0;
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        const int lp = wv * pw + r * TZR_WAVE + lane;
        if (r < rounds && lp < nv) {
          vmask |= 1u << r;
          kreg[r] = S.pk[lp];
          sreg[r] = S.ps[lp];
        }
      }
    }
  }
}

