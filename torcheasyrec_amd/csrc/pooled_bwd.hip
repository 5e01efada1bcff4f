// K6: backward index plan for gfx950.
//
// Replaces fbgemm transpose_embedding_input (linearize + cub radix sort + run-length), reached from
// autograd of self.ebc(kjt) (/root/reference/tzrec/modules/embedding.py:930) with the optimizer
// fused by apply_optimizer_in_backward (/root/reference/tzrec/main.py:774-781).
//
// The apply (K7, pooled_bwd_apply.hip) wants, per table, the lookups ordered by (row, original
// lookup position): every summation order is then a function of the ids alone => bit-reproducible
// updates, no float atomics anywhere.  Round 1: 3-pass LSD radix sort, 10 dependent launches (95 us at
// B = 65536 for ~20 us of traffic).  Round 2: hist / scan / scatter / sort, 4 launches (57 us: each
// launch is a chain of dependent round trips, >= 10 us whatever it moves).  Now TWO launches, and the
// first depends on the ids only, so it can ride inside the pooled forward's launch (pooled_fwd.hip):
//
//   part   a workgroup per CHUNK (<= 1024 table-major input positions of one table): buckets its
//          lookups (<= 512 buckets per table of ~rows/512 consecutive row ids, evenly filled by
//          uniform ids -- bwd_bucket_params), orders them by bucket inside the chunk (stable, in LDS)
//          and writes them back as the chunk's SLAB with the slab-local bucket starts (1 KB).  No
//          global histogram, no cross-chunk prefix: nothing in the pass waits for another workgroup.
//          The LAST chunk of a table to arrive (one agent-scope counter per table, bucket starts
//          published write-through) sums the table's bucket counts and derives what the old scan
//          launch did: global bucket starts, the unit grid of the apply, heavy buckets and their tiles.
//   sort   a workgroup per UNIT of the apply gathers its buckets from every chunk slab (chunk order =
//          table-major order, so equal rows stay in lookup order), sorts them in LDS (< 1280 lookups)
//          and writes the final order, ks[1] -> ks[0].  Buckets with more than BWD_TH lookups (hot
//          rows: Zipf heads, default ids, every row of a tiny table) are handled by extra workgroups
//          of the same launch, one per TILE = the bucket's lookups in a run of whole chunks: a
//          counting pass when the bucket holds <= 512 row ids, a split around the hot row when it is
//          wide, a plain copy when the bucket is one row.
//
// A workgroup only ever reads what a PREVIOUS launch wrote, except the table scan of `part`, which
// uses the publish / arrive / consume protocol of tzr_gfx950.h.
#include <tzr_gfx950.h>

#include "pooled_bwd.h"

extern int g_tzr_bwd_prof;

extern "C" size_t tzr_pooled_bwd_workspace(int64_t n_values, int64_t n_positions, int n_feats,
                                           int n_tables, int64_t B, int max_dim) {
  (void)B;
  if (n_values < 0 || n_positions < 0 || n_feats <= 0 || n_tables <= 0 || max_dim <= 0) return 0;
  return bwd_layout(nullptr, nullptr, n_values, n_positions, n_feats, n_tables, max_dim) + 256;
}

// Inspection of a plan (tests, debugging): byte offsets into `ws` of
//   out[0] ks[0]  sorted pairs (uint2 {row, lookup position} per table-major position)
//   out[1] ks[1]  the chunk slabs
//   out[2] feat_start (uint32[F+1])  out[3] ucut (uint32[max_chunks+1])  out[4] cdesc (64 B each)
//   out[5] max_chunks  out[6] tcount (uint32[T]: work items per table)  out[7] hlist (32 B each)
extern "C" int tzr_pooled_bwd_plan_view(int64_t n_values, int64_t n_positions, int n_feats,
                                        int n_tables, int max_dim, int64_t* out8) {
  if (!out8 || n_values < 0 || n_positions < 0 || n_feats <= 0 || n_tables <= 0 || max_dim <= 0)
    return TZR_ERR_INVALID;
  BwdPlan P;
  bwd_layout(&P, nullptr, n_values, n_positions, n_feats, n_tables, max_dim);
  const char* base = nullptr;
  out8[0] = reinterpret_cast<const char*>(P.ks[0]) - base;
  out8[1] = reinterpret_cast<const char*>(P.ks[1]) - base;
  out8[2] = reinterpret_cast<const char*>(P.feat_start) - base;
  out8[3] = reinterpret_cast<const char*>(P.ucut) - base;
  out8[4] = reinterpret_cast<const char*>(P.cdesc) - base;
  out8[5] = P.max_chunks;
  out8[6] = reinterpret_cast<const char*>(P.tcount) - base;
  out8[7] = g_tzr_bwd_prof ? reinterpret_cast<const char*>(P.prof) - base : reinterpret_cast<const char*>(P.hlist) - base;
  return TZR_OK;
}

#include "pooled_bwd_part.h"

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
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += BWD_THREADS) {
    const TzrTable tb = tables[t];
    const uint32_t s = tb.n_feats > 0 ? P.feat_start[tb.first_order] : 0u;
    const uint32_t e = tb.n_feats > 0 ? P.feat_start[tb.first_order + tb.n_feats] : 0u;
    P.tab_chunk[t] = (int32_t)((e - s + (uint32_t)P.ch - 1) / (uint32_t)P.ch);
    P.tab_pchunk[t] = (int32_t)((e - s + (uint32_t)P.pch - 1) / (uint32_t)P.pch);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0, prun = 0;
    for (int t = 0; t < T; ++t) {
      const int32_t n = P.tab_chunk[t], pn = P.tab_pchunk[t];
      P.tab_chunk[t] = run;
      P.tab_pchunk[t] = prun;
      run += n;
      prun += pn;
    }
    P.tab_chunk[T] = run;
    P.tab_pchunk[T] = prun;
  }
}

template <bool FUSED>
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_part_kernel(
    const TzrTable* __restrict__ tables, int T, int F, BwdSrcArgs A, int one_wg_heavy, BwdPlan P) {
  __shared__ BwdPartLds S;
  bwd_part_body<FUSED>(tables, T, F, A, P, one_wg_heavy, S, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// sort: ks[1] (chunk slabs) -> ks[0] (final order), ONE launch
// ------------------------------------------------------------------------------------------
// Workgroups [0, max_chunks): one UNIT each -- the light buckets of the unit (whole buckets of
// consecutive row ids) are gathered from every chunk slab of the table and get a stable LSD sort of
// (row id - smallest row id of the unit) over the bits that difference needs, in LDS; uniform ids at
// B = 65536 on a 40M-row table: ~8 buckets of ~128 lookups, 20 bits.  Lookups of heavy buckets inside
// the unit's position range are written by the heavy workers below.
// Workgroups [max_chunks, ...): heavy-bucket workers, looping over the work items of the table scans.

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
#define BWD_MAXSUB 16  // light sub-ranges of a unit's bucket range (heavy buckets cut it; <= UMAX / TH + 2 can occur)

struct BwdSortLds {  // ~22 KB: 7 workgroups per CU
  BwdRankLds<BWD_NB> L;
  unsigned gstart[BWD_NB + 1];
  unsigned pre[BWD_NB + 1];      // heavy tile: bucket counts ahead of the tile; unit: (bucket start - light lookups ahead)
  uint32_t pk[BWD_UMAX], ps[BWD_UMAX];  // exchange buffer of the LDS-resident passes / gathered lookups
  uint32_t sbeg[BWD_SEGB];       // segment table of one batch of chunks: slab position of the segment ...
  uint32_t spre[BWD_SEGB + 1];   // ... and lookups of the batch ahead of it
  uint32_t wtot[BWD_WAVES];
  uint32_t smm[2 * BWD_WAVES];
  uint32_t sub_lo[BWD_MAXSUB], sub_hi[BWD_MAXSUB];
  uint32_t nsub, misc;
};
static_assert(BWD_HT <= BWD_UMAX, "the exchange buffer holds a heavy tile");

// ---- gathering a bucket range from the chunk slabs ----------------------------------------------
// Buckets [lo, hi) of PARTITION chunks [ca, ca + nc) of a table (relative chunk indices, nc <= BWD_SEGB): the
// lookups of chunk j are ks[1][sbeg[j] ... + (spre[j+1] - spre[j])).  Returns their number.  All
// threads call; ends with a barrier.
__device__ __forceinline__ int bwd_segs_build(BwdSortLds& S, const BwdPlan& P, int first_chunk, uint32_t ts,
                                              int ca, int nc, uint32_t lo, uint32_t hi) {
  for (int j = threadIdx.x; j < nc; j += BWD_THREADS) {
    const uint16_t* row = P.lst + (size_t)(first_chunk + ca + j) * BWD_LROW;
    const uint32_t a = row[lo], b = row[hi];
    S.spre[j] = b - a;
    S.sbeg[j] = ts + (uint32_t)(ca + j) * (uint32_t)P.pch + a;
  }
  __syncthreads();
  bwd_block_scan(S.spre, nc, S.wtot);
  return (int)S.spre[nc];
}
// lookup i (< spre[nc]) of the batch, in chunk order
__device__ __forceinline__ uint2 bwd_segs_get(const BwdSortLds& S, const uint2* __restrict__ ks1, int nc, uint32_t i) {
  int lo = 0, hi = nc;  // last j with spre[j] <= i (spre[j+1] > i: the segment that holds i)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (S.spre[mid] <= i) lo = mid; else hi = mid;
  }
  return ks1[S.sbeg[lo] + (i - S.spre[lo])];
}

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

__device__ __forceinline__ void bwd_sort_unit(const TzrTable* __restrict__ tables, const BwdPlan& P,
                                              BwdSortLds& S, int c) {
  BwdChunkDesc cd;
  if (!bwd_chunk(P, c, &cd)) return;
  if (P.uflag[c]) return;
  const bool last_unit = c + 1 >= cd.last_chunk;
  const int64_t s = P.ucut[c];
  const int64_t e = last_unit ? cd.te : (int64_t)P.ucut[c + 1];
  const int n_pos = (int)(e - s);
  if (n_pos <= 0 || n_pos > BWD_UMAX) return;
  // light buckets of the unit: all light buckets of [b_lo, b_hi) (a light bucket is never split over units)
  const uint32_t b_lo = P.ub0[c];
  const uint32_t b_hi = last_unit ? (uint32_t)cd.nb : min(P.ub0[c + 1], (uint32_t)cd.nb);
  if (b_hi <= b_lo) return;
  const int tid = threadIdx.x;
  const int lane = tid & (TZR_WAVE - 1);
  const int wv = tid / TZR_WAVE;
  const int C = (int)((cd.te - cd.ts + P.pch - 1) / P.pch);  // partition chunks of the table
  const uint32_t* hbits = P.hbits + (size_t)cd.t * (BWD_NB / 32);
  // heavy buckets inside the range cut it into light sub-ranges (rare: none under uniform ids)
  if (tid < BWD_NB / 32) {
    const uint32_t w = hbits[tid];
    uint32_t m = 0xFFFFFFFFu;
    const uint32_t w0 = (uint32_t)tid * 32u;
    if (b_lo > w0) m &= b_lo - w0 >= 32u ? 0u : (0xFFFFFFFFu << (b_lo - w0));
    if (b_hi < w0 + 32u) m &= b_hi <= w0 ? 0u : (0xFFFFFFFFu >> (w0 + 32u - b_hi));
    S.gstart[tid] = w & m;
  }
  __syncthreads();
  uint32_t any_heavy = 0;
#pragma unroll
  for (int i = 0; i < BWD_NB / 32; ++i) any_heavy |= S.gstart[i];
  if (tid == 0) {
    if (!any_heavy) {
      S.sub_lo[0] = b_lo;
      S.sub_hi[0] = b_hi;
      S.nsub = 1;
    } else {  // runs of zero bits of the heavy bitmap inside [b_lo, b_hi)
      uint32_t ns = 0, open_at = 0;
      bool open = false;
      for (uint32_t b = b_lo; b < b_hi; ++b) {
        const bool hv = (S.gstart[b >> 5] >> (b & 31)) & 1u;
        if (!hv && !open) {
          open = true;
          open_at = b;
        }
        if (hv && open) {
          open = false;
          if (ns < BWD_MAXSUB) {
            S.sub_lo[ns] = open_at;
            S.sub_hi[ns] = b;
          }
          ++ns;
        }
      }
      if (open) {
        if (ns < BWD_MAXSUB) {
          S.sub_lo[ns] = open_at;
          S.sub_hi[ns] = b_hi;
        }
        ++ns;
      }
      S.nsub = ns;
    }
  }
  __syncthreads();
  const int nsub = (int)S.nsub;
  if (nsub == 0 || nsub > BWD_MAXSUB) return;  // (more than BWD_MAXSUB cannot happen: each cut is a heavy bucket inside the unit)
  // gather: sub-range by sub-range, chunk batch by chunk batch, into the exchange buffer, in
  // (sub-range, chunk, bucket, position) order: equal rows keep their table-major order
  const uint2* __restrict__ ks1 = P.ks[1];
  int n = 0;
  for (int q = 0; q < nsub; ++q) {
    const uint32_t lo = S.sub_lo[q], hi = S.sub_hi[q];
    for (int ca = 0; ca < C; ca += BWD_SEGB) {
      const int nc = min(BWD_SEGB, C - ca);
      const int m = bwd_segs_build(S, P, cd.first_pchunk, (uint32_t)cd.ts, ca, nc, lo, hi);
      if (n + m <= BWD_UMAX) {
        for (int i = tid; i < m; i += BWD_THREADS) {
          const uint2 v = bwd_segs_get(S, ks1, nc, (uint32_t)i);
          S.pk[n + i] = v.x;
          S.ps[n + i] = v.y;
        }
      }
      n += m;
      __syncthreads();
    }
  }
  if (n <= 0 || n > BWD_UMAX) return;  // (n <= n_pos by construction)
  uint2* __restrict__ dst = P.ks[0];
  constexpr int kRounds = BWD_UMAX / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds], dest[kRounds];
  uint32_t vmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    kreg[r] = sreg[r] = 0u;
    if (r < rounds && lp < n) {
      vmask |= 1u << r;
      kreg[r] = S.pk[lp];
      sreg[r] = S.ps[lp];
      kmin = min(kmin, kreg[r]);
      kmax = max(kmax, kreg[r]);
    }
  }
  for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, m, TZR_WAVE));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, m, TZR_WAVE));
  }
  if (lane == 0) {
    S.smm[wv] = kmin;
    S.smm[BWD_WAVES + wv] = kmax;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    kmin = min(kmin, S.smm[w]);
    kmax = max(kmax, S.smm[BWD_WAVES + w]);
  }
  if (any_heavy) {
    // final position of a light lookup = start of its bucket + its rank inside the bucket; the sort below
    // yields the rank among ALL light lookups of the unit (ascending rows = ascending buckets):
    // pre[b - b_lo] = bucket start - light lookups of the unit in buckets below b
    const uint32_t* bb = P.binbase + (size_t)cd.t * (BWD_NB + 1);
    const int nbk = (int)(b_hi - b_lo);
    for (int i = tid; i < nbk; i += BWD_THREADS) {
      const uint32_t b = b_lo + (uint32_t)i;
      const bool hv = (hbits[b >> 5] >> (b & 31)) & 1u;
      S.gstart[i] = hv ? 0u : bb[b + 1] - bb[b];
    }
    __syncthreads();
    bwd_block_scan(S.gstart, nbk, S.wtot);
    for (int i = tid; i < nbk; i += BWD_THREADS) S.pre[i] = bb[b_lo + (uint32_t)i] - S.gstart[i];
  }
  __syncthreads();  // pk / ps / smm / gstart are the core's from here on
  bwd_sort_core<kRounds>(kreg, sreg, vmask, pw, rounds, kmin, max(1, bwd_bits(kmax - kmin)), !any_heavy,
                         S, dest);
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if ((vmask >> r) & 1u) {
      const uint32_t at = any_heavy ? S.pre[bwd_bucket(kreg[r], cd.mult) - b_lo] + dest[r] : (uint32_t)s + dest[r];
      dst[at] = make_uint2(kreg[r], sreg[r]);
    }
}

// ---- heavy buckets --------------------------------------------------------------------------------
struct BwdHeavyCtx {  // the table of a work item
  int first_chunk, C;
  uint32_t ts, klo;
  int n;      // lookups of the bucket
  int wbits;  // bits of (row id - klo) inside the bucket
  int bits;   // ... at least 1
};

__device__ __forceinline__ BwdHeavyCtx bwd_heavy_ctx(const TzrTable* __restrict__ tables, const BwdPlan& P,
                                                     const BwdHeavy& H) {
  BwdHeavyCtx X;
  const BwdChunkDesc cd = P.cdesc[P.tab_chunk[H.t]];
  X.first_chunk = cd.first_pchunk;  // partition chunks: where the bucket's lookups sit
  X.C = (int)((cd.te - cd.ts + P.pch - 1) / P.pch);
  X.ts = (uint32_t)cd.ts;
  const int64_t rows = tables[H.t].rows;
  const uint64_t klo64 = (((uint64_t)H.bin << 32) + cd.mult - 1) / cd.mult;
  uint64_t khi = (((uint64_t)(H.bin + 1) << 32) + cd.mult - 1) / cd.mult;
  if (khi > (uint64_t)rows) khi = (uint64_t)rows;
  X.klo = (uint32_t)klo64;
  X.wbits = bwd_bits((uint32_t)(khi - klo64 - 1));
  X.bits = max(1, X.wbits);
  const uint32_t* bb = P.binbase + (size_t)H.t * (BWD_NB + 1);
  X.n = (int)(bb[H.bin + 1] - H.start);
  return X;
}

// The most frequent row id among 64 evenly spaced lookups of the bucket's first chunk batch (ties: the
// earliest sample): every wave of every workgroup that walks the bucket gets the same answer.
__device__ __forceinline__ uint32_t bwd_sample_mode(const BwdSortLds& S, const uint2* __restrict__ ks1, int nc,
                                                    int m, int lane) {
  if (m <= 0) return BWD_SENT;
  const uint32_t sk = bwd_segs_get(S, ks1, nc, (uint32_t)(((int64_t)lane * m) / TZR_WAVE)).x;
  unsigned long long peers = ~0ull;
  for (int bit = 0; bit < 32; ++bit) {
    const int on = (sk >> bit) & 1;
    const unsigned long long bm = __ballot(on);
    peers &= on ? bm : ~bm;
  }
  uint32_t best = ((uint32_t)__popcll(peers) << 6) | (uint32_t)(TZR_WAVE - 1 - lane);
  for (int mm = TZR_WAVE >> 1; mm > 0; mm >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, mm, TZR_WAVE));
  return (uint32_t)__shfl((int)sk, TZR_WAVE - 1 - (int)(best & 63u), TZR_WAVE);
}

// Lookups [w0, w0 + wn) of the current chunk batch (wn <= MAXR * 256), dealt wave-contiguously (lookup lp of
// the window at round r of lane l of wave w with lp = w*pw + r*64 + l).  Returns the mask of valid rounds.
template <int MAXR>
__device__ __forceinline__ uint32_t bwd_window_load(const BwdSortLds& S, const uint2* __restrict__ ks1, int nc,
                                                    int w0, int wn, int pw, int rounds,
                                                    uint32_t (&kreg)[MAXR], uint32_t (&sreg)[MAXR]) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  uint32_t vmask = 0;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    kreg[r] = sreg[r] = 0u;
    if (r < rounds && lp < wn) {
      const uint2 v = bwd_segs_get(S, ks1, nc, (uint32_t)(w0 + lp));
      kreg[r] = v.x;
      sreg[r] = v.y;
      vmask |= 1u << r;
    }
  }
  return vmask;
}

// lookups of bucket `bin` in chunks [0, c_end) of the table (workgroup-uniform; ends with a barrier)
__device__ __forceinline__ uint32_t bwd_count_before(BwdSortLds& S, const BwdPlan& P, const BwdHeavyCtx& X,
                                                     uint32_t bin, int c_end) {
  uint32_t tot = 0;
  for (int ca = 0; ca < c_end; ca += BWD_SEGB) {
    const int nc = min(BWD_SEGB, c_end - ca);
    tot += (uint32_t)bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, bin, bin + 1);
    __syncthreads();
  }
  return tot;
}

// bucket = one row (exact table): chunk order is the final order
__device__ __forceinline__ void bwd_heavy_copy(const TzrTable* __restrict__ tables, const BwdPlan& P,
                                               BwdSortLds& S, const BwdHeavy& H) {
  const BwdHeavyCtx X = bwd_heavy_ctx(tables, P, H);
  uint32_t off = bwd_count_before(S, P, X, H.bin, H.c_begin);
  uint2* __restrict__ dst = P.ks[0] + H.start;
  for (int ca = H.c_begin; ca < H.c_end; ca += BWD_SEGB) {
    const int nc = min(BWD_SEGB, H.c_end - ca);
    const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
    for (int i = threadIdx.x; i < m; i += BWD_THREADS) dst[off + i] = bwd_segs_get(S, P.ks[1], nc, (uint32_t)i);
    off += (uint32_t)m;
    __syncthreads();
  }
}

// bucket of <= BWD_NB row ids: the tile's workgroup counts the whole bucket per row id on its own
// (no inter-workgroup traffic; the counts of the lookups in the chunks ahead of its tile are the
// cross-tile prefix), then ranks and writes its tile, BWD_HT lookups at a time: the tiles of a hot
// bucket are sorted in parallel.
__device__ __forceinline__ void bwd_heavy_onepass(const TzrTable* __restrict__ tables, const BwdPlan& P,
                                                  BwdSortLds& S, const BwdHeavy& H) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const BwdHeavyCtx X = bwd_heavy_ctx(tables, P, H);
  const uint2* __restrict__ ks1 = P.ks[1];
  uint2* __restrict__ dst = P.ks[0] + H.start;
  for (int i = threadIdx.x; i <= BWD_NB; i += BWD_THREADS) S.gstart[i] = S.pre[i] = 0;
  // (a heavy bucket usually is heavy because of ONE row: its lookups are counted with ballots into
  // wave registers, the others -- few per wave, on different counters -- with one LDS atomic each)
  uint32_t hot = BWD_SENT, hot_tot = 0, hot_pre = 0;
  constexpr int kU = 4;
  for (int ca = 0; ca < X.C; ca += BWD_SEGB) {
    const int nc = min(BWD_SEGB, X.C - ca);
    const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
    if (ca == 0) hot = bwd_sample_mode(S, ks1, nc, m, lane);
    // lookups of the batch that sit in chunks ahead of the tile: a prefix of the batch
    const int nah = H.c_begin <= ca ? 0 : (H.c_begin >= ca + nc ? m : (int)S.spre[H.c_begin - ca]);
    for (int base = 0; base < m; base += BWD_THREADS * kU) {
      uint32_t k8[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = base + u * BWD_THREADS + (int)threadIdx.x;
        k8[u] = i < m ? bwd_segs_get(S, ks1, nc, (uint32_t)i).x : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = base + u * BWD_THREADS + (int)threadIdx.x;
        const bool v = i < m;
        const bool ahead = i < nah;
        const bool is_hot = v && k8[u] == hot;
        hot_tot += (uint32_t)__popcll(__ballot(is_hot));
        hot_pre += (uint32_t)__popcll(__ballot(is_hot && ahead));
        if (v && !is_hot) {
          atomicAdd(&S.gstart[k8[u] - X.klo], 1u);
          if (ahead) atomicAdd(&S.pre[k8[u] - X.klo], 1u);
        }
      }
    }
    __syncthreads();
  }
  if (lane == 0 && hot_tot) {
    atomicAdd(&S.gstart[hot - X.klo], hot_tot);
    if (hot_pre) atomicAdd(&S.pre[hot - X.klo], hot_pre);
  }
  __syncthreads();
  bwd_block_scan(S.gstart, BWD_NB, S.wtot);
  constexpr int kRounds = BWD_HT / BWD_THREADS;
  for (int ca = H.c_begin; ca < H.c_end; ca += BWD_SEGB) {
    const int nc = min(BWD_SEGB, H.c_end - ca);
    const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
    for (int w0 = 0; w0 < m; w0 += BWD_HT) {
      const int wn = min(BWD_HT, m - w0);
      const int pw = bwd_wave_span(wn);
      const int rounds = pw / TZR_WAVE;
      uint32_t kreg[kRounds], sreg[kRounds], dig[kRounds], dest[kRounds];
      const uint32_t vmask = bwd_window_load<kRounds>(S, ks1, nc, w0, wn, pw, rounds, kreg, sreg);
#pragma unroll
      for (int r = 0; r < kRounds; ++r) dig[r] = ((vmask >> r) & 1u) ? kreg[r] - X.klo : 0u;
      bwd_rank_tile<BWD_NB, kRounds>(dig, vmask, rounds, X.wbits, S.L, dest);
#pragma unroll
      for (int r = 0; r < kRounds; ++r)
        if ((vmask >> r) & 1u)
          dst[S.gstart[dig[r]] + S.pre[dig[r]] + (dest[r] - (uint32_t)S.L.lstart[dig[r]])] =
              make_uint2(kreg[r], sreg[r]);
      __syncthreads();
      for (int d = threadIdx.x; d < BWD_NB; d += BWD_THREADS)  // the next window's lookups come behind these
        S.pre[d] += (unsigned)S.L.lstart[d + 1] - (unsigned)S.L.lstart[d];
      __syncthreads();
    }
  }
}

// The whole bucket by one workgroup.  One tile: every pass in LDS.  More: the lookups are first copied
// out of the slabs into ks[2] (chunk order), then stable LSD passes ks[2] -> ks[0] -> ks[2] -> ks[0], tile by
// tile; a single workgroup walks the tiles in order, so the running per-digit offsets ARE the cross-tile prefix.
__device__ __forceinline__ void bwd_heavy_serial(const TzrTable* __restrict__ tables, const BwdPlan& P,
                                                 BwdSortLds& S, const BwdHeavy& H) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const BwdHeavyCtx X = bwd_heavy_ctx(tables, P, H);
  const int n = X.n;
  constexpr int kRounds = BWD_HT / BWD_THREADS;
  if (n <= BWD_HT && X.C <= BWD_SEGB) {  // one batch of chunks, one tile: every pass in LDS
    uint2* __restrict__ dst = P.ks[0] + H.start;
    const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, 0, X.C, H.bin, H.bin + 1);
    const int pw = bwd_wave_span(m);
    const int rounds = pw / TZR_WAVE;
    uint32_t kreg[kRounds], sreg[kRounds], dest[kRounds];
    uint32_t vmask = bwd_window_load<kRounds>(S, P.ks[1], X.C, 0, m, pw, rounds, kreg, sreg);
    __syncthreads();
    bwd_sort_core<kRounds>(kreg, sreg, vmask, pw, rounds, X.klo, X.bits, false, S, dest);
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
      if ((vmask >> r) & 1u) dst[dest[r]] = make_uint2(kreg[r], sreg[r]);
    __syncthreads();
    return;
  }
  {  // chunk order -> contiguous
    uint2* __restrict__ flat = P.ks[2] + H.start;
    int off = 0;
    for (int ca = 0; ca < X.C; ca += BWD_SEGB) {
      const int nc = min(BWD_SEGB, X.C - ca);
      const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
      for (int i = threadIdx.x; i < m; i += BWD_THREADS) flat[off + i] = bwd_segs_get(S, P.ks[1], nc, (uint32_t)i);
      off += m;
      __syncthreads();
    }
    __threadfence();  // this workgroup reads it back below
    __syncthreads();
  }
  // an odd number of passes ks[2] -> ks[0] -> ks[2] -> ks[0] ends where the apply reads
  const int bits = X.bits;
  const int npass = bits <= BWD_RB ? 1 : (bits <= 3 * BWD_RB ? 3 : 5);
  const int width = (bits + npass - 1) / npass;
  const unsigned mask = (1u << width) - 1u;
  const uint32_t klo = X.klo;
  for (int pass = 0; pass < npass; ++pass) {
    const uint2* __restrict__ src = ((pass & 1) ? P.ks[0] : P.ks[2]) + H.start;
    uint2* __restrict__ dst = ((pass & 1) ? P.ks[2] : P.ks[0]) + H.start;
    const int shift = pass * width;
    for (int i = threadIdx.x; i <= BWD_NB; i += BWD_THREADS) S.gstart[i] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += BWD_HT) {
      uint32_t k8[kRounds];
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
        const int i = base + r * BWD_THREADS + (int)threadIdx.x;
        k8[r] = i < n ? src[i].x : 0u;
      }
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
        if (base + r * BWD_THREADS >= n) break;  // wave-uniform
        const int i = base + r * BWD_THREADS + (int)threadIdx.x;
        bwd_wave_count(S.gstart, ((k8[r] - klo) >> shift) & mask, i < n, width, lane);
      }
    }
    __syncthreads();
    bwd_block_scan(S.gstart, BWD_NB, S.wtot);
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
          dig[r] = ((v.x - klo) >> shift) & mask;
        }
      }
      bwd_rank_tile<BWD_NB, kRounds>(dig, vmask, rounds, width, S.L, dest);
#pragma unroll
      for (int r = 0; r < kRounds; ++r)
        if ((vmask >> r) & 1u)
          dst[S.gstart[dig[r]] + (dest[r] - (uint32_t)S.L.lstart[dig[r]])] = make_uint2(kreg[r], sreg[r]);
      __syncthreads();
      for (int d = threadIdx.x; d < BWD_NB; d += BWD_THREADS)
        S.gstart[d] += (unsigned)S.L.lstart[d + 1] - (unsigned)S.L.lstart[d];
      __syncthreads();
    }
    __threadfence();  // this workgroup reads the pass's output back in the next pass
    __syncthreads();
  }
}

// Tile of a WIDE heavy bucket (more than BWD_NB row ids, more than one tile of lookups).  Such a
// bucket is almost always one hot row plus a sprinkle of cold ones, so instead of LSD passes by one
// workgroup (26 tiles x 3 passes for the clipped tail of a 40M-row table) every tile workgroup
//   1. picks the same candidate row: the most frequent of 64 evenly spaced samples of the bucket's first chunks;
//   2. walks the bucket once counting lookups below / equal to the candidate (ballots only), in the
//      whole bucket and ahead of its own tile: they place its hot lookups, in table-major order, after
//      the cold rows below the candidate;
//   3. the tile that starts at the table's first chunk also gathers the cold lookups (at most BWD_UMAX,
//      else see below) in table-major order, sorts them in LDS and writes them around the hot run.
// Every workgroup derives the same counts, so all of them take the same decision: when the cold
// lookups do not fit, the first tile falls back to the passes over the whole bucket and the others
// leave (the fallback never writes ks[1]: the others may still be reading it).  Resulting order:
// ascending row ids, table-major order inside a row, like everywhere else.
__device__ __forceinline__ void bwd_heavy_hot(const TzrTable* __restrict__ tables, const BwdPlan& P,
                                              BwdSortLds& S, const BwdHeavy& H) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const BwdHeavyCtx X = bwd_heavy_ctx(tables, P, H);
  const uint2* __restrict__ ks1 = P.ks[1];
  uint2* __restrict__ dst = P.ks[0] + H.start;
  const int n = X.n;
  constexpr int kRounds = BWD_HT / BWD_THREADS;
  constexpr int kU = 4;
  uint32_t hot = BWD_SENT;
  uint32_t lt_tot = 0, eq_tot = 0, eq_pre = 0;
  for (int ca = 0; ca < X.C; ca += BWD_SEGB) {
    const int nc = min(BWD_SEGB, X.C - ca);
    const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
    if (ca == 0) hot = bwd_sample_mode(S, ks1, nc, m, lane);
    // lookups of the batch that sit in chunks ahead of the tile: a prefix of the batch
    const int nah = H.c_begin <= ca ? 0 : (H.c_begin >= ca + nc ? m : (int)S.spre[H.c_begin - ca]);
    for (int base = 0; base < m; base += BWD_THREADS * kU) {
      uint32_t k4[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = base + u * BWD_THREADS + (int)threadIdx.x;
        k4[u] = i < m ? bwd_segs_get(S, ks1, nc, (uint32_t)i).x : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = base + u * BWD_THREADS + (int)threadIdx.x;
        const bool v = i < m;
        lt_tot += (uint32_t)__popcll(__ballot(v && k4[u] < hot));
        eq_tot += (uint32_t)__popcll(__ballot(v && k4[u] == hot));
        eq_pre += (uint32_t)__popcll(__ballot(v && k4[u] == hot && i < nah));
      }
    }
    __syncthreads();
  }
  if (lane == 0) {  // wave-level partial counts (a batch's lookups are dealt over the four waves)
    S.gstart[wv] = lt_tot;
    S.gstart[BWD_WAVES + wv] = eq_tot;
    S.gstart[2 * BWD_WAVES + wv] = eq_pre;
  }
  __syncthreads();
  uint32_t n_lt = 0, n_eq = 0, eq_ahead = 0;
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    n_lt += S.gstart[w];
    n_eq += S.gstart[BWD_WAVES + w];
    eq_ahead += S.gstart[2 * BWD_WAVES + w];
  }
  __syncthreads();
  const int n_cold = n - (int)n_eq;
  if (n_cold > BWD_UMAX) {  // not one hot row: every tile workgroup of the bucket sees the same numbers
    if (H.c_begin == 0) bwd_heavy_serial(tables, P, S, H);
    return;
  }
  // the hot lookups of this tile, in table-major order, behind the hot lookups of the tiles before
  {
    uint32_t ahead0 = n_lt + eq_ahead;  // workgroup-uniform: first position for the next window's hot lookups
    for (int ca = H.c_begin; ca < H.c_end; ca += BWD_SEGB) {
      const int nc = min(BWD_SEGB, H.c_end - ca);
      const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
      for (int w0 = 0; w0 < m; w0 += BWD_HT) {
        const int wn = min(BWD_HT, m - w0);
        const int pw = bwd_wave_span(wn);
        const int rounds = pw / TZR_WAVE;
        uint32_t kreg[kRounds], sreg[kRounds], hpos[kRounds];
        const uint32_t vmask = bwd_window_load<kRounds>(S, ks1, nc, w0, wn, pw, rounds, kreg, sreg);
        uint32_t hmask = 0, run = 0;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
          hpos[r] = 0;
          if (r < rounds) {
            const bool is_hot = ((vmask >> r) & 1u) && kreg[r] == hot;
            const unsigned long long hm = __ballot(is_hot);
            hpos[r] = run + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
            run += (uint32_t)__popcll(hm);
            if (is_hot) hmask |= 1u << r;
          }
        }
        if (lane == 0) S.wtot[wv] = run;
        __syncthreads();
        uint32_t ahead = ahead0, win_hot = 0;
#pragma unroll
        for (int w = 0; w < BWD_WAVES; ++w) {
          if (w < wv) ahead += S.wtot[w];
          win_hot += S.wtot[w];
        }
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
          if ((hmask >> r) & 1u) dst[ahead + hpos[r]] = make_uint2(hot, sreg[r]);
        ahead0 += win_hot;
        __syncthreads();
      }
    }
  }
  if (H.c_begin != 0 || n_cold == 0) return;
  // 3. the cold lookups of the whole bucket: ordered gather into LDS, stable sort, write
  uint32_t gathered = 0;  // cold lookups walked so far (workgroup-uniform)
  for (int ca = 0; ca < X.C; ca += BWD_SEGB) {
    const int nc = min(BWD_SEGB, X.C - ca);
    const int m = bwd_segs_build(S, P, X.first_chunk, X.ts, ca, nc, H.bin, H.bin + 1);
    for (int base = 0; base < m; base += BWD_HT) {
      const int nt = min(BWD_HT, m - base);
      const int pw = bwd_wave_span(nt);
      const int rounds = pw / TZR_WAVE;
      uint32_t kc[kRounds], sc[kRounds], cpos[kRounds];
      uint32_t cmask = 0, run = 0;
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
        const int lp = wv * pw + r * TZR_WAVE + lane;
        bool cold = false;
        kc[r] = sc[r] = cpos[r] = 0;
        if (r < rounds && lp < nt) {
          const uint2 v = bwd_segs_get(S, ks1, nc, (uint32_t)(base + lp));
          kc[r] = v.x;
          sc[r] = v.y;
          cold = v.x != hot;
        }
        if (r < rounds) {
          const unsigned long long cm = __ballot(cold);
          cpos[r] = run + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
          run += (uint32_t)__popcll(cm);
          if (cold) cmask |= 1u << r;
        }
      }
      if (lane == 0) S.wtot[wv] = run;
      __syncthreads();
      uint32_t ahead = gathered, tile_cold = 0;
#pragma unroll
      for (int w = 0; w < BWD_WAVES; ++w) {
        if (w < wv) ahead += S.wtot[w];
        tile_cold += S.wtot[w];
      }
#pragma unroll
      for (int r = 0; r < kRounds; ++r)
        if ((cmask >> r) & 1u) {
          S.pk[ahead + cpos[r]] = kc[r];
          S.ps[ahead + cpos[r]] = sc[r];
        }
      gathered += tile_cold;
      __syncthreads();
    }
  }
  {
    constexpr int cRounds = BWD_UMAX / BWD_THREADS;
    const int pw = bwd_wave_span(n_cold);
    const int rounds = pw / TZR_WAVE;
    uint32_t kreg[cRounds], sreg[cRounds], dest[cRounds];
    uint32_t vmask = 0;
#pragma unroll
    for (int r = 0; r < cRounds; ++r) {
      const int lp = wv * pw + r * TZR_WAVE + lane;
      kreg[r] = sreg[r] = 0;
      if (r < rounds && lp < n_cold) {
        vmask |= 1u << r;
        kreg[r] = S.pk[lp];
        sreg[r] = S.ps[lp];
      }
    }
    __syncthreads();  // the exchange buffer is the core's from here on
    bwd_sort_core<cRounds>(kreg, sreg, vmask, pw, rounds, X.klo, X.bits, false, S, dest);
#pragma unroll
    for (int r = 0; r < cRounds; ++r)
      if ((vmask >> r) & 1u) dst[dest[r] + (kreg[r] > hot ? n_eq : 0u)] = make_uint2(kreg[r], sreg[r]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(BWD_THREADS) TZR_WAVES_PER_EU(7) void tzr_bwd_sort_kernel(
    const TzrTable* __restrict__ tables, int n_tables, int n_units, BwdPlan P) {
  __shared__ BwdSortLds S;
  if ((int)blockIdx.x < n_units) {
    bwd_sort_unit(tables, P, S, (int)blockIdx.x);
    return;
  }
  // the heavy work items: tcount[t] of them in the region of table t; this worker takes items w, w + W, ...
  // of the concatenation (table by table: the prefix over <= a few hundred tables lives in LDS)
  const unsigned workers = gridDim.x - (unsigned)n_units;
  const unsigned w = blockIdx.x - (unsigned)n_units;
  unsigned base = 0;  // items of the tables before `t0`
  for (int t0 = 0; t0 < n_tables; t0 += BWD_SEGB) {
    const int nt = min(BWD_SEGB, n_tables - t0);
    for (int j = threadIdx.x; j < nt; j += BWD_THREADS) S.spre[j] = P.tcount[t0 + j];
    __syncthreads();
    bwd_block_scan(S.spre, nt, S.wtot);
    const unsigned tot = S.spre[nt];
    // first item of this worker at or after `base`
    unsigned hi = base + ((w + workers - base % workers) % workers);
    for (; hi < base + tot; hi += workers) {
      const unsigned rel = hi - base;
      int lo = 0, up = nt;
      while (up - lo > 1) {
        const int mid = (lo + up) >> 1;
        if (S.spre[mid] <= rel) lo = mid; else up = mid;
      }
      const int t = t0 + lo;
      const unsigned idx = rel - S.spre[lo];
      __syncthreads();  // the heavy paths reuse the segment tables
      const BwdHeavy H = P.hlist[bwd_hbase(P.feat_start[tables[t].first_order], (uint32_t)t) + idx];
      if (H.kind == BWD_HK_COPY) bwd_heavy_copy(tables, P, S, H);
      else if (H.kind == BWD_HK_ONEPASS) bwd_heavy_onepass(tables, P, S, H);
      else if (H.kind == BWD_HK_HOT) bwd_heavy_hot(tables, P, S, H);
      else bwd_heavy_serial(tables, P, S, H);
      __syncthreads();
      // the table prefix was overwritten: rebuild it
      for (int j = threadIdx.x; j < nt; j += BWD_THREADS) S.spre[j] = P.tcount[t0 + j];
      __syncthreads();
      bwd_block_scan(S.spre, nt, S.wtot);
    }
    base += tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

int g_tzr_bwd_force_prep = 0;  // tzr_tune("bwd_force_prep"): take the > BWD_GEO geometry path
int g_tzr_bwd_ch = 0;          // tzr_tune("bwd_ch"): positions per unit block (0 = by problem size)
int g_tzr_bwd_pk = 0;          // tzr_tune("bwd_pk"): unit blocks per partition chunk (0 = by problem size)
int g_tzr_bwd_one_wg_heavy = 0;  // tzr_tune("bwd_one_wg_heavy"): no tile-parallel heavy buckets
int g_tzr_bwd_prof = 0;          // tzr_tune("bwd_prof"): the partition pass records phase timestamps per chunk (BwdPlan.prof)

static void bwd_launch_sort(const TzrTable* d_tables, int n_tables, const BwdPlan& P, hipStream_t s) {
  const unsigned chunks = (unsigned)P.max_chunks;
  const unsigned workers = (unsigned)std::min<int64_t>(P.max_heavy, 1024);
  hipLaunchKernelGGL(tzr_bwd_sort_kernel, dim3(chunks + workers), dim3(BWD_THREADS), 0, s, d_tables,
                     n_tables, (int)chunks, P);
}

extern "C" int tzr_pooled_bwd_plan(const TzrTable* d_tables, int n_tables,
                                   const TzrFeature* d_feats, int n_feats, int n_keys,
                                   int64_t max_rows, int max_dim, const int64_t* d_values, const int64_t* d_offsets,
                                   int64_t n_values, int64_t n_positions, int64_t B,
                                   int uniform_bag_len, void* ws,
                                   size_t ws_bytes, void* stream) {
  BwdPlan P;
  const int rc = bwd_plan_check(d_tables, n_tables, d_feats, n_feats, n_keys, max_rows, max_dim, d_offsets, n_values,
                                n_positions, B, uniform_bag_len, ws, ws_bytes, &P);
  if (rc != TZR_OK) return rc;
  if (n_values == 0 || B == 0) return TZR_OK;
  if (!d_values) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned chunks = (unsigned)P.max_pchunks;
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = d_offsets;
  A.B = B;
  A.uniform = (int)(uniform_bag_len == 1);
  // the per-table arrival counters and item counts start at zero (a launch of our own: hipMemsetAsync of
  // these 200 bytes runs as two 5 us fill kernels on this stack)
  hipLaunchKernelGGL(tzr_bwd_zero_kernel, dim3(1), dim3(BWD_THREADS), 0, s, P.tarr, P.tcount, n_tables);
  if (n_feats > BWD_GEO || n_tables > BWD_GEO || g_tzr_bwd_force_prep) {
    hipLaunchKernelGGL(tzr_bwd_prep_kernel, dim3(1), dim3(BWD_THREADS), 0, s, d_tables, n_tables, A,
                       n_feats, P);
    hipLaunchKernelGGL(tzr_bwd_part_kernel<false>, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, n_feats, A, (g_tzr_bwd_one_wg_heavy & 1) | (g_tzr_bwd_prof ? 2 : 0), P);
  } else {
    hipLaunchKernelGGL(tzr_bwd_part_kernel<true>, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables,
                       n_tables, n_feats, A, (g_tzr_bwd_one_wg_heavy & 1) | (g_tzr_bwd_prof ? 2 : 0), P);
  }
  bwd_launch_sort(d_tables, n_tables, P, s);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// The second launch of the plan on its own: after tzr_pooled_fwd_plan ran the partition pass inside the
// forward's launch.
extern "C" int tzr_pooled_bwd_plan_finish(const TzrTable* d_tables, int n_tables, int n_feats, int max_dim,
                                          int64_t n_values, int64_t n_positions, void* ws, size_t ws_bytes,
                                          void* stream) {
  if (!d_tables || n_tables <= 0 || n_feats <= 0 || n_values < 0 || max_dim <= 0) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  if (n_positions < 0 || n_positions >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  BwdPlan P;
  if (bwd_layout(&P, ws, n_values, n_positions, n_feats, n_tables, max_dim) > ws_bytes) return TZR_ERR_WORKSPACE;
  if (n_values == 0) return TZR_OK;
  bwd_launch_sort(d_tables, n_tables, P, static_cast<hipStream_t>(stream));
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
