// K7: fused backward + sparse optimizer, wavefront-level duplicate-row reduction (gfx950).
//
// Replaces fbgemm split_embedding_backward_codegen_{sgd,adagrad,rowwise_adagrad}_*_exact_
// {warp,cta}_per_row_1 (optimizer fused into backward by apply_optimizer_in_backward,
// /root/reference/tzrec/main.py:774-781).
//
// Input: the plan of pooled_bwd.hip -- per table, lookups sorted by (row, original position).
// A workgroup owns one UNIT of the table's sorted positions (< BWD_CH + BWD_TH lookups: whole light
// buckets and/or block-sized slices of heavy ones; keys + sources staged once in LDS, coalesced);
// each of its 4 waves reduces a quarter of the unit tile by tile:
//   * a tile = 64/(D/4) consecutive sorted lookups, one per lane group: ALL gradient gathers of a
//     tile are independent loads in flight together (the HBM/MALL latency is paid once per tile,
//     not once per duplicate);
//   * duplicates are summed with a segmented inclusive scan over the lane groups (shuffles,
//     log2 steps, fixed tree => bit-reproducible), runs crossing tiles ride a register carry;
//   * the lane group on a run's LAST lookup performs the single read-modify-write of the row
//     (weights + optimizer state).  Exactly one lane group in the whole grid touches a given row:
//     no atomics, no cross-XCD L2 coherence hazard.
//   * runs crossing a wave range are stitched through LDS records by wave 0, runs crossing a
//     unit (only possible inside a heavy bucket or an exact table) through per-unit records by
//     the last unit of the table to finish (a 65536-lookup run of a 3-row table is 64 unit records
//     long): ONE launch for the whole apply.
#include <tzr_gfx950.h>

#ifdef IT_PROF  // scripts/build_prof_lib.sh: wall-clock stamps (100 MHz) of every workgroup's phases, read back by tzr_bwd_prof_dump
#define BWD_PROF_WGS 4096
__device__ uint64_t g_bwd_prof[BWD_PROF_WGS * 8];
#define BWD_PROF_MARK(i) do { if (blockIdx.x < BWD_PROF_WGS && (threadIdx.x & (TZR_WAVE - 1)) == 0) { \
    const int w_ = threadIdx.x / TZR_WAVE; if ((i) < 2 ? w_ == 0 : true) g_bwd_prof[blockIdx.x * 8 + ((i) < 2 ? (i) : (i) == 2 ? 2 + w_ : 6 + ((i) - 3))] = wall_clock64(); } } while (0)
extern "C" int tzr_bwd_prof_dump(uint64_t* h_out, int n_wg) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_bwd_prof), (size_t)std::min(n_wg, BWD_PROF_WGS) * 8 * sizeof(uint64_t)) == hipSuccess ? 0 : -1;
}
#else
#define BWD_PROF_MARK(i)
#endif

#include "pooled_bwd_apply.h"
#include "pooled_bwd_sort.h"

// LDS of an apply workgroup: the unit's reduction (U), and -- before it, in the same bytes -- the unit's sort when the plan
// left it to the apply (S).
union BwdApplyLds {
  BwdSortLds S;
  BwdUnitLds U;
};

// The unit's lookups into U.sK[1 .. n] / U.sS[0 .. n) with a neighbour (or BWD_SENT) on either side, SORTED: from ks[0] (the
// sort launch's output; ks[1] for an exact table, final after the partition pass) -- or, for a unit of a bucketed table whose
// position range holds no heavy lookup when the plan was built fused (P.hcount[1]; pooled_bwd.hip: the sort launch skips those
// units), from ks[1] through the sort launch's own LDS sort (bwd_sort_core: one LDS atomic per lookup into 512 groups by the
// low bits of the row id, in-group ranking by (row id, position)), here.  That takes the unit sort's dependent chain --
// descriptor -> cuts -> lookups -> LDS -> 8-byte scattered stores -> the apply's re-read -- off the step: with uniform ids the
// sort launch was 15 us behind a 2.7 us launch gap, each of its workgroups 12.7 us (profiles/r05an/plan_phase_profile.txt).
// A unit made of whole light buckets has no run that continues outside it: sentinels on both sides.  All threads call; ends
// with a barrier.
__device__ __forceinline__ void bwd_stage_unit(const BwdPlan& P, const BwdChunkDesc& cd, int64_t s, int64_t e, int n, bool fused,
                                               BwdApplyLds& L, TzrDst* sG, const BwdGrads& G) {
  const int64_t ts = cd.ts, te = cd.te;
  if (!fused) {
    const uint2* __restrict__ KS = cd.exact ? P.ks[1] : P.ks[0];
    for (int i = threadIdx.x; i < n; i += BWD_THREADS) {
      const uint2 v = KS[s + i];
      L.U.sK[i + 1] = v.x;
      L.U.sS[i] = v.y;
    }
    if (threadIdx.x == 0) {
      // the neighbours outside the unit are either in another bucket (another row id) or in the same
      // sorted bucket: comparing with them is always meaningful
      L.U.sK[0] = s > ts ? KS[s - 1].x : BWD_SENT;
      L.U.sK[n + 1] = e < te ? KS[e].x : BWD_SENT;
#pragma unroll
      for (int i = 0; i < TZR_MAX_DST; ++i) sG[i] = G.d[i];  // static indices: straight from kernarg
    }
    __syncthreads();
    return;
  }
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const uint2* __restrict__ src = P.ks[1] + s;
  constexpr int kRounds = BWD_UMAX / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds], dest[kRounds];
  uint32_t vmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    const bool in = r < rounds && lp < n;
    const uint2 v = src[lp < n ? lp : n - 1];  // (clamped, unconditional: see bwd_elem_one)
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
    L.U.sK[0] = BWD_SENT;
    L.U.sK[n + 1] = BWD_SENT;
#pragma unroll
    for (int i = 0; i < TZR_MAX_DST; ++i) sG[i] = G.d[i];
  }
  __syncthreads();
}

// Runs crossing unit boundaries: the unit holding the run's first lookup adds the leading pieces
// of the following units (in order) and updates the row.  One wave.
template <bool ADAM>
__device__ __forceinline__ void bwd_stitch_unit(const TzrTable& tb, const BwdOpt& opt, float lr,
                                                int max_dim, const BwdPlan& P, int chunk,
                                                int last_chunk, int lane) {
  // The run that is open at the end of unit `chunk` continues through the LEADING pieces of the units behind it.  Their
  // records are fetched a wave's worth at a time -- 64 / (D / 4) units per round trip, flag and piece together -- and added in
  // unit order by shuffles.  (One unit after the other, flag then piece, this walk was two dependent cache-bypassing loads
  // per unit: 21 units x ~1.5 us for a row of the 3-row Criteo table at B = 65 536, and the workgroup doing it was the LAST
  // of the launch to finish -- 34 of the kernel's 78 us, profiles/r05ai/apply_phase_profile_fast1.txt.)
  const int lg = tb.dim >> 2;    // lanes per record
  const int gw = TZR_WAVE / lg;  // records per round trip
  const int gi = lane / lg, c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const uint32_t key = tzr_consume_u32(P.ctkey + chunk);
  float4 sum = tzr_zero4();
  if (lane < lg) sum = bwd_consume4(P.ctrail + (size_t)chunk * max_dim + 4 * lane);
  bool done = false;
  for (int c0 = chunk + 1; c0 < last_chunk && !done; c0 += gw) {
    const int c2 = c0 + gi;
    const bool in = lane_on && c2 < last_chunk;
    unsigned f = 0;
    float4 v = tzr_zero4();
    if (in) {
      f = tzr_consume_u32(P.cflags + c2);
      v = bwd_consume4(P.clead + (size_t)c2 * max_dim + 4 * c);
    }
    const unsigned long long lead = __ballot(in && c == 0 && (f & BWD_LEAD));
    const unsigned long long whole = __ballot(in && c == 0 && (f & BWD_LEAD) && (f & BWD_LEAD_WHOLE));
    for (int g = 0; g < gw; ++g) {  // (wave-uniform)
      if (!((lead >> (g * lg)) & 1ull)) {
        done = true;
        break;
      }
      const float4 piece = bwd_shfl4(v, g * lg + (lane < lg ? lane : 0));
      if (lane < lg) sum = tzr_add4(sum, piece);
      if (!((whole >> (g * lg)) & 1ull)) {
        done = true;
        break;
      }
    }
  }
  bwd_apply_row_wave<ADAM>(tb, opt, lr, key, sum, lane);
}

// Called by wave 0 of every unit once its boundary record is published.  Runs can only cross unit
// boundaries inside a sorted bucket (a row of an exact table, a heavy bucket): the units that
// overlap such a bucket meet at the bucket's counter, and the LAST of them to arrive stitches the
// bucket's open runs -- the pieces are summed in unit order whoever does it, so the result does
// not depend on the arrival order; buckets are stitched in parallel, and no second launch is
// needed (round 1: tzr_bwd_stitch_kernel, 10 us at B = 65536).  A unit touches at most two such
// buckets: the one its first lookup and the one its last lookup falls in.
template <bool ADAM>
__device__ __forceinline__ void bwd_arrive_and_stitch(const TzrTable& tb, const BwdOpt& opt, float lr,
                                                      int max_dim, const BwdPlan& P,
                                                      const BwdChunkDesc& cd, uint32_t b_first,
                                                      uint32_t b_last, int lane) {
  if (!P.tab_stitch[cd.t]) return;
  const size_t base = (size_t)cd.t * BWD_NB;
  const uint32_t e0 = P.sexp[base + b_first];
  const uint32_t e1 = b_last != b_first ? P.sexp[base + b_last] : 0u;
  if (!(e0 | e1)) return;
  tzr_drain_stores();  // the record (write-through stores of this wave) before the arrivals
  const int c_first = P.tab_chunk[cd.t];
  const uint32_t* bb = P.binbase + (size_t)cd.t * (BWD_NB + 1);
  for (int side = 0; side < 2; ++side) {
    const uint32_t b = side ? b_last : b_first;
    const uint32_t expect = side ? e1 : e0;
    if (!expect) continue;
    int last = 0;
    if (lane == 0) last = tzr_arrive(P.sarr + base + b) == expect - 1 ? 1 : 0;
    last = __shfl(last, 0, TZR_WAVE);
    if (!last) continue;
      if (lane == 0) tzr_publish_u32(P.sarr + base + b, 0u);  // the plan can be applied again
    const int u0 = c_first + (int)((bb[b] - (uint32_t)cd.ts) / (uint32_t)P.ch);
    const int u1 = c_first + (int)((bb[b + 1] - 1 - (uint32_t)cd.ts) / (uint32_t)P.ch);
    for (int cb = u0; cb <= u1; cb += TZR_WAVE) {
      const int c = cb + lane;
      bool mine = false;
      if (c <= u1 && (tzr_consume_u32(P.cflags + c) & BWD_TRAIL))
        mine = bwd_bucket(tzr_consume_u32(P.ctkey + c), cd.mult) == b;  // u1 may end in another bucket
      unsigned long long open = __ballot(mine);
      while (open) {
        const int k = __ffsll(open) - 1;
        open &= open - 1;
        bwd_stitch_unit<ADAM>(tb, opt, lr, max_dim, P, cb + k, u1 + 1, lane);
      }
    }
  }
}

template <bool ADAM>
__device__ __forceinline__ void bwd_reduce_body(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    int grad_mode, const BwdGrads& G, const BwdOpt& opt, int max_dim, const BwdPlan& P) {
  __shared__ BwdApplyLds L;
  __shared__ TzrDst sG[TZR_MAX_DST];
  uint32_t* const sK = L.U.sK;  // K[s-1], K[s..e), K[e] (sentinels at table ends)
  uint32_t* const sS = L.U.sS;
  uint32_t* const rflags = L.U.rflags;
  uint32_t* const rlkey = L.U.rlkey;
  uint32_t* const rtkey = L.U.rtkey;
  float(*const rlead)[BWD_MAXDIM] = L.U.rlead;
  float(*const rtrail)[BWD_MAXDIM] = L.U.rtrail;
  const uint32_t uf = P.uflag[blockIdx.x], um = P.umix[blockIdx.x], fz = P.hcount[1];
  BwdChunkDesc cd;
  if (!bwd_chunk(P, blockIdx.x, &cd)) return;
  const int t = cd.t;
  const int64_t ts = cd.ts, te = cd.te;
  (void)ts;
  // the unit: sorted positions [s, e) of the table (pooled_bwd.hip, scan kernel)
  const int64_t s = P.ucut[blockIdx.x];
  const int64_t e = (int)blockIdx.x + 1 < cd.last_chunk ? (int64_t)P.ucut[blockIdx.x + 1] : te;
  const int n = (int)(e - s);
  const TzrTable tb = tables[t];
  if (n <= 0 || n > BWD_UMAX) {  // an empty tail unit (n > BWD_UMAX cannot happen by construction)
    if (threadIdx.x == 0) P.cflags[blockIdx.x] = 0;  // overlaps no bucket: nobody waits for it
    return;
  }
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  bwd_stage_unit(P, cd, s, e, n, fz && !cd.exact && !uf && !um, L, sG, G);

  const int lg = tb.dim >> 2;    // lanes per row
  const int gw = TZR_WAVE / lg;  // lookups per tile
  const int gi = lane / lg;
  const int c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const float lr = *opt.lr;
  const bool single = tb.n_feats == 1;
  const BwdSrc one = bwd_resolve(feats + P.feat_by_order[tb.first_order], sG);

  const int range = (n + BWD_WAVES - 1) / BWD_WAVES;  // sorted positions reduced by one wave
  const int r0 = min(n, wv * range);                  // range of this wave, unit-relative
  const int r1 = min(n, r0 + range);
  unsigned flags = 0;
  const uint32_t leadkey = r0 < r1 ? sK[r0 + 1] : BWD_SENT;
  bool lead_open = r0 < r1 && sK[r0] == leadkey;  // first run started before this range
  bool cvalid = false;                            // carry: run continuing from the previous tile
  uint32_t ckey = BWD_SENT;
  float4 csum = tzr_zero4();

  for (int t0 = r0; t0 < r1; t0 += gw) {
    const int idx = t0 + gi;
    const bool valid = lane_on && idx < r1;
    const uint32_t key = valid ? sK[idx + 1] : BWD_SENT;
    const uint32_t nxt = valid ? sK[idx + 2] : BWD_SENT;
    const bool tail = valid && key != nxt;
    float4 g = tzr_zero4();
    if (valid)
      g = bwd_lookup_grad(feats, tb, P.feat_by_order, sG, one, single, grad_mode, offsets, weights,
                          P.bag_of, B, uniform, sS[idx], c);
    const bool in_lead = lead_open && key == leadkey;
    const bool do_apply = tail && !in_lead;
    float4 w4 = tzr_zero4();
    if (do_apply)  // issued before the scan: overlaps the gradient gathers
      w4 = tzr_ldw4(reinterpret_cast<const void*>(tb.w), tb.w_dtype, (int64_t)key * tb.w_stride + 4 * c);
    const float4 m4 = bwd_load_state<ADAM>(tb, opt, (int64_t)key, c, do_apply);
    // segmented inclusive scan over the lane groups of the tile (keys are sorted, so equality at
    // distance d implies one run in between)
    for (int d = 1; d < gw; d <<= 1) {
      const uint32_t ok = __shfl_up(key, d * lg, 64);
      const float4 ov = make_float4(__shfl_up(g.x, d * lg, 64), __shfl_up(g.y, d * lg, 64),
                                    __shfl_up(g.z, d * lg, 64), __shfl_up(g.w, d * lg, 64));
      if (gi >= d && ok == key) g = tzr_add4(ov, g);
    }
    if (cvalid && key == ckey) g = tzr_add4(csum, g);  // earlier lookups first
    if (tail && in_lead) {  // the run inherited from the previous range ends here
      rlead[wv][4 * c + 0] = g.x; rlead[wv][4 * c + 1] = g.y;
      rlead[wv][4 * c + 2] = g.z; rlead[wv][4 * c + 3] = g.w;
    }
    if (__any(tail && in_lead)) {
      flags |= BWD_LEAD;
      lead_open = false;
    }
    bwd_apply_row<ADAM>(tb, opt, lr, (int64_t)key, c, g, w4, m4, do_apply, lg, c, lane);
    // carry out of the tile: its last valid lookup, if that run goes on
    const int nv = min(gw, r1 - t0);
    const int last = (nv - 1) * lg;
    ckey = __shfl(key, last, 64);
    cvalid = __shfl((int)(valid && !tail), last, 64) != 0;
    csum = bwd_shfl4(g, last + (lane_on ? c : 0));
  }
  if (r0 < r1 && cvalid) {  // the last run continues past this range
    float* dst = lead_open ? rlead[wv] : rtrail[wv];
    if (lane_on && gi == 0) {
      dst[4 * c + 0] = csum.x; dst[4 * c + 1] = csum.y; dst[4 * c + 2] = csum.z; dst[4 * c + 3] = csum.w;
    }
    flags |= lead_open ? (BWD_LEAD | BWD_LEAD_WHOLE) : BWD_TRAIL;
  }
  if (lane == 0) {
    rflags[wv] = flags;
    rlkey[wv] = leadkey;
    rtkey[wv] = ckey;
  }
  __syncthreads();

  // stitch the 4 ranges of the chunk (wave 0; control flow is wave-uniform)
  if (wv != 0) return;
  const bool on = lane < lg;
  bool open = false;
  uint32_t okey = BWD_SENT;
  float4 osum = tzr_zero4();
  unsigned cf = 0;
  float4 clead = tzr_zero4();
  for (int r = 0; r < BWD_WAVES; ++r) {
    const unsigned f = rflags[r];
    if (f & BWD_LEAD) {
      float4 lv = tzr_zero4();
      if (on) lv = make_float4(rlead[r][4 * lane], rlead[r][4 * lane + 1], rlead[r][4 * lane + 2],
                               rlead[r][4 * lane + 3]);
      if (open) {
        osum = tzr_add4(osum, lv);
        if (!(f & BWD_LEAD_WHOLE)) {
          bwd_apply_row_wave<ADAM>(tb, opt, lr, okey, osum, lane);
          open = false;
        }
      } else {  // still inside the run inherited from the previous chunk
        clead = tzr_add4(clead, lv);
        cf = BWD_LEAD | (f & BWD_LEAD_WHOLE);
      }
    }
    if (f & BWD_TRAIL) {
      open = true;
      okey = rtkey[r];
      osum = tzr_zero4();
      if (on) osum = make_float4(rtrail[r][4 * lane], rtrail[r][4 * lane + 1], rtrail[r][4 * lane + 2],
                                 rtrail[r][4 * lane + 3]);
    }
  }
  if (open) cf |= BWD_TRAIL;
  if (on && cf) {  // only units with an open boundary have a payload
    bwd_publish4(P.clead + (size_t)blockIdx.x * max_dim + 4 * lane, clead);
    bwd_publish4(P.ctrail + (size_t)blockIdx.x * max_dim + 4 * lane, osum);
  }
  if (lane == 0) {
    tzr_publish_u32(P.cflags + blockIdx.x, cf);
    tzr_publish_u32(P.clkey + blockIdx.x, sK[1]);
    tzr_publish_u32(P.ctkey + blockIdx.x, okey);
  }
  bwd_arrive_and_stitch<ADAM>(tb, opt, lr, max_dim, P, cd, bwd_bucket(sK[1], cd.mult),
                              bwd_bucket(sK[n], cd.mult), lane);
}

template <bool ADAM>
__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_reduce_kernel(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    int grad_mode, BwdGrads G, BwdOpt opt, int max_dim, BwdPlan P) {
  bwd_reduce_body<ADAM>(tables, T, feats, offsets, weights, B, uniform, grad_mode, G, opt, max_dim, P);
}

// the same body compiled for 7 / 8 waves per SIMD (72 / 64 VGPRs); tzr_tune("bwd_apply_waves") = 6 | 7 | 8, 0 = 7
__global__ __launch_bounds__(BWD_THREADS) TZR_WAVES_PER_EU(7) void tzr_bwd_reduce_w7_kernel(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    int grad_mode, BwdGrads G, BwdOpt opt, int max_dim, BwdPlan P) {
  bwd_reduce_body<false>(tables, T, feats, offsets, weights, B, uniform, grad_mode, G, opt, max_dim, P);
}
__global__ __launch_bounds__(BWD_THREADS) TZR_WAVES_PER_EU(8) void tzr_bwd_reduce_w8_kernel(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    int grad_mode, BwdGrads G, BwdOpt opt, int max_dim, BwdPlan P) {
  bwd_reduce_body<false>(tables, T, feats, offsets, weights, B, uniform, grad_mode, G, opt, max_dim, P);
}
int g_tzr_bwd_apply_waves = 0;

// ---- the same unit through bwd_reduce_unit's FAST memory side (pooled_bwd_apply.h: bwd_apply_row_fast) ----------------------
// Round 5.  The ISA of the loop above (profiles/r05ai/apply_w7_loop.s) runs a tile as a CHAIN: LDS keys -> FLAT gradient gather
// -> `s_waitcnt lgkmcnt(0)` (an LDS wait: it also waits for the FLAT gather) -> weights (a dtype branch around the load: wait)
// -> FLAT state load -> wait -> scan -> FLAT stores, whose acknowledgement the next tile's LDS reads wait for: four dependent
// round trips per 16 lookups, 16 tiles per wave -- ~4.5 us per tile whatever the data's home (tables capped at 100 k rows, i.e.
// cache resident: 78 vs 80 us; gradients cache resident: 77 vs 80, profiles/r05ah).  Units whose table is fp32, read by one key
// with one gradient buffer take bwd_reduce_unit<.., FK>: global instructions only, nothing loaded conditionally, ONE round trip
// per NT tiles; every other unit takes the general form of the same function.  Same arithmetic, same summation order.
template <int FK, int NT>
__device__ __forceinline__ void bwd_reduce_body_fast(
    const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform,
    int grad_mode, const BwdGrads& G, const BwdOpt& opt, int max_dim, const BwdPlan& P) {
  __shared__ BwdApplyLds L;
  __shared__ TzrDst sG[TZR_MAX_DST];
  BwdUnitLds& U = L.U;
  BWD_PROF_MARK(0);  // workgroup started
  // (the unit's cuts and flags are fetched WITH its descriptor, not behind it: the staging below is a chain of dependent round
  // trips -- descriptor -> cuts -> keys -> LDS, 4.9 us per workgroup in profiles/r05ai/apply_phase_profile.txt; the grid is
  // max_chunks workgroups and ucut holds max_chunks + 1 entries)
  const uint32_t u0 = P.ucut[blockIdx.x], u1 = P.ucut[blockIdx.x + 1];
  const uint32_t uf = P.uflag[blockIdx.x], um = P.umix[blockIdx.x], fz = P.hcount[1];
  BwdChunkDesc cd;
  if (!bwd_chunk(P, blockIdx.x, &cd)) return;
  const int t = cd.t;
  const int64_t te = cd.te;
  const int64_t s = u0;
  const int64_t e = (int)blockIdx.x + 1 < cd.last_chunk ? (int64_t)u1 : te;
  const int n = (int)(e - s);
  const TzrTable tb = tables[t];
  if (n <= 0 || n > BWD_UMAX) {
    if (threadIdx.x == 0) P.cflags[blockIdx.x] = 0;
    return;
  }
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const TzrFeature* const ft = feats + P.feat_by_order[tb.first_order];
  const int ft_dst = ft->n_dst;  // (issued with the unit's keys: no round trip of its own)
  bwd_stage_unit(P, cd, s, e, n, fz && !cd.exact && !uf && !um, L, sG, G);
  BWD_PROF_MARK(1);  // unit staged in LDS
  const float lr = *opt.lr;
  const int lg = tb.dim >> 2;
  auto tail = [&](unsigned cf, uint32_t okey, const float4& clead, const float4& osum) {  // wave 0: the unit's open ends
    BWD_PROF_MARK(3);  // tiles + stitch of the wave ranges done
    if (lane < lg && cf) {
      bwd_publish4(P.clead + (size_t)blockIdx.x * max_dim + 4 * lane, clead);
      bwd_publish4(P.ctrail + (size_t)blockIdx.x * max_dim + 4 * lane, osum);
    }
    if (lane == 0) {
      tzr_publish_u32(P.cflags + blockIdx.x, cf);
      tzr_publish_u32(P.clkey + blockIdx.x, U.sK[1]);
      tzr_publish_u32(P.ctkey + blockIdx.x, okey);
    }
    bwd_arrive_and_stitch<false>(tb, opt, lr, max_dim, P, cd, bwd_bucket(U.sK[1], cd.mult), bwd_bucket(U.sK[n], cd.mult), lane);
    BWD_PROF_MARK(4);  // unit boundaries done: the workgroup ends
  };
  const bool fast = tb.w_dtype == TZR_DT_F32 && (grad_mode == 1 || (tb.n_feats == 1 && ft_dst == 1));  // (workgroup-uniform)
  if (fast)
    bwd_reduce_unit<false, NT, FK>(tb, feats, P.feat_by_order, P.bag_of, offsets, weights, B, uniform, grad_mode, opt, U, sG, n, tail);
  else
    bwd_reduce_unit<false, 1, 0>(tb, feats, P.feat_by_order, P.bag_of, offsets, weights, B, uniform, grad_mode, opt, U, sG, n, tail);
}

#define TZR_REDUCE_FAST_KERNEL(NAME, FK_, NT_, ATTR)                                                                  \
  __global__ __launch_bounds__(BWD_THREADS) ATTR void NAME(                                                          \
      const TzrTable* __restrict__ tables, int T, const TzrFeature* __restrict__ feats,                              \
      const int64_t* __restrict__ offsets, const float* __restrict__ weights, int64_t B, int uniform, int grad_mode, \
      BwdGrads G, BwdOpt opt, int max_dim, BwdPlan P) {                                                               \
    bwd_reduce_body_fast<FK_, NT_>(tables, T, feats, offsets, weights, B, uniform, grad_mode, G, opt, max_dim, P);    \
  }
// one tile per round trip at 7 waves per SIMD (the whole unit grid of a 65 536-sample Criteo step resident at once) ...
TZR_REDUCE_FAST_KERNEL(tzr_bwd_reduce_fast_adagrad_kernel, TZR_OPT_ADAGRAD, 1, TZR_WAVES_PER_EU(7))
TZR_REDUCE_FAST_KERNEL(tzr_bwd_reduce_fast_rowwise_kernel, TZR_OPT_ROWWISE_ADAGRAD, 1, TZR_WAVES_PER_EU(7))
TZR_REDUCE_FAST_KERNEL(tzr_bwd_reduce_fast_sgd_kernel, TZR_OPT_SGD, 1, TZR_WAVES_PER_EU(7))
// ... or two tiles per round trip, also at 7 (tzr_tune "bwd_apply_fast" = 2)
TZR_REDUCE_FAST_KERNEL(tzr_bwd_reduce_fast2_adagrad_kernel, TZR_OPT_ADAGRAD, 2, TZR_WAVES_PER_EU(7))
TZR_REDUCE_FAST_KERNEL(tzr_bwd_reduce_fast2_rowwise_kernel, TZR_OPT_ROWWISE_ADAGRAD, 2, TZR_WAVES_PER_EU(7))
TZR_REDUCE_FAST_KERNEL(tzr_bwd_reduce_fast2_sgd_kernel, TZR_OPT_SGD, 2, TZR_WAVES_PER_EU(7))
int g_tzr_bwd_apply_fast = 0;  // tzr_tune("bwd_apply_fast"): 0 = one tile per round trip, 2 = two, -1 = the general loop only

extern "C" int tzr_pooled_bwd_apply(const TzrTable* d_tables, const TzrFeature* d_feats,
                                    int n_feats, int n_tables, int max_dim,
                                    const int64_t* d_offsets, const float* d_weights,
                                    int64_t n_values, int64_t n_positions, int64_t B,
                                    int uniform_bag_len, int grad_mode, const TzrDst* h_grads,
                                    int n_dst, const TzrSparseOptim* h_optim, void* ws,
                                    size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || !h_grads || !h_optim || n_tables <= 0 || n_feats <= 0 ||
      n_values < 0 || B < 0 || n_dst <= 0 || n_dst > TZR_MAX_DST || max_dim <= 0 ||
      max_dim > BWD_MAXDIM || (max_dim & 3) || (grad_mode != 0 && grad_mode != 1))
    return TZR_ERR_INVALID;
  const bool uniform = uniform_bag_len == 1;
  if (!uniform && !d_offsets && grad_mode == 0) return TZR_ERR_INVALID;
  if (!h_optim->d_lr) return TZR_ERR_INVALID;
  if (h_optim->kind != TZR_OPT_SGD && h_optim->kind != TZR_OPT_ADAGRAD &&
      h_optim->kind != TZR_OPT_ROWWISE_ADAGRAD && h_optim->kind != TZR_OPT_ACCUMULATE &&
      h_optim->kind != TZR_OPT_ADAM)
    return TZR_ERR_UNSUPPORTED;
  if (h_optim->kind == TZR_OPT_ADAM && !h_optim->d_adam) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  if (n_positions < 0 || n_positions >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  BwdPlan P;
  if (bwd_layout(&P, ws, n_values, n_positions, n_feats, n_tables, max_dim) > ws_bytes)
    return TZR_ERR_WORKSPACE;
  if (n_values == 0 || n_positions == 0 || B == 0) return TZR_OK;
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
  const unsigned chunks = (unsigned)P.max_chunks;
#define TZR_REDUCE_LAUNCH(K)                                                                       \
  hipLaunchKernelGGL(K, dim3(chunks), dim3(BWD_THREADS), 0, s, d_tables, n_tables, d_feats, d_offsets, \
                     d_weights, B, (int)uniform, grad_mode, G, opt, max_dim, P)
  // bags of one id with pooled gradients, or one gradient row per id (the sharded owners' and the sequence lookup's backward), no
  // per-sample weights: the shapes the fast memory side of the tile loop is written for
  const bool fast_shape = ((grad_mode == 0 && uniform) || grad_mode == 1) && !d_weights && g_tzr_bwd_apply_fast >= 0 &&
                          g_tzr_bwd_apply_waves == 0 &&
                          (opt.kind == TZR_OPT_ADAGRAD || opt.kind == TZR_OPT_ROWWISE_ADAGRAD || opt.kind == TZR_OPT_SGD);
  if (fast_shape) {
    const bool two = g_tzr_bwd_apply_fast == 2;
    if (opt.kind == TZR_OPT_ADAGRAD) {
      if (two) TZR_REDUCE_LAUNCH(tzr_bwd_reduce_fast2_adagrad_kernel); else TZR_REDUCE_LAUNCH(tzr_bwd_reduce_fast_adagrad_kernel);
    } else if (opt.kind == TZR_OPT_ROWWISE_ADAGRAD) {
      if (two) TZR_REDUCE_LAUNCH(tzr_bwd_reduce_fast2_rowwise_kernel); else TZR_REDUCE_LAUNCH(tzr_bwd_reduce_fast_rowwise_kernel);
    } else {
      if (two) TZR_REDUCE_LAUNCH(tzr_bwd_reduce_fast2_sgd_kernel); else TZR_REDUCE_LAUNCH(tzr_bwd_reduce_fast_sgd_kernel);
    }
  } else if (opt.kind == TZR_OPT_ADAM) {
    TZR_REDUCE_LAUNCH((tzr_bwd_reduce_kernel<true>));  // Adam holds two state rows per lane: no registers for a second tile
  } else if (g_tzr_bwd_apply_waves == 6) {
    TZR_REDUCE_LAUNCH((tzr_bwd_reduce_kernel<false>));
  } else if (g_tzr_bwd_apply_waves == 8) {
    TZR_REDUCE_LAUNCH(tzr_bwd_reduce_w8_kernel);
  } else {
    // 7 waves per SIMD = 1792 workgroups resident: the whole unit grid of a B = 65536 Criteo step (1691) runs in
    // one wave of workgroups.  At 6 (77 VGPRs, what the compiler picks unasked) the last 155 units waited for a
    // free slot and finished a full workgroup time after the rest: 88.4 -> 78.9 us (row-wise Adagrad 92.9 -> 78.6),
    // profiles/r03n.  8 waves (64 VGPRs) spills: 88.0 us.
    TZR_REDUCE_LAUNCH(tzr_bwd_reduce_w7_kernel);
  }
#undef TZR_REDUCE_LAUNCH
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// Dense update of replicated tables: lane group <-> row of the concatenated row space; the row's
// gradient comes from the all-reduced accumulation buffer; rows with an all-zero gradient are
// skipped (a sparse update never visits them).  Same per-row arithmetic as the sparse path.
template <bool ADAM>
__global__ __launch_bounds__(BWD_THREADS) void tzr_dense_rows_update_kernel(
    const TzrTable* __restrict__ tables, int T, const int64_t* __restrict__ row_start,
    int64_t total_rows, float* __restrict__ acc, int dim, BwdOpt opt, int clear) {
  const int lg = dim >> 2;
  const int gw = TZR_WAVE / lg;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int gi = lane / lg;
  const int c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const int gpb = gw * BWD_WAVES;
  const float lr = *opt.lr;
  const int64_t iters = (total_rows + (int64_t)gridDim.x * gpb - 1) / ((int64_t)gridDim.x * gpb);
  for (int64_t it = 0; it < iters; ++it) {  // uniform trip count: the row update is wave-collective
    const int64_t r = (it * gridDim.x + blockIdx.x) * gpb + wv * gw + gi;
    const bool valid = lane_on && r < total_rows;
    float4 g = tzr_zero4();
    if (valid) g = tzr_ld4(acc + r * (int64_t)dim + 4 * c);
    float nz = (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f) ? 1.f : 0.f;
    nz = bwd_group_sum(nz, lg, c, lane);
    const bool active = valid && nz > 0.f;
    // `clear`: leave the accumulation buffer zero behind the update -- the next step's ACCUMULATE pass writes only the rows
    // it touches, and a whole-buffer memset in front of it was a launch of its own (4 us of a 0.4 ms sharded step)
    if (clear && active) tzr_st4(acc + r * (int64_t)dim + 4 * c, tzr_zero4());
    int t = 0;
    int64_t row = 0;
    if (valid) {
      t = (int)tzr_last_le(row_start, T, r);
      row = r - row_start[t];
    }
    const TzrTable tb = tables[t];
    float4 w4 = tzr_zero4();
    if (active) w4 = tzr_ldw4(reinterpret_cast<const void*>(tb.w), tb.w_dtype, row * (int64_t)tb.w_stride + 4 * c);
    const float4 m4 = bwd_load_state<ADAM>(tb, opt, row, c, active);
    bwd_apply_row<ADAM>(tb, opt, lr, row, c, g, w4, m4, active, lg, c, lane);
  }
}

static int dense_rows_update(const TzrTable* d_tables, int n_tables, const int64_t* d_row_start, int64_t total_rows,
                             float* d_acc, int dim, const TzrSparseOptim* h_optim, int clear, void* stream);

extern "C" int tzr_dense_rows_update(const TzrTable* d_tables, int n_tables,
                                     const int64_t* d_row_start, int64_t total_rows,
                                     const float* d_acc, int dim, const TzrSparseOptim* h_optim,
                                     void* stream) {
  return dense_rows_update(d_tables, n_tables, d_row_start, total_rows, const_cast<float*>(d_acc), dim, h_optim, 0, stream);
}

extern "C" int tzr_dense_rows_update_clear(const TzrTable* d_tables, int n_tables,
                                           const int64_t* d_row_start, int64_t total_rows, float* d_acc,
                                           int dim, const TzrSparseOptim* h_optim, void* stream) {
  return dense_rows_update(d_tables, n_tables, d_row_start, total_rows, d_acc, dim, h_optim, 1, stream);
}

static int dense_rows_update(const TzrTable* d_tables, int n_tables, const int64_t* d_row_start, int64_t total_rows,
                             float* d_acc, int dim, const TzrSparseOptim* h_optim, int clear, void* stream) {
  if (!d_tables || n_tables <= 0 || !d_row_start || total_rows < 0 || !h_optim || !h_optim->d_lr ||
      dim <= 0 || (dim & 3) || dim > BWD_MAXDIM)
    return TZR_ERR_INVALID;
  if (h_optim->kind == TZR_OPT_ADAM && !h_optim->d_adam) return TZR_ERR_INVALID;
  if (h_optim->kind != TZR_OPT_SGD && h_optim->kind != TZR_OPT_ADAGRAD &&
      h_optim->kind != TZR_OPT_ROWWISE_ADAGRAD && h_optim->kind != TZR_OPT_ADAM)
    return TZR_ERR_UNSUPPORTED;
  if (total_rows == 0) return TZR_OK;
  if (!d_acc || (reinterpret_cast<uintptr_t>(d_acc) & 15)) return TZR_ERR_INVALID;
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
  const int gpb = (TZR_WAVE / (dim >> 2)) * BWD_WAVES;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, (total_rows + gpb - 1) / gpb);
  if (opt.kind == TZR_OPT_ADAM) {
    hipLaunchKernelGGL((tzr_dense_rows_update_kernel<true>), dim3(grid), dim3(BWD_THREADS), 0,
                       static_cast<hipStream_t>(stream), d_tables, n_tables, d_row_start, total_rows,
                       d_acc, dim, opt, clear);
  } else {
    hipLaunchKernelGGL((tzr_dense_rows_update_kernel<false>), dim3(grid), dim3(BWD_THREADS), 0,
                       static_cast<hipStream_t>(stream), d_tables, n_tables, d_row_start, total_rows,
                       d_acc, dim, opt, clear);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- sparse Adam step counter --------------------------------------------------------------------
__global__ void tzr_sparse_adam_tick_kernel(float* st, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float t = st[0] + 1.0f;
    st[0] = t;
    st[1] = 1.0f - powf(beta1, t);
    st[2] = 1.0f - powf(beta2, t);
  }
}

extern "C" int tzr_sparse_adam_tick(float* d_adam, float beta1, float beta2, void* stream) {
  if (!d_adam || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return TZR_ERR_INVALID;
  hipLaunchKernelGGL(tzr_sparse_adam_tick_kernel, dim3(1), dim3(TZR_WAVE), 0,
                     static_cast<hipStream_t>(stream), d_adam, beta1, beta2);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
