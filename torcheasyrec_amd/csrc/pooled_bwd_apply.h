// Shared between the planned apply (pooled_bwd_apply.hip, K7) and the one-launch backward of small batches
// (pooled_bwd_direct.hip): gradient addressing of a lookup, the per-row optimizer arithmetic, and the reduction of one
// unit of sorted lookups held in LDS (wave-level segmented sums, single read-modify-write per row).
#pragma once
#include "pooled_bwd.h"

struct BwdGrads {
  TzrDst d[TZR_MAX_DST];
};

struct BwdOpt {
  int kind, wd_mode, clip;
  const float* lr;
  float eps, wd, max_grad;
  float beta1, beta2;
  const float* adam;  // {step, 1 - beta1^step, 1 - beta2^step}
};

// Gradient sources of one lookup (key -> table), resolved once per workgroup when the table is
// read by a single key (the common case).
// (scalar fields, no arrays: a runtime-indexed array in this struct lands in scratch memory and
// turned into 4x write amplification on the first version of this kernel -- profiles/r01b)
struct BwdSrc {
  const float *gp0, *gp1, *gp2, *gp3;  // group gradient buffer + first column
  int64_t gs0, gs1, gs2, gs3;          // sample stride
  int n_dst;
  int mean;
};

// `sG` = the gradient-buffer descriptors copied to LDS once per workgroup: run-time selection
// indexes LDS, never a private copy of the kernel arguments.
__device__ __forceinline__ BwdSrc bwd_resolve(const TzrFeature* __restrict__ ft,
                                              const TzrDst* sG) {
  BwdSrc s;
  const int n = ft->n_dst;
  s.n_dst = n;
  s.mean = ft->pooling == TZR_POOL_MEAN;
  const int d0 = n > 0 ? ft->dst[0] : 0, d1 = n > 1 ? ft->dst[1] : 0;
  const int d2 = n > 2 ? ft->dst[2] : 0, d3 = n > 3 ? ft->dst[3] : 0;
  s.gp0 = reinterpret_cast<const float*>(sG[d0].ptr) + ft->col[0];
  s.gp1 = reinterpret_cast<const float*>(sG[d1].ptr) + ft->col[1];
  s.gp2 = reinterpret_cast<const float*>(sG[d2].ptr) + ft->col[2];
  s.gp3 = reinterpret_cast<const float*>(sG[d3].ptr) + ft->col[3];
  s.gs0 = sG[d0].stride;
  s.gs1 = sG[d1].stride;
  s.gs2 = sG[d2].stride;
  s.gs3 = sG[d3].stride;
  return s;
}

// dL/d(row contribution) of the lookup at original position i, float4 chunk c of its row.
//   grad_mode 0: pooled-output gradients per feature group (bag (key,b) -> grad[g][b, col..])
//   grad_mode 1: one gradient row per id: G.d[0][i, :]
__device__ __forceinline__ float4 bwd_lookup_grad(
    const TzrFeature* __restrict__ feats, const TzrTable& tb, const int32_t* __restrict__ feat_by_order,
    const TzrDst* sG, const BwdSrc& one, bool single, int grad_mode,
    const int64_t* __restrict__ offsets, const float* __restrict__ weights,
    const uint32_t* __restrict__ bag_of, int64_t B, int uniform, uint32_t i, int c) {
  if (grad_mode == 1)
    return tzr_ld4(reinterpret_cast<const float*>(sG[0].ptr) + (int64_t)i * sG[0].stride + 4 * c);
  const uint32_t bag = uniform ? i : bag_of[i];
  const uint32_t key = bag / (uint32_t)B;
  const int64_t b = bag - key * (uint32_t)B;
  BwdSrc s = one;
  if (!single) {  // the lookup of this table that reads `key` (a table is read once per key)
    int o = tb.first_order;
    while (o + 1 < tb.first_order + tb.n_feats && feats[feat_by_order[o]].key != (int32_t)key) ++o;
    s = bwd_resolve(feats + feat_by_order[o], sG);
  }
  float4 g = tzr_ld4(s.gp0 + b * s.gs0 + 4 * c);
  if (s.n_dst > 1) g = tzr_add4(g, tzr_ld4(s.gp1 + b * s.gs1 + 4 * c));
  if (s.n_dst > 2) g = tzr_add4(g, tzr_ld4(s.gp2 + b * s.gs2 + 4 * c));
  if (s.n_dst > 3) g = tzr_add4(g, tzr_ld4(s.gp3 + b * s.gs3 + 4 * c));
  const bool mean = !uniform && s.mean;
  if (weights || mean) {
    float sc = weights ? weights[i] : 1.0f;
    if (mean) {
      const int64_t len = offsets[(int64_t)bag + 1] - offsets[bag];
      if (len > 1) sc = sc / (float)len;
    }
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

// Prefetch of the elementwise optimizer state of (row, chunk c): issued together with the weight
// load, before the reduction, so the update itself waits on no memory.
// (ADAM is a template parameter of everything below: with the Adam arithmetic as one more run-time
// branch of the row update the reduce kernel needed 99 instead of 78 VGPRs, one wave per SIMD less,
// and the DLRM-Criteo Adagrad step lost 17 us -- profiles/r01k.  The <false> instantiations are the
// code that was there before.)
template <bool ADAM>
__device__ __forceinline__ float4 bwd_load_state(const TzrTable& tb, const BwdOpt& opt, int64_t row,
                                                 int c, bool active) {
  if (active && (ADAM || opt.kind == TZR_OPT_ADAGRAD))  // Adam: exp_avg
    return tzr_ld4(reinterpret_cast<const float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c);
  // row-wise Adagrad: the row's scalar, fetched by the group's first lane (c == lane in group at every call site)
  // together with the weights -- not after the gradient reduction, where its latency was exposed once per run
  if (!ADAM && active && c == 0 && opt.kind == TZR_OPT_ROWWISE_ADAGRAD)
    return make_float4(reinterpret_cast<const float*>(tb.m)[row * (int64_t)tb.m_stride], 0.f, 0.f, 0.f);
  return tzr_zero4();
}

// The same with NO lane-dependent condition around the load (every lane of the wave loads; the caller passes a row that
// exists for the lanes it will not use): a load inside `if (active)` is a branch to hipcc and a branch between two loads a
// wait between them -- the gradient gather, the weights and the state of a tile ran as three dependent round trips
// (pooled_bwd_direct.hip: "branch-free").  Row-wise Adagrad: all lanes of the group read the row's scalar (one address).
template <bool ADAM>
__device__ __forceinline__ float4 bwd_load_state_all(const TzrTable& tb, const BwdOpt& opt, int64_t row, int c) {
  if (ADAM || opt.kind == TZR_OPT_ADAGRAD)  // (kernel-uniform)
    return tzr_ld4(reinterpret_cast<const float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c);
  if (!ADAM && opt.kind == TZR_OPT_ROWWISE_ADAGRAD)
    return make_float4(reinterpret_cast<const float*>(tb.m)[row * (int64_t)tb.m_stride], 0.f, 0.f, 0.f);
  return tzr_zero4();
}

// ONE update of row `row`, chunk c; `active` lanes hold the summed gradient g, the row's current
// weights w4 and (elementwise adagrad) state m4.  All 64 lanes of the wave must call (row-wise
// adagrad reduces in the group).
template <bool ADAM>
__device__ __forceinline__ void bwd_apply_row(const TzrTable& tb, const BwdOpt& opt, float lr,
                                              int64_t row, int c, float4 g, float4 w4, float4 m4,
                                              bool active, int lg, int lane_in_group, int lane) {
  if (opt.clip) {
    g.x = fminf(fmaxf(g.x, -opt.max_grad), opt.max_grad);
    g.y = fminf(fmaxf(g.y, -opt.max_grad), opt.max_grad);
    g.z = fminf(fmaxf(g.z, -opt.max_grad), opt.max_grad);
    g.w = fminf(fmaxf(g.w, -opt.max_grad), opt.max_grad);
  }
  void* const wbase = reinterpret_cast<void*>(tb.w);
  const int64_t woff = row * (int64_t)tb.w_stride + 4 * c;
  if constexpr (ADAM) {
    // fbgemm split Adam [upstream]: m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2,
    // w -= lr * ((m / (1-b1^t)) / (sqrt(v / (1-b2^t)) + eps) + wd * w); only touched rows move
    if (active) {
      float* mp = reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c;
      float* vp = mp + tb.dim;
      float4 v4 = tzr_ld4(vp);
      const float b1 = opt.beta1, b2 = opt.beta2;
      const float c1 = opt.adam[1], c2 = opt.adam[2];
      m4.x = b1 * m4.x + (1.0f - b1) * g.x; m4.y = b1 * m4.y + (1.0f - b1) * g.y;
      m4.z = b1 * m4.z + (1.0f - b1) * g.z; m4.w = b1 * m4.w + (1.0f - b1) * g.w;
      v4.x = b2 * v4.x + (1.0f - b2) * g.x * g.x; v4.y = b2 * v4.y + (1.0f - b2) * g.y * g.y;
      v4.z = b2 * v4.z + (1.0f - b2) * g.z * g.z; v4.w = b2 * v4.w + (1.0f - b2) * g.w * g.w;
      tzr_st4(mp, m4);
      tzr_st4(vp, v4);
      w4.x -= lr * ((m4.x / c1) / (sqrtf(v4.x / c2) + opt.eps) + opt.wd * w4.x);
      w4.y -= lr * ((m4.y / c1) / (sqrtf(v4.y / c2) + opt.eps) + opt.wd * w4.y);
      w4.z -= lr * ((m4.z / c1) / (sqrtf(v4.z / c2) + opt.eps) + opt.wd * w4.z);
      w4.w -= lr * ((m4.w / c1) / (sqrtf(v4.w / c2) + opt.eps) + opt.wd * w4.w);
      tzr_stw4(wbase, tb.w_dtype, woff, w4);
    }
    return;
  }
  if (opt.kind == TZR_OPT_ADAGRAD) {
    if (active) {
      float* mp = reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c;
      m4.x += g.x * g.x; m4.y += g.y * g.y; m4.z += g.z * g.z; m4.w += g.w * g.w;
      tzr_st4(mp, m4);
      w4.x -= lr * g.x / (sqrtf(m4.x) + opt.eps);
      w4.y -= lr * g.y / (sqrtf(m4.y) + opt.eps);
      w4.z -= lr * g.z / (sqrtf(m4.z) + opt.eps);
      w4.w -= lr * g.w / (sqrtf(m4.w) + opt.eps);
      tzr_stw4(wbase, tb.w_dtype, woff, w4);
    }
  } else if (opt.kind == TZR_OPT_ROWWISE_ADAGRAD) {
    float4 gl = g;
    if (opt.wd_mode == TZR_WD_L2) gl = tzr_fma4(opt.wd, w4, g);
    float ss = active ? (gl.x * gl.x + gl.y * gl.y + gl.z * gl.z + gl.w * gl.w) : 0.f;
    ss = bwd_group_sum(ss, lg, lane_in_group, lane);
    // the row's scalar state is read by the group's first lane only and broadcast, so no lane
    // can observe the store below
    float* mp = reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride;
    float mold = (active && lane_in_group == 0) ? m4.x : 0.f;  // loaded by bwd_load_state
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
      tzr_stw4(wbase, tb.w_dtype, woff, w4);
      if (lane_in_group == 0) *mp = mnew;
    }
  } else if (opt.kind == TZR_OPT_ACCUMULATE) {
    // replicated table: hand the summed row gradient to the all-reduce (tb.m = dense [rows, dim])
    if (active) tzr_st4(reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c, g);
  } else {  // SGD
    if (active) {
      w4.x -= lr * g.x; w4.y -= lr * g.y; w4.z -= lr * g.z; w4.w -= lr * g.w;
      tzr_stw4(wbase, tb.w_dtype, woff, w4);
    }
  }
}

// The FAST form of the tile loop's memory side (bwd_reduce_unit<.., FK != 0>): fp32 tables read by ONE key whose gradient comes
// from ONE buffer, bags of exactly one id, no per-sample weights, the optimizer kind FK known at compile time.  Every access is
// a GLOBAL instruction through a pointer that is known to be device memory (the general forms go through descriptor pointers:
// FLAT instructions, which count in the LDS counter as well -- every `s_waitcnt lgkmcnt(0)` in front of an LDS read then waits
// for the gradient gather and the row stores in flight) and nothing is loaded conditionally (a load inside `if` is a branch, a
// branch between two loads a wait): the ISA of the general loop ran the gradient gather, the weights, the state and the stores
// of a tile as FOUR dependent round trips (profiles/r05ai), this one as one.
template <int FK>
__device__ __forceinline__ void bwd_apply_row_fast(const TzrTable& tb, const BwdOpt& opt, float lr, int64_t row, int c, float4 g,
                                                   float4 w4, float4 m4, bool active, int lg, int lane_in_group, int lane) {
  if (opt.clip) {
    g.x = fminf(fmaxf(g.x, -opt.max_grad), opt.max_grad);
    g.y = fminf(fmaxf(g.y, -opt.max_grad), opt.max_grad);
    g.z = fminf(fmaxf(g.z, -opt.max_grad), opt.max_grad);
    g.w = fminf(fmaxf(g.w, -opt.max_grad), opt.max_grad);
  }
  float* const wp = reinterpret_cast<float*>(tb.w) + row * (int64_t)tb.w_stride + 4 * c;
  if constexpr (FK == TZR_OPT_ADAGRAD) {
    if (active) {
      m4.x += g.x * g.x; m4.y += g.y * g.y; m4.z += g.z * g.z; m4.w += g.w * g.w;
      tzr_stg4(reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride + 4 * c, m4);
      w4.x -= lr * g.x / (sqrtf(m4.x) + opt.eps);
      w4.y -= lr * g.y / (sqrtf(m4.y) + opt.eps);
      w4.z -= lr * g.z / (sqrtf(m4.z) + opt.eps);
      w4.w -= lr * g.w / (sqrtf(m4.w) + opt.eps);
      tzr_stg4(wp, w4);
    }
  } else if constexpr (FK == TZR_OPT_ROWWISE_ADAGRAD) {
    float4 gl = g;
    if (opt.wd_mode == TZR_WD_L2) gl = tzr_fma4(opt.wd, w4, g);
    float ss = active ? (gl.x * gl.x + gl.y * gl.y + gl.z * gl.z + gl.w * gl.w) : 0.f;
    ss = bwd_group_sum(ss, lg, lane_in_group, lane);
    if (active) {  // (every lane of the group loaded the row's scalar itself: m4.x)
      const float mnew = m4.x + ss / (float)tb.dim;
      const float mult = lr / (sqrtf(mnew) + opt.eps);
      float corr = 1.0f;
      if (opt.wd_mode == TZR_WD_L2) corr = 1.0f - mult * opt.wd;
      else if (opt.wd_mode == TZR_WD_DECOUPLE) corr = 1.0f - lr * opt.wd;
      w4.x = corr * w4.x - mult * g.x;
      w4.y = corr * w4.y - mult * g.y;
      w4.z = corr * w4.z - mult * g.z;
      w4.w = corr * w4.w - mult * g.w;
      tzr_stg4(wp, w4);
      if (lane_in_group == 0) tzr_stg(reinterpret_cast<float*>(tb.m) + row * (int64_t)tb.m_stride, mnew);
    }
  } else {  // SGD
    if (active) {
      w4.x -= lr * g.x; w4.y -= lr * g.y; w4.z -= lr * g.z; w4.w -= lr * g.w;
      tzr_stg4(wp, w4);
    }
  }
}

__device__ __forceinline__ float4 bwd_shfl4(float4 v, int src) {
  return make_float4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64),
                     __shfl(v.w, src, 64));
}

// One row update done by a whole wave acting as a single group (lanes >= D/4 idle): used by the
// stitching steps, where runs are few.
template <bool ADAM>
__device__ __forceinline__ void bwd_apply_row_wave(const TzrTable& tb, const BwdOpt& opt, float lr,
                                                   uint32_t key, float4 g, int lane) {
  const bool on = lane < (tb.dim >> 2);
  float4 w4 = tzr_zero4();
  if (on) w4 = tzr_ldw4(reinterpret_cast<const void*>(tb.w), tb.w_dtype, (int64_t)key * tb.w_stride + 4 * lane);
  const float4 m4 = bwd_load_state<ADAM>(tb, opt, (int64_t)key, lane, on);
  bwd_apply_row<ADAM>(tb, opt, lr, (int64_t)key, lane, g, w4, m4, on, TZR_WAVE, lane, lane);
}

// Boundary record of a unit: written by wave 0 of its workgroup, read by whichever workgroup
// stitches the table (another CU, usually another XCD): agent-scope stores and loads (tzr_gfx950.h).
__device__ __forceinline__ void bwd_publish4(float* p, float4 v) {
  uint64_t* q = reinterpret_cast<uint64_t*>(p);
  tzr_publish_u64(q, ((uint64_t)__float_as_uint(v.y) << 32) | __float_as_uint(v.x));
  tzr_publish_u64(q + 1, ((uint64_t)__float_as_uint(v.w) << 32) | __float_as_uint(v.z));
}
__device__ __forceinline__ float4 bwd_consume4(const float* p) {
  const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
  const uint64_t a = tzr_consume_u64(q), b = tzr_consume_u64(q + 1);
  return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)),
                     __uint_as_float((uint32_t)b), __uint_as_float((uint32_t)(b >> 32)));
}

// LDS of one unit's reduction: the sorted keys with one neighbour (or BWD_SENT) on either side, the lookup positions, and
// the per-wave boundary records the workgroup's wave 0 stitches.
struct BwdUnitLds {
  uint32_t sK[BWD_UMAX + 2];  // K[s-1], K[s..e), K[e] (sentinels at table ends)
  uint32_t sS[BWD_UMAX];
  uint32_t rflags[BWD_WAVES], rlkey[BWD_WAVES], rtkey[BWD_WAVES];
  float rlead[BWD_WAVES][BWD_MAXDIM], rtrail[BWD_WAVES][BWD_MAXDIM];
};

// The reduction of ONE unit held in LDS (U.sK[1 .. n], U.sS[0 .. n), neighbours in U.sK[0] / U.sK[n + 1]; sG = the
// gradient-buffer descriptors, in LDS as well; a barrier behind the fills): every wave reduces its quarter and updates the
// rows whose runs end inside it, wave 0 then stitches the runs that cross wave ranges and hands what stays open at the
// unit's two ends to `tail(cf, okey, clead, osum)` (wave 0 only: cf = BWD_LEAD / BWD_LEAD_WHOLE / BWD_TRAIL, clead = the
// piece of the run inherited from the unit before, okey / osum = the run that continues past the unit).  All threads of the
// workgroup call; waves 1.. return behind the barrier in front of the stitch.
//
// Same arithmetic and the same summation order as `bwd_reduce_body` of pooled_bwd_apply.hip (segmented scan inside a
// tile, carry across tiles, wave records), restated for the one-launch backward of small batches (pooled_bwd_direct.hip)
// with the memory side turned around: that kernel runs grids of ~1 700 units at 7 waves per SIMD and walks its tiles one
// round trip after the other (two tiles in flight cost it a wave of occupancy: 114 vs 85 us, NOTES.md); a small batch has
// fewer workgroups than the chip has slots, so here NT tiles are in flight TOGETHER -- gradient rows, weights and state of
// 64 lookups per wave in one round trip instead of four -- and the up to three runs wave 0 closes are loaded together too.
// (The predicates of the loads are known up front: equal keys are adjacent, so "this lookup belongs to the run inherited
// from the range before" is `key == leadkey` for the whole range.)  The planned apply keeps its own copy: it is compiled
// for exactly 7 waves per SIMD (71 of 72 VGPRs) and any re-arrangement of its source moved live ranges into scratch.
template <bool ADAM, int NT, int FK = 0, class Tail>
__device__ __forceinline__ void bwd_reduce_unit(
    const TzrTable& tb, const TzrFeature* __restrict__ feats, const int32_t* __restrict__ feat_by_order,
    const uint32_t* __restrict__ bag_of, const int64_t* __restrict__ offsets, const float* __restrict__ weights,
    int64_t B, int uniform, int grad_mode, const BwdOpt& opt, BwdUnitLds& U, const TzrDst* sG, int n, Tail&& tail) {
  uint32_t* const sK = U.sK;
  uint32_t* const sS = U.sS;
  uint32_t* const rflags = U.rflags;
  uint32_t* const rlkey = U.rlkey;
  uint32_t* const rtkey = U.rtkey;
  float(*const rlead)[BWD_MAXDIM] = U.rlead;
  float(*const rtrail)[BWD_MAXDIM] = U.rtrail;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int lg = tb.dim >> 2;    // lanes per row
  const int gw = TZR_WAVE / lg;  // lookups per tile
  const int gi = lane / lg;
  const int c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const float lr = *opt.lr;
  const bool single = tb.n_feats == 1;
  const BwdSrc one = bwd_resolve(feats + feat_by_order[tb.first_order], sG);
  // FK != 0 (see bwd_apply_row_fast): lookup position i of the table's one key is bag (key, b = i - key B) with ONE gradient row
  // (grad_mode 1: one gradient row per lookup position, whatever the table's keys)
  const float* const fgp = grad_mode == 1 ? reinterpret_cast<const float*>(sG[0].ptr) : one.gp0;
  const int64_t fgs = grad_mode == 1 ? sG[0].stride : one.gs0;
  const uint32_t fkb = (FK != 0 && grad_mode == 0) ? (uint32_t)feats[feat_by_order[tb.first_order]].key * (uint32_t)B : 0u;
  const float* const fwp = reinterpret_cast<const float*>(tb.w);
  const float* const fmp = reinterpret_cast<const float*>(tb.m);

  const int range = (n + BWD_WAVES - 1) / BWD_WAVES;  // sorted positions reduced by one wave
  const int r0 = min(n, wv * range);                  // range of this wave, unit-relative
  const int r1 = min(n, r0 + range);
  unsigned flags = 0;
  const uint32_t leadkey = r0 < r1 ? sK[r0 + 1] : BWD_SENT;
  const bool lead0 = r0 < r1 && sK[r0] == leadkey;  // first run started before this range
  bool lead_open = lead0;
  bool cvalid = false;                              // carry: run continuing from the previous tile
  uint32_t ckey = BWD_SENT;
  float4 csum = tzr_zero4();

  for (int t0 = r0; t0 < r1; t0 += NT * gw) {
    uint32_t key[NT];
    float4 g[NT], w4[NT], m4[NT];
    unsigned vmask = 0, tmask = 0, lmask = 0;  // per tile u: valid lookup / last of its run / in the inherited run
    // every load of the NT tiles first -- UNCONDITIONALLY: lanes without a lookup (the range's tail, idle lane groups) read
    // the range's last lookup and its row instead, lookups that are not the last of their run read their row all the
    // same (an L2 hit next to the run's last lookup); what is used is decided by the masks below
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int idx = t0 + u * gw + gi;
      const bool valid = lane_on && idx < r1;
      const int idc = valid ? idx : r1 - 1;  // (r0 < r1 inside this loop)
      const uint32_t kc = sK[idc + 1];
      const uint32_t nxt = sK[idc + 2];
      key[u] = valid ? kc : BWD_SENT;
      const bool tl = valid && kc != nxt;
      const bool inl = lead0 && kc == leadkey;
      if constexpr (FK != 0) {
        g[u] = tzr_ldg4(fgp + (int64_t)(sS[idc] - fkb) * fgs + 4 * c);
        w4[u] = tzr_ldg4(fwp + (int64_t)kc * tb.w_stride + 4 * c);
        if constexpr (FK == TZR_OPT_ADAGRAD) m4[u] = tzr_ldg4(fmp + (int64_t)kc * tb.m_stride + 4 * c);
        else if constexpr (FK == TZR_OPT_ROWWISE_ADAGRAD) m4[u] = make_float4(tzr_ldg(fmp + (int64_t)kc * tb.m_stride), 0.f, 0.f, 0.f);
        else m4[u] = tzr_zero4();
      } else {
        g[u] = bwd_lookup_grad(feats, tb, feat_by_order, sG, one, single, grad_mode, offsets, weights, bag_of, B, uniform,
                               sS[idc], c);
        w4[u] = tzr_ldw4(reinterpret_cast<const void*>(tb.w), tb.w_dtype, (int64_t)kc * tb.w_stride + 4 * c);
        m4[u] = bwd_load_state_all<ADAM>(tb, opt, (int64_t)kc, c);
      }
      vmask |= valid ? 1u << u : 0u;
      tmask |= tl ? 1u << u : 0u;
      lmask |= inl ? 1u << u : 0u;
    }
#pragma unroll
    for (int u = 0; u < NT; ++u)
      if (!((vmask >> u) & 1u)) g[u] = tzr_zero4();
    // ... then the tiles in order: scan, carry, records, update
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (t0 + u * gw >= r1) break;  // wave-uniform
      const bool valid = (vmask >> u) & 1u, tl = (tmask >> u) & 1u, in_lead = (lmask >> u) & 1u;
      const bool do_apply = tl && !in_lead;
      float4 gg = g[u];
      // segmented inclusive scan over the lane groups of the tile (equal keys are adjacent, so equality at
      // distance d implies one run in between)
      for (int d = 1; d < gw; d <<= 1) {
        const uint32_t ok = __shfl_up(key[u], d * lg, 64);
        const float4 ov = make_float4(__shfl_up(gg.x, d * lg, 64), __shfl_up(gg.y, d * lg, 64),
                                      __shfl_up(gg.z, d * lg, 64), __shfl_up(gg.w, d * lg, 64));
        if (gi >= d && ok == key[u]) gg = tzr_add4(ov, gg);
      }
      if (cvalid && key[u] == ckey) gg = tzr_add4(csum, gg);  // earlier lookups first
      if (tl && in_lead) {  // the run inherited from the previous range ends here
        rlead[wv][4 * c + 0] = gg.x; rlead[wv][4 * c + 1] = gg.y;
        rlead[wv][4 * c + 2] = gg.z; rlead[wv][4 * c + 3] = gg.w;
      }
      if (__any(tl && in_lead)) {
        flags |= BWD_LEAD;
        lead_open = false;
      }
      if constexpr (FK != 0) bwd_apply_row_fast<FK>(tb, opt, lr, (int64_t)key[u], c, gg, w4[u], m4[u], do_apply, lg, c, lane);
      else bwd_apply_row<ADAM>(tb, opt, lr, (int64_t)key[u], c, gg, w4[u], m4[u], do_apply, lg, c, lane);
      // carry out of the tile: its last valid lookup, if that run goes on
      const int nv = min(gw, r1 - (t0 + u * gw));
      const int last = (nv - 1) * lg;
      ckey = __shfl(key[u], last, 64);
      cvalid = __shfl((int)(valid && !tl), last, 64) != 0;
      csum = bwd_shfl4(gg, last + (lane_on ? c : 0));
    }
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
#ifdef BWD_PROF_MARK
  BWD_PROF_MARK(2);  // this wave's tiles done
#endif
  __syncthreads();

  // stitch the 4 ranges of the chunk (wave 0; control flow is wave-uniform).  Pass 1 walks the records and notes the runs
  // that close inside the unit (at most one per boundary between two ranges); pass 2 loads their rows TOGETHER and
  // updates them.
  if (wv != 0) return;
  const bool on = lane < lg;
  bool open = false;
  uint32_t okey = BWD_SENT;
  float4 osum = tzr_zero4();
  unsigned cf = 0;
  float4 clead = tzr_zero4();
  uint32_t ck[BWD_WAVES - 1];
  float4 cs[BWD_WAVES - 1];
  int nc = 0;
#pragma unroll
  for (int r = 0; r < BWD_WAVES; ++r) {
    const unsigned f = rflags[r];
    if (f & BWD_LEAD) {
      float4 lv = tzr_zero4();
      if (on) lv = make_float4(rlead[r][4 * lane], rlead[r][4 * lane + 1], rlead[r][4 * lane + 2],
                               rlead[r][4 * lane + 3]);
      if (open) {
        osum = tzr_add4(osum, lv);
        if (!(f & BWD_LEAD_WHOLE)) {
#pragma unroll
          for (int q = 0; q < BWD_WAVES - 1; ++q)  // (static indices: the arrays stay in registers)
            if (q == nc) {
              ck[q] = okey;
              cs[q] = osum;
            }
          ++nc;
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
  {
    float4 cw[BWD_WAVES - 1], cm[BWD_WAVES - 1];
#pragma unroll
    for (int q = 0; q < BWD_WAVES - 1; ++q) {
      const bool act = on && q < nc;
      cw[q] = tzr_zero4();
      if (act) cw[q] = tzr_ldw4(reinterpret_cast<const void*>(tb.w), tb.w_dtype, (int64_t)ck[q] * tb.w_stride + 4 * lane);
      cm[q] = bwd_load_state<ADAM>(tb, opt, (int64_t)(q < nc ? ck[q] : 0u), lane, act);
    }
#pragma unroll
    for (int q = 0; q < BWD_WAVES - 1; ++q) {
      if (q >= nc) break;  // wave-uniform
      bwd_apply_row<ADAM>(tb, opt, lr, (int64_t)ck[q], lane, cs[q], cw[q], cm[q], on, TZR_WAVE, lane, lane);
    }
  }
  if (open) cf |= BWD_TRAIL;
  tail(cf, okey, clead, osum);
}
