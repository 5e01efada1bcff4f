// The mixing step of a multi-gate mixture of experts (/root/reference/tzrec/modules/mmoe.py:63-76):
//
//     out_t[b, :] = sum_e softmax(logits_t[b, :])_e * expert_e[b, :]          for every task t
//
// The reference stacks the expert outputs ([B, E, H], a copy), and per task runs softmax, unsqueeze and a batched matmul of
// B products [1, E] x [E, H]; autograd adds a batched product each for the gate and the experts, the softmax backward, the
// sum of the tasks' expert gradients and the un-stacking copies.  At B = 8192, E = 3, H = 128 that is ~20 launches of 5-19 us
// for 12 MB of data (profiles/r05y: the three strided-batched GEMMs alone 18.6 + 14.1 + 11.8 us per task).  Here: ONE launch per
// direction over all tasks; every expert row is read once for all tasks, nothing is stacked, the probabilities are kept for
// the backward ([B, E] per task).
//
// Mapping: a group of lgp = 2^k >= H / 4 lanes (at most 64) per row, a lane its float4 column(s); the softmax of a row's E
// logits is computed redundantly by every lane of the group (E <= 8 cached loads); the backward's E x T dot products
// <g_t[b, :], expert_e[b, :]> are reduced over the group by shuffles.  HBM-light; what it buys is launches.
#include "tzr_common.h"

#define MX_THREADS 256
#define MX_E TZR_MOE_MAX_EXPERTS
#define MX_T TZR_MOE_MAX_TASKS

struct MxArgs {
  const float* expert[MX_E];
  int64_t expert_stride[MX_E];
  float* d_expert[MX_E];  // backward
  int64_t d_expert_stride[MX_E];
  const float* logits[MX_T];  // forward: gate logits; backward: unused
  int64_t logits_stride[MX_T];
  float* probs[MX_T];  // [B, E] contiguous: written by the forward, read by the backward
  float* out[MX_T];    // forward: mixed rows; backward: d(loss)/d(logits)
  int64_t out_stride[MX_T];
  const float* g[MX_T];  // backward: d(loss)/d(out_t)
  int64_t g_stride[MX_T];
};

// TT / EE: compile-time bounds of the task / expert loops (the smallest instantiated pair covering T, E: the per-lane arrays are TT x EE)
template <int TT, int EE>
__global__ __launch_bounds__(MX_THREADS) void tzr_moe_mix_fwd_kernel(MxArgs A, int64_t B, int H, int E, int T, int lgp) {
  const int N4 = H >> 2;
  const int rpw = TZR_WAVE / lgp;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int gi = lane / lgp, c = lane - gi * lgp;
  const int64_t wave = (int64_t)blockIdx.x * (MX_THREADS / TZR_WAVE) + threadIdx.x / TZR_WAVE;
  const int64_t n_waves = (int64_t)gridDim.x * (MX_THREADS / TZR_WAVE);
  for (int64_t b = wave * rpw + gi; b < B; b += n_waves * rpw) {
    float p[TT][EE];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t >= T) break;
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < EE; ++e) {
        p[t][e] = e < E ? A.logits[t][b * A.logits_stride[t] + e] : -INFINITY;
        mx = fmaxf(mx, p[t][e]);
      }
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < EE; ++e) {
        p[t][e] = e < E ? expf(p[t][e] - mx) : 0.f;
        s += p[t][e];
      }
      const float inv = 1.0f / s;
#pragma unroll
      for (int e = 0; e < EE; ++e) {
        p[t][e] *= inv;
        if (c == 0 && e < E) A.probs[t][b * E + e] = p[t][e];
      }
    }
    for (int cc = c; cc < N4; cc += lgp) {
      float4 x[EE];
#pragma unroll
      for (int e = 0; e < EE; ++e) x[e] = e < E ? tzr_ld4(A.expert[e] + b * A.expert_stride[e] + 4 * cc) : tzr_zero4();
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t >= T) break;
        float4 o = tzr_zero4();
#pragma unroll
        for (int e = 0; e < EE; ++e) o = tzr_fma4(p[t][e], x[e], o);
        tzr_st4(A.out[t] + b * A.out_stride[t] + 4 * cc, o);
      }
    }
  }
}

template <int TT, int EE>
__global__ __launch_bounds__(MX_THREADS) void tzr_moe_mix_bwd_kernel(MxArgs A, int64_t B, int H, int E, int T, int lgp) {
  const int N4 = H >> 2;
  const int rpw = TZR_WAVE / lgp;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int gi = lane / lgp, c = lane - gi * lgp;
  const int64_t wave = (int64_t)blockIdx.x * (MX_THREADS / TZR_WAVE) + threadIdx.x / TZR_WAVE;
  const int64_t n_waves = (int64_t)gridDim.x * (MX_THREADS / TZR_WAVE);
  // (every lane of a wave runs the same number of iterations: the shuffles below are wave-wide)
  for (int64_t b0 = wave * rpw; b0 < B; b0 += n_waves * rpw) {
    const int64_t b = b0 + gi;
    const bool on = b < B;
    float p[TT][EE], dp[TT][EE];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int e = 0; e < EE; ++e) {
        p[t][e] = (on && t < T && e < E) ? A.probs[t][b * E + e] : 0.f;
        dp[t][e] = 0.f;
      }
    for (int cc = c; cc < N4 + c; cc += lgp) {  // (c < lgp: the same trip count for every lane; columns >= N4 are idle)
      const bool col = on && cc < N4;
      float4 x[EE], g[TT];
#pragma unroll
      for (int e = 0; e < EE; ++e) x[e] = (col && e < E) ? tzr_ld4(A.expert[e] + b * A.expert_stride[e] + 4 * cc) : tzr_zero4();
#pragma unroll
      for (int t = 0; t < TT; ++t) g[t] = (col && t < T) ? tzr_ld4(A.g[t] + b * A.g_stride[t] + 4 * cc) : tzr_zero4();
#pragma unroll
      for (int e = 0; e < EE; ++e) {
        if (e >= E) break;
        float4 o = tzr_zero4();
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          o = tzr_fma4(p[t][e], g[t], o);
          dp[t][e] += g[t].x * x[e].x + g[t].y * x[e].y + g[t].z * x[e].z + g[t].w * x[e].w;
        }
        if (col) tzr_st4(A.d_expert[e] + b * A.d_expert_stride[e] + 4 * cc, o);
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t >= T) break;
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < EE; ++e) {
        float v = dp[t][e];
        for (int m = lgp >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, TZR_WAVE);
        dp[t][e] = v;
        dot += p[t][e] * v;
      }
#pragma unroll
      for (int e = 0; e < EE; ++e)
        if (on && c == 0 && e < E) A.out[t][b * A.out_stride[t] + e] = p[t][e] * (dp[t][e] - dot);
    }
  }
}

static int mx_check(const TzrMoeMix* m) {
  if (!m || m->B <= 0 || m->n_experts <= 0 || m->n_tasks <= 0 || m->H <= 0) return TZR_ERR_INVALID;
  if (m->n_experts > MX_E || m->n_tasks > MX_T || (m->H & 3) || m->H > 4096) return TZR_ERR_UNSUPPORTED;
  for (int e = 0; e < m->n_experts; ++e)
    if (!m->expert[e] || (m->expert_stride[e] & 3) || (m->expert[e] & 15)) return TZR_ERR_INVALID;
  for (int t = 0; t < m->n_tasks; ++t)
    if (!m->probs[t]) return TZR_ERR_INVALID;
  return TZR_OK;
}

static void mx_geometry(const TzrMoeMix* m, int* lgp, unsigned* grid) {
  int l = 1;
  while (l < (m->H >> 2) && l < TZR_WAVE) l <<= 1;
  const int rpw = TZR_WAVE / l;
  const int64_t rows_per_wg = (int64_t)rpw * (MX_THREADS / TZR_WAVE);
  *lgp = l;
  *grid = (unsigned)std::min<int64_t>((m->B + rows_per_wg - 1) / rows_per_wg, 4096);
}

#define MX_LAUNCH(K, TT_, EE_)                                                                                              \
  hipLaunchKernelGGL((K<TT_, EE_>), dim3(grid), dim3(MX_THREADS), 0, static_cast<hipStream_t>(stream), A, h_mix->B, h_mix->H, \
                     h_mix->n_experts, h_mix->n_tasks, lgp)
#define MX_BY_E(K, TT_)                        \
  do {                                         \
    if (h_mix->n_experts <= 2) MX_LAUNCH(K, TT_, 2);      \
    else if (h_mix->n_experts <= 4) MX_LAUNCH(K, TT_, 4); \
    else MX_LAUNCH(K, TT_, 8);                 \
  } while (0)
#define MX_DISPATCH(K)                         \
  do {                                         \
    if (h_mix->n_tasks <= 1) MX_BY_E(K, 1);    \
    else if (h_mix->n_tasks <= 2) MX_BY_E(K, 2); \
    else MX_BY_E(K, 4);                        \
  } while (0)

extern "C" int tzr_moe_mix_fwd(const TzrMoeMix* h_mix, void* stream) {
  const int rc = mx_check(h_mix);
  if (rc != TZR_OK) return rc;
  MxArgs A = {};
  for (int e = 0; e < h_mix->n_experts; ++e) {
    A.expert[e] = reinterpret_cast<const float*>(h_mix->expert[e]);
    A.expert_stride[e] = h_mix->expert_stride[e];
  }
  for (int t = 0; t < h_mix->n_tasks; ++t) {
    if (!h_mix->logits[t] || !h_mix->out[t] || (h_mix->out_stride[t] & 3) || (h_mix->out[t] & 15) || h_mix->logits_stride[t] < h_mix->n_experts)
      return TZR_ERR_INVALID;
    A.logits[t] = reinterpret_cast<const float*>(h_mix->logits[t]);
    A.logits_stride[t] = h_mix->logits_stride[t];
    A.probs[t] = reinterpret_cast<float*>(h_mix->probs[t]);
    A.out[t] = reinterpret_cast<float*>(h_mix->out[t]);
    A.out_stride[t] = h_mix->out_stride[t];
  }
  int lgp;
  unsigned grid;
  mx_geometry(h_mix, &lgp, &grid);
  MX_DISPATCH(tzr_moe_mix_fwd_kernel);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_moe_mix_bwd(const TzrMoeMix* h_mix, void* stream) {
  const int rc = mx_check(h_mix);
  if (rc != TZR_OK) return rc;
  MxArgs A = {};
  for (int e = 0; e < h_mix->n_experts; ++e) {
    if (!h_mix->d_expert[e] || (h_mix->d_expert_stride[e] & 3) || (h_mix->d_expert[e] & 15)) return TZR_ERR_INVALID;
    A.expert[e] = reinterpret_cast<const float*>(h_mix->expert[e]);
    A.expert_stride[e] = h_mix->expert_stride[e];
    A.d_expert[e] = reinterpret_cast<float*>(h_mix->d_expert[e]);
    A.d_expert_stride[e] = h_mix->d_expert_stride[e];
  }
  for (int t = 0; t < h_mix->n_tasks; ++t) {
    if (!h_mix->grad_out[t] || !h_mix->d_logits[t] || (h_mix->grad_out_stride[t] & 3) || (h_mix->grad_out[t] & 15) ||
        h_mix->d_logits_stride[t] < h_mix->n_experts)
      return TZR_ERR_INVALID;
    A.probs[t] = reinterpret_cast<float*>(h_mix->probs[t]);
    A.g[t] = reinterpret_cast<const float*>(h_mix->grad_out[t]);
    A.g_stride[t] = h_mix->grad_out_stride[t];
    A.out[t] = reinterpret_cast<float*>(h_mix->d_logits[t]);
    A.out_stride[t] = h_mix->d_logits_stride[t];
  }
  int lgp;
  unsigned grid;
  mx_geometry(h_mix, &lgp, &grid);
  MX_DISPATCH(tzr_moe_mix_bwd_kernel);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
