// Dense Adam that takes its gradients as they lie: finished tensors, rows of per-workgroup partial sums (tzr_mlp2_bwd_parts) or
// the batch slices of the first top-MLP layer's weight gradient (tzr_dot_interaction_top_wgrad_parts) -- ONE launch for every
// tensor of the step.
//
// Replaces TZRecOptimizer.step() of the reference for the dense parameters (/root/reference/tzrec/optim/optimizer.py:56-68:
// torch.optim.Adam built by optim/optimizer_builder.py; same arithmetic as tzr_dense_adam, csrc/dense_ops.hip) AND the
// launches that used to stand between the backward and it in the DLRM step (models/dlrm.py:101-135): two column-sum
// finishes, the weight gradient's slice reduction and the optimizer's own step-counter launch were 4 of the step's 17
// launches, 20 us of dependent round trips for < 14 MB (profiles/r05final3/kernel_stats.csv).  Here a gradient element is
// summed (same order as the finish it replaces: bit-identical) by the threads that then update its parameter; the step
// counter of a tensor is read by every workgroup and moved on by the LAST of the tensor's workgroups to finish.
#include <tzr_gfx950.h>

#include <algorithm>
#include <cstring>

#include "tzr_common.h"
#include "wgrad_reduce.h"

namespace {

struct FusedSrc {
  int kind;  // 0 finished gradient, 1 partial-sum rows, 2 weight-gradient slices
  int G, P, col;
  const float* parts;
};

struct FusedTable {
  TzrAdamTensor t[TZR_ADAM_MAX_TENSORS];
  FusedSrc s[TZR_ADAM_MAX_TENSORS];
  int first[TZR_ADAM_MAX_TENSORS + 1];  // first workgroup of every tensor
  int n;
  WgReduceArgs wg;
};

struct AdamK {
  float lr, b1, b2, eps, wd, step_size, bc2_sqrt;
};

// torch.optim.Adam (amsgrad off, L2 weight decay), element i of tensor a with gradient gi; a.param == 0: the gradient is only
// stored (a caller that wants the finished tensor after all)
__device__ __forceinline__ void adam_element(const TzrAdamTensor& a, const AdamK& k, int64_t i, float gi) {
  if (!a.param) {
    reinterpret_cast<float*>(a.grad)[i] = gi;
    return;
  }
  float* __restrict__ p = reinterpret_cast<float*>(a.param);
  float* __restrict__ m = reinterpret_cast<float*>(a.exp_avg);
  float* __restrict__ v = reinterpret_cast<float*>(a.exp_avg_sq);
  const float pi = p[i];
  if (k.wd != 0.f) gi = fmaf(k.wd, pi, gi);
  const float mi = fmaf(1.0f - k.b1, gi - m[i], m[i]);  // lerp, as torch's fused kernel
  const float vi = fmaf(k.b2, v[i], (1.0f - k.b2) * gi * gi);
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / k.bc2_sqrt + k.eps;
  p[i] = pi - k.step_size * (mi / denom);
}

}  // namespace

__global__ __launch_bounds__(256) void tzr_adam_fused_kernel(FusedTable T, const float* __restrict__ lr_ptr, float lr_host, float b1,
                                                             float b2, float eps, float wd) {
  __shared__ float sl[16][17];
  int y = 0;
  while (y + 1 < T.n && (int)blockIdx.x >= T.first[y + 1]) ++y;  // (workgroup-uniform, <= 32 steps through kernel arguments)
  const TzrAdamTensor a = T.t[y];
  const FusedSrc src = T.s[y];
  const int lb = (int)blockIdx.x - T.first[y], nblk = T.first[y + 1] - T.first[y];
  // every workgroup of the tensor reads the SAME step count (it only moves when the last of them is done)
  float* const state = reinterpret_cast<float*>(a.state);
  AdamK k;
  k.lr = lr_ptr ? *lr_ptr : lr_host;
  k.b1 = b1; k.b2 = b2; k.eps = eps; k.wd = wd;
  float step = 1.0f;
  if (a.param) {
    step = state[0] + 1.0f;
    k.step_size = k.lr / (1.0f - powf(b1, step));
    k.bc2_sqrt = sqrtf(1.0f - powf(b2, step));
  }
  if (src.kind == 0) {
    // (src.parts != 0 with a finished tensor: the gradient is read from THERE -- a store-only call then copies it to `grad`, which is
    // how a set of gradients is packed into one flat buffer for a collective, partial sums and finished tensors in one launch)
    const float* __restrict__ g = src.parts ? src.parts : reinterpret_cast<const float*>(a.grad);
    for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < a.numel; i += (int64_t)nblk * 256) adam_element(a, k, i, g[i]);
  } else if (src.kind == 1) {
    // tzr_mlp_finish_kernel's sum: 16 outputs per workgroup x 16 slices of the partial rows, a thread adds its slice's partials
    // (8 loads in flight), thread (o, slice 0) adds the 16 slice sums in slice order
    const int ol = threadIdx.x & 15, sq = threadIdx.x >> 4;
    const int64_t o = (int64_t)lb * 16 + ol;
    const int per = (src.G + 15) / 16;
    const int g0 = sq * per, g1 = min(src.G, g0 + per);
    float v = 0.f;
    if (o < a.numel) {
      const float* col = src.parts + src.col + o;
      for (int g = g0; g < g1; g += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = g + j < g1 ? col[(size_t)(g + j) * src.P] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
      }
    }
    sl[sq][ol] = v;
    __syncthreads();
    if (sq == 0 && o < a.numel) {
      v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += sl[q][ol];
      adam_element(a, k, o, v);
    }
  } else {
    const int lane = threadIdx.x & 63;
    const int o = (lb * 4 + (int)(threadIdx.x >> 6)) * 16 + (lane & 15);
    int h, c;
    bool live;
    const float v = wg_reduce_output(T.wg, o, lane, &h, &c, &live);
    if (live && (lane >> 4) == 0) adam_element(a, k, (int64_t)o, v);  // (the parameter is [64, width] contiguous: element o)
  }
  if (!a.param) return;
  // The tensor's step count moves on when its last workgroup is done.  Arrivals go through a two-level tree -- groups of 32
  // workgroups (state[2 + group]), the last of a group then arrives at state[1]: every counter is zero between launches, and none
  // of them sees more than 32 arrivals (783 workgroups arriving at ONE word cost 11 us of a 20 us launch: profiles/r06k).
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* cnt = reinterpret_cast<uint32_t*>(state + 1);
    const int grp = lb >> 5, ngrp = (nblk + 31) >> 5;
    const uint32_t in_grp = (uint32_t)min(32, nblk - (grp << 5));
    if (tzr_arrive(cnt + 1 + grp) == in_grp - 1u) {
      tzr_publish_u32(cnt + 1 + grp, 0u);
      if (tzr_arrive(cnt) == (uint32_t)ngrp - 1u) {
        tzr_publish_u32(cnt, 0u);
        state[0] = step;
      }
    }
  }
}

// h_sources (nullable: every gradient is a finished tensor) runs parallel to h_tensors; at most one source of kind
// TZR_ADAM_SRC_WGRAD, described by h_wgrad.  A tensor with param == 0 only gets its finished gradient stored into `grad`.
extern "C" int tzr_dense_adam_fused(const TzrAdamTensor* h_tensors, const TzrAdamSource* h_sources, int n_tensors,
                                    const TzrWgradParts* h_wgrad, const float* d_lr, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, void* stream) {
  if (!h_tensors || n_tensors <= 0) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int base = 0; base < n_tensors; base += TZR_ADAM_MAX_TENSORS) {
    FusedTable T;
    T.n = std::min(TZR_ADAM_MAX_TENSORS, n_tensors - base);
    std::memset(&T.wg, 0, sizeof(T.wg));
    int blocks = 0;
    for (int i = 0; i < T.n; ++i) {
      const TzrAdamTensor& a = h_tensors[base + i];
      T.t[i] = a;
      if (a.numel < 0 || !a.grad && !(h_sources && h_sources[base + i].kind != TZR_ADAM_SRC_TENSOR)) return TZR_ERR_INVALID;
      if (a.param && (!a.exp_avg || !a.exp_avg_sq || !a.state)) return TZR_ERR_INVALID;
      FusedSrc& fs = T.s[i];
      fs.kind = 0; fs.G = fs.P = fs.col = 0; fs.parts = nullptr;
      T.first[i] = blocks;
      int nb;
      if (h_sources && h_sources[base + i].kind == TZR_ADAM_SRC_ROWS) {
        const TzrAdamSource& src = h_sources[base + i];
        if (!src.parts || src.G <= 0 || src.P <= 0 || src.col < 0 || src.col + a.numel > src.P) return TZR_ERR_INVALID;
        fs.kind = 1; fs.G = src.G; fs.P = src.P; fs.col = src.col; fs.parts = reinterpret_cast<const float*>(src.parts);
        nb = (int)((a.numel + 15) / 16);
      } else if (h_sources && h_sources[base + i].kind == TZR_ADAM_SRC_WGRAD) {
        if (!h_wgrad || T.wg.part) return TZR_ERR_INVALID;
        std::memcpy(&T.wg, h_wgrad, sizeof(T.wg));
        const int n = T.wg.n, width = n * (n - 1) / 2 + WG_D * n;
        if (!T.wg.part || a.numel != (int64_t)WG_H * width) return TZR_ERR_INVALID;
        fs.kind = 2;
        nb = (WG_H * width + 63) / 64;
      } else {
        if (h_sources && h_sources[base + i].parts) fs.parts = reinterpret_cast<const float*>(h_sources[base + i].parts);
        nb = (int)std::min<int64_t>(1024, (a.numel + 255) / 256);
      }
      blocks += std::max(nb, a.numel > 0 ? 1 : 0);
      if (a.numel == 0) T.first[i] = blocks;  // (no workgroup: its step count does not move, as a tensor without a gradient's)
    }
    T.first[T.n] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(tzr_adam_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T, d_lr, lr, beta1, beta2, eps, weight_decay);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
