// The reduction of tzr_ia_wgrad_kernel's partial sums (csrc/interaction_wgrad.hip), shared with the dense optimizer that takes
// its gradients straight from partial sums (csrc/adam_fused.hip): dW1[h][c] = scale * sum over the batch slices, in slice order.
#pragma once
#include "tzr_common.h"

#define WG_H 64
#define WG_D 16
#define WG_NG 4
#define WG_MAXSLICES 64

struct WgGroup {
  int src;    // block of the pair matrix this group produces: 0 none, 1 c00, 2 c01, 3 c11
  int npb;    // its pair columns, in blocks of 16 (the last one padded)
  int x0, xn; // X rows [x0, x0 + xn) behind them, one block each
  int nb;     // npb + xn <= WG_MAXB
  int vbase;  // first column of the group in a row of the partial sums
};

// dW1[h][c] = scale * sum over the slices of the partial column that holds z column c.  Four lanes per output, lane j sums the
// slices j, j + 4, ... in order, then (s0 + s1) + (s2 + s3): one fixed order.  A wave = 16 outputs x 4 (j = lane >> 4).
struct WgReduceArgs {
  const float *part, *scale;
  float* dW;
  int64_t ldw;
  int n, slices, vw;
  WgGroup g[WG_NG];
};

// Output o = h * width + c of the weight gradient (all 64 lanes of a wave call with lane & 15 selecting one of 16 consecutive
// outputs, lane >> 4 the slice residue): the sum, scaled, in every lane of the output's four (the caller takes lane >> 4 == 0).
// *live = o exists.
__device__ __forceinline__ float wg_reduce_output(const WgReduceArgs& a, int o, int lane, int* h_out, int* c_out, bool* live_out) {
  const int n = a.n, P = n * (n - 1) / 2, width = P + WG_D * n;
  const int j = lane >> 4;
  const bool live = o < WG_H * width;
  o = live ? o : WG_H * width - 1;
  const int h = o / width, c = o - h * width;
  const int n0 = n < 16 ? n : 16, n1 = n > 16 ? n - 16 : 0;
  int v;
  if (c >= P) {
    const int x = (c - P) >> 4, d = (c - P) & 15;
    int k = 0;
    while (k < WG_NG - 1 && !(x >= a.g[k].x0 && x < a.g[k].x0 + a.g[k].xn)) ++k;
    v = a.g[k].vbase + 16 * (a.g[k].npb + x - a.g[k].x0) + d;
  } else {
    int i = 0, p = c;
    while (p >= n - 1 - i) { p -= n - 1 - i; ++i; }
    const int jj = i + 1 + p;
    if (jj < 16) v = a.g[0].vbase + (i * (2 * n0 - i - 1)) / 2 + jj - i - 1;
    else if (i < 16) v = a.g[1].vbase + i * n1 + (jj - 16);
    else v = a.g[2].vbase + ((i - 16) * (2 * n1 - (i - 16) - 1)) / 2 + jj - i - 1;
  }
  const int64_t step = (int64_t)WG_H * a.vw;
  const float* p = a.part + (int64_t)h * a.vw + v + j * step;
  float sum = 0.f;
  static_assert(WG_MAXSLICES <= 64, "sixteen slices per lane");
  {  // (slices: a multiple of 8, at most 64: all of a lane's loads in flight at once -- it was two dependent rounds of eight)
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int s = 4 * k;
      x[k] = p[(int64_t)(s < a.slices ? s : 0) * step];
      x[k] = s < a.slices ? x[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += x[k];
  }
  // (adding the zeros of slices beyond the count changes nothing: x + 0 = x, and -0 never arises from a sum started at +0)
  const float s1 = __shfl_xor(sum, 16), t01 = j & 1 ? s1 + sum : sum + s1;   // lanes j = 0 / 1: s0 + s1; j = 2 / 3: s2 + s3
  const float t23 = __shfl_xor(t01, 32);
  const float tot = j < 2 ? t01 + t23 : t23 + t01;
  const float sc = a.scale ? *a.scale : 1.f;
  *h_out = h;
  *c_out = c;
  *live_out = live;
  return sc * tot;
}
