// Dense glue of a training step that PyTorch issues as dozens of 3-5 us launches at B = 65536:
//
//   tzr_bce_logits   BCEWithLogitsLoss(reduction="mean") forward AND d(loss)/d(logits) in two launches
//                    (reference: /root/reference/tzrec/models/rank_model.py:190-191,233-240 builds
//                    torch.nn.BCEWithLogitsLoss; PyTorch runs it as ~12 elementwise / reduce kernels
//                    forward + backward, ~55 us per step)
//   tzr_dense_adam   Adam over every dense parameter tensor in two launches (reference: the dense
//                    optimizer stepped by TZRecOptimizer, /root/reference/tzrec/optim/optimizer.py:56-68,
//                    built from `adam_optimizer` in optimizer_builder.py; torch's fused multi-tensor
//                    Adam + its foreach helpers cost ~40 us for DLRM's 54 k parameters)
//
//   tzr_relu_bwd_colsum  ReLU backward and the bias gradient of a Linear+ReLU layer from one pass
//
// They are bandwidth-light; what they buy is launch count and one re-read of the gradient.  Deterministic: fixed-order
// reductions, no float atomics.
#include "tzr_common.h"

#define DN_THREADS 256
#define DN_ITEMS 4  // logits per thread
#define DN_MAX_PARTS 4096

// Stage 1: one logit per thread-item, per-workgroup partial sums in a fixed tree; stage 2: one
// workgroup adds the partials in index order.  (A single workgroup walking all B logits is
// latency-bound: 64 dependent round trips at B = 65536, measured slower than torch's 12 launches.)
template <typename LabelT>
__global__ __launch_bounds__(DN_THREADS) void tzr_bce_logits_kernel(
    const float* __restrict__ logits, const LabelT* __restrict__ labels,
    const float* __restrict__ sample_weight, int64_t B, int64_t per_wg, float* __restrict__ parts,
    float* __restrict__ grad) {
  __shared__ float part[DN_THREADS / TZR_WAVE];
  const float inv = 1.0f / (float)B;
  float acc = 0.f;
  const int64_t lo = (int64_t)blockIdx.x * per_wg;
  const int64_t hi = min(B, lo + per_wg);
  for (int64_t base = lo; base < hi; base += DN_THREADS * DN_ITEMS) {
    float x[DN_ITEMS], y[DN_ITEMS], w[DN_ITEMS];
#pragma unroll
    for (int j = 0; j < DN_ITEMS; ++j) {  // independent loads first
      const int64_t i = base + (int64_t)j * DN_THREADS + threadIdx.x;
      const bool ok = i < hi;
      x[j] = ok ? logits[i] : 0.f;
      y[j] = ok ? (float)labels[i] : 0.f;
      w[j] = ok ? (sample_weight ? sample_weight[i] : 1.0f) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < DN_ITEMS; ++j) {
      const int64_t i = base + (int64_t)j * DN_THREADS + threadIdx.x;
      if (i >= hi) continue;
      // loss = max(x,0) - x*y + log1p(exp(-|x|));  d/dx = sigmoid(x) - y
      const float e = expf(-fabsf(x[j]));
      acc += w[j] * (fmaxf(x[j], 0.f) - x[j] * y[j] + log1pf(e));
      const float sig = x[j] >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      grad[i] = w[j] * (sig - y[j]) * inv;
    }
  }
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  for (int d = TZR_WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, TZR_WAVE);
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int i = 0; i < DN_THREADS / TZR_WAVE; ++i) v += part[i];
    parts[blockIdx.x] = v;
  }
}

__global__ __launch_bounds__(DN_THREADS) void tzr_bce_finish_kernel(const float* __restrict__ parts, int n,
                                                                     float inv, float* __restrict__ loss) {
  __shared__ float part[DN_THREADS / TZR_WAVE];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += DN_THREADS) acc += parts[i];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  for (int d = TZR_WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, TZR_WAVE);
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int i = 0; i < DN_THREADS / TZR_WAVE; ++i) v += part[i];
    *loss = v * inv;
  }
}

extern "C" size_t tzr_bce_logits_workspace(int64_t B) {
  (void)B;
  return (size_t)DN_MAX_PARTS * sizeof(float) + 256;
}

extern "C" int tzr_bce_logits(const float* d_logits, const void* d_labels, int labels_itemsize,
                              int labels_are_float, const float* d_sample_weight, int64_t B,
                              float* d_loss, float* d_grad_logits, void* ws, size_t ws_bytes,
                              void* stream) {
  if (!d_logits || !d_labels || !d_loss || !d_grad_logits || B <= 0) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < (size_t)DN_MAX_PARTS * sizeof(float))
    return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* parts = static_cast<float*>(ws);
  const int64_t tile = DN_THREADS * DN_ITEMS;
  int64_t n_wg = (B + tile - 1) / tile;
  int64_t per_wg = tile;
  if (n_wg > DN_MAX_PARTS) {  // very large batches: several tiles per workgroup
    per_wg = ((B + DN_MAX_PARTS - 1) / DN_MAX_PARTS + tile - 1) / tile * tile;
    n_wg = (B + per_wg - 1) / per_wg;
  }
#define TZR_BCE_LAUNCH(T)                                                                          \
  hipLaunchKernelGGL(tzr_bce_logits_kernel<T>, dim3((unsigned)n_wg), dim3(DN_THREADS), 0, s, d_logits, \
                     static_cast<const T*>(d_labels), d_sample_weight, B, per_wg, parts, d_grad_logits)
  if (labels_are_float && labels_itemsize == 4) TZR_BCE_LAUNCH(float);
  else if (!labels_are_float && labels_itemsize == 8) TZR_BCE_LAUNCH(int64_t);
  else if (!labels_are_float && labels_itemsize == 4) TZR_BCE_LAUNCH(int32_t);
  else return TZR_ERR_UNSUPPORTED;
#undef TZR_BCE_LAUNCH
  hipLaunchKernelGGL(tzr_bce_finish_kernel, dim3(1), dim3(DN_THREADS), 0, s, parts, (int)n_wg,
                     1.0f / (float)B, d_loss);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- ReLU backward + bias gradient ---------------------------------------------------------------
// g = gy * (y > 0) and colsum[n] = sum_b g[b, n] from one pass over gy / y (PyTorch: threshold_backward
// then a separate column reduction that re-reads g).  Row tiles accumulate per-thread column sums in a
// fixed order, workgroup partials go to the workspace, a second launch adds them in index order.
#define RB_THREADS 256
#define RB_MAX_WG 1024

#define RB_UNROLL 4
__global__ __launch_bounds__(RB_THREADS) void tzr_relu_bwd_colsum_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ y, int64_t y_stride,
    int64_t B, int N, int64_t rows_per_wg, float* __restrict__ g, int64_t g_stride,
    float* __restrict__ parts) {
  // thread -> (row lane r, float4 column c): N4 = N/4 column groups, RB_THREADS / N4 rows per sweep,
  // RB_UNROLL sweeps loaded before any is used (these kernels are latency-, not bandwidth-bound)
  __shared__ float4 red[RB_THREADS];
  const int N4 = N >> 2;
  const int rl = RB_THREADS / N4;
  const int c = threadIdx.x % N4;
  const int r = threadIdx.x / N4;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = min(B, lo + rows_per_wg);
  float4 acc = tzr_zero4();
  if (r < rl) {
    for (int64_t b0 = lo + r; b0 < hi; b0 += (int64_t)rl * RB_UNROLL) {
      float4 a[RB_UNROLL], v[RB_UNROLL];
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        a[u] = b < hi ? tzr_ld4(gy + b * gy_stride + 4 * c) : tzr_zero4();
        v[u] = b < hi ? tzr_ld4(y + b * y_stride + 4 * c) : tzr_zero4();
      }
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        if (b >= hi) continue;
        float4 o;
        o.x = v[u].x > 0.f ? a[u].x : 0.f;
        o.y = v[u].y > 0.f ? a[u].y : 0.f;
        o.z = v[u].z > 0.f ? a[u].z : 0.f;
        o.w = v[u].w > 0.f ? a[u].w : 0.f;
        tzr_st4(g + b * g_stride + 4 * c, o);
        acc = tzr_add4(acc, o);
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (r == 0 && c < N4) {  // fixed order over the row lanes
    float4 t = red[c];
    for (int k = 1; k < rl; ++k) t = tzr_add4(t, red[k * N4 + c]);
    tzr_st4(parts + (size_t)blockIdx.x * N + 4 * c, t);
  }
}

// Column sums of parts[n_wg][N]: one workgroup per 64 columns, 16 slices of the partials per column
// summed concurrently with 8 independent loads in flight each (a serial walk over 1024 partials is
// 60 us of pure latency), then combined in slice order.
#define RB_FIN_THREADS 1024
__global__ __launch_bounds__(RB_FIN_THREADS) void tzr_colsum_finish_kernel(const float* __restrict__ parts,
                                                                            int n_wg, int N,
                                                                            float* __restrict__ out) {
  __shared__ float red[RB_FIN_THREADS];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;  // 0..15
  float t = 0.f;
  if (col < N) {
    for (int k0 = slice; k0 < n_wg; k0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 16 * u;
        v[u] = k < n_wg ? parts[(size_t)k * N + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (slice == 0 && col < N) {
    float r = 0.f;
    for (int sl = 0; sl < 16; ++sl) r += red[sl * 64 + (threadIdx.x & 63)];
    out[col] = r;
  }
}

extern "C" size_t tzr_relu_bwd_colsum_workspace(int64_t B, int N) {
  (void)B;
  return (size_t)RB_MAX_WG * (size_t)std::max(N, 4) * sizeof(float) + 256;
}

// d_colsum == NULL with out_G != NULL: no finishing launch -- the column sums stay *out_G rows of N partial sums at the head of
// `ws` (tzr_relu_bwd_colsum_parts: for tzr_dense_adam_fused's TZR_ADAM_SRC_ROWS)
static int relu_bwd_colsum_impl(const float* d_grad_y, int64_t grad_y_stride, const float* d_y, int64_t y_stride, int64_t B, int N,
                                float* d_grad, int64_t grad_stride, float* d_colsum, int* out_G, void* ws, size_t ws_bytes,
                                void* stream) {
  if (!d_grad_y || !d_y || !d_grad || (!d_colsum && !out_G) || B <= 0 || N <= 0) return TZR_ERR_INVALID;
  if ((N & 3) || N > 4 * RB_THREADS || (grad_y_stride & 3) || (y_stride & 3) || (grad_stride & 3))
    return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_relu_bwd_colsum_workspace(B, N) - 256)
    return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rl = RB_THREADS / (N >> 2);
  // one unrolled batch of sweeps per workgroup, at most RB_MAX_WG workgroups
  int64_t rows_per_wg = (int64_t)rl * RB_UNROLL;
  int64_t n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  if (n_wg > RB_MAX_WG) {
    rows_per_wg = ((B + RB_MAX_WG - 1) / RB_MAX_WG + rl * RB_UNROLL - 1) / (rl * RB_UNROLL) * (rl * RB_UNROLL);
    n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  }
  float* parts = static_cast<float*>(ws);
  hipLaunchKernelGGL(tzr_relu_bwd_colsum_kernel, dim3((unsigned)n_wg), dim3(RB_THREADS), 0, s, d_grad_y,
                     grad_y_stride, d_y, y_stride, B, N, rows_per_wg, d_grad, grad_stride, parts);
  if (d_colsum)
    hipLaunchKernelGGL(tzr_colsum_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(RB_FIN_THREADS), 0, s, parts,
                       (int)n_wg, N, d_colsum);
  if (out_G) *out_G = (int)n_wg;
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_relu_bwd_colsum(const float* d_grad_y, int64_t grad_y_stride, const float* d_y,
                                   int64_t y_stride, int64_t B, int N, float* d_grad, int64_t grad_stride,
                                   float* d_colsum, void* ws, size_t ws_bytes, void* stream) {
  if (!d_colsum) return TZR_ERR_INVALID;
  return relu_bwd_colsum_impl(d_grad_y, grad_y_stride, d_y, y_stride, B, N, d_grad, grad_stride, d_colsum, nullptr, ws, ws_bytes, stream);
}

extern "C" int tzr_relu_bwd_colsum_parts(const float* d_grad_y, int64_t grad_y_stride, const float* d_y, int64_t y_stride,
                                         int64_t B, int N, float* d_grad, int64_t grad_stride, void* ws, size_t ws_bytes,
                                         int* out_G, void* stream) {
  if (!out_G) return TZR_ERR_INVALID;
  return relu_bwd_colsum_impl(d_grad_y, grad_y_stride, d_y, y_stride, B, N, d_grad, grad_stride, nullptr, out_G, ws, ws_bytes, stream);
}

// ---- backward of the one-unit logits layer ---------------------------------------------------------
// y = x w^T + b with ONE output unit: gx[b,:] = gy[b] * w, gw[n] = sum_b gy[b] * x[b,n], gb = sum_b gy[b]
// from one pass over x (PyTorch: a GEMM, a broadcast multiply and two reductions, ~35 us at B = 65536).
__global__ __launch_bounds__(RB_THREADS) void tzr_head_bwd_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ x, int64_t x_stride,
    const float* __restrict__ w, int64_t B, int N, int64_t rows_per_wg, float* __restrict__ gx,
    int64_t gx_stride, float* __restrict__ parts /*[n_wg][N + 4]*/) {
  __shared__ float4 red[RB_THREADS];
  __shared__ float redb[RB_THREADS];
  const int N4 = N >> 2;
  const int rl = RB_THREADS / N4;
  const int c = threadIdx.x % N4;
  const int r = threadIdx.x / N4;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = min(B, lo + rows_per_wg);
  float4 acc = tzr_zero4();
  float accb = 0.f;
  if (r < rl) {
    const float4 w4 = tzr_ld4(w + 4 * c);
    for (int64_t b0 = lo + r; b0 < hi; b0 += (int64_t)rl * RB_UNROLL) {
      float4 v[RB_UNROLL];
      float gs[RB_UNROLL];
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        v[u] = b < hi ? tzr_ld4(x + b * x_stride + 4 * c) : tzr_zero4();
        gs[u] = b < hi ? gy[b * gy_stride] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        if (b >= hi) continue;
        if (gx) tzr_st4(gx + b * gx_stride + 4 * c, make_float4(gs[u] * w4.x, gs[u] * w4.y, gs[u] * w4.z, gs[u] * w4.w));
        acc = tzr_fma4(gs[u], v[u], acc);
        if (c == 0) accb += gs[u];
      }
    }
  }
  red[threadIdx.x] = acc;
  redb[threadIdx.x] = accb;
  __syncthreads();
  if (r == 0 && c < N4) {
    float4 t = red[c];
    for (int k = 1; k < rl; ++k) t = tzr_add4(t, red[k * N4 + c]);
    tzr_st4(parts + (size_t)blockIdx.x * (N + 4) + 4 * c, t);
    if (c == 0) {
      float tb = 0.f;
      for (int k = 0; k < rl; ++k) tb += redb[k * N4];
      float* pb = parts + (size_t)blockIdx.x * (N + 4) + N;
      pb[0] = tb;
      pb[1] = pb[2] = pb[3] = 0.f;
    }
  }
}

extern "C" size_t tzr_head_bwd_workspace(int64_t B, int N) {
  (void)B;
  return (size_t)RB_MAX_WG * (size_t)(N + 4) * sizeof(float) + 256;
}

extern "C" int tzr_head_bwd(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                            const float* d_w, int64_t B, int N, float* d_grad_x, int64_t grad_x_stride,
                            float* d_grad_wb, void* ws, size_t ws_bytes, void* stream) {
  if (!d_grad_y || !d_x || !d_w || !d_grad_wb || B <= 0 || N <= 0) return TZR_ERR_INVALID;
  if ((N & 3) || N > 4 * RB_THREADS || (x_stride & 3) || (d_grad_x && (grad_x_stride & 3))) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_head_bwd_workspace(B, N) - 256)
    return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rl = RB_THREADS / (N >> 2);
  int64_t rows_per_wg = (int64_t)rl * RB_UNROLL;
  int64_t n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  if (n_wg > RB_MAX_WG) {
    rows_per_wg = ((B + RB_MAX_WG - 1) / RB_MAX_WG + rl * RB_UNROLL - 1) / (rl * RB_UNROLL) * (rl * RB_UNROLL);
    n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  }
  float* parts = static_cast<float*>(ws);
  hipLaunchKernelGGL(tzr_head_bwd_kernel, dim3((unsigned)n_wg), dim3(RB_THREADS), 0, s, d_grad_y, grad_y_stride,
                     d_x, x_stride, d_w, B, N, rows_per_wg, d_grad_x, grad_x_stride, parts);
  hipLaunchKernelGGL(tzr_colsum_finish_kernel, dim3((unsigned)((N + 4 + 63) / 64)), dim3(RB_FIN_THREADS), 0, s,
                     parts, (int)n_wg, N + 4, d_grad_wb);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// The same layer when x is the OUTPUT OF A RELU (the last hidden layer of an MLP in front of a one-unit score layer, e.g.
// DIN's attention MLP, /root/reference/tzrec/modules/sequence.py:101-128): the input gradient is masked on the way out,
// g[b, n] = gy[b] * w[n] * (x[b, n] > 0), and its column sums -- the bias gradient of the layer that produced x -- come
// from the same pass.  Saves writing gx unmasked and tzr_relu_bwd_colsum's re-read of gx and x.
// parts row: [gw (N) | gb, 0, 0, 0 | colsum g (N)]
__global__ __launch_bounds__(RB_THREADS) void tzr_head_bwd_relu_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ x, int64_t x_stride,
    const float* __restrict__ w, int64_t B, int N, int64_t rows_per_wg, float* __restrict__ g,
    int64_t g_stride, float* __restrict__ parts /*[n_wg][2 N + 4]*/) {
  __shared__ float4 red[RB_THREADS];
  __shared__ float4 redc[RB_THREADS];
  __shared__ float redb[RB_THREADS];
  const int N4 = N >> 2;
  const int rl = RB_THREADS / N4;
  const int c = threadIdx.x % N4;
  const int r = threadIdx.x / N4;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = min(B, lo + rows_per_wg);
  float4 acc = tzr_zero4(), accc = tzr_zero4();
  float accb = 0.f;
  if (r < rl) {
    const float4 w4 = tzr_ld4(w + 4 * c);
    for (int64_t b0 = lo + r; b0 < hi; b0 += (int64_t)rl * RB_UNROLL) {
      float4 v[RB_UNROLL];
      float gs[RB_UNROLL];
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        v[u] = b < hi ? tzr_ld4(x + b * x_stride + 4 * c) : tzr_zero4();
        gs[u] = b < hi ? gy[b * gy_stride] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        if (b >= hi) continue;
        float4 o;
        o.x = v[u].x > 0.f ? gs[u] * w4.x : 0.f;
        o.y = v[u].y > 0.f ? gs[u] * w4.y : 0.f;
        o.z = v[u].z > 0.f ? gs[u] * w4.z : 0.f;
        o.w = v[u].w > 0.f ? gs[u] * w4.w : 0.f;
        tzr_st4(g + b * g_stride + 4 * c, o);
        accc = tzr_add4(accc, o);
        acc = tzr_fma4(gs[u], v[u], acc);
        if (c == 0) accb += gs[u];
      }
    }
  }
  red[threadIdx.x] = acc;
  redc[threadIdx.x] = accc;
  redb[threadIdx.x] = accb;
  __syncthreads();
  if (r == 0 && c < N4) {
    float4 t = red[c], tc = redc[c];
    for (int k = 1; k < rl; ++k) {
      t = tzr_add4(t, red[k * N4 + c]);
      tc = tzr_add4(tc, redc[k * N4 + c]);
    }
    float* const row = parts + (size_t)blockIdx.x * (2 * N + 4);
    tzr_st4(row + 4 * c, t);
    tzr_st4(row + N + 4 + 4 * c, tc);
    if (c == 0) {
      float tb = 0.f;
      for (int k = 0; k < rl; ++k) tb += redb[k * N4];
      row[N] = tb;
      row[N + 1] = row[N + 2] = row[N + 3] = 0.f;
    }
  }
}

extern "C" size_t tzr_head_bwd_relu_workspace(int64_t B, int N) {
  (void)B;
  return (size_t)RB_MAX_WG * (size_t)(2 * N + 4) * sizeof(float) + 256;
}

extern "C" int tzr_head_bwd_relu(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                                 const float* d_w, int64_t B, int N, float* d_grad, int64_t grad_stride,
                                 float* d_sums /*[2 N + 4]: gw | gb, 0, 0, 0 | colsum*/, void* ws, size_t ws_bytes, void* stream) {
  if (!d_grad_y || !d_x || !d_w || !d_grad || !d_sums || B <= 0 || N <= 0) return TZR_ERR_INVALID;
  if ((N & 3) || N > 4 * RB_THREADS || (x_stride & 3) || (grad_stride & 3)) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_head_bwd_relu_workspace(B, N) - 256)
    return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rl = RB_THREADS / (N >> 2);
  int64_t rows_per_wg = (int64_t)rl * RB_UNROLL;
  int64_t n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  if (n_wg > RB_MAX_WG) {
    rows_per_wg = ((B + RB_MAX_WG - 1) / RB_MAX_WG + rl * RB_UNROLL - 1) / (rl * RB_UNROLL) * (rl * RB_UNROLL);
    n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  }
  float* parts = static_cast<float*>(ws);
  hipLaunchKernelGGL(tzr_head_bwd_relu_kernel, dim3((unsigned)n_wg), dim3(RB_THREADS), 0, s, d_grad_y, grad_y_stride,
                     d_x, x_stride, d_w, B, N, rows_per_wg, d_grad, grad_stride, parts);
  hipLaunchKernelGGL(tzr_colsum_finish_kernel, dim3((unsigned)((2 * N + 4 + 63) / 64)), dim3(RB_FIN_THREADS), 0, s,
                     parts, (int)n_wg, 2 * N + 4, d_sums);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- Linear layers with a handful of output units (logits, MMoE gates) -----------------------------------------------
// y = x W^T + b with n_out <= 8 (the output layer of every rank model, /root/reference/tzrec/models/rank_model.py:190-191; the
// gates of MMoE, /root/reference/tzrec/modules/mmoe.py: Linear(in, num_expert) + softmax).  As GEMMs these are one output tile
// wide: hipBLASLt takes 31 us for [8192, 64] x [64, 1] and 26 us for the [3, 8192] x [8192, 256] weight gradient (profiles/r05j)
// -- 2 MB of input each.  Forward: one pass over x, a row per lane group, weights in registers, shuffle reduction, no LDS.
// Backward: tzr_head_bwd's scheme with NO accumulators per thread (gx = sum_j gy_j w_j, gw_j += gy_j x, gb_j += gy_j).
#define SK_MAX_OUT 8

template <int NO>
__global__ __launch_bounds__(RB_THREADS) void tzr_skinny_linear_fwd_kernel(
    const float* __restrict__ x, int64_t x_stride, const float* __restrict__ w, int64_t w_stride,
    const float* __restrict__ bias, int64_t B, int K, int n_out, int lgp, float* __restrict__ y, int64_t y_stride) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int N4 = K >> 2;
  const int rpw = TZR_WAVE / lgp;  // rows of a wave's sweep
  const int gi = lane / lgp, c = lane - gi * lgp;
  const int64_t wave = (int64_t)blockIdx.x * (RB_THREADS / TZR_WAVE) + threadIdx.x / TZR_WAVE;
  const int64_t n_waves = (int64_t)gridDim.x * (RB_THREADS / TZR_WAVE);
  const bool one_col = N4 <= lgp;
  float4 wr[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) wr[j] = (one_col && c < N4 && j < n_out) ? tzr_ld4(w + (int64_t)j * w_stride + 4 * c) : tzr_zero4();
  for (int64_t row0 = wave * rpw * RB_UNROLL; row0 < B; row0 += n_waves * rpw * RB_UNROLL) {
    float acc[RB_UNROLL][NO];
    if (one_col) {
      float4 v[RB_UNROLL];
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = row0 + (int64_t)u * rpw + gi;
        v[u] = (b < B && c < N4) ? tzr_ld4(x + b * x_stride + 4 * c) : tzr_zero4();
      }
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u)
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[u][j] = v[u].x * wr[j].x + v[u].y * wr[j].y + v[u].z * wr[j].z + v[u].w * wr[j].w;
    } else {  // K > 256: a lane walks its columns, the weights come from the cache
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = row0 + (int64_t)u * rpw + gi;
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[u][j] = 0.f;
        if (b < B)
          for (int cc = c; cc < N4; cc += lgp) {
            const float4 v = tzr_ld4(x + b * x_stride + 4 * cc);
#pragma unroll
            for (int j = 0; j < NO; ++j) {
              if (j >= n_out) break;
              const float4 ww = tzr_ld4(w + (int64_t)j * w_stride + 4 * cc);
              acc[u][j] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
            }
          }
      }
    }
#pragma unroll
    for (int u = 0; u < RB_UNROLL; ++u) {
      const int64_t b = row0 + (int64_t)u * rpw + gi;
#pragma unroll
      for (int j = 0; j < NO; ++j) {
        float t = acc[u][j];
        for (int m = lgp >> 1; m > 0; m >>= 1) t += __shfl_xor(t, m, TZR_WAVE);
        if (c == 0 && b < B && j < n_out) y[b * y_stride + j] = t + (bias ? bias[j] : 0.f);
      }
    }
  }
}

extern "C" int tzr_skinny_linear_fwd(const float* d_x, int64_t x_stride, const float* d_w, int64_t w_stride, const float* d_bias,
                                     int64_t B, int K, int n_out, float* d_y, int64_t y_stride, void* stream) {
  if (!d_x || !d_w || !d_y || B <= 0 || K <= 0 || n_out <= 0) return TZR_ERR_INVALID;
  if ((K & 3) || K > 4 * RB_THREADS || n_out > SK_MAX_OUT || (x_stride & 3) || (w_stride & 3) || y_stride < n_out ||
      ((uintptr_t)d_x & 15) || ((uintptr_t)d_w & 15))
    return TZR_ERR_UNSUPPORTED;
  int lgp = 1;
  while (lgp < (K >> 2) && lgp < TZR_WAVE) lgp <<= 1;
  const int rpw = TZR_WAVE / lgp;
  const int64_t rows_per_wg = (int64_t)rpw * RB_UNROLL * (RB_THREADS / TZR_WAVE);
  const unsigned grid = (unsigned)std::min<int64_t>((B + rows_per_wg - 1) / rows_per_wg, 2048);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define SK_FWD(NO_)                                                                                                      \
  hipLaunchKernelGGL((tzr_skinny_linear_fwd_kernel<NO_>), dim3(grid), dim3(RB_THREADS), 0, s, d_x, x_stride, d_w, w_stride, d_bias, B, K, \
                     n_out, lgp, d_y, y_stride)
  if (n_out == 1) SK_FWD(1);
  else if (n_out == 2) SK_FWD(2);
  else if (n_out <= 4) SK_FWD(4);
  else SK_FWD(8);
#undef SK_FWD
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// parts row of a workgroup: [gw (n_out x K, row-major) | gb (n_out, padded to a multiple of 4)]
template <int NO>
__global__ __launch_bounds__(RB_THREADS) void tzr_skinny_linear_bwd_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ x, int64_t x_stride,
    const float* __restrict__ w, int64_t w_stride, int64_t B, int K, int n_out, int64_t rows_per_wg, float* __restrict__ gx,
    int64_t gx_stride, float* __restrict__ parts, int row_len) {
  __shared__ float4 red[RB_THREADS];
  __shared__ float redb[RB_THREADS];
  const int N4 = K >> 2;
  const int rl = RB_THREADS / N4;
  const int c = threadIdx.x % N4;
  const int r = threadIdx.x / N4;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t hi = min(B, lo + rows_per_wg);
  float4 acc[NO], w4[NO];
  float accb[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    acc[j] = tzr_zero4();
    accb[j] = 0.f;
    w4[j] = (r < rl && j < n_out) ? tzr_ld4(w + (int64_t)j * w_stride + 4 * c) : tzr_zero4();
  }
  if (r < rl) {
    for (int64_t b0 = lo + r; b0 < hi; b0 += (int64_t)rl * RB_UNROLL) {
      float4 v[RB_UNROLL];
      float gs[RB_UNROLL][NO];
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        v[u] = b < hi ? tzr_ld4(x + b * x_stride + 4 * c) : tzr_zero4();
#pragma unroll
        for (int j = 0; j < NO; ++j) gs[u][j] = (b < hi && j < n_out) ? gy[b * gy_stride + j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RB_UNROLL; ++u) {
        const int64_t b = b0 + (int64_t)u * rl;
        if (b >= hi) continue;
        float4 o = tzr_zero4();
#pragma unroll
        for (int j = 0; j < NO; ++j) {
          o = tzr_fma4(gs[u][j], w4[j], o);
          acc[j] = tzr_fma4(gs[u][j], v[u], acc[j]);
          if (c == 0) accb[j] += gs[u][j];
        }
        if (gx) tzr_st4(gx + b * gx_stride + 4 * c, o);
      }
    }
  }
  float* const row = parts + (size_t)blockIdx.x * row_len;
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    if (j >= n_out) break;  // uniform
    __syncthreads();
    red[threadIdx.x] = acc[j];
    redb[threadIdx.x] = accb[j];
    __syncthreads();
    if (r == 0 && c < N4) {
      float4 t = red[c];
      for (int k = 1; k < rl; ++k) t = tzr_add4(t, red[k * N4 + c]);
      tzr_st4(row + (size_t)j * K + 4 * c, t);
      if (c == 0) {
        float tb = 0.f;
        for (int k = 0; k < rl; ++k) tb += redb[k * N4];
        row[(size_t)n_out * K + j] = tb;
      }
    }
  }
  if (threadIdx.x < (unsigned)(row_len - n_out * K - n_out)) row[(size_t)n_out * K + n_out + threadIdx.x] = 0.f;  // padding of gb
}

static inline int sk_row_len(int K, int n_out) { return n_out * K + (n_out + 3) / 4 * 4; }

extern "C" size_t tzr_skinny_linear_bwd_workspace(int64_t B, int K, int n_out) {
  (void)B;
  return (size_t)RB_MAX_WG * (size_t)sk_row_len(K, std::max(n_out, 1)) * sizeof(float) + 256;
}

// d_grad_wb == NULL with out_G / out_P: no finishing launch, the sums stay *out_G rows of *out_P partial sums at the head of `ws`
static int skinny_linear_bwd_impl(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                                  const float* d_w, int64_t w_stride, int64_t B, int K, int n_out, float* d_grad_x,
                                  int64_t grad_x_stride, float* d_grad_wb /*[n_out K + n_out padded to 4]*/, int* out_G, int* out_P,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (!d_grad_y || !d_x || !d_w || (!d_grad_wb && !(out_G && out_P)) || B <= 0 || K <= 0 || n_out <= 0) return TZR_ERR_INVALID;
  if ((K & 3) || K > 4 * RB_THREADS || n_out > SK_MAX_OUT || (x_stride & 3) || (w_stride & 3) || grad_y_stride < n_out ||
      (d_grad_x && (grad_x_stride & 3)))
    return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_skinny_linear_bwd_workspace(B, K, n_out) - 256)
    return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rl = RB_THREADS / (K >> 2);
  int64_t rows_per_wg = (int64_t)rl * RB_UNROLL;
  int64_t n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  if (n_wg > RB_MAX_WG) {
    rows_per_wg = ((B + RB_MAX_WG - 1) / RB_MAX_WG + rl * RB_UNROLL - 1) / (rl * RB_UNROLL) * (rl * RB_UNROLL);
    n_wg = (B + rows_per_wg - 1) / rows_per_wg;
  }
  const int row_len = sk_row_len(K, n_out);
  float* parts = static_cast<float*>(ws);
#define SK_BWD(NO_)                                                                                                          \
  hipLaunchKernelGGL((tzr_skinny_linear_bwd_kernel<NO_>), dim3((unsigned)n_wg), dim3(RB_THREADS), 0, s, d_grad_y, grad_y_stride, d_x, \
                     x_stride, d_w, w_stride, B, K, n_out, rows_per_wg, d_grad_x, grad_x_stride, parts, row_len)
  if (n_out == 1) SK_BWD(1);
  else if (n_out == 2) SK_BWD(2);
  else if (n_out <= 4) SK_BWD(4);
  else SK_BWD(8);
#undef SK_BWD
  if (d_grad_wb)
    hipLaunchKernelGGL(tzr_colsum_finish_kernel, dim3((unsigned)((row_len + 63) / 64)), dim3(RB_FIN_THREADS), 0, s, parts, (int)n_wg,
                       row_len, d_grad_wb);
  if (out_G) *out_G = (int)n_wg;
  if (out_P) *out_P = row_len;
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_skinny_linear_bwd(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                                     const float* d_w, int64_t w_stride, int64_t B, int K, int n_out, float* d_grad_x,
                                     int64_t grad_x_stride, float* d_grad_wb /*[n_out K + n_out padded to 4]*/, void* ws,
                                     size_t ws_bytes, void* stream) {
  if (!d_grad_wb) return TZR_ERR_INVALID;
  return skinny_linear_bwd_impl(d_grad_y, grad_y_stride, d_x, x_stride, d_w, w_stride, B, K, n_out, d_grad_x, grad_x_stride, d_grad_wb,
                                nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int tzr_skinny_linear_bwd_parts(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                                           const float* d_w, int64_t w_stride, int64_t B, int K, int n_out, float* d_grad_x,
                                           int64_t grad_x_stride, void* ws, size_t ws_bytes, int* out_G, int* out_P, void* stream) {
  if (!out_G || !out_P) return TZR_ERR_INVALID;
  return skinny_linear_bwd_impl(d_grad_y, grad_y_stride, d_x, x_stride, d_w, w_stride, B, K, n_out, d_grad_x, grad_x_stride, nullptr,
                                out_G, out_P, ws, ws_bytes, stream);
}

// ---- Adam ---------------------------------------------------------------------------------------

struct AdamTable {
  TzrAdamTensor t[TZR_ADAM_MAX_TENSORS];
  int n;
};

// Per tensor, state[0] = step (float, as torch keeps it for capturable optimizers; torch counts
// steps per PARAMETER, so a tensor skipped for lack of a gradient does not age), [1] = 1 - b1^step,
// [2] = 1 - b2^step.  A launch of its own so every workgroup of the apply reads ONE consistent step.
__global__ __launch_bounds__(256) void tzr_adam_begin_kernel(AdamTable T, float b1, float b2) {
  if ((int)threadIdx.x >= T.n) return;
  float* state = reinterpret_cast<float*>(T.t[threadIdx.x].state);
  const float step = state[0] + 1.0f;
  state[0] = step;
  state[1] = 1.0f - powf(b1, step);
  state[2] = 1.0f - powf(b2, step);
}

__global__ __launch_bounds__(256) void tzr_adam_apply_kernel(AdamTable T, const float* __restrict__ lr_ptr,
                                                             float lr_host, float b1, float b2, float eps,
                                                             float weight_decay) {
  const TzrAdamTensor a = T.t[blockIdx.y];
  float* __restrict__ p = reinterpret_cast<float*>(a.param);
  const float* __restrict__ g = reinterpret_cast<const float*>(a.grad);
  float* __restrict__ m = reinterpret_cast<float*>(a.exp_avg);
  float* __restrict__ v = reinterpret_cast<float*>(a.exp_avg_sq);
  const float* __restrict__ state = reinterpret_cast<const float*>(a.state);
  const float lr = lr_ptr ? *lr_ptr : lr_host;
  const float bc1 = state[1], bc2 = state[2];
  const float step_size = lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.numel; i += (int64_t)gridDim.x * 256) {
    float gi = g[i];
    const float pi = p[i];
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);  // L2 (torch.optim.Adam semantics)
    const float mi = fmaf(1.0f - b1, gi - m[i], m[i]);          // lerp, as torch's fused kernel
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

extern "C" int tzr_dense_adam(const TzrAdamTensor* h_tensors, int n_tensors, const float* d_lr,
                              float lr, float beta1, float beta2, float eps, float weight_decay,
                              void* stream) {
  if (!h_tensors || n_tensors <= 0) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int base = 0; base < n_tensors; base += TZR_ADAM_MAX_TENSORS) {
    AdamTable T;
    T.n = std::min(TZR_ADAM_MAX_TENSORS, n_tensors - base);
    int64_t mx = 0;
    for (int i = 0; i < T.n; ++i) {
      T.t[i] = h_tensors[base + i];
      if (!T.t[i].param || !T.t[i].grad || !T.t[i].exp_avg || !T.t[i].exp_avg_sq || !T.t[i].state ||
          T.t[i].numel < 0)
        return TZR_ERR_INVALID;
      mx = std::max(mx, T.t[i].numel);
    }
    hipLaunchKernelGGL(tzr_adam_begin_kernel, dim3(1), dim3(256), 0, s, T, beta1, beta2);
    if (mx == 0) continue;
    const unsigned gx = (unsigned)std::min<int64_t>(1024, (mx + 255) / 256);
    hipLaunchKernelGGL(tzr_adam_apply_kernel, dim3(gx, (unsigned)T.n), dim3(256), 0, s, T, d_lr, lr, beta1,
                       beta2, eps, weight_decay);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
