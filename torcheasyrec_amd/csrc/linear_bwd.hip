// Input gradient of a Linear layer chained with the ReLU mask and the bias gradient of the layer BELOW it, on the matrix
// cores (gfx950, exact-fp32 v_mfma_f32_16x16x4_f32):
//
//     g_out[n, h] = (sum_k g_in[n, k] W[k, h]) * (y[n, h] > 0)        colsum[h] = sum_n g_out[n, h]
//
// i.e. autograd of `Linear -> ReLU -> Linear` (the Perceptron stack of /root/reference/tzrec/modules/mlp.py:58-83) between
// two hidden layers: torch runs a GEMM that writes d(loss)/d(y) [N, H], threshold_backward that re-reads it together with y,
// and a column reduction that re-reads the result.  For the attention MLP of DIN on the jagged positions
// (/root/reference/tzrec/modules/sequence.py:101-128; N = 450 k positions at the Taobao config, K = 64, H = 256) that is
// 166 us of GEMM + 290 us of mask / column sums: 461 MB written and read back for nothing.  Here d(loss)/d(y) never leaves
// the registers: HBM traffic = g_in + y + g_out once (1.03 GB -> HBM-bound, ~0.2 ms; the product is 14.7 GFLOP = 94 us of
// MFMA issue, hidden under it by three workgroups per CU).
//
// Mapping.  A workgroup of four waves walks 16-row tiles of g_in (persistent, grid-strided); wave w owns the output columns
// [w H / 4, (w + 1) H / 4).  The roles of the MFMA operands are SWAPPED against the textbook product -- A = a 16-column
// block of W^T (resident in registers for the whole kernel: K / 4 x H / 64 registers per lane), B = the tile of g_in^T --
// so that accumulator register j of lane (r, q) is out[row r][column 16 jb + 4 q + j]: every lane holds FOUR CONSECUTIVE
// columns of one row, and y is read / g_out written as 16-byte pieces straight from / to global memory (no transposing
// pass through LDS).  The contraction index is permuted (k(s, q) = 16 (s / 4) + 4 q + s % 4) so that a lane's B operands
// of four MFMA steps are one ds_read_b128 of the g_in tile; that tile is the only thing in LDS (double-buffered: one
// workgroup barrier per tile, the next tile's global load in flight across it).
//
// Column sums: per-lane running sums over the workgroup's tiles, reduced over the 16 row lanes at the end, one row of
// partial sums per workgroup, added in workgroup order by a second launch: no float atomics, bit-reproducible.
#include <tzr_gfx950.h>

#include "tzr_common.h"

#define LB_WAVES 4
#define LB_THREADS (LB_WAVES * TZR_WAVE)
#define LB_TS 16
#define LB_MAX_WG 768  // 3 workgroups per CU resident on 256 CUs

typedef float lb_f32x4 __attribute__((ext_vector_type(4)));

int g_tzr_linear_bwd_wg = 0;  // tzr_tune("linear_bwd_wg"): > 0 caps the workgroups (tests: many tiles per workgroup on small inputs)

template <int KS /* K / 4 */, int HB /* H / 64 */>
__global__ __launch_bounds__(LB_THREADS) TZR_WAVES_PER_EU(3) void tzr_linear_bwd_relu_kernel(
    const float* __restrict__ gin, int64_t gin_stride, const float* __restrict__ W, int64_t w_stride,
    const float* __restrict__ y, int64_t y_stride, int64_t N, float* __restrict__ gout, int64_t gout_stride,
    float* __restrict__ parts) {
  constexpr int K = 4 * KS, H = 64 * HB, P = K + 4, K4 = K / 4;
  __shared__ __attribute__((aligned(16))) float TA[2][LB_TS * P];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  const int cb = wv * 16 * HB;
  // W^T blocks: operand A of step s is W[k(s, q)][cb + 16 jb + r]
  float Wa[HB][KS];
#pragma unroll
  for (int jb = 0; jb < HB; ++jb)
#pragma unroll
    for (int s = 0; s < KS; ++s) Wa[jb][s] = tzr_ldg(W + (int64_t)(16 * (s >> 2) + 4 * q + (s & 3)) * w_stride + cb + 16 * jb + r);
  const int trow = threadIdx.x / K4, tc4 = threadIdx.x % K4;  // the thread's 16 bytes of a g_in tile
  const bool stager = (int)threadIdx.x < LB_TS * K4;
  const int64_t ntiles = (N + LB_TS - 1) / LB_TS;
  float4 cs[HB];
#pragma unroll
  for (int jb = 0; jb < HB; ++jb) cs[jb] = tzr_zero4();
  int64_t t = blockIdx.x;
  if (t < ntiles && stager) {
    const int64_t row = t * LB_TS + trow;
    tzr_st4(&TA[0][trow * P + 4 * tc4], row < N ? tzr_ldg4(gin + row * gin_stride + 4 * tc4) : tzr_zero4());
  }
  __syncthreads();
  int buf = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int64_t tn = t + gridDim.x;
    float4 nx = tzr_zero4();
    if (tn < ntiles && stager) {
      const int64_t row = tn * LB_TS + trow;
      if (row < N) nx = tzr_ldg4(gin + row * gin_stride + 4 * tc4);
    }
    const int64_t row = t * LB_TS + r;
    const bool ok = row < N;
    float4 yv[HB];
#pragma unroll
    for (int jb = 0; jb < HB; ++jb) yv[jb] = ok ? tzr_ldg4(y + row * y_stride + cb + 16 * jb + 4 * q) : tzr_zero4();
    lb_f32x4 acc[HB];
#pragma unroll
    for (int jb = 0; jb < HB; ++jb) acc[jb] = lb_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* const A = &TA[buf][r * P + 4 * q];
#pragma unroll
    for (int e = 0; e < KS / 4; ++e) {
      const float4 av = tzr_ld4(A + 16 * e);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int jb = 0; jb < HB; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[jb][4 * e + c], a4[c], acc[jb], 0, 0, 0);
    }
    // (staged BEFORE this tile's stores are issued: behind stores under a row test the wait for nx would count short and become a
    // wait for the stores themselves -- gemm_rows.hip)
    if (tn < ntiles && stager) tzr_st4(&TA[buf ^ 1][trow * P + 4 * tc4], nx);
#pragma unroll
    for (int jb = 0; jb < HB; ++jb) {
      float4 o;
      o.x = yv[jb].x > 0.f ? acc[jb][0] : 0.f;
      o.y = yv[jb].y > 0.f ? acc[jb][1] : 0.f;
      o.z = yv[jb].z > 0.f ? acc[jb][2] : 0.f;
      o.w = yv[jb].w > 0.f ? acc[jb][3] : 0.f;
      cs[jb] = tzr_add4(cs[jb], o);  // (rows beyond N: y read as 0 -> o = 0)
      if (ok) tzr_stg4(gout + row * gout_stride + cb + 16 * jb + 4 * q, o);
    }
    tzr_lds_barrier();  // every wave is done with TA[buf]; TA[buf ^ 1] is complete (global loads / stores stay in flight)
    buf ^= 1;
  }
  // column sums of this workgroup: over the 16 row lanes (fixed tree), lane r == 0 of every q writes its four columns
#pragma unroll
  for (int jb = 0; jb < HB; ++jb) {
    float v[4] = {cs[jb].x, cs[jb].y, cs[jb].z, cs[jb].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] += __shfl_xor(v[c], 1);
      v[c] += __shfl_xor(v[c], 2);
      v[c] += __shfl_xor(v[c], 4);
      v[c] += __shfl_xor(v[c], 8);
    }
    if (r == 0) tzr_stg4(parts + (size_t)blockIdx.x * H + cb + 16 * jb + 4 * q, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// out[c] = sum over the workgroups' rows of parts[.][c], in row order (16 slices of the rows summed concurrently with 8
// loads in flight each, then combined in slice order: the arrangement of tzr_colsum_finish_kernel, dense_ops.hip)
#define LB_FIN_THREADS 1024
__global__ __launch_bounds__(LB_FIN_THREADS) void tzr_linear_bwd_finish_kernel(const float* __restrict__ parts, int n_wg, int H,
                                                                                float* __restrict__ out) {
  __shared__ float red[LB_FIN_THREADS];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  float t = 0.f;
  if (col < H) {
    for (int k0 = slice; k0 < n_wg; k0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 16 * u;
        v[u] = k < n_wg ? parts[(size_t)k * H + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (slice == 0 && col < H) {
    float s = 0.f;
    for (int sl = 0; sl < 16; ++sl) s += red[sl * 64 + (threadIdx.x & 63)];
    out[col] = s;
  }
}

extern "C" int tzr_linear_bwd_relu_supported(int K, int H) {
  return (K == 16 || K == 32 || K == 64) && (H == 64 || H == 128 || H == 256) ? 1 : 0;
}

extern "C" size_t tzr_linear_bwd_relu_workspace(int64_t N, int H) {
  (void)N;
  return (size_t)LB_MAX_WG * (size_t)std::max(H, 4) * sizeof(float) + 256;
}

template <int KS, int HB>
static void lb_launch(hipStream_t s, int grid, const float* gin, int64_t gs, const float* W, int64_t wst, const float* y, int64_t ys,
                      int64_t N, float* gout, int64_t os, float* parts) {
  hipLaunchKernelGGL((tzr_linear_bwd_relu_kernel<KS, HB>), dim3((unsigned)grid), dim3(LB_THREADS), 0, s, gin, gs, W, wst, y, ys, N, gout,
                     os, parts);
}

extern "C" int tzr_linear_bwd_relu(const float* d_grad_in, int64_t grad_in_stride, const float* d_w, int64_t w_stride,
                                   const float* d_y, int64_t y_stride, int64_t N, int K, int H, float* d_grad_out,
                                   int64_t grad_out_stride, float* d_colsum, void* ws, size_t ws_bytes, void* stream) {
  if (!d_grad_in || !d_w || !d_y || !d_grad_out || !d_colsum || N <= 0 || K <= 0 || H <= 0) return TZR_ERR_INVALID;
  if (!tzr_linear_bwd_relu_supported(K, H)) return TZR_ERR_UNSUPPORTED;
  if ((grad_in_stride & 3) || (y_stride & 3) || (grad_out_stride & 3) || grad_in_stride < K || y_stride < H || grad_out_stride < H ||
      w_stride < H)
    return TZR_ERR_UNSUPPORTED;
  if (((uintptr_t)d_grad_in | (uintptr_t)d_y | (uintptr_t)d_grad_out) & 15) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_linear_bwd_relu_workspace(N, H) - 256) return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t ntiles = (N + LB_TS - 1) / LB_TS;
  const int cap = g_tzr_linear_bwd_wg > 0 ? std::min(g_tzr_linear_bwd_wg, LB_MAX_WG) : LB_MAX_WG;
  const int grid = (int)std::min<int64_t>(ntiles, cap);
  float* parts = static_cast<float*>(ws);
#define LB_CASE(KS_, HB_)                                                                                                    \
  if (K == 4 * KS_ && H == 64 * HB_) {                                                                                       \
    lb_launch<KS_, HB_>(s, grid, d_grad_in, grad_in_stride, d_w, w_stride, d_y, y_stride, N, d_grad_out, grad_out_stride, parts); \
  } else
  LB_CASE(4, 1) LB_CASE(4, 2) LB_CASE(4, 4) LB_CASE(8, 1) LB_CASE(8, 2) LB_CASE(8, 4) LB_CASE(16, 1) LB_CASE(16, 2) LB_CASE(16, 4) {
    return TZR_ERR_UNSUPPORTED;
  }
#undef LB_CASE
  hipLaunchKernelGGL(tzr_linear_bwd_finish_kernel, dim3((unsigned)((H + 63) / 64)), dim3(LB_FIN_THREADS), 0, s, parts, grid, H,
                     d_colsum);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
