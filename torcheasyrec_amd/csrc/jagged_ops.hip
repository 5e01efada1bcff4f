// K12 (sequence path): jagged <-> padded dense.
//
// Replaces fbgemm jagged_to_padded_dense reached from JaggedTensor.to_padded_dense in
// SequenceEmbeddingGroupImpl (/root/reference/tzrec/modules/embedding.py:1429,1480): the unpooled
// EmbeddingCollection output [N, D] (N = sum of sequence lengths) becomes [B, max_len, D], rows
// beyond a sample's length filled with `padding_value`, sequences longer than max_len truncated.
// The backward gathers the gradient back to jagged form (truncated positions get zero).
// Pure HBM streaming: float4 lanes, consecutive lanes walk consecutive float4s of the dense tensor.
#include "tzr_common.h"

#define JG_THREADS 256

template <bool TO_DENSE>
__global__ __launch_bounds__(JG_THREADS) void tzr_jagged_dense_kernel(
    float* __restrict__ jagged, int64_t jagged_stride, const int64_t* __restrict__ offsets,
    int64_t B, int64_t max_len, int lg, float* __restrict__ dense, float pad) {
  const int64_t total = B * max_len * lg;
  for (int64_t k = (int64_t)blockIdx.x * JG_THREADS + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * JG_THREADS) {
    const int c = (int)(k % lg);
    const int64_t r = k / lg;
    const int64_t l = r % max_len;
    const int64_t b = r / max_len;
    const int64_t s = offsets[b], e = offsets[b + 1];
    const bool in = s + l < e;
    float* dp = dense + r * (int64_t)(lg * 4) + 4 * c;
    if (TO_DENSE) {
      tzr_st4(dp, in ? tzr_ld4(jagged + (s + l) * jagged_stride + 4 * c) : make_float4(pad, pad, pad, pad));
    } else if (in) {
      tzr_st4(jagged + (s + l) * jagged_stride + 4 * c, tzr_ld4(dp));
    }
  }
}

// rows of sequences longer than max_len are not covered by the dense tensor: zero their gradient
__global__ __launch_bounds__(JG_THREADS) void tzr_jagged_zero_tail_kernel(
    float* __restrict__ jagged, int64_t jagged_stride, const int64_t* __restrict__ offsets,
    int64_t B, int64_t max_len, int lg) {
  const int64_t b = blockIdx.x;
  const int64_t s = offsets[b] + max_len, e = offsets[b + 1];
  for (int64_t k = threadIdx.x; k < (e - s) * lg; k += JG_THREADS)
    if (s < e) tzr_st4(jagged + (s + k / lg) * jagged_stride + 4 * (k % lg), tzr_zero4());
}

static int jg_check(const float* jagged, int64_t js, const int64_t* offsets, int64_t B,
                    int64_t max_len, int dim, const float* dense) {
  if (!offsets || B < 0 || max_len < 0 || dim <= 0) return TZR_ERR_INVALID;
  if ((dim & 3) || (js & 3) || js < dim) return TZR_ERR_UNSUPPORTED;
  if (B * max_len > 0 && (!dense || (reinterpret_cast<uintptr_t>(dense) & 15))) return TZR_ERR_INVALID;
  if (jagged && (reinterpret_cast<uintptr_t>(jagged) & 15)) return TZR_ERR_INVALID;
  return TZR_OK;
}

extern "C" int tzr_jagged_to_padded_dense(const float* d_values, int64_t values_stride,
                                          const int64_t* d_offsets, int64_t B, int64_t max_len,
                                          int dim, float padding_value, float* d_out,
                                          void* stream) {
  int rc = jg_check(d_values, values_stride, d_offsets, B, max_len, dim, d_out);
  if (rc != TZR_OK) return rc;
  const int64_t total = B * max_len * (dim >> 2);
  if (total == 0) return TZR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + JG_THREADS - 1) / JG_THREADS);
  hipLaunchKernelGGL((tzr_jagged_dense_kernel<true>), dim3(grid), dim3(JG_THREADS), 0,
                     static_cast<hipStream_t>(stream), const_cast<float*>(d_values), values_stride,
                     d_offsets, B, max_len, dim >> 2, d_out, padding_value);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_padded_dense_to_jagged(const float* d_dense, const int64_t* d_offsets,
                                          int64_t B, int64_t max_len, int dim, float* d_values,
                                          int64_t values_stride, void* stream) {
  int rc = jg_check(d_values, values_stride, d_offsets, B, max_len, dim, d_dense);
  if (rc != TZR_OK) return rc;
  if (B == 0) return TZR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tzr_jagged_zero_tail_kernel, dim3((unsigned)B), dim3(JG_THREADS), 0, s,
                     d_values, values_stride, d_offsets, B, max_len, dim >> 2);
  const int64_t total = B * max_len * (dim >> 2);
  if (total > 0) {
    const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + JG_THREADS - 1) / JG_THREADS);
    hipLaunchKernelGGL((tzr_jagged_dense_kernel<false>), dim3(grid), dim3(JG_THREADS), 0, s,
                       d_values, values_stride, d_offsets, B, max_len, dim >> 2,
                       const_cast<float*>(d_dense), 0.0f);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- segment reduce (multi-valued sequence steps) ----------------------------------------------
// Replaces torch.segment_reduce(jt.values(), pooling, lengths=key_lengths) in
// SequenceEmbeddingGroupImpl (/root/reference/tzrec/modules/embedding.py:1353-1366): the rows of
// the ids of one sequence STEP are pooled (sum / mean) into the step's row; an empty step gives a
// zero row (the reference's nan_to_num after a mean over nothing).  Thread = (segment, float4
// chunk): the lanes of a segment read its rows as 16-byte pieces; steps hold a few ids.
template <bool BWD>
__global__ __launch_bounds__(JG_THREADS) void tzr_segment_reduce_kernel(
    float* __restrict__ values, int64_t values_stride, const int64_t* __restrict__ offsets,
    int64_t S, int lg, int mean, float* __restrict__ seg, int64_t seg_stride) {
  const int64_t total = S * lg;
  for (int64_t k = (int64_t)blockIdx.x * JG_THREADS + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * JG_THREADS) {
    const int c = (int)(k % lg);
    const int64_t s = k / lg;
    const int64_t b = offsets[s], e = offsets[s + 1];
    const float scale = (mean && e > b) ? 1.0f / (float)(e - b) : 1.0f;
    float* sp = seg + s * seg_stride + 4 * c;
    if (!BWD) {
      float4 acc = tzr_zero4();
      for (int64_t i = b; i < e; ++i) acc = tzr_add4(acc, tzr_ld4(values + i * values_stride + 4 * c));
      acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
      tzr_st4(sp, acc);
    } else {
      float4 g = tzr_ld4(sp);
      g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
      for (int64_t i = b; i < e; ++i) tzr_st4(values + i * values_stride + 4 * c, g);
    }
  }
}

static int sr_check(const float* values, int64_t vs, const int64_t* offsets, int64_t S, int dim,
                    int mode, const float* seg, int64_t ss) {
  if (!offsets || S < 0 || dim <= 0 || (mode != 0 && mode != 1)) return TZR_ERR_INVALID;
  if ((dim & 3) || (vs & 3) || (ss & 3) || vs < dim || ss < dim) return TZR_ERR_UNSUPPORTED;
  if (S > 0 && (!seg || (reinterpret_cast<uintptr_t>(seg) & 15))) return TZR_ERR_INVALID;
  if (values && (reinterpret_cast<uintptr_t>(values) & 15)) return TZR_ERR_INVALID;
  return TZR_OK;
}

extern "C" int tzr_segment_reduce_fwd(const float* d_values, int64_t values_stride,
                                      const int64_t* d_offsets, int64_t S, int dim, int mode,
                                      float* d_out, int64_t out_stride, void* stream) {
  int rc = sr_check(d_values, values_stride, d_offsets, S, dim, mode, d_out, out_stride);
  if (rc != TZR_OK) return rc;
  const int64_t total = S * (dim >> 2);
  if (total == 0) return TZR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + JG_THREADS - 1) / JG_THREADS);
  hipLaunchKernelGGL((tzr_segment_reduce_kernel<false>), dim3(grid), dim3(JG_THREADS), 0,
                     static_cast<hipStream_t>(stream), const_cast<float*>(d_values), values_stride,
                     d_offsets, S, dim >> 2, mode, d_out, out_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_segment_reduce_bwd(const float* d_grad_out, int64_t grad_out_stride,
                                      const int64_t* d_offsets, int64_t S, int dim, int mode,
                                      float* d_grad_values, int64_t grad_values_stride,
                                      void* stream) {
  int rc = sr_check(d_grad_values, grad_values_stride, d_offsets, S, dim, mode, d_grad_out,
                    grad_out_stride);
  if (rc != TZR_OK) return rc;
  const int64_t total = S * (dim >> 2);
  if (total == 0) return TZR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + JG_THREADS - 1) / JG_THREADS);
  hipLaunchKernelGGL((tzr_segment_reduce_kernel<true>), dim3(grid), dim3(JG_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_grad_values, grad_values_stride, d_offsets,
                     S, dim >> 2, mode, const_cast<float*>(d_grad_out), grad_out_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
