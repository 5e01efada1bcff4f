// Row-wise INT8 export of embedding tables.
//
// Replaces _quantize_quint8_rowwise_f16 / dequantize_quint8_rowwise_f16
// (/root/reference/tzrec/utils/quant_util.py:25-131,158-196), the numpy encoder reached from the
// distributed-sparse export (tzrec/utils/export_util.py:2353) and the delta-embedding dump: the
// QUint8RowwiseF16 row is [dim uint8 values][float16 scale][float16 offset], dim + 4 bytes.
// Byte-exact with the reference: every step is the same IEEE single-precision operation numpy
// performs (min / max, round-to-nearest-even float16 conversions, a correctly rounded division,
// rint, clip), none of them contracted into an fma.
//
// HBM streaming: D/4 consecutive lanes own one row (16-byte loads), the row's min / max travel
// through lane shuffles, each lane emits its four bytes as one 32-bit store and the group's first
// lane the {scale, offset} word, so a wave writes one contiguous run of (64 / (D/4)) rows.
#include "tzr_common.h"

#define EX_THREADS 256
#define EX_FP16_MAX 65504.0f
#define EX_U 4

__device__ __forceinline__ float ex_group_min(float v, int lg, int lig, int lane) {
  if ((lg & (lg - 1)) == 0) {
    for (int m = lg >> 1; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
  }
  float r = v;
  const int g0 = lane - lig;
  for (int l = 0; l < lg; ++l) r = fminf(r, __shfl(v, g0 + l, 64));
  return r;
}
__device__ __forceinline__ float ex_group_max(float v, int lg, int lig, int lane) {
  if ((lg & (lg - 1)) == 0) {
    for (int m = lg >> 1; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
  }
  float r = v;
  const int g0 = lane - lig;
  for (int l = 0; l < lg; ++l) r = fmaxf(r, __shfl(v, g0 + l, 64));
  return r;
}
__device__ __forceinline__ int ex_group_or(int v, int lg, int lig, int lane) {
  if ((lg & (lg - 1)) == 0) {
    for (int m = lg >> 1; m > 0; m >>= 1) v |= __shfl_xor(v, m, 64);
    return v;
  }
  int r = 0;
  const int g0 = lane - lig;
  for (int l = 0; l < lg; ++l) r |= __shfl(v, g0 + l, 64);
  return r;
}

__device__ __forceinline__ unsigned ex_f16_bits(float v) {
  const _Float16 h = (_Float16)v;  // round to nearest even, overflow -> inf (as numpy astype)
  unsigned short b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}
__device__ __forceinline__ float ex_f16_round(float v) { return (float)(_Float16)v; }

__device__ __forceinline__ unsigned ex_q(float x, float offset, float scale) {
  // np.subtract, np.divide, np.rint, np.clip in float32
  float q = rintf((x - offset) / scale);
  q = fminf(fmaxf(q, 0.0f), 255.0f);
  return (unsigned)q;
}

// bad[0] / bad[1] / bad[2]: smallest row with a non-finite value / an offset / a scale outside the
// finite float16 range (initialised to INT64_MAX by the launcher)
__global__ __launch_bounds__(EX_THREADS) void tzr_quantize_rows_kernel(
    const void* __restrict__ w, int w_dtype, int64_t w_stride, int64_t rows, int lg,
    unsigned char* __restrict__ out, unsigned long long* __restrict__ bad) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int gpw = TZR_WAVE / lg;            // rows per wave
  const int lig = lane % lg;                // float4 chunk of the row
  const int gi = lane / lg;                 // row of the wave (gi >= gpw: idle tail lanes)
  const int D = lg * 4;
  const int64_t row_bytes = D + 4;
  const int64_t waves = ((int64_t)gridDim.x * EX_THREADS) / TZR_WAVE;
  const int64_t wave0 = ((int64_t)blockIdx.x * EX_THREADS + threadIdx.x) / TZR_WAVE;
  // EX_U row slots per wave and iteration: all their loads are issued before the first reduction
  for (int64_t r0 = wave0 * gpw * EX_U; r0 < rows; r0 += waves * gpw * EX_U) {
    float4 v[EX_U];
#pragma unroll
    for (int u = 0; u < EX_U; ++u) {
      const int64_t row = r0 + (int64_t)u * gpw + gi;
      v[u] = (gi < gpw && row < rows) ? tzr_ldw4(w, w_dtype, row * w_stride + 4 * lig) : tzr_zero4();
    }
#pragma unroll
    for (int u = 0; u < EX_U; ++u) {
      const int64_t row = r0 + (int64_t)u * gpw + gi;
      const bool on = gi < gpw && row < rows;
      const float4 x = v[u];
      const int nonfinite = on && !(__builtin_isfinite(x.x) && __builtin_isfinite(x.y) &&
                                    __builtin_isfinite(x.z) && __builtin_isfinite(x.w));
      float mn = fminf(fminf(x.x, x.y), fminf(x.z, x.w));
      float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
      // every lane takes part in the shuffles; groups are lg-aligned, the tail lanes of a wave
      // (gi >= gpw) reduce among themselves and are dropped
      mn = ex_group_min(mn, lg, lig, lane);
      mx = ex_group_max(mx, lg, lig, lane);
      const int bad_row = ex_group_or(nonfinite, lg, lig, lane);
      if (on) {
        const float offset = ex_f16_round(mn);
        const double vr64 = (double)mx - (double)offset;
        const bool bad_offset = fabsf(mn) > EX_FP16_MAX;
        const bool bad_scale = (!__builtin_isfinite(vr64) || fabs(vr64) > (double)EX_FP16_MAX * 255.0) && vr64 != 0.0;
        if (lig == 0) {
          if (bad_row) atomicMin(&bad[0], (unsigned long long)row);
          else if (bad_offset) atomicMin(&bad[1], (unsigned long long)row);
          else if (bad_scale) atomicMin(&bad[2], (unsigned long long)row);
        }
        const float vr = mx - offset;
        float scale = (vr != 0.0f) ? vr / 255.0f : 1.0f;
        scale = ex_f16_round(scale);
        if (scale == 0.0f) scale = 1.0f;
        const unsigned sb = ex_f16_bits(scale);
        unsigned char* o = out + row * row_bytes;
        const unsigned word = ex_q(x.x, offset, scale) | (ex_q(x.y, offset, scale) << 8) |
                              (ex_q(x.z, offset, scale) << 16) | (ex_q(x.w, offset, scale) << 24);
        *reinterpret_cast<unsigned*>(o + 4 * lig) = word;
        if (lig == 0) *reinterpret_cast<unsigned*>(o + D) = sb | (ex_f16_bits(mn) << 16);
      }
    }
  }
}

__global__ void tzr_export_init_kernel(unsigned long long* bad) {
  if (threadIdx.x < 3) bad[threadIdx.x] = 0x7fffffffffffffffULL;
}

// value * scale + offset with TWO roundings (numpy evaluates the product first): contraction into
// an fma is switched off for this expression (__fmul_rn / __fadd_rn do not stop it)
__device__ __forceinline__ float ex_mul_then_add(float a, float b, float c) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const float p = a * b;
  return p + c;
}

__global__ __launch_bounds__(EX_THREADS) void tzr_dequantize_rows_kernel(
    const unsigned char* __restrict__ q, int64_t rows, int lg, float* __restrict__ out,
    int64_t out_stride) {
  const int D = lg * 4;
  const int64_t row_bytes = D + 4;
  const int64_t total = rows * lg;
  for (int64_t k = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * EX_THREADS) {
    const int64_t row = k / lg;
    const int c = (int)(k - row * lg);
    const unsigned char* p = q + row * row_bytes;
    const unsigned word = *reinterpret_cast<const unsigned*>(p + 4 * c);
    const unsigned meta = *reinterpret_cast<const unsigned*>(p + D);
    unsigned short sb = (unsigned short)(meta & 0xffff), ob = (unsigned short)(meta >> 16);
    _Float16 sh, oh;
    __builtin_memcpy(&sh, &sb, 2);
    __builtin_memcpy(&oh, &ob, 2);
    const float s = (float)sh, off = (float)oh;
    float4 v;
    v.x = ex_mul_then_add((float)(word & 255u), s, off);
    v.y = ex_mul_then_add((float)((word >> 8) & 255u), s, off);
    v.z = ex_mul_then_add((float)((word >> 16) & 255u), s, off);
    v.w = ex_mul_then_add((float)(word >> 24), s, off);
    tzr_st4(out + row * out_stride + 4 * c, v);
  }
}

extern "C" int tzr_quantize_rows_q8f16(const void* d_w, int w_dtype, int64_t w_stride, int64_t rows,
                                       int dim, uint8_t* d_out, int64_t* d_first_bad,
                                       void* stream) {
  if (rows < 0 || dim <= 0 || !d_first_bad) return TZR_ERR_INVALID;
  if ((dim & 3) || dim > 256 || (w_dtype != TZR_DT_F32 && w_dtype != TZR_DT_F16)) return TZR_ERR_UNSUPPORTED;
  if ((w_stride & 3) || w_stride < dim) return TZR_ERR_UNSUPPORTED;
  if (rows > 0 && (!d_w || !d_out)) return TZR_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(d_w) & 15) || (reinterpret_cast<uintptr_t>(d_out) & 3)) return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tzr_export_init_kernel, dim3(1), dim3(TZR_WAVE), 0, s,
                     reinterpret_cast<unsigned long long*>(d_first_bad));
  if (rows > 0) {
    const int lg = dim >> 2;
    const int64_t gpw = TZR_WAVE / lg;
    const int64_t waves = (rows + gpw * EX_U - 1) / (gpw * EX_U);
    const int64_t wg = (waves + (EX_THREADS / TZR_WAVE) - 1) / (EX_THREADS / TZR_WAVE);
    hipLaunchKernelGGL(tzr_quantize_rows_kernel, dim3((unsigned)std::min<int64_t>(wg, 65536)),
                       dim3(EX_THREADS), 0, s, d_w, w_dtype, w_stride, rows, lg, d_out,
                       reinterpret_cast<unsigned long long*>(d_first_bad));
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_dequantize_rows_q8f16(const uint8_t* d_rows, int64_t rows, int dim, float* d_out,
                                         int64_t out_stride, void* stream) {
  if (rows < 0 || dim <= 0) return TZR_ERR_INVALID;
  if ((dim & 3) || (out_stride & 3) || out_stride < dim) return TZR_ERR_UNSUPPORTED;
  if (rows == 0) return TZR_OK;
  if (!d_rows || !d_out || (reinterpret_cast<uintptr_t>(d_rows) & 3) || (reinterpret_cast<uintptr_t>(d_out) & 15))
    return TZR_ERR_INVALID;
  const int64_t total = rows * (dim >> 2);
  const unsigned grid = (unsigned)std::min<int64_t>(65536, (total + EX_THREADS - 1) / EX_THREADS);
  hipLaunchKernelGGL(tzr_dequantize_rows_kernel, dim3(grid), dim3(EX_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_rows, rows, dim >> 2, d_out, out_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
