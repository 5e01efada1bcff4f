// DIN target attention over JAGGED positions (BASELINE configs[3] multi_tower_din).
//
// Replaces DINEncoder.forward of the reference (/root/reference/tzrec/modules/sequence.py:101-128), which pads every
// sample's click sequence to the batch's longest (SequenceEmbeddingGroupImpl: to_padded_dense,
// /root/reference/tzrec/modules/embedding.py:1429-1480) and runs the attention MLP, the masked softmax and the weighted
// sum over [B, L, .] tensors: at `sequence_length: 100` with histories of 10..100 clicks ~45 % of the positions are padding.
// Here a position is a ROW of the unpooled lookup's output, [N = sum of lengths, D], and nothing is ever padded:
//
//   tzr_jagged_segment_ids   position -> sample (binary search in the offsets, once per batch)
//   tzr_din_assemble_fwd     X[n] = [ k_n | q_b * k_n | q_b ]                         (b = sample of position n)
//                            The reference's first layer W [q, k, q - k, q * k] is (Wb - Wc) k + Wd (q * k) + (Wa + Wc) q:
//                            the same sums over three D-wide blocks instead of four (the caller folds the weights).
//   (the attention MLP runs on the [N, 3 D] rows: plain products, the GEMM library)
//   tzr_din_attn_fwd         s_n = h_n . w + c;  p = softmax of s over a sample's positions;  out_b = sum_n p_n k_n
//   tzr_din_attn_bwd         ds_n = p_n (g_b . k_n - sum_m p_m g_b . k_m);  dk_n = p_n g_b
//   tzr_din_assemble_bwd     dk_n += dX_n[0:D] + q_b * dX_n[D:2D];  dq_b = sum_n (k_n * dX_n[D:2D] + dX_n[2D:3D])
//
// Masking semantics of the reference, restated for rows: positions behind `max_len` (the padded length / max_seq_length)
// do not exist; the scores of padding positions are -(2^31 - 1), i.e. exp() = 0 exactly in fp32 next to any real score, so
// the softmax of a sample with >= 1 position is the softmax over its positions; a sample with NO position gets a uniform
// softmax over padding rows that are all zero: out = 0 and no gradient.
//
// One wave per sample in the attention kernels (a sample's positions are consecutive rows): every sum over a sample's
// positions has a fixed order -- no atomics, bit-reproducible.  All of it is HBM streaming of [N, .] rows.
#include "tzr_common.h"

#define DA_THREADS 256
#define DA_WAVES (DA_THREADS / TZR_WAVE)
#define DA_MAXLEN 2048  // positions of one sample whose scores a wave keeps in LDS

__global__ __launch_bounds__(DA_THREADS) void tzr_jagged_segment_ids_kernel(const int64_t* __restrict__ offsets, int64_t B,
                                                                            int64_t N, int32_t* __restrict__ seg) {
  for (int64_t n = (int64_t)blockIdx.x * DA_THREADS + threadIdx.x; n < N; n += (int64_t)gridDim.x * DA_THREADS) {
    // last b with offsets[b] <= n (offsets[0] = 0 <= n < offsets[B]); positions behind offsets[B] get B
    seg[n] = n >= offsets[B] ? (int32_t)B : (int32_t)tzr_last_le(offsets, B, n);
  }
}

extern "C" int tzr_jagged_segment_ids(const int64_t* d_offsets, int64_t B, int64_t N, int32_t* d_seg, void* stream) {
  if (!d_offsets || B < 0 || N < 0 || B >= (1LL << 31)) return TZR_ERR_INVALID;
  if (N == 0) return TZR_OK;
  if (!d_seg || B == 0) return TZR_ERR_INVALID;
  const unsigned grid = (unsigned)std::min<int64_t>(8192, (N + DA_THREADS - 1) / DA_THREADS);
  hipLaunchKernelGGL(tzr_jagged_segment_ids_kernel, dim3(grid), dim3(DA_THREADS), 0, static_cast<hipStream_t>(stream), d_offsets, B, N,
                     d_seg);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// thread = (position n, float4 chunk c of D)
template <bool Q3 /* the query as a third block of the row */>
__global__ __launch_bounds__(DA_THREADS) void tzr_din_assemble_fwd_kernel(
    const float* __restrict__ kv, int64_t kvs, const float* __restrict__ q, int64_t qs, const int32_t* __restrict__ seg,
    int64_t B, int64_t N, int lg, float* __restrict__ X, int64_t xs) {
  const int64_t total = N * lg;
  for (int64_t k = (int64_t)blockIdx.x * DA_THREADS + threadIdx.x; k < total; k += (int64_t)gridDim.x * DA_THREADS) {
    const int c = (int)(k % lg);
    const int64_t n = k / lg;
    const int64_t b = seg[n];
    float4 kk = tzr_zero4(), qq = tzr_zero4();
    if (b < B) {  // (a row behind the last sample -- capacity padding -- is zero)
      kk = tzr_ld4(kv + n * kvs + 4 * c);
      qq = tzr_ld4(q + b * qs + 4 * c);
    }
    float* xp = X + n * xs + 4 * c;
    tzr_st4(xp, kk);
    tzr_st4(xp + 4 * lg, make_float4(qq.x * kk.x, qq.y * kk.y, qq.z * kk.z, qq.w * kk.w));
    if (Q3) tzr_st4(xp + 8 * lg, qq);
  }
}

static int din_assemble_fwd(bool q3, const float* d_kv, int64_t kv_stride, const float* d_q, int64_t q_stride, const int32_t* d_seg,
                            int64_t B, int64_t N, int D, float* d_X, int64_t x_stride, void* stream) {
  if (B < 0 || N < 0 || D <= 0) return TZR_ERR_INVALID;
  if ((D & 3) || (kv_stride & 3) || (q_stride & 3) || (x_stride & 3) || kv_stride < D || q_stride < D || x_stride < (q3 ? 3 : 2) * D)
    return TZR_ERR_UNSUPPORTED;
  if (N == 0) return TZR_OK;
  if (!d_kv || !d_q || !d_seg || !d_X ||
      ((reinterpret_cast<uintptr_t>(d_kv) | reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_X)) & 15))
    return TZR_ERR_INVALID;
  const int64_t total = N * (D >> 2);
  const unsigned grid = (unsigned)std::min<int64_t>(16384, (total + DA_THREADS - 1) / DA_THREADS);
  if (q3)
    hipLaunchKernelGGL(tzr_din_assemble_fwd_kernel<true>, dim3(grid), dim3(DA_THREADS), 0, static_cast<hipStream_t>(stream), d_kv, kv_stride,
                       d_q, q_stride, d_seg, B, N, D >> 2, d_X, x_stride);
  else
    hipLaunchKernelGGL(tzr_din_assemble_fwd_kernel<false>, dim3(grid), dim3(DA_THREADS), 0, static_cast<hipStream_t>(stream), d_kv,
                       kv_stride, d_q, q_stride, d_seg, B, N, D >> 2, d_X, x_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_din_assemble_fwd(const float* d_kv, int64_t kv_stride, const float* d_q, int64_t q_stride,
                                    const int32_t* d_seg, int64_t B, int64_t N, int D, float* d_X, int64_t x_stride,
                                    void* stream) {
  return din_assemble_fwd(true, d_kv, kv_stride, d_q, q_stride, d_seg, B, N, D, d_X, x_stride, stream);
}

// X[n] = [ k_n | q_b * k_n ] only: the query's own block of the first layer is one product per SAMPLE (tzr_linear_rows' row vector)
extern "C" int tzr_din_assemble2_fwd(const float* d_kv, int64_t kv_stride, const float* d_q, int64_t q_stride,
                                     const int32_t* d_seg, int64_t B, int64_t N, int D, float* d_X, int64_t x_stride,
                                     void* stream) {
  return din_assemble_fwd(false, d_kv, kv_stride, d_q, q_stride, d_seg, B, N, D, d_X, x_stride, stream);
}

// dk: thread = (position, chunk).  `acc` != 0: dk is added to what d_dkv holds (the attention's direct part).
__global__ __launch_bounds__(DA_THREADS) void tzr_din_assemble_bwd_k_kernel(
    const float* __restrict__ dX, int64_t xs, const float* __restrict__ q, int64_t qs, const int32_t* __restrict__ seg, int64_t B,
    int64_t N, int lg, float* __restrict__ dkv, int64_t dks, int acc) {
  const int64_t total = N * lg;
  for (int64_t k = (int64_t)blockIdx.x * DA_THREADS + threadIdx.x; k < total; k += (int64_t)gridDim.x * DA_THREADS) {
    const int c = (int)(k % lg);
    const int64_t n = k / lg;
    const int64_t b = seg[n];
    float4 g = tzr_zero4();
    if (b < B) {
      const float* xp = dX + n * xs + 4 * c;
      const float4 g0 = tzr_ld4(xp), g1 = tzr_ld4(xp + 4 * lg), qq = tzr_ld4(q + b * qs + 4 * c);
      g = make_float4(fmaf(qq.x, g1.x, g0.x), fmaf(qq.y, g1.y, g0.y), fmaf(qq.z, g1.z, g0.z), fmaf(qq.w, g1.w, g0.w));
      if (acc) g = tzr_add4(tzr_ld4(dkv + n * dks + 4 * c), g);
    }
    tzr_st4(dkv + n * dks + 4 * c, g);
  }
}

// dq: one WAVE per sample (a thread per (sample, chunk) walking the sample's positions made the lanes of a wave wait for the
// longest of their five samples: 77 us, profiles/r05x).  Lane (g, c): position group g = lane / lg of 64 / lg, chunk c; the
// groups take positions g, g + G, ... and are added in group order at the end (fixed order: a function of the lengths alone).
template <bool Q3>
__global__ __launch_bounds__(DA_THREADS) void tzr_din_assemble_bwd_q_kernel(
    const float* __restrict__ dX, int64_t xs, const float* __restrict__ kv, int64_t kvs, const int64_t* __restrict__ offsets,
    int64_t B, int lg, const float* __restrict__ add, int64_t adds, float* __restrict__ dq, int64_t dqs) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int G = TZR_WAVE / lg;  // position groups of a wave (lg <= 64)
  const int g = lane / lg, c = lane - g * lg;
  const bool on = g < G;
  for (int64_t b = (int64_t)blockIdx.x * DA_WAVES + wv; b < B; b += (int64_t)gridDim.x * DA_WAVES) {
    const int64_t s = offsets[b], e = offsets[b + 1];
    float4 a = tzr_zero4();
    if (on)
      for (int64_t n = s + g; n < e; n += G) {
        const float* xp = dX + n * xs + 4 * c;
        const float4 g1 = tzr_ld4(xp + 4 * lg), g2 = Q3 ? tzr_ld4(xp + 8 * lg) : tzr_zero4(), kk = tzr_ld4(kv + n * kvs + 4 * c);
        a.x += fmaf(kk.x, g1.x, g2.x); a.y += fmaf(kk.y, g1.y, g2.y); a.z += fmaf(kk.z, g1.z, g2.z); a.w += fmaf(kk.w, g1.w, g2.w);
      }
    float4 t = a;  // group 0's lanes collect the groups in order
    for (int q = 1; q < G; ++q) {
      const int src = q * lg + (lane < lg ? lane : 0);
      const float4 o = make_float4(__shfl(a.x, src), __shfl(a.y, src), __shfl(a.z, src), __shfl(a.w, src));
      if (lane < lg) t = tzr_add4(t, o);
    }
    if (lane < lg) {
      if (add) t = tzr_add4(t, tzr_ld4(add + b * adds + 4 * lane));
      tzr_st4(dq + b * dqs + 4 * lane, t);
    }
  }
}

static int din_assemble_bwd(bool q3, const float* d_dX, int64_t x_stride, const float* d_kv, int64_t kv_stride, const float* d_q,
                            int64_t q_stride, const int32_t* d_seg, const int64_t* d_offsets, int64_t B, int64_t N, int D, float* d_dkv,
                            int64_t dkv_stride, int accumulate_dkv, const float* d_dq_add, int64_t dq_add_stride, float* d_dq,
                            int64_t dq_stride, void* stream) {
  if (B < 0 || N < 0 || D <= 0) return TZR_ERR_INVALID;
  if ((D & 3) || D > 4 * TZR_WAVE || ((kv_stride | q_stride | x_stride | dkv_stride | dq_stride | dq_add_stride) & 3) || kv_stride < D ||
      q_stride < D || x_stride < (q3 ? 3 : 2) * D || dkv_stride < D || dq_stride < D || (d_dq_add && dq_add_stride < D))
    return TZR_ERR_UNSUPPORTED;  // (D <= 256: a wave holds a row of the query gradient)
  if (d_dq_add && (reinterpret_cast<uintptr_t>(d_dq_add) & 15)) return TZR_ERR_INVALID;
  if (B > 0 && (!d_offsets || !d_dq || (reinterpret_cast<uintptr_t>(d_dq) & 15))) return TZR_ERR_INVALID;
  if (N > 0 && (!d_dX || !d_kv || !d_q || !d_seg || !d_dkv ||
                ((reinterpret_cast<uintptr_t>(d_dX) | reinterpret_cast<uintptr_t>(d_kv) | reinterpret_cast<uintptr_t>(d_q) |
                  reinterpret_cast<uintptr_t>(d_dkv)) & 15)))
    return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int lg = D >> 2;
  if (N > 0) {
    const unsigned grid = (unsigned)std::min<int64_t>(16384, (N * lg + DA_THREADS - 1) / DA_THREADS);
    hipLaunchKernelGGL(tzr_din_assemble_bwd_k_kernel, dim3(grid), dim3(DA_THREADS), 0, s, d_dX, x_stride, d_q, q_stride, d_seg, B, N,
                       lg, d_dkv, dkv_stride, accumulate_dkv);
  }
  if (B > 0) {
    const unsigned grid = (unsigned)std::min<int64_t>(16384, (B + DA_WAVES - 1) / DA_WAVES);
    if (q3)
      hipLaunchKernelGGL(tzr_din_assemble_bwd_q_kernel<true>, dim3(grid), dim3(DA_THREADS), 0, s, d_dX, x_stride, d_kv, kv_stride,
                         d_offsets, B, lg, d_dq_add, dq_add_stride, d_dq, dq_stride);
    else
      hipLaunchKernelGGL(tzr_din_assemble_bwd_q_kernel<false>, dim3(grid), dim3(DA_THREADS), 0, s, d_dX, x_stride, d_kv, kv_stride,
                         d_offsets, B, lg, d_dq_add, dq_add_stride, d_dq, dq_stride);
  }
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_din_assemble_bwd(const float* d_dX, int64_t x_stride, const float* d_kv, int64_t kv_stride,
                                    const float* d_q, int64_t q_stride, const int32_t* d_seg, const int64_t* d_offsets,
                                    int64_t B, int64_t N, int D, float* d_dkv, int64_t dkv_stride, int accumulate_dkv,
                                    float* d_dq, int64_t dq_stride, void* stream) {
  return din_assemble_bwd(true, d_dX, x_stride, d_kv, kv_stride, d_q, q_stride, d_seg, d_offsets, B, N, D, d_dkv, dkv_stride,
                          accumulate_dkv, nullptr, 0, d_dq, dq_stride, stream);
}

// backward of tzr_din_assemble2_fwd: dk_n (+)= dX_n[0:D] + q_b * dX_n[D:2D];  dq_b = sum_n k_n * dX_n[D:2D] + d_dq_add[b]
// (d_dq_add: the query's gradient through its own per-sample block of the first layer, or NULL)
extern "C" int tzr_din_assemble2_bwd(const float* d_dX, int64_t x_stride, const float* d_kv, int64_t kv_stride,
                                     const float* d_q, int64_t q_stride, const int32_t* d_seg, const int64_t* d_offsets,
                                     int64_t B, int64_t N, int D, float* d_dkv, int64_t dkv_stride, int accumulate_dkv,
                                     const float* d_dq_add, int64_t dq_add_stride, float* d_dq, int64_t dq_stride, void* stream) {
  return din_assemble_bwd(false, d_dX, x_stride, d_kv, kv_stride, d_q, q_stride, d_seg, d_offsets, B, N, D, d_dkv, dkv_stride,
                          accumulate_dkv, d_dq_add, dq_add_stride, d_dq, dq_stride, stream);
}

// ---- scores, softmax over a sample's positions, weighted sum of its rows: one wave per sample ------------------------
// 16 lanes per position (float4 pieces of its row: a 64-float hidden row is one 256-byte read of the group), four positions
// per wave pass; the group's dot product by a fixed shuffle tree.
__device__ __forceinline__ float da_group16_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
__device__ __forceinline__ float da_wave_max(float v) {
  for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float da_wave_sum(float v) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// row_n . w for the positions [s, s + len) of one sample -> sc[0 .. len) (LDS of this wave); lanes (g = lane / 16, j = lane % 16)
__device__ __forceinline__ void da_row_dots(const float* __restrict__ rows, int64_t stride, int64_t s, int len, int cols4,
                                            const float* __restrict__ w, float bias, float* sc, int lane) {
  const int g = lane >> 4, j = lane & 15;
  // sixteen positions per pass: four per 16-lane group, their row loads in flight TOGETHER (a position behind the sample's last
  // re-reads the last: no load sits inside a branch).  One position per pass was one dependent round trip per four positions,
  // 14 in a row for a 55-click history (72 us for the forward kernel, profiles/r05x).  Same arithmetic per position.
  for (int i0 = 0; i0 < len; i0 += 16) {
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u + g;
      const int ic = i < len ? i : len - 1;
      a[u] = 0.f;
      for (int c = j; c < cols4; c += 16) {
        const float4 x = tzr_ld4(rows + (s + ic) * stride + 4 * c), ww = tzr_ld4(w + 4 * c);
        a[u] = fmaf(x.x, ww.x, a[u]); a[u] = fmaf(x.y, ww.y, a[u]); a[u] = fmaf(x.z, ww.z, a[u]); a[u] = fmaf(x.w, ww.w, a[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u + g;
      const float t = da_group16_sum(a[u]);
      if (i < len && j == 0) sc[i] = t + bias;
    }
  }
}

__global__ __launch_bounds__(DA_THREADS) void tzr_din_attn_fwd_kernel(
    const float* __restrict__ h, int64_t hs, int H, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ kv, int64_t kvs, int D, const int64_t* __restrict__ offsets, int64_t B, int64_t max_len,
    float* __restrict__ out, int64_t outs, float* __restrict__ p) {
  __shared__ float scs[DA_WAVES][DA_MAXLEN];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  float* sc = scs[wv];
  const float b3 = bias ? bias[0] : 0.f;
  for (int64_t b = (int64_t)blockIdx.x * DA_WAVES + wv; b < B; b += (int64_t)gridDim.x * DA_WAVES) {
    const int64_t s = offsets[b], e = offsets[b + 1];
    const int len = (int)min(e - s, max_len);
    // positions behind max_len do not exist for the encoder: probability 0
    for (int64_t n = s + len + lane; n < e; n += TZR_WAVE) p[n] = 0.f;
    da_row_dots(h, hs, s, len, H >> 2, w, b3, sc, lane);
    __builtin_amdgcn_wave_barrier();
    float mx = -3.402823466e38f;
    for (int i = lane; i < len; i += TZR_WAVE) mx = fmaxf(mx, sc[i]);
    mx = da_wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < len; i += TZR_WAVE) {
      const float ex = expf(sc[i] - mx);
      sc[i] = ex;
      sum += ex;
    }
    sum = da_wave_sum(sum);  // (fixed tree)
    const float inv = len > 0 ? 1.0f / sum : 0.f;
    for (int i = lane; i < len; i += TZR_WAVE) {
      const float pi = sc[i] * inv;
      sc[i] = pi;
      p[s + i] = pi;
    }
    __builtin_amdgcn_wave_barrier();
    // out_b = sum_i p_i k_i: lane (g, j) sums positions g, g + 4, ... of chunk j (+ 16, ...), the four groups are added in order
    const int g = lane >> 4, j = lane & 15;
    for (int c = j; c < (D >> 2); c += 16) {
      float4 a = tzr_zero4();
      for (int i0 = g; i0 < len; i0 += 16) {  // four of the group's positions per pass, loads together, added in position order
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 4 * u;
          r[u] = tzr_ld4(kv + (s + (i < len ? i : len - 1)) * kvs + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + 4 * u < len) a = tzr_fma4(sc[i0 + 4 * u], r[u], a);
      }
      a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
      a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
      if (g == 0) tzr_st4(out + b * outs + 4 * c, a);
    }
    __builtin_amdgcn_wave_barrier();  // (the next sample overwrites sc)
  }
}

extern "C" int tzr_din_attn_fwd(const float* d_h, int64_t h_stride, int H, const float* d_w, const float* d_bias,
                                const float* d_kv, int64_t kv_stride, int D, const int64_t* d_offsets, int64_t B,
                                int64_t max_len, float* d_out, int64_t out_stride, float* d_p, void* stream) {
  if (B < 0 || H <= 0 || D <= 0 || max_len < 0) return TZR_ERR_INVALID;
  if ((H & 3) || (D & 3) || ((h_stride | kv_stride | out_stride) & 3) || h_stride < H || kv_stride < D || out_stride < D ||
      max_len > DA_MAXLEN)
    return TZR_ERR_UNSUPPORTED;
  if (B == 0) return TZR_OK;
  if (!d_offsets || !d_out || !d_w || (reinterpret_cast<uintptr_t>(d_out) & 15) || (reinterpret_cast<uintptr_t>(d_w) & 15))
    return TZR_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(d_h) | reinterpret_cast<uintptr_t>(d_kv)) & 15) return TZR_ERR_INVALID;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, (B + DA_WAVES - 1) / DA_WAVES);
  hipLaunchKernelGGL(tzr_din_attn_fwd_kernel, dim3(grid), dim3(DA_THREADS), 0, static_cast<hipStream_t>(stream), d_h, h_stride, H, d_w,
                     d_bias, d_kv, kv_stride, D, d_offsets, B, max_len, d_out, out_stride, d_p);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ds_n = p_n (g_b . k_n - sum_m p_m g_b . k_m), dk_n = p_n g_b (the direct path of the weighted sum); one wave per sample
__global__ __launch_bounds__(DA_THREADS) void tzr_din_attn_bwd_kernel(
    const float* __restrict__ gout, int64_t gos, const float* __restrict__ p, const float* __restrict__ kv, int64_t kvs, int D,
    const int64_t* __restrict__ offsets, int64_t B, int64_t max_len, float* __restrict__ ds, float* __restrict__ dkv, int64_t dks) {
  __shared__ float scs[DA_WAVES][DA_MAXLEN];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  float* sc = scs[wv];
  for (int64_t b = (int64_t)blockIdx.x * DA_WAVES + wv; b < B; b += (int64_t)gridDim.x * DA_WAVES) {
    const int64_t s = offsets[b], e = offsets[b + 1];
    const int len = (int)min(e - s, max_len);
    const int lg = D >> 2;
    for (int64_t n = s + len + lane; n < e; n += TZR_WAVE) ds[n] = 0.f;
    for (int64_t k = (int64_t)len * lg + lane; k < (e - s) * lg; k += TZR_WAVE) tzr_st4(dkv + (s + k / lg) * dks + 4 * (k % lg), tzr_zero4());
    da_row_dots(kv, kvs, s, len, lg, gout + b * gos, 0.f, sc, lane);  // sc[i] = g_b . k_i
    __builtin_amdgcn_wave_barrier();
    float dot = 0.f;
    for (int i = lane; i < len; i += TZR_WAVE) dot = fmaf(p[s + i], sc[i], dot);
    dot = da_wave_sum(dot);
    for (int i = lane; i < len; i += TZR_WAVE) ds[s + i] = p[s + i] * (sc[i] - dot);
    for (int k = lane; k < len * lg; k += TZR_WAVE) {
      const int i = k / lg, c = k - i * lg;
      const float pi = p[s + i];
      const float4 g = tzr_ld4(gout + b * gos + 4 * c);
      tzr_st4(dkv + (s + i) * dks + 4 * c, make_float4(pi * g.x, pi * g.y, pi * g.z, pi * g.w));
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int tzr_din_attn_bwd(const float* d_grad_out, int64_t grad_out_stride, const float* d_p, const float* d_kv,
                                int64_t kv_stride, int D, const int64_t* d_offsets, int64_t B, int64_t max_len, float* d_ds,
                                float* d_dkv, int64_t dkv_stride, void* stream) {
  if (B < 0 || D <= 0 || max_len < 0) return TZR_ERR_INVALID;
  if ((D & 3) || ((grad_out_stride | kv_stride | dkv_stride) & 3) || grad_out_stride < D || kv_stride < D || dkv_stride < D ||
      max_len > DA_MAXLEN)
    return TZR_ERR_UNSUPPORTED;
  if (B == 0) return TZR_OK;
  if (!d_offsets || !d_grad_out || (reinterpret_cast<uintptr_t>(d_grad_out) & 15)) return TZR_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(d_kv) | reinterpret_cast<uintptr_t>(d_dkv)) & 15) return TZR_ERR_INVALID;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, (B + DA_WAVES - 1) / DA_WAVES);
  hipLaunchKernelGGL(tzr_din_attn_bwd_kernel, dim3(grid), dim3(DA_THREADS), 0, static_cast<hipStream_t>(stream), d_grad_out,
                     grad_out_stride, d_p, d_kv, kv_stride, D, d_offsets, B, max_len, d_ds, d_dkv, dkv_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
