// K9 / K10: DLRM dot interaction and FM second-order term for gfx950.
//
// K9 replaces InteractionArch.forward (/root/reference/tzrec/modules/interaction.py:80-91:
// torch.bmm(X, X^T) + strict-upper-triangle gather) fused with the concatenations of
// DLRM.predict (/root/reference/tzrec/models/dlrm.py:123-130).  K10 replaces
// FactorizationMachine.forward (/root/reference/tzrec/modules/fm.py:27-42).
//
// Both ops are HBM-bound (K9: 1728 B in + up to 3132 B out per sample for 23 kflop).  MFMA is used
// only for the dense pairwise contraction: one wave per sample, n <= 32 rows padded to two 16-row
// blocks, K = D = 16, exact fp32 v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain, no TF32 on
// gfx950).  Fragment maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15]; accumulator reg r of lane l is D[row=(l>>4)*4+r][col=l&15].
#include "tzr_common.h"

#define IA_THREADS 256
#define IA_WAVES (IA_THREADS / TZR_WAVE)
#define IA_MAXN 32
#define IA_D 16
#define IA_MAXP (IA_MAXN * (IA_MAXN - 1) / 2)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(4))) tzr_f4u {
  float x, y, z, w;
};
__device__ __forceinline__ void tzr_st4_a4(float* p, float4 v) {  // 4-byte aligned 16-byte store
  tzr_f4u u;
  u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
  *reinterpret_cast<tzr_f4u*>(p) = u;
}
__device__ __forceinline__ float4 tzr_ld4_a4(const float* p) {
  const tzr_f4u u = *reinterpret_cast<const tzr_f4u*>(p);
  return make_float4(u.x, u.y, u.z, u.w);
}

// row i of X[b]: the dense row (if any) is row 0, sparse rows follow.
__device__ __forceinline__ const float* ia_row(const float* dense, int64_t dense_stride,
                                               const float* sparse, int64_t sparse_stride,
                                               int64_t b, int i, int hd) {
  return (hd && i == 0) ? dense + b * dense_stride
                        : sparse + b * sparse_stride + (int64_t)(i - hd) * IA_D;
}

// Each wave stages through its own LDS slice, so producer and consumer lanes are always in the same
// wave: LDS instructions of one wave execute in issue order, and all that is needed is that the
// compiler keeps them in order -- not a workgroup barrier that would stall the other three waves.
__device__ __forceinline__ void ia_wave_sync() { __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(IA_THREADS) void tzr_dot_interaction_fwd_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int64_t B, float* __restrict__ out, int64_t out_stride,
    int cat_dense, int cat_sparse) {
  __shared__ float tri[IA_WAVES][IA_MAXP + 16];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  const int P = n * (n - 1) / 2;
  const int64_t step = (int64_t)gridDim.x * IA_WAVES;
  // the rows of the wave's NEXT sample are fetched before the current one is contracted and stored (a wave walks
  // several samples when the grid is smaller than the batch)
  float4 n0 = tzr_zero4(), n1 = tzr_zero4();
  {
    const int64_t b = (int64_t)blockIdx.x * IA_WAVES + wv;
    if (b < B) {
      if (r < n) n0 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b, r, hd) + 4 * q);
      if (16 + r < n) n1 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b, 16 + r, hd) + 4 * q);
    }
  }
  for (int64_t b0 = (int64_t)blockIdx.x * IA_WAVES; b0 < B; b0 += step) {
    const int64_t b = b0 + wv;
    const bool on = b < B;
    const float4 a0 = n0, a1 = n1;
    n0 = n1 = tzr_zero4();
    if (b + step < B) {
      if (r < n) n0 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b + step, r, hd) + 4 * q);
      if (16 + r < n) n1 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b + step, 16 + r, hd) + 4 * q);
    }
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c11 = c00;
    const float x0[4] = {a0.x, a0.y, a0.z, a0.w};
    const float x1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], x0[e], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], x1[e], c01, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], x1[e], c11, 0, 0, 0);
    }
    // strict upper triangle, row-major (i<j): idx(i,j) = i*(2n-i-1)/2 + j-i-1, staged in LDS so
    // the global write is one contiguous run per sample
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i0 = 4 * q + reg, i1 = 16 + i0;
      const int j0 = r, j1 = 16 + r;
      if (i0 < j0 && j0 < n) tri[wv][i0 * (2 * n - i0 - 1) / 2 + j0 - i0 - 1] = c00[reg];
      if (j1 < n) tri[wv][i0 * (2 * n - i0 - 1) / 2 + j1 - i0 - 1] = c01[reg];
      if (i1 < j1 && j1 < n) tri[wv][i1 * (2 * n - i1 - 1) / 2 + j1 - i1 - 1] = c11[reg];
    }
    ia_wave_sync();  // tri[wv] is private to this wave: no workgroup barrier needed
    if (on) {
      float* o = out + b * out_stride;
      for (int idx = lane; idx < P; idx += TZR_WAVE) o[idx] = tri[wv][idx];
      int col = P;
      if (cat_dense && hd) {
        if (r == 0) tzr_st4_a4(o + col + 4 * q, a0);
        col += IA_D;
      }
      if (cat_sparse) {
        if (r >= hd && r < n) tzr_st4_a4(o + col + (r - hd) * IA_D + 4 * q, a0);
        if (16 + r < n) tzr_st4_a4(o + col + (16 + r - hd) * IA_D + 4 * q, a1);
      }
    }
    ia_wave_sync();
  }
}

// Backward: dX = (G + G^T) X.  Computed TRANSPOSED, dX^T = X^T S with S = G + G^T symmetric, so the
// MFMA accumulator of lane (r = l&15, q = l>>4) holds dX[row r][cols 4q..4q+3]: the same float4-per-
// lane image the forward loads, i.e. 16-byte loads AND stores on both sides (the first version used
// 4-byte accesses everywhere and ran at 2.2 TB/s against the forward's 5.9 TB/s).
//   A'[c][k] = X[k][c]  (lane: c = l&15, k = 4*ks + (l>>4))  read from an LDS image of X
//   B'[k][i] = S[k][i]  (lane: k = 4*ks + (l>>4), i = l&15)  read from the LDS S matrix
__global__ __launch_bounds__(IA_THREADS) void tzr_dot_interaction_bwd_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int64_t B, const float* __restrict__ gout,
    int64_t gout_stride, int cat_dense, int cat_sparse, float* __restrict__ gdense,
    int64_t gdense_stride, float* __restrict__ gsparse, int64_t gsparse_stride) {
  // per wave: S 32 x 33 floats, X 32 x 17 floats (odd strides: conflict-free column reads)
  __shared__ float S[IA_WAVES][IA_MAXN * (IA_MAXN + 1)];
  __shared__ float Xs[IA_WAVES][IA_MAXN * (IA_D + 1)];
  __shared__ unsigned short ij[IA_MAXP];  // idx -> (i << 8) | j
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  const int P = n * (n - 1) / 2;
  for (int idx = threadIdx.x; idx < P; idx += IA_THREADS) {
    int i = 0, rem = idx;
    while (rem >= n - 1 - i) {
      rem -= n - 1 - i;
      ++i;
    }
    ij[idx] = (unsigned short)((i << 8) | (i + 1 + rem));
  }
  for (int k = threadIdx.x; k < IA_WAVES * IA_MAXN * (IA_MAXN + 1); k += IA_THREADS)
    (&S[0][0])[k] = 0.f;
  for (int k = threadIdx.x; k < IA_WAVES * IA_MAXN * (IA_D + 1); k += IA_THREADS)
    (&Xs[0][0])[k] = 0.f;
  __syncthreads();
  const int pd = P;                                   // pass-through dense columns of gout
  const int ps = P + ((cat_dense && hd) ? IA_D : 0);  // pass-through sparse columns
  for (int64_t b0 = (int64_t)blockIdx.x * IA_WAVES; b0 < B; b0 += (int64_t)gridDim.x * IA_WAVES) {
    const int64_t b = b0 + wv;
    const bool on = b < B;
    const bool row0 = on && r < n, row1 = on && 16 + r < n;
    float4 p0 = tzr_zero4(), p1 = tzr_zero4();  // pass-through gradients of rows r / 16+r
    if (on) {
      const float* g = gout + b * gout_stride;
      for (int idx = lane; idx < P; idx += TZR_WAVE) {
        const float v = g[idx];
        const int i = ij[idx] >> 8, j = ij[idx] & 255;
        S[wv][i * (IA_MAXN + 1) + j] = v;
        S[wv][j * (IA_MAXN + 1) + i] = v;
      }
      float4 a0 = tzr_zero4(), a1 = tzr_zero4();
      if (row0) a0 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b, r, hd) + 4 * q);
      if (row1) a1 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b, 16 + r, hd) + 4 * q);
      float* xr0 = &Xs[wv][r * (IA_D + 1) + 4 * q];
      float* xr1 = &Xs[wv][(16 + r) * (IA_D + 1) + 4 * q];
      xr0[0] = a0.x; xr0[1] = a0.y; xr0[2] = a0.z; xr0[3] = a0.w;
      xr1[0] = a1.x; xr1[1] = a1.y; xr1[2] = a1.z; xr1[3] = a1.w;
      if (row0) {
        if (hd && r == 0) { if (cat_dense) p0 = tzr_ld4_a4(g + pd + 4 * q); }
        else if (cat_sparse) p0 = tzr_ld4_a4(g + ps + (r - hd) * IA_D + 4 * q);
      }
      if (row1 && cat_sparse) p1 = tzr_ld4_a4(g + ps + (16 + r - hd) * IA_D + 4 * q);
    }
    ia_wave_sync();  // S[wv] / Xs[wv] are private to this wave
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
    for (int ks = 0; ks < IA_MAXN / 4; ++ks) {
      const int k = 4 * ks + q;                       // contraction index = row of X / S
      const float xa = Xs[wv][k * (IA_D + 1) + r];     // A'[c=r][k]
      const float s0 = S[wv][k * (IA_MAXN + 1) + r];   // B'[k][i=r]
      const float s1 = S[wv][k * (IA_MAXN + 1) + 16 + r];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, s0, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, s1, d1, 0, 0, 0);
    }
    // accumulator reg of lane (r, q) = D'[row c = 4q+reg][col i = r] = dX[i = r][c = 4q+reg]
    if (row0) {
      const float4 v = make_float4(d0[0] + p0.x, d0[1] + p0.y, d0[2] + p0.z, d0[3] + p0.w);
      if (hd && r == 0) tzr_st4(gdense + b * gdense_stride + 4 * q, v);
      else tzr_st4(gsparse + b * gsparse_stride + (int64_t)(r - hd) * IA_D + 4 * q, v);
    }
    if (row1) {
      const float4 v = make_float4(d1[0] + p1.x, d1[1] + p1.y, d1[2] + p1.z, d1[3] + p1.w);
      tzr_st4(gsparse + b * gsparse_stride + (int64_t)(16 + r - hd) * IA_D + 4 * q, v);
    }
    ia_wave_sync();
  }
}

// The same backward with the NEXT sample's global loads in flight while the current one is contracted: a wave walks
// many samples (grid sized to the chip, not to the batch), so the prologue above is paid once per ~10 samples and the
// dependent chain  load -> LDS -> MFMA -> store  of one sample overlaps the loads of the next.
struct IaBwdRegs {
  float gv[(IA_MAXP + TZR_WAVE - 1) / TZR_WAVE];  // pair gradients idx = lane + 64 k
  float4 a0, a1, p0, p1;                          // X rows r / 16+r, pass-through gradients of those rows
};

__device__ __forceinline__ void ia_bwd_fetch(IaBwdRegs& R, const float* __restrict__ dense, int64_t dense_stride,
                                             const float* __restrict__ sparse, int64_t sparse_stride, int n, int hd,
                                             int64_t b, bool on, const float* __restrict__ gout, int64_t gout_stride,
                                             int cat_dense, int cat_sparse, int P, int pd, int ps, int lane) {
  const int r = lane & 15, q = lane >> 4;
  R.a0 = R.a1 = R.p0 = R.p1 = tzr_zero4();
#pragma unroll
  for (int k = 0; k < (IA_MAXP + TZR_WAVE - 1) / TZR_WAVE; ++k) R.gv[k] = 0.f;
  if (!on) return;
  const float* g = gout + b * gout_stride;
#pragma unroll
  for (int k = 0; k < (IA_MAXP + TZR_WAVE - 1) / TZR_WAVE; ++k) {
    const int idx = lane + TZR_WAVE * k;
    if (idx < P) R.gv[k] = g[idx];
  }
  const bool row0 = r < n, row1 = 16 + r < n;
  if (row0) R.a0 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b, r, hd) + 4 * q);
  if (row1) R.a1 = tzr_ld4(ia_row(dense, dense_stride, sparse, sparse_stride, b, 16 + r, hd) + 4 * q);
  if (row0) {
    if (hd && r == 0) { if (cat_dense) R.p0 = tzr_ld4_a4(g + pd + 4 * q); }
    else if (cat_sparse) R.p0 = tzr_ld4_a4(g + ps + (r - hd) * IA_D + 4 * q);
  }
  if (row1 && cat_sparse) R.p1 = tzr_ld4_a4(g + ps + (16 + r - hd) * IA_D + 4 * q);
}

__global__ __launch_bounds__(IA_THREADS) void tzr_dot_interaction_bwd_pipe_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int64_t B, const float* __restrict__ gout,
    int64_t gout_stride, int cat_dense, int cat_sparse, float* __restrict__ gdense,
    int64_t gdense_stride, float* __restrict__ gsparse, int64_t gsparse_stride) {
  __shared__ float S[IA_WAVES][IA_MAXN * (IA_MAXN + 1)];
  __shared__ float Xs[IA_WAVES][IA_MAXN * (IA_D + 1)];
  __shared__ unsigned short ij[IA_MAXP];  // idx -> (i << 8) | j
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  const int P = n * (n - 1) / 2;
  for (int idx = threadIdx.x; idx < P; idx += IA_THREADS) {
    int i = 0, rem = idx;
    while (rem >= n - 1 - i) {
      rem -= n - 1 - i;
      ++i;
    }
    ij[idx] = (unsigned short)((i << 8) | (i + 1 + rem));
  }
  for (int k = threadIdx.x; k < IA_WAVES * IA_MAXN * (IA_MAXN + 1); k += IA_THREADS)
    (&S[0][0])[k] = 0.f;
  for (int k = threadIdx.x; k < IA_WAVES * IA_MAXN * (IA_D + 1); k += IA_THREADS)
    (&Xs[0][0])[k] = 0.f;
  __syncthreads();
  const int pd = P;
  const int ps = P + ((cat_dense && hd) ? IA_D : 0);
  const int64_t step = (int64_t)gridDim.x * IA_WAVES;
  int64_t b = (int64_t)blockIdx.x * IA_WAVES + wv;
  IaBwdRegs cur;
  ia_bwd_fetch(cur, dense, dense_stride, sparse, sparse_stride, n, hd, b, b < B, gout, gout_stride, cat_dense,
               cat_sparse, P, pd, ps, lane);
  // (all waves of the workgroup make the same number of trips: the loop runs on the workgroup's first sample)
  for (int64_t b0 = (int64_t)blockIdx.x * IA_WAVES; b0 < B; b0 += step, b += step) {
    const bool on = b < B;
    const bool row0 = on && r < n, row1 = on && 16 + r < n;
    if (on) {
#pragma unroll
      for (int k = 0; k < (IA_MAXP + TZR_WAVE - 1) / TZR_WAVE; ++k) {
        const int idx = lane + TZR_WAVE * k;
        if (idx < P) {
          const int i = ij[idx] >> 8, j = ij[idx] & 255;
          S[wv][i * (IA_MAXN + 1) + j] = cur.gv[k];
          S[wv][j * (IA_MAXN + 1) + i] = cur.gv[k];
        }
      }
      float* xr0 = &Xs[wv][r * (IA_D + 1) + 4 * q];
      float* xr1 = &Xs[wv][(16 + r) * (IA_D + 1) + 4 * q];
      xr0[0] = cur.a0.x; xr0[1] = cur.a0.y; xr0[2] = cur.a0.z; xr0[3] = cur.a0.w;
      xr1[0] = cur.a1.x; xr1[1] = cur.a1.y; xr1[2] = cur.a1.z; xr1[3] = cur.a1.w;
    }
    const float4 p0 = cur.p0, p1 = cur.p1;
    IaBwdRegs nxt;  // the next sample of this wave: its loads fly over the contraction below
    ia_bwd_fetch(nxt, dense, dense_stride, sparse, sparse_stride, n, hd, b + step, b + step < B, gout, gout_stride,
                 cat_dense, cat_sparse, P, pd, ps, lane);
    ia_wave_sync();
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
    for (int ks = 0; ks < IA_MAXN / 4; ++ks) {
      const int k = 4 * ks + q;
      const float xa = Xs[wv][k * (IA_D + 1) + r];
      const float s0 = S[wv][k * (IA_MAXN + 1) + r];
      const float s1 = S[wv][k * (IA_MAXN + 1) + 16 + r];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, s0, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, s1, d1, 0, 0, 0);
    }
    if (row0) {
      const float4 v = make_float4(d0[0] + p0.x, d0[1] + p0.y, d0[2] + p0.z, d0[3] + p0.w);
      if (hd && r == 0) tzr_st4(gdense + b * gdense_stride + 4 * q, v);
      else tzr_st4(gsparse + b * gsparse_stride + (int64_t)(r - hd) * IA_D + 4 * q, v);
    }
    if (row1) {
      const float4 v = make_float4(d1[0] + p1.x, d1[1] + p1.y, d1[2] + p1.z, d1[3] + p1.w);
      tzr_st4(gsparse + b * gsparse_stride + (int64_t)(16 + r - hd) * IA_D + 4 * q, v);
    }
    ia_wave_sync();
    cur = nxt;
  }
}

// ---- general shapes --------------------------------------------------------------------------
// The MFMA kernels above are specialised for the DLRM-Criteo shape (D = 16, n <= 32).  Any other
// (n, D) the reference's InteractionArch accepts (D % 4 == 0) takes this path: one workgroup per
// sample, X staged once in LDS with an odd row pitch, one output pair per thread per round.  Still
// HBM-bound work (n*D floats in, n(n-1)/2 (+ pass-through) out per sample); the contraction is
// n^2 D / 2 FMAs per sample on the VALU, fed from LDS: lanes of consecutive pairs share row i
// (broadcast) and read consecutive rows j (pitch D + 1: conflict-free).
#define IAG_CAP 15360  // floats of LDS per workgroup (60 KB)

// pair index -> (i, j), i < j, row-major over the strict upper triangle
__device__ __forceinline__ void iag_pair(int idx, int n, int* pi, int* pj) {
  const float t = (float)(2 * n - 1);
  int i = (int)((t - sqrtf(fmaxf(t * t - 8.f * (float)idx, 0.f))) * 0.5f);
  i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
  while (i + 1 <= n - 2 && (i + 1) * (2 * n - i - 2) / 2 <= idx) ++i;
  while (i > 0 && i * (2 * n - i - 1) / 2 > idx) --i;
  *pi = i;
  *pj = idx - i * (2 * n - i - 1) / 2 + i + 1;
}

__device__ __forceinline__ void iag_stage_x(float* Xs, const float* dense, int64_t dense_stride,
                                            const float* sparse, int64_t sparse_stride, int64_t b,
                                            int n, int hd, int D) {
  const int lg = D >> 2;
  for (int e = threadIdx.x; e < n * lg; e += IA_THREADS) {
    const int row = e / lg, c = e - row * lg;
    const float* src = (hd && row == 0) ? dense + b * dense_stride
                                        : sparse + b * sparse_stride + (int64_t)(row - hd) * D;
    const float4 v = tzr_ld4(src + 4 * c);
    float* d = Xs + row * (D + 1) + 4 * c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

__global__ __launch_bounds__(IA_THREADS) void tzr_dot_interaction_fwd_general_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int D, int64_t B, float* __restrict__ out,
    int64_t out_stride, int cat_dense, int cat_sparse) {
  __shared__ float Xs[IAG_CAP];
  const int P = n * (n - 1) / 2;
  const int pitch = D + 1;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    iag_stage_x(Xs, dense, dense_stride, sparse, sparse_stride, b, n, hd, D);
    __syncthreads();
    float* o = out + b * out_stride;
    for (int idx = threadIdx.x; idx < P; idx += IA_THREADS) {
      int i, j;
      iag_pair(idx, n, &i, &j);
      const float* xi = Xs + i * pitch;
      const float* xj = Xs + j * pitch;
      float acc = 0.f;
      for (int k = 0; k < D; ++k) acc = fmaf(xi[k], xj[k], acc);
      o[idx] = acc;
    }
    int col = P;
    if (cat_dense && hd) {
      for (int k = threadIdx.x; k < D; k += IA_THREADS) o[col + k] = Xs[k];
      col += D;
    }
    if (cat_sparse) {
      const int F = n - hd;
      for (int e = threadIdx.x; e < F * D; e += IA_THREADS) {
        const int row = e / D, k = e - row * D;
        o[col + e] = Xs[(row + hd) * pitch + k];
      }
    }
    __syncthreads();
  }
}

// dX[i][c] = sum_j S[i][j] X[j][c] (+ pass-through), S = G + G^T with a zero diagonal.
__global__ __launch_bounds__(IA_THREADS) void tzr_dot_interaction_bwd_general_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int D, int64_t B, const float* __restrict__ gout,
    int64_t gout_stride, int cat_dense, int cat_sparse, float* __restrict__ gdense,
    int64_t gdense_stride, float* __restrict__ gsparse, int64_t gsparse_stride) {
  __shared__ float lds[IAG_CAP];
  float* Xs = lds;                    // n x (D + 1)
  float* S = lds + n * (D + 1);       // n x (n + 1)
  const int P = n * (n - 1) / 2;
  const int pitch = D + 1, sp = n + 1;
  const int pd = P;
  const int ps = P + ((cat_dense && hd) ? D : 0);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const float* g = gout + b * gout_stride;
    iag_stage_x(Xs, dense, dense_stride, sparse, sparse_stride, b, n, hd, D);
    for (int i = threadIdx.x; i < n; i += IA_THREADS) S[i * sp + i] = 0.f;
    for (int idx = threadIdx.x; idx < P; idx += IA_THREADS) {
      int i, j;
      iag_pair(idx, n, &i, &j);
      const float v = g[idx];
      S[i * sp + j] = v;
      S[j * sp + i] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n * D; e += IA_THREADS) {
      const int i = e / D, c = e - i * D;
      const float* si = S + i * sp;
      float acc = 0.f;
      for (int j = 0; j < n; ++j) acc = fmaf(si[j], Xs[j * pitch + c], acc);
      if (hd && i == 0) {
        if (cat_dense) acc += g[pd + c];
        gdense[b * gdense_stride + c] = acc;
      } else {
        if (cat_sparse) acc += g[ps + (i - hd) * D + c];
        gsparse[b * gsparse_stride + (int64_t)(i - hd) * D + c] = acc;
      }
    }
    __syncthreads();
  }
}

// ---- MFMA kernels for any D % 4 == 0 and n <= 64 ------------------------------------------------
// Same wave-per-sample scheme as the D = 16 kernels above, generalised: NB row blocks of 16
// (NB = 2: n <= 32, NB = 4: n <= 64) and a loop over 16-column blocks of D (the contraction index).
// A lane (r = l & 15, q = l >> 4) loads the float4 X[16 bi + r][cb + 4q .. cb + 4q + 3] of every
// row block; MFMA step e consumes element e of it, so the four steps of a column block cover its 16
// contraction indices (lanes past D supply zeros).  The D = 16 kernels stay as they are: they are
// on the measured DLRM-Criteo step.
__device__ __forceinline__ const float* iam_row(const float* dense, int64_t dense_stride,
                                                const float* sparse, int64_t sparse_stride,
                                                int64_t b, int i, int hd, int D) {
  return (hd && i == 0) ? dense + b * dense_stride
                        : sparse + b * sparse_stride + (int64_t)(i - hd) * D;
}

template <int G, int NB>
__device__ __forceinline__ void iam_load_group(float4 (&a)[G][NB], const float* dense,
                                               int64_t dense_stride, const float* sparse,
                                               int64_t sparse_stride, int64_t b, bool on, int n,
                                               int hd, int D, int cg, int r, int q) {
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      const int row = 16 * bi + r, c = cg + 16 * gi + 4 * q;
      a[gi][bi] = (on && c < D && row < n)
                      ? tzr_ld4(iam_row(dense, dense_stride, sparse, sparse_stride, b, row, hd, D) + c)
                      : tzr_zero4();
    }
}

template <int NB>
__global__ __launch_bounds__(IA_THREADS) void tzr_dot_interaction_fwd_mfma_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int D, int64_t B, float* __restrict__ out,
    int64_t out_stride, int cat_dense, int cat_sparse) {
  constexpr int MAXN = 16 * NB;
  constexpr int MAXP = MAXN * (MAXN - 1) / 2;
  constexpr int NPAIR = NB * (NB + 1) / 2;
  __shared__ float tri[IA_WAVES][MAXP + 16];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  const int P = n * (n - 1) / 2;
  const int colD = P;
  const int colS = P + ((cat_dense && hd) ? D : 0);
  for (int64_t b0 = (int64_t)blockIdx.x * IA_WAVES; b0 < B; b0 += (int64_t)gridDim.x * IA_WAVES) {
    const int64_t b = b0 + wv;
    const bool on = b < B;
    float* o = out + b * out_stride;
    f32x4 acc[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column blocks go in groups of G: all loads of a group are issued before its pass-through stores
    // and MFMAs.  NB = 2: G = 4 (64 columns, 8 float4 per lane in flight, 4 waves per SIMD);
    // NB = 4: G = 1 with the next block prefetched (the 10 accumulators leave fewer registers).
    constexpr int G = NB == 2 ? 4 : 1;
    constexpr bool PF = NB != 2;
    float4 nxt[G][NB];
    if (PF) iam_load_group<G, NB>(nxt, dense, dense_stride, sparse, sparse_stride, b, on, n, hd, D, 0, r, q);
    for (int cg = 0; cg < D; cg += 16 * G) {
      float4 a[G][NB];
      if (PF) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
#pragma unroll
          for (int bi = 0; bi < NB; ++bi) a[gi][bi] = nxt[gi][bi];
        if (cg + 16 * G < D)
          iam_load_group<G, NB>(nxt, dense, dense_stride, sparse, sparse_stride, b, on, n, hd, D, cg + 16 * G, r, q);
      } else {
        iam_load_group<G, NB>(a, dense, dense_stride, sparse, sparse_stride, b, on, n, hd, D, cg, r, q);
      }
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const int c = cg + 16 * gi + 4 * q;
        if (cg + 16 * gi < D) {  // wave-uniform
          float x[NB][4];
#pragma unroll
          for (int bi = 0; bi < NB; ++bi) {
            const int row = 16 * bi + r;
            const float4 v = a[gi][bi];
            if (on && c < D && row < n) {
              if (hd && row == 0) {
                if (cat_dense) tzr_st4_a4(o + colD + c, v);
              } else if (cat_sparse) {
                tzr_st4_a4(o + colS + (int64_t)(row - hd) * D + c, v);
              }
            }
            x[bi][0] = v.x; x[bi][1] = v.y; x[bi][2] = v.z; x[bi][3] = v.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int p = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
              for (int bj = bi; bj < NB; ++bj, ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[bi][e], x[bj][e], acc[p], 0, 0, 0);
          }
        }
      }
    }
    {
      int p = 0;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = bi; bj < NB; ++bj, ++p)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int i = 16 * bi + 4 * q + reg, j = 16 * bj + r;
            if (i < j && j < n) tri[wv][i * (2 * n - i - 1) / 2 + j - i - 1] = acc[p][reg];
          }
    }
    ia_wave_sync();
    if (on)
      for (int idx = lane; idx < P; idx += TZR_WAVE) o[idx] = tri[wv][idx];
    ia_wave_sync();
  }
}

// dX^T = X^T S per 16-column block of D: A'[c][k] = X[k][cb + c] straight from global (each element
// of X is read exactly once per sample; a lane quad reads four 64-byte row pieces), B'[k][i] = S[k][i]
// from the wave's LDS image of S = G + G^T.  The accumulator of lane (r, q) is dX[16 bi + r][cb + 4q ..
// cb + 4q + 3]: float4 stores.
// SCAP = floats of S per wave; the pitch is n | 1 (odd: conflict-free column reads), so up to 48 rows
// fit a 4-wave workgroup.  NB = 1 (up to 16 rows), 2 (32), 4 (48 in a 4-wave, 64 in a 2-wave workgroup).
template <int NB, int WAVES, int SCAP, bool XS>
__global__ __launch_bounds__(WAVES * TZR_WAVE, (NB <= 2 ? 4 : (WAVES == 4 ? 3 : 2))) void tzr_dot_interaction_bwd_mfma_kernel(
    const float* __restrict__ dense, int64_t dense_stride, const float* __restrict__ sparse,
    int64_t sparse_stride, int n, int hd, int D, int64_t B, const float* __restrict__ gout,
    int64_t gout_stride, int cat_dense, int cat_sparse, float* __restrict__ gdense,
    int64_t gdense_stride, float* __restrict__ gsparse, int64_t gsparse_stride) {
  constexpr int MAXN = 16 * NB;
  constexpr int MAXP = MAXN * (MAXN - 1) / 2;
  constexpr int THREADS = WAVES * TZR_WAVE;
  // full-size S image: compile-time pitch, rows / columns past n stay zero; reduced image (33-48
  // rows in a 4-wave workgroup): run-time pitch, reads past n are masked
  constexpr bool RT = SCAP != MAXN * (MAXN + 1);
  __shared__ float S[WAVES][SCAP];
  const int SP = RT ? (n | 1) : MAXN + 1;
  __shared__ unsigned short ij[MAXP];  // idx -> (i << 8) | j
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / TZR_WAVE);  // scalar: sample bases stay in SGPRs
  const int r = lane & 15, q = lane >> 4;
  const int P = n * (n - 1) / 2;
  for (int idx = threadIdx.x; idx < P; idx += THREADS) {
    int i, j;
    iag_pair(idx, n, &i, &j);
    ij[idx] = (unsigned short)((i << 8) | j);
  }
  for (int k = threadIdx.x; k < WAVES * SCAP; k += THREADS) (&S[0][0])[k] = 0.f;
  __syncthreads();
  const int pd = P;
  const int ps = P + ((cat_dense && hd) ? D : 0);
  const int ksteps = (n + 3) >> 2;
  // One wave walks its samples alone (2-3 waves per SIMD at 33-64 rows), so every global round trip it
  // waits for is exposed; the order of the memory operations is the design (profiles/r03bt, n = 64 / D = 32:
  // 1 131 -> 523 us):
  //  - the pair gradients arrive GB x 64 at a time: one round trip per batch, not one per value;
  //  - every operand load of a column block is issued before its MFMA chain starts (inside the chain
  //    each one would be a dependent round trip): the first block's ahead of the S image, the next
  //    block's between the chain and the stores of the current one, into the registers the chain has released;
  //  - the pass-through gradients of a block are loaded ahead of its chain, not between chain and store;
  //  - XS: the next sample's first batch and first operands are issued ahead of this sample's last
  //    stores.  It costs ~90 VGPRs, so it pays only where LDS, not the register file, sets the occupancy.
  constexpr int GB = NB == 1 ? 2 : (WAVES == 4 ? 8 : 16);
  float xa[MAXN / 4];
  float gv[GB];
  auto load_xa = [&](int64_t bb, int cb) {
    const bool cin = bb < B && (cb + r < D);  // operand column of this lane
    const float* sp = sparse + bb * sparse_stride;  // wave-uniform bases, 32-bit lane offsets
    const float* dp = hd ? dense + bb * dense_stride : sp;
    const int c = cb + r;
#pragma unroll
    for (int ks = 0; ks < MAXN / 4; ++ks) {
      const int k = 4 * ks + q;  // contraction index = row of X / S
      const float* p = sp + (unsigned)((k - (ks ? hd : (k ? hd : 0))) * D + c);
      if (ks == 0 && hd && k == 0) p = dp + (unsigned)c;
      xa[ks] = (cin && k < n) ? *p : 0.f;
    }
  };
  auto load_gv = [&](int64_t bb, int base) {
    if (bb < B) {
      const float* gb = gout + bb * gout_stride;
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int idx = base + u * TZR_WAVE + lane;
        gv[u] = gb[idx < P ? idx : P - 1];
      }
    }
  };
  const int64_t bstep = (int64_t)gridDim.x * WAVES;
  if (XS) {
    load_xa((int64_t)blockIdx.x * WAVES + wv, 0);
    load_gv((int64_t)blockIdx.x * WAVES + wv, 0);
  }
  for (int64_t b0 = (int64_t)blockIdx.x * WAVES; b0 < B; b0 += bstep) {
    const int64_t b = b0 + wv;
    const bool on = b < B;
    const float* g = gout + b * gout_stride;
    float* gsp = gsparse + b * gsparse_stride;
    float* gdp = hd ? gdense + b * gdense_stride : gsp;
    if (!XS) load_xa(b, 0);
    if (on) {
      for (int base = 0; base < P; base += GB * TZR_WAVE) {
        if (!XS || base) load_gv(b, base);
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int idx = base + u * TZR_WAVE + lane;
          if (idx < P) {
            const int i = ij[idx] >> 8, j = ij[idx] & 255;
            S[wv][i * SP + j] = gv[u];
            S[wv][j * SP + i] = gv[u];
          }
        }
      }
    }
    ia_wave_sync();  // S[wv] is private to this wave
    for (int cb = 0; cb < D; cb += 16) {
      const bool kin = on && (cb + 4 * q < D);      // output float4 of this lane
      f32x4 d[NB];
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) d[bi] = f32x4{0.f, 0.f, 0.f, 0.f};
      float4 pt[NB];
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        const int row = 16 * bi + r;
        pt[bi] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kin && row < n) {
          if (hd && row == 0) {
            if (cat_dense) pt[bi] = tzr_ld4_a4(g + (unsigned)(pd + cb + 4 * q));
          } else if (cat_sparse) {
            pt[bi] = tzr_ld4_a4(g + (unsigned)(ps + (row - hd) * D + cb + 4 * q));
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < MAXN / 4; ++ks) {
        if (ks < ksteps) {  // wave-uniform
          const int k = 4 * ks + q;
#pragma unroll
          for (int bi = 0; bi < NB; ++bi) {
            const int col = 16 * bi + r;
            const float sv = (!RT || (k < n && col < n)) ? S[wv][k * SP + col] : 0.f;
            d[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks], sv, d[bi], 0, 0, 0);
          }
        }
      }
      // the chain has released xa: the next operands are requested ahead of the stores (S is rewritten
      // only after them)
      if (!XS && cb + 16 < D) load_xa(b, cb + 16);
      if (XS && cb + 16 >= D) {
        load_xa(b + bstep, 0);
        load_gv(b + bstep, 0);
      }
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        const int row = 16 * bi + r;
        if (kin && row < n) {
          const float4 v = tzr_add4(make_float4(d[bi][0], d[bi][1], d[bi][2], d[bi][3]), pt[bi]);
          if (hd && row == 0) {
            tzr_st4(gdp + (unsigned)(cb + 4 * q), v);
          } else {
            tzr_st4(gsp + (unsigned)((row - hd) * D + cb + 4 * q), v);
          }
        }
      }
      if (XS && cb + 16 < D) load_xa(b, cb + 16);  // (ahead of the stores it would not fit the register file)
    }
    ia_wave_sync();
  }
}

static unsigned iam_grid(int64_t B, int waves) {
  const int64_t wg = (B + waves - 1) / waves;
  return (unsigned)(wg < 1 ? 1 : (wg > 16384 ? 16384 : wg));
}

static bool iag_fits(int n, int D, bool bwd) {
  const int64_t need = (int64_t)n * (D + 1) + (bwd ? (int64_t)n * (n + 1) : 0);
  return need <= IAG_CAP && n <= 2048;
}

int g_tzr_ia_gen_wgs = 0;  // tzr_tune("ia_gen_wgs"): workgroups of the 2-64-row MFMA backward for D != 16 (0 = the resident set)

// persistent grid: exactly the workgroups that are resident at once (256 CUs x per_cu) walk the batch --
// the pair-index table and the zeroed S image are set up once per workgroup, and no second, partial round
// of workgroups trails the first (profiles/r03bt: n = 40: 402 -> 285 us from the grid alone)
static unsigned iam_bwd_grid(int64_t B, int waves, int per_cu) {
  unsigned g = iam_grid(B, waves);
  const unsigned cap = g_tzr_ia_gen_wgs > 0 ? (unsigned)g_tzr_ia_gen_wgs : 256u * (unsigned)per_cu;
  return g < cap ? g : cap;
}

static unsigned iag_grid(int64_t B) { return (unsigned)(B < 1 ? 1 : (B > 16384 ? 16384 : B)); }

int g_tzr_ia_fwd_wgs = 0;    // tzr_tune("ia_fwd_wgs"): workgroups of the D = 16 forward (0 = one per 4 samples, <= 8192)
int g_tzr_ia_bwd_plain = 0;  // tzr_tune("ia_bwd_plain"): 1 = the backward without the software pipeline (A/B; D = 16, n <= 32)
int g_tzr_ia_bwd_wgs = 0;    // tzr_tune("ia_bwd_wgs"): workgroups of that backward (0 = one per 4 samples, at most 3 072
                             // pipelined / 8 192 plain -- profiles/r02x: 114.7 -> 91.5 us per forward + backward pair)

static unsigned ia_grid(int64_t B) {
  const int64_t wg = (B + IA_WAVES - 1) / IA_WAVES;
  return (unsigned)(wg < 1 ? 1 : (wg > 8192 ? 8192 : wg));
}

extern "C" int tzr_dot_interaction_fwd(const float* d_dense, int64_t dense_stride,
                                       const float* d_sparse, int64_t sparse_stride, int F, int D,
                                       int64_t B, float* d_out, int64_t out_stride, int cat_dense,
                                       int cat_sparse, void* stream) {
  const int hd = d_dense ? 1 : 0;
  const int n = F + hd;
  if (!d_sparse || !d_out || F <= 0 || B < 0) return TZR_ERR_INVALID;
  const bool mfma = (D == IA_D && n <= IA_MAXN);
  if (n < 2 || D <= 0 || (D & 3) || (!mfma && n > 64 && !iag_fits(n, D, false))) return TZR_ERR_UNSUPPORTED;
  if ((sparse_stride & 3) || (hd && (dense_stride & 3)) ||
      (reinterpret_cast<uintptr_t>(d_sparse) & 15) || (reinterpret_cast<uintptr_t>(d_dense) & 15))
    return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  if (!mfma && n <= 64) {
    if (n <= 32)
      hipLaunchKernelGGL((tzr_dot_interaction_fwd_mfma_kernel<2>), dim3(iam_grid(B, IA_WAVES)),
                         dim3(IA_THREADS), 0, static_cast<hipStream_t>(stream), d_dense,
                         dense_stride, d_sparse, sparse_stride, n, hd, D, B, d_out, out_stride,
                         cat_dense, cat_sparse);
    else
      hipLaunchKernelGGL((tzr_dot_interaction_fwd_mfma_kernel<4>), dim3(iam_grid(B, IA_WAVES)),
                         dim3(IA_THREADS), 0, static_cast<hipStream_t>(stream), d_dense,
                         dense_stride, d_sparse, sparse_stride, n, hd, D, B, d_out, out_stride,
                         cat_dense, cat_sparse);
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  if (!mfma) {
    hipLaunchKernelGGL(tzr_dot_interaction_fwd_general_kernel, dim3(iag_grid(B)), dim3(IA_THREADS),
                       0, static_cast<hipStream_t>(stream), d_dense, dense_stride, d_sparse,
                       sparse_stride, n, hd, D, B, d_out, out_stride, cat_dense, cat_sparse);
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  unsigned fgrid = ia_grid(B);
  if (g_tzr_ia_fwd_wgs > 0 && (unsigned)g_tzr_ia_fwd_wgs < fgrid) fgrid = (unsigned)g_tzr_ia_fwd_wgs;
  hipLaunchKernelGGL(tzr_dot_interaction_fwd_kernel, dim3(fgrid), dim3(IA_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_dense, dense_stride, d_sparse,
                     sparse_stride, n, hd, B, d_out, out_stride, cat_dense, cat_sparse);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_dot_interaction_bwd(const float* d_dense, int64_t dense_stride,
                                       const float* d_sparse, int64_t sparse_stride, int F, int D,
                                       int64_t B, const float* d_grad_out, int64_t grad_out_stride,
                                       int cat_dense, int cat_sparse, float* d_grad_dense,
                                       int64_t grad_dense_stride, float* d_grad_sparse,
                                       int64_t grad_sparse_stride, void* stream) {
  const int hd = d_dense ? 1 : 0;
  const int n = F + hd;
  if (!d_sparse || !d_grad_out || !d_grad_sparse || F <= 0 || B < 0) return TZR_ERR_INVALID;
  if (hd && !d_grad_dense) return TZR_ERR_INVALID;
  const bool mfma = (D == IA_D && n <= IA_MAXN);
  if (n < 2 || D <= 0 || (D & 3) || (!mfma && n > 64 && !iag_fits(n, D, true))) return TZR_ERR_UNSUPPORTED;
  if ((sparse_stride & 3) || (grad_sparse_stride & 3) || (hd && ((dense_stride | grad_dense_stride) & 3)) ||
      ((reinterpret_cast<uintptr_t>(d_sparse) | reinterpret_cast<uintptr_t>(d_grad_sparse) |
        reinterpret_cast<uintptr_t>(d_dense) | reinterpret_cast<uintptr_t>(d_grad_dense)) & 15))
    return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  if (!mfma && n <= 64) {
#define IAB_LAUNCH(NB_, W_, SCAP_, XS_, PER_CU_)                                                               \
  hipLaunchKernelGGL((tzr_dot_interaction_bwd_mfma_kernel<NB_, W_, SCAP_, XS_>),                                 \
                     dim3(iam_bwd_grid(B, W_, PER_CU_)), dim3(W_ * TZR_WAVE), 0, static_cast<hipStream_t>(stream), \
                     d_dense, dense_stride, d_sparse, sparse_stride, n, hd, D, B, d_grad_out, grad_out_stride,  \
                     cat_dense, cat_sparse, d_grad_dense, grad_dense_stride, d_grad_sparse, grad_sparse_stride)
    // row blocks, cross-sample prefetch and workgroups per CU as measured (profiles/r03bt): the prefetch
    // pays only where the LDS image, not the register file, limits the occupancy (49-64 rows)
    if (n <= 16) {
      IAB_LAUNCH(1, 4, 16 * 17, false, 4);
    } else if (n <= 32) {
      IAB_LAUNCH(2, 4, 32 * 33, false, 4);
    } else if (n <= 48) {
      IAB_LAUNCH(4, 4, 48 * 49, false, 3);
    } else {
      IAB_LAUNCH(4, 2, 64 * 65, true, 4);
    }
#undef IAB_LAUNCH
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  if (!mfma) {
    hipLaunchKernelGGL(tzr_dot_interaction_bwd_general_kernel, dim3(iag_grid(B)), dim3(IA_THREADS),
                       0, static_cast<hipStream_t>(stream), d_dense, dense_stride, d_sparse,
                       sparse_stride, n, hd, D, B, d_grad_out, grad_out_stride, cat_dense,
                       cat_sparse, d_grad_dense, grad_dense_stride, d_grad_sparse,
                       grad_sparse_stride);
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  unsigned grid = ia_grid(B);
  const unsigned cap = g_tzr_ia_bwd_wgs > 0 ? (unsigned)g_tzr_ia_bwd_wgs : (g_tzr_ia_bwd_plain ? grid : 3072u);
  if (cap < grid) grid = cap;
  if (!g_tzr_ia_bwd_plain) {
    hipLaunchKernelGGL(tzr_dot_interaction_bwd_pipe_kernel, dim3(grid), dim3(IA_THREADS), 0,
                       static_cast<hipStream_t>(stream), d_dense, dense_stride, d_sparse, sparse_stride, n, hd, B,
                       d_grad_out, grad_out_stride, cat_dense, cat_sparse, d_grad_dense, grad_dense_stride,
                       d_grad_sparse, grad_sparse_stride);
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  hipLaunchKernelGGL(tzr_dot_interaction_bwd_kernel, dim3(grid), dim3(IA_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_dense, dense_stride, d_sparse,
                     sparse_stride, n, hd, B, d_grad_out, grad_out_stride, cat_dense, cat_sparse,
                     d_grad_dense, grad_dense_stride, d_grad_sparse, grad_sparse_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ---- K10: FM ---------------------------------------------------------------------------------
// thread = (sample, float4 chunk of D); F sequential float4 loads per thread, D/4 consecutive lanes
// read one contiguous embedding row.

template <bool BWD>
__global__ __launch_bounds__(IA_THREADS) void tzr_fm_kernel(
    const float* __restrict__ x, int64_t x_stride, int F, int D, int64_t B,
    const float* __restrict__ gout, int64_t gout_stride, float* __restrict__ out,
    int64_t out_stride) {
  const int lg = D >> 2;
  const int64_t total = B * lg;
  for (int64_t k = (int64_t)blockIdx.x * IA_THREADS + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * IA_THREADS) {
    const int64_t b = k / lg;
    const int c = (int)(k - b * lg);
    const float* xp = x + b * x_stride + 4 * c;
    float4 s = tzr_zero4(), ss = tzr_zero4();
    for (int f = 0; f < F; ++f) {
      const float4 v = tzr_ld4(xp + (int64_t)f * D);
      s = tzr_add4(s, v);
      ss.x = fmaf(v.x, v.x, ss.x); ss.y = fmaf(v.y, v.y, ss.y);
      ss.z = fmaf(v.z, v.z, ss.z); ss.w = fmaf(v.w, v.w, ss.w);
    }
    if (!BWD) {
      float4 o;
      o.x = 0.5f * (s.x * s.x - ss.x); o.y = 0.5f * (s.y * s.y - ss.y);
      o.z = 0.5f * (s.z * s.z - ss.z); o.w = 0.5f * (s.w * s.w - ss.w);
      tzr_st4(out + b * out_stride + 4 * c, o);
    } else {
      const float4 g = tzr_ld4(gout + b * gout_stride + 4 * c);
      float* op = out + b * out_stride + 4 * c;
      for (int f = 0; f < F; ++f) {
        const float4 v = tzr_ld4(xp + (int64_t)f * D);
        float4 o;
        o.x = g.x * (s.x - v.x); o.y = g.y * (s.y - v.y);
        o.z = g.z * (s.z - v.z); o.w = g.w * (s.w - v.w);
        tzr_st4(op + (int64_t)f * D, o);
      }
    }
  }
}

static int fm_check(const float* x, int64_t xs, int F, int D, int64_t B) {
  if (!x || F <= 0 || D <= 0 || B < 0) return TZR_ERR_INVALID;
  if ((D & 3) || (xs & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return TZR_ERR_UNSUPPORTED;
  return TZR_OK;
}

extern "C" int tzr_fm_fwd(const float* d_x, int64_t x_stride, int F, int D, int64_t B,
                          float* d_out, int64_t out_stride, void* stream) {
  int rc = fm_check(d_x, x_stride, F, D, B);
  if (rc != TZR_OK) return rc;
  if (!d_out || (out_stride & 3) || (reinterpret_cast<uintptr_t>(d_out) & 15)) return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  const int64_t total = B * (D >> 2);
  const unsigned grid = (unsigned)std::min<int64_t>(8192, (total + IA_THREADS - 1) / IA_THREADS);
  hipLaunchKernelGGL((tzr_fm_kernel<false>), dim3(grid), dim3(IA_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_x, x_stride, F, D, B,
                     (const float*)nullptr, (int64_t)0, d_out, out_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_fm_bwd(const float* d_x, int64_t x_stride, int F, int D, int64_t B,
                          const float* d_grad_out, int64_t grad_out_stride, float* d_grad_x,
                          int64_t grad_x_stride, void* stream) {
  int rc = fm_check(d_x, x_stride, F, D, B);
  if (rc != TZR_OK) return rc;
  if (!d_grad_out || !d_grad_x || (grad_out_stride & 3) || (grad_x_stride & 3) ||
      (reinterpret_cast<uintptr_t>(d_grad_out) & 15) || (reinterpret_cast<uintptr_t>(d_grad_x) & 15))
    return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  const int64_t total = B * (D >> 2);
  const unsigned grid = (unsigned)std::min<int64_t>(8192, (total + IA_THREADS - 1) / IA_THREADS);
  hipLaunchKernelGGL((tzr_fm_kernel<true>), dim3(grid), dim3(IA_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_x, x_stride, F, D, B, d_grad_out,
                     grad_out_stride, d_grad_x, grad_x_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
