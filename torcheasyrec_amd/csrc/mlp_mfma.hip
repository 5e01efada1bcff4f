// The small dense stacks of DLRM on the matrix cores (gfx950): the shapes of examples/dlrm_criteo.config
// (/root/reference/tzrec/models/dlrm.py:101-135: bottom MLP 13 -> 64 -> 16, top MLP tail 64 -> 32 -> 1) as ONE WAVE PER
// 16-SAMPLE TILE, exact-fp32 v_mfma_f32_16x16x4_f32, weights resident in registers, no workgroup barrier in the tile
// loop.  The general kernels of mlp_ops.hip (tiles of 64 samples through LDS, a block barrier between every phase, VALU
// products) take 25 + 45 + 48 us for < 1 GFLOP at B = 65 536 -- latency, not work; these take over when the shape
// matches and leave every other shape to them (same C entry points, same partial-sum rows, same finish kernel).
//
// Fragment maps (cdna_hip_programming.md section 3): lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// accumulator reg j of lane (r = l & 15, q = l >> 4) is D[row 4 q + j][col r].  A contraction index may be permuted freely
// as long as both operands agree: rows of activations are read as one ds_read_b128 per four k-steps.
#include "tzr_common.h"

#define MM_WAVES 4
#define MM_THREADS (MM_WAVES * TZR_WAVE)
#define MM_TS 16
#define MM_MAX_WG 512  // = ML_MAX_WG of mlp_ops.hip: rows of the partial-sum workspace

extern int g_tzr_mlp_mfma;  // mlp_ops.hip: > 0 also caps the workgroups (tests: many tiles per wave on small batches)

typedef float mm_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mm_sum16(float v) {  // over the 16 lanes that share l >> 4 (all of them get the sum)
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
__device__ __forceinline__ float mm_sum_q(float v) {  // over the 4 lanes that share l & 15
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ---- top MLP tail: y2 = relu(y1 W2^T + b2), logit = y2 . w3 + b3, mean BCE-with-logits, and the whole backward ----------
// H1 = 64, H2 = 32.  Per tile: 32 + 32 + 32 MFMAs (y2; dW2 += g2^T y1; g1 = g2 W2), layouts changed through two
// wave-private LDS tiles.  Partial-sum row of a workgroup (the layout tzr_mlp_finish_kernel reduces):
// [dW2 (32 x 64) | db2 (32) | dw3 (32) | db3, loss (2) | db1 (64)].
#define MT_H1 64
#define MT_H2 32
#define MT_P1 (MT_H1 + 4)  // row pitches: 16-byte aligned rows, b128 reads of 16 rows spread over the banks
#define MT_P2 (MT_H2 + 4)
#define MT_ROW (MT_H2 * MT_H1 + 2 * MT_H2 + 2 + MT_H1)

template <typename LabelT>
__global__ __launch_bounds__(MM_THREADS) TZR_WAVES_PER_EU(2) void tzr_mlp_tail64_kernel(
    const float* __restrict__ y1, int64_t y1s, const LabelT* __restrict__ labels, int64_t B, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ logits,
    float* __restrict__ g1, int64_t g1s, float* __restrict__ parts) {
  __shared__ __attribute__((aligned(16))) float T1s[MM_WAVES][MM_TS * MT_P1];  // y1 tile, later the g1 tile
  __shared__ __attribute__((aligned(16))) float T2s[MM_WAVES][MM_TS * MT_P2];  // g2 tile
  __shared__ float red[MM_WAVES][MT_ROW];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TZR_WAVE));
  const int r = lane & 15, q = lane >> 4;
  float* T1 = &T1s[wv][0];
  float* T2 = &T2s[wv][0];
  // W2 twice: B operand of y1 W2^T (k = 16 q + ks over H1: W2[16 jb + r][k]) and of g2 W2 (k = 8 q + ks over H2: W2[k][16 hb + r])
  float Wa[2][16], Wb[4][8];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = tzr_ld4(W2 + (16 * jb + r) * MT_H1 + 16 * q + 4 * i);
      Wa[jb][4 * i] = v.x; Wa[jb][4 * i + 1] = v.y; Wa[jb][4 * i + 2] = v.z; Wa[jb][4 * i + 3] = v.w;
    }
#pragma unroll
  for (int hb = 0; hb < 4; ++hb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) Wb[hb][ks] = W2[(8 * q + ks) * MT_H1 + 16 * hb + r];
  float b2v[2], w3v[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    b2v[jb] = b2 ? b2[16 * jb + r] : 0.f;
    w3v[jb] = w3[16 * jb + r];
  }
  const float bias3 = b3 ? b3[0] : 0.f;
  const float inv = 1.0f / (float)B;
  mm_f32x4 accW[2][4];  // dW2[c = 16 jb + 4 q + j][h = 16 hb + r]
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) accW[jb][hb] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
  float db2a[2] = {0.f, 0.f}, dw3a[2] = {0.f, 0.f}, db1a[4] = {0.f, 0.f, 0.f, 0.f}, db3a = 0.f, lossa = 0.f;

  const int64_t tiles = (B + MM_TS - 1) / MM_TS;
  const int64_t nw = (int64_t)gridDim.x * MM_WAVES;
  // a wave's y1 tile (4 KB contiguous when y1 is dense: four 16-byte pieces per lane), fetched a turn ahead: a wave has two
  // or more turns at the step's batch, and each was a chain that began with an HBM round trip (rows behind the batch read
  // row B - 1 and are zeroed when the tile is stored)
  auto fetch = [&](float4 (&v)[4], int64_t tt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + TZR_WAVE * i, row = e >> 4, c4 = e & 15;
      const int64_t b = tt * MM_TS + row;
      v[i] = tzr_ld4(y1 + (b < B ? b : B - 1) * y1s + 4 * c4);
    }
  };
  int64_t t = (int64_t)blockIdx.x * MM_WAVES + wv;
  float4 yt[4];
  if (t < tiles) fetch(yt, t);
  for (; t < tiles; t += nw) {
    const int64_t b0 = t * MM_TS;
    // ---- y1 tile -> LDS; the next turn's takes off
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + TZR_WAVE * i, row = e >> 4, c4 = e & 15;
      tzr_st4(T1 + row * MT_P1 + 4 * c4, b0 + row < B ? yt[i] : tzr_zero4());
    }
    fetch(yt, t + nw < tiles ? t + nw : t);
    __builtin_amdgcn_wave_barrier();
    // ---- y2 = relu(y1 W2^T + b2): A = y1[sample r][16 q + ks]
    mm_f32x4 y2a[2] = {mm_f32x4{0.f, 0.f, 0.f, 0.f}, mm_f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 av = tzr_ld4(T1 + r * MT_P1 + 16 * q + 4 * i);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) y2a[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], Wa[jb][4 * i + e], y2a[jb], 0, 0, 0);
    }
    // reg j of lane (r, q): sample 4 q + j, unit 16 jb + r
    float y2v[2][4], z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y2v[0][j] = fmaxf(y2a[0][j] + b2v[0], 0.f);
      y2v[1][j] = fmaxf(y2a[1][j] + b2v[1], 0.f);
      z[j] = mm_sum16(fmaf(y2v[0][j], w3v[0], y2v[1][j] * w3v[1])) + bias3;  // (a fixed shuffle tree: deterministic)
    }
    // ---- loss and d(loss)/d(logit) of samples 4 q .. 4 q + 3 (every lane of the group computes them)
    float dl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t b = b0 + 4 * q + j;
      float li = 0.f;
      dl[j] = 0.f;
      if (b < B) {
        const float y = (float)labels[b];
        const float e = expf(-fabsf(z[j]));
        li = fmaxf(z[j], 0.f) - z[j] * y + log1pf(e);  // the formula of tzr_bce_logits (dense_ops.hip)
        const float sig = z[j] >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        dl[j] = (sig - y) * inv;
        if (r == 0) logits[b] = z[j];
      }
      if (r == 0) {
        db3a += dl[j];
        lossa += li;
      }
    }
    // ---- g2 = dl w3 masked by y2 > 0 (same layout), its column sums, dw3; g2 tile -> LDS
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = y2v[jb][j] > 0.f ? dl[j] * w3v[jb] : 0.f;
        db2a[jb] += g;
        dw3a[jb] = fmaf(dl[j], y2v[jb][j], dw3a[jb]);
        T2[(4 * q + j) * MT_P2 + 16 * jb + r] = g;
      }
    __builtin_amdgcn_wave_barrier();
    // ---- dW2 += g2^T y1: A = g2[sample 4 ks + q][unit 16 jb + r], B = y1[sample 4 ks + q][16 hb + r]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float ga[2], yb[4];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) ga[jb] = T2[(4 * ks + q) * MT_P2 + 16 * jb + r];
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) yb[hb] = T1[(4 * ks + q) * MT_P1 + 16 * hb + r];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) accW[jb][hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[jb], yb[hb], accW[jb][hb], 0, 0, 0);
    }
    // ---- g1 = (g2 W2) masked by y1 > 0: A = g2[sample r][8 q + ks]
    mm_f32x4 g1a[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) g1a[hb] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 av = tzr_ld4(T2 + r * MT_P2 + 8 * q + 4 * i);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) g1a[hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], Wb[hb][4 * i + e], g1a[hb], 0, 0, 0);
    }
    float g1v[4][4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h = T1[(4 * q + j) * MT_P1 + 16 * hb + r];
        g1v[hb][j] = h > 0.f ? g1a[hb][j] : 0.f;
        db1a[hb] += g1v[hb][j];
      }
    __builtin_amdgcn_wave_barrier();  // every read of the y1 tile is done: g1 replaces it
#pragma unroll
    for (int hb = 0; hb < 4; ++hb)
#pragma unroll
      for (int j = 0; j < 4; ++j) T1[(4 * q + j) * MT_P1 + 16 * hb + r] = g1v[hb][j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + TZR_WAVE * i, row = e >> 4, c4 = e & 15;
      if (b0 + row < B) tzr_st4(g1 + (b0 + row) * g1s + 4 * c4, tzr_ld4(T1 + row * MT_P1 + 4 * c4));
    }
    __builtin_amdgcn_wave_barrier();  // (the next tile overwrites T1)
  }
  // ---- this wave's sums -> LDS row; the workgroup's four rows are added in wave order into its partial-sum row
  float* my = &red[wv][0];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int hb = 0; hb < 4; ++hb)
#pragma unroll
      for (int j = 0; j < 4; ++j) my[(16 * jb + 4 * q + j) * MT_H1 + 16 * hb + r] = accW[jb][hb][j];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const float s2 = mm_sum_q(db2a[jb]), s3 = mm_sum_q(dw3a[jb]);
    if (q == 0) {
      my[MT_H2 * MT_H1 + 16 * jb + r] = s2;
      my[MT_H2 * MT_H1 + MT_H2 + 16 * jb + r] = s3;
    }
  }
  {
    const float s3 = mm_sum_q(db3a), sl = mm_sum_q(lossa);
    if (lane == 0) {
      my[MT_H2 * MT_H1 + 2 * MT_H2] = s3;
      my[MT_H2 * MT_H1 + 2 * MT_H2 + 1] = sl * inv;
    }
  }
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const float s1 = mm_sum_q(db1a[hb]);
    if (q == 0) my[MT_H2 * MT_H1 + 2 * MT_H2 + 2 + 16 * hb + r] = s1;
  }
  __syncthreads();
  float* row = parts + (size_t)blockIdx.x * MT_ROW;
  for (int i = threadIdx.x; i < MT_ROW; i += MM_THREADS) {
    float v = red[0][i];
#pragma unroll
    for (int w = 1; w < MM_WAVES; ++w) v += red[w][i];
    row[i] = v;
  }
}

// launchers called by tzr_mlp_tail (mlp_ops.hip) when the shape matches; return the number of partial-sum rows written
int tzr_mlp_tail64_launch(const float* d_y1, int64_t y1_stride, const void* d_labels, int labels_itemsize, int labels_are_float,
                          int64_t B, const float* d_W2, const float* d_b2, const float* d_w3, const float* d_b3, float* d_logits,
                          float* d_g1, int64_t g1_stride, float* parts, hipStream_t s) {
  const int64_t tiles = (B + MM_TS - 1) / MM_TS;
  const int G = (int)std::min<int64_t>((tiles + MM_WAVES - 1) / MM_WAVES, g_tzr_mlp_mfma > 0 ? std::min(g_tzr_mlp_mfma, MM_MAX_WG) : MM_MAX_WG);
#define TZR_TAIL64_LAUNCH(T)                                                                                              \
  hipLaunchKernelGGL(tzr_mlp_tail64_kernel<T>, dim3(G), dim3(MM_THREADS), 0, s, d_y1, y1_stride, static_cast<const T*>(d_labels), \
                     B, d_W2, d_b2, d_w3, d_b3, d_logits, d_g1, g1_stride, parts)
  if (labels_are_float && labels_itemsize == 4) TZR_TAIL64_LAUNCH(float);
  else if (!labels_are_float && labels_itemsize == 8) TZR_TAIL64_LAUNCH(int64_t);
  else if (!labels_are_float && labels_itemsize == 4) TZR_TAIL64_LAUNCH(int32_t);
  else return -1;
#undef TZR_TAIL64_LAUNCH
  return G;
}

// ---- bottom MLP: ha = relu(x Wa^T + ba) [B, 64], hb = relu(ha Wb^T + bb) [B, 16]; K0 <= 16 ---------------------------------
#define MB_H1 64
#define MB_H2 16
#define MB_PX 20  // row pitch of the [16 x 16] tiles (x, masked dhb)

__global__ __launch_bounds__(MM_THREADS) void tzr_mlp2_fwd16_kernel(
    const float* __restrict__ x, int64_t xs, int64_t B, int K0, const float* __restrict__ Wa, const float* __restrict__ ba,
    const float* __restrict__ Wb, const float* __restrict__ bb, float* __restrict__ ha, int64_t has, float* __restrict__ hb,
    int64_t hbs) {
  __shared__ __attribute__((aligned(16))) float T1s[MM_WAVES][MM_TS * MT_P1];  // ha tile
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TZR_WAVE));
  const int r = lane & 15, q = lane >> 4;
  float* T1 = &T1s[wv][0];
  // B operands: first layer k = 4 ks + q over the inputs (zero beyond K0): Wa[16 hb + r][k]; second layer k = 16 q + ks over
  // the 64 hidden units: Wb[r][k]
  float Wfa[4][4], Wfb[16], bav[4];
#pragma unroll
  for (int hbk = 0; hbk < 4; ++hbk) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = 4 * ks + q;
      const float w = Wa[(16 * hbk + r) * K0 + (k < K0 ? k : 0)];
      Wfa[hbk][ks] = k < K0 ? w : 0.f;
    }
    bav[hbk] = ba ? ba[16 * hbk + r] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 v = tzr_ld4(Wb + r * MB_H1 + 16 * q + 4 * i);
    Wfb[4 * i] = v.x; Wfb[4 * i + 1] = v.y; Wfb[4 * i + 2] = v.z; Wfb[4 * i + 3] = v.w;
  }
  const float bbv = bb ? bb[r] : 0.f;
  const int64_t tiles = (B + MM_TS - 1) / MM_TS;
  const int64_t nw = (int64_t)gridDim.x * MM_WAVES;
  for (int64_t t = (int64_t)blockIdx.x * MM_WAVES + wv; t < tiles; t += nw) {
    const int64_t b0 = t * MM_TS;
    // ---- A = x[sample r][4 ks + q] straight from HBM (the whole tile is 16 x K0 floats)
    float xa[4];
    {
      const int64_t b = b0 + r < B ? b0 + r : B - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = 4 * ks + q;
        const float v = x[b * xs + (k < K0 ? k : 0)];
        xa[ks] = (k < K0 && b0 + r < B) ? v : 0.f;
      }
    }
    mm_f32x4 acc[4];
#pragma unroll
    for (int hbk = 0; hbk < 4; ++hbk) acc[hbk] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int hbk = 0; hbk < 4; ++hbk) acc[hbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks], Wfa[hbk][ks], acc[hbk], 0, 0, 0);
    // reg j of lane (r, q): sample 4 q + j, unit 16 hbk + r -> the ha tile in LDS
#pragma unroll
    for (int hbk = 0; hbk < 4; ++hbk)
#pragma unroll
      for (int j = 0; j < 4; ++j) T1[(4 * q + j) * MT_P1 + 16 * hbk + r] = fmaxf(acc[hbk][j] + bav[hbk], 0.f);
    __builtin_amdgcn_wave_barrier();
    // ---- ha out (rows of 256 bytes, 16 bytes per lane) and hb = relu(ha Wb^T + bb): A = ha[sample r][16 q + ks]
    mm_f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + TZR_WAVE * i, row = e >> 4, c4 = e & 15;
      if (b0 + row < B) tzr_st4(ha + (b0 + row) * has + 4 * c4, tzr_ld4(T1 + row * MT_P1 + 4 * c4));
      const float4 av = tzr_ld4(T1 + r * MT_P1 + 16 * q + 4 * i);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, Wfb[4 * i], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, Wfb[4 * i + 1], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, Wfb[4 * i + 2], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, Wfb[4 * i + 3], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (b0 + 4 * q + j < B) hb[(b0 + 4 * q + j) * hbs + r] = fmaxf(acc2[j] + bbv, 0.f);
    __builtin_amdgcn_wave_barrier();  // (the next tile overwrites T1)
  }
}

void tzr_mlp2_fwd16_launch(const float* d_x, int64_t xs, int64_t B, int K0, const float* d_Wa, const float* d_ba, const float* d_Wb,
                           const float* d_bb, float* d_ha, int64_t has, float* d_hb, int64_t hbs, hipStream_t s) {
  const int64_t tiles = (B + MM_TS - 1) / MM_TS;
  const int G = (int)std::min<int64_t>((tiles + MM_WAVES - 1) / MM_WAVES, g_tzr_mlp_mfma > 0 ? g_tzr_mlp_mfma : 1024);
  hipLaunchKernelGGL(tzr_mlp2_fwd16_kernel, dim3(G), dim3(MM_THREADS), 0, s, d_x, xs, B, K0, d_Wa, d_ba, d_Wb, d_bb, d_ha, has,
                     d_hb, hbs);
}

// backward: the four parameter gradients from dhb = d(loss)/d(hb) (x is data: no input gradient).  Partial-sum row of a
// workgroup: [dWb (16 x 64) | dbb (16) | dWa (64 x K0) | dba (64)].
__global__ __launch_bounds__(MM_THREADS) void tzr_mlp2_bwd16_kernel(
    const float* __restrict__ dhb, int64_t dhbs, const float* __restrict__ hb, int64_t hbs, const float* __restrict__ ha,
    int64_t has, const float* __restrict__ x, int64_t xs, int64_t B, int K0, const float* __restrict__ Wb,
    float* __restrict__ parts, int P) {
  __shared__ __attribute__((aligned(16))) float Thas[MM_WAVES][MM_TS * MT_P1];  // ha tile
  __shared__ __attribute__((aligned(16))) float Tgas[MM_WAVES][MM_TS * MT_P1];  // d(loss)/d(pre-activation of ha)
  __shared__ __attribute__((aligned(16))) float Tgbs[MM_WAVES][MM_TS * MB_PX];  // dhb masked by hb > 0
  __shared__ __attribute__((aligned(16))) float Txs[MM_WAVES][MM_TS * MB_PX];   // x tile, zero beyond K0
  __shared__ float red[MM_WAVES][MB_H2 * MB_H1 + MB_H2 + MB_H1 * 16 + MB_H1];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TZR_WAVE));
  const int r = lane & 15, q = lane >> 4;
  float *Tha = &Thas[wv][0], *Tga = &Tgas[wv][0], *Tgb = &Tgbs[wv][0], *Tx = &Txs[wv][0];
  // B operand of gb Wb (k = 4 q + ks over the 16 units of hb): Wb[k][16 hbk + r]
  float Wf[4][4];
#pragma unroll
  for (int hbk = 0; hbk < 4; ++hbk)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) Wf[hbk][ks] = Wb[(4 * q + ks) * MB_H1 + 16 * hbk + r];
  mm_f32x4 accWb[4], accWa[4];  // dWb[unit 4 q + j][16 hbk + r]; dWa[16 hbk + 4 q + j][input r]
#pragma unroll
  for (int hbk = 0; hbk < 4; ++hbk) accWb[hbk] = accWa[hbk] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
  float dbba[4] = {0.f, 0.f, 0.f, 0.f}, dbaa[4] = {0.f, 0.f, 0.f, 0.f};
  const int trow = lane >> 2, tc4 = lane & 3;  // this lane's 16-byte piece of a [16 x 16] tile
  const int64_t tiles = (B + MM_TS - 1) / MM_TS;
  const int64_t nw = (int64_t)gridDim.x * MM_WAVES;
  // a wave's tiles, fetched a turn ahead (two or more turns per wave at the step's batch, each a chain that began with an HBM
  // round trip): ha (four pieces per lane), dhb, hb and x (one piece per lane each).  Rows behind the batch and inputs
  // behind K0 read a clamped address and are zeroed at the LDS stores.
  struct In {
    float4 ha[4], d, h;
    float x[4];
  };
  auto fetch = [&](In& v, int64_t tt) {
    const int64_t b0 = tt * MM_TS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + TZR_WAVE * i, row = e >> 4, c4 = e & 15;
      const int64_t b = b0 + row < B ? b0 + row : B - 1;
      v.ha[i] = tzr_ld4(ha + b * has + 4 * c4);
    }
    const int64_t b = b0 + trow < B ? b0 + trow : B - 1;
    v.d = tzr_ld4(dhb + b * dhbs + 4 * tc4);
    v.h = tzr_ld4(hb + b * hbs + 4 * tc4);
    const float* xp = x + b * xs;
#pragma unroll
    for (int j = 0; j < 4; ++j) v.x[j] = xp[4 * tc4 + j < K0 ? 4 * tc4 + j : 0];
  };
  int64_t t = (int64_t)blockIdx.x * MM_WAVES + wv;
  In in;
  if (t < tiles) fetch(in, t);
  for (; t < tiles; t += nw) {
    const int64_t b0 = t * MM_TS;
    // ---- tiles -> LDS: ha, gb = dhb masked by hb > 0, x; the next turn's take off
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + TZR_WAVE * i, row = e >> 4, c4 = e & 15;
      tzr_st4(Tha + row * MT_P1 + 4 * c4, b0 + row < B ? in.ha[i] : tzr_zero4());
    }
    {
      float4 g = tzr_zero4(), xv = tzr_zero4();
      if (b0 + trow < B) {
        const float4 d = in.d, h = in.h;
        g = make_float4(h.x > 0.f ? d.x : 0.f, h.y > 0.f ? d.y : 0.f, h.z > 0.f ? d.z : 0.f, h.w > 0.f ? d.w : 0.f);
        const int k = 4 * tc4;
        xv.x = k < K0 ? in.x[0] : 0.f;
        xv.y = k + 1 < K0 ? in.x[1] : 0.f;
        xv.z = k + 2 < K0 ? in.x[2] : 0.f;
        xv.w = k + 3 < K0 ? in.x[3] : 0.f;
      }
      dbba[0] += g.x; dbba[1] += g.y; dbba[2] += g.z; dbba[3] += g.w;  // columns 4 tc4 .. of dbb, rows trow + 16 n
      tzr_st4(Tgb + trow * MB_PX + 4 * tc4, g);
      tzr_st4(Tx + trow * MB_PX + 4 * tc4, xv);
    }
    fetch(in, t + nw < tiles ? t + nw : t);
    __builtin_amdgcn_wave_barrier();
    // ---- ga = (gb Wb) masked by ha > 0: A = gb[sample r][4 q + ks]
    mm_f32x4 acc[4];
#pragma unroll
    for (int hbk = 0; hbk < 4; ++hbk) acc[hbk] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const float4 av = tzr_ld4(Tgb + r * MB_PX + 4 * q);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int hbk = 0; hbk < 4; ++hbk) acc[hbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[ks], Wf[hbk][ks], acc[hbk], 0, 0, 0);
    }
#pragma unroll
    for (int hbk = 0; hbk < 4; ++hbk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h = Tha[(4 * q + j) * MT_P1 + 16 * hbk + r];
        const float g = h > 0.f ? acc[hbk][j] : 0.f;
        dbaa[hbk] += g;
        Tga[(4 * q + j) * MT_P1 + 16 * hbk + r] = g;
      }
    __builtin_amdgcn_wave_barrier();
    // ---- dWb += gb^T ha (A = gb[sample 4 ks + q][unit r], B = ha[sample 4 ks + q][16 hbk + r]);
    //      dWa += ga^T x  (A = ga[sample 4 ks + q][16 hbk + r], B = x[sample 4 ks + q][input r])
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float gbv = Tgb[(4 * ks + q) * MB_PX + r], xv = Tx[(4 * ks + q) * MB_PX + r];
#pragma unroll
      for (int hbk = 0; hbk < 4; ++hbk) {
        accWb[hbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(gbv, Tha[(4 * ks + q) * MT_P1 + 16 * hbk + r], accWb[hbk], 0, 0, 0);
        accWa[hbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tga[(4 * ks + q) * MT_P1 + 16 * hbk + r], xv, accWa[hbk], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the next tile overwrites the LDS tiles)
  }
  // ---- this wave's sums -> LDS row; the workgroup's four rows are added in wave order into its partial-sum row
  float* my = &red[wv][0];
  const int oWa = MB_H2 * MB_H1 + MB_H2, oba = oWa + MB_H1 * K0;
#pragma unroll
  for (int hbk = 0; hbk < 4; ++hbk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      my[(4 * q + j) * MB_H1 + 16 * hbk + r] = accWb[hbk][j];
      if (r < K0) my[oWa + (16 * hbk + 4 * q + j) * K0 + r] = accWa[hbk][j];
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) {  // dbb: over the 16 lanes that share l & 3 (rows of the tile)
    float v = dbba[e];
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (trow == 0) my[MB_H2 * MB_H1 + 4 * tc4 + e] = v;
  }
#pragma unroll
  for (int hbk = 0; hbk < 4; ++hbk) {
    const float v = mm_sum_q(dbaa[hbk]);
    if (q == 0) my[oba + 16 * hbk + r] = v;
  }
  __syncthreads();
  float* row = parts + (size_t)blockIdx.x * P;
  for (int i = threadIdx.x; i < P; i += MM_THREADS) {
    float v = red[0][i];
#pragma unroll
    for (int w = 1; w < MM_WAVES; ++w) v += red[w][i];
    row[i] = v;
  }
}

int tzr_mlp2_bwd16_launch(const float* d_dhb, int64_t dhbs, const float* d_hb, int64_t hbs, const float* d_ha, int64_t has,
                          const float* d_x, int64_t xs, int64_t B, int K0, const float* d_Wb, float* parts, int P, hipStream_t s) {
  const int64_t tiles = (B + MM_TS - 1) / MM_TS;
  const int G = (int)std::min<int64_t>((tiles + MM_WAVES - 1) / MM_WAVES, g_tzr_mlp_mfma > 0 ? std::min(g_tzr_mlp_mfma, MM_MAX_WG) : MM_MAX_WG);
  hipLaunchKernelGGL(tzr_mlp2_bwd16_kernel, dim3(G), dim3(MM_THREADS), 0, s, d_dhb, dhbs, d_hb, hbs, d_ha, has, d_x, xs, B, K0, d_Wb,
                     parts, P);
  return G;
}
