// Small dense layers of DLRM / DeepFM as whole-layer-stack kernels (gfx950).
//
// tzrec builds them as tzrec.modules.mlp.MLP stacks (/root/reference/tzrec/modules/mlp.py:58-83, used by
// /root/reference/tzrec/models/dlrm.py:101-135): the bottom MLP 13 -> 64 -> 16 on the dense features and the tail of
// the top MLP (64 -> 32 -> 1 logit -> BCE-with-logits, rank_model.py:190-191,233-240).  On a GPU PyTorch runs each as a
// string of skinny GEMMs (N = 64 / 32 / 16 / 1) plus activation / bias-gradient / loss kernels: ~17 launches of
// 5-17 us at B = 65536 (profiles/r03q/kernel_stats.csv: ~160 us of a 0.80 ms step) and the same ~17 launches at
// B = 8192, where a launch cannot cost less than ~4.5 us whatever it does.  The arithmetic is tiny (< 1 GFLOP per
// step); what costs is launches and the [B, 64] / [B, 32] intermediates crossing HBM between them.  Here:
//
//   tzr_mlp2_fwd   y = relu(relu(x Wa^T + ba) Wb^T + bb) in one launch (both activations are kept for the backward)
//   tzr_mlp2_bwd   its weight / bias gradients from d(loss)/dy in one launch + a finish
//   tzr_mlp_tail   given y1 = relu(z W1^T + b1): y2 = relu(y1 W2^T + b2), logit = y2 w3 + b3, mean BCE-with-logits
//                  AND the whole backward down to g1 = d(loss)/d(pre-activation of y1) and its column sums (= the bias
//                  gradient of the 783 -> 64 layer), W2 / b2 / w3 / b3 gradients: one launch + a finish.
//
// VALU kernels (no MFMA: K <= 64 products on tiles of 64 samples, operands in LDS, weights read as LDS broadcasts);
// every reduction over the batch runs in a fixed order (per-workgroup partial sums over a fixed tile assignment, then
// one finish pass in workgroup order): bit-reproducible, no float atomics.
#include "tzr_common.h"

int g_tzr_mlp_mfma = 0;  // tzr_tune("mlp_mfma"): -1 = the general LDS-tiled kernels for every shape (A/B, tests); 0 = MFMA kernels where the shape fits
int tzr_mlp_tail64_launch(const float* d_y1, int64_t y1_stride, const void* d_labels, int labels_itemsize, int labels_are_float,
                          int64_t B, const float* d_W2, const float* d_b2, const float* d_w3, const float* d_b3, float* d_logits,
                          float* d_g1, int64_t g1_stride, float* parts, hipStream_t s);  // mlp_mfma.hip
void tzr_mlp2_fwd16_launch(const float* d_x, int64_t xs, int64_t B, int K0, const float* d_Wa, const float* d_ba, const float* d_Wb,
                           const float* d_bb, float* d_ha, int64_t has, float* d_hb, int64_t hbs, hipStream_t s);
int tzr_mlp2_bwd16_launch(const float* d_dhb, int64_t dhbs, const float* d_hb, int64_t hbs, const float* d_ha, int64_t has,
                          const float* d_x, int64_t xs, int64_t B, int K0, const float* d_Wb, float* parts, int P, hipStream_t s);

#define ML_THREADS 256
#define ML_TS 64      // samples per tile
#define ML_K0 32      // max input width of tzr_mlp2
#define ML_H1 64      // max hidden width (tzr_mlp2 first layer; tzr_mlp_tail input)
#define ML_H2 32      // max output width (tzr_mlp2 second layer; tzr_mlp_tail hidden)
#define ML_MAX_WG 512  // workgroups = partial-sum rows of the backward kernels (two per CU)
#define ML_PK (ML_K0 + 4)  // LDS pitches of the activation tiles [sample][feature]: rows 16-byte aligned
#define ML_P1 (ML_H1 + 4)
#define ML_P2 (ML_H2 + 4)

__device__ __forceinline__ int ml_up4(int n) { return (n + 3) & ~3; }

// Register-blocked products on LDS tiles (a thread owns a 4 x 4 or 2 x 4 block of outputs, so an LDS read feeds
// 4 FMAs: the first version -- one output column per thread, two LDS reads per FMA -- was LDS-bound at ~100 us).
//
// acc[i][c] += sum_{k < nin} In[(s0 + 16 i) * pin + k] * W[k * pw + c0 + c]: 4 samples x 4 output columns.  The four
// samples of a thread are 16 apart: the lanes of a wave then read rows s0 = 0 .. 15 of the tile -- `pin` = 4 mod 64
// floats apart, 16 different LDS banks -- instead of rows 4 apart (16 banks apart: 4-way conflicts on every read).
// W is k-major ([nin][pw], 16-byte aligned rows): the transposed nn.Linear weight for a forward layer, the weight as
// stored ([out][in]) for the back-projection g_in = g_out W.
__device__ __forceinline__ void ml_gemm_4x4(float (&acc)[4][4], const float* In, int pin, int s0, const float* W, int pw,
                                            int c0, int nin) {
#pragma unroll 4
  for (int k = 0; k < nin; ++k) {
    const float4 w = *reinterpret_cast<const float4*>(W + k * pw + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = In[(s0 + 16 * i) * pin + k];
      acc[i][0] = fmaf(v, w.x, acc[i][0]);
      acc[i][1] = fmaf(v, w.y, acc[i][1]);
      acc[i][2] = fmaf(v, w.z, acc[i][2]);
      acc[i][3] = fmaf(v, w.w, acc[i][3]);
    }
  }
}
// block mapping of a [ML_TS samples] x [nout columns] product: thread -> (samples s0, s0 + 16, s0 + 32, s0 + 48; 4 columns from c0)
__device__ __forceinline__ bool ml_block(int nout, int* s0, int* c0) {
  const int ogs = ml_up4(nout) >> 2;
  const int og = (int)threadIdx.x % ogs, sg = (int)threadIdx.x / ogs;
  *s0 = sg;
  *c0 = 4 * og;
  return sg < ML_TS / 4;
}

// acc[a][b] += sum_{s < ML_TS} G[s * pg + j0 + a] * A[s * pa + k0 + b]: a 2 x 4 block of a weight gradient
__device__ __forceinline__ void ml_outer_2x4(float (&acc)[2][4], const float* G, int pg, int j0, const float* A, int pa,
                                             int k0) {
#pragma unroll 4
  for (int s = 0; s < ML_TS; ++s) {
    const float2 g = *reinterpret_cast<const float2*>(G + s * pg + j0);
    const float4 a = *reinterpret_cast<const float4*>(A + s * pa + k0);
    acc[0][0] = fmaf(g.x, a.x, acc[0][0]); acc[0][1] = fmaf(g.x, a.y, acc[0][1]);
    acc[0][2] = fmaf(g.x, a.z, acc[0][2]); acc[0][3] = fmaf(g.x, a.w, acc[0][3]);
    acc[1][0] = fmaf(g.y, a.x, acc[1][0]); acc[1][1] = fmaf(g.y, a.y, acc[1][1]);
    acc[1][2] = fmaf(g.y, a.z, acc[1][2]); acc[1][3] = fmaf(g.y, a.w, acc[1][3]);
  }
}
// weight-gradient blocks of a [nj] x [nk] gradient (padded to even / multiple-of-4 sizes in LDS): block m of this
// thread = rows j0 .. j0+1, columns k0 .. k0+3; at most ML_OB blocks per thread (64 x 64 / 8 / 256 = 2)
#define ML_OB 2
__device__ __forceinline__ bool ml_oblock(int m, int nj, int nk, int* j0, int* k0) {
  const int kbs = ml_up4(nk) >> 2, jbs = (nj + 1) >> 1;
  const int o = (int)threadIdx.x + m * ML_THREADS;
  *k0 = 4 * (o % kbs);
  *j0 = 2 * (o / kbs);
  return o < kbs * jbs;
}

// column sums of an LDS tile: thread j < n adds the tile's samples in order
__device__ __forceinline__ float ml_colsum(const float* A, int pa, int n) {
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;  // four independent chains, combined in a fixed order
  if ((int)threadIdx.x < n)
    for (int s = 0; s < ML_TS; s += 4) {
      v0 += A[s * pa + threadIdx.x];
      v1 += A[(s + 1) * pa + threadIdx.x];
      v2 += A[(s + 2) * pa + threadIdx.x];
      v3 += A[(s + 3) * pa + threadIdx.x];
    }
  return (v0 + v1) + (v2 + v3);
}

// ------------------------------------------------------------------------------------------------------------------
// two-layer ReLU MLP, forward
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ML_THREADS) void tzr_mlp2_fwd_kernel(
    const float* __restrict__ x, int64_t xs, int64_t B, int K0, const float* __restrict__ Wa,
    const float* __restrict__ ba, int H1, const float* __restrict__ Wb, const float* __restrict__ bb, int H2,
    float* __restrict__ ha, int64_t has, float* __restrict__ hb, int64_t hbs) {
  __shared__ __attribute__((aligned(16))) float sWaT[ML_K0 * ML_H1];  // [k][H1p]
  __shared__ __attribute__((aligned(16))) float sWbT[ML_H1 * ML_H2];  // [k][H2p]
  __shared__ float sba[ML_H1], sbb[ML_H2];
  __shared__ __attribute__((aligned(16))) float sx[ML_TS * ML_PK];
  __shared__ __attribute__((aligned(16))) float sh[ML_TS * ML_P1];
  const int H1p = ml_up4(H1), H2p = ml_up4(H2);
  for (int i = threadIdx.x; i < K0 * H1p; i += ML_THREADS) {
    const int k = i / H1p, j = i - k * H1p;
    sWaT[i] = j < H1 ? Wa[j * K0 + k] : 0.f;
  }
  for (int i = threadIdx.x; i < H1 * H2p; i += ML_THREADS) {
    const int k = i / H2p, j = i - k * H2p;
    sWbT[i] = j < H2 ? Wb[j * H1 + k] : 0.f;
  }
  if ((int)threadIdx.x < H1p) sba[threadIdx.x] = (ba && (int)threadIdx.x < H1) ? ba[threadIdx.x] : 0.f;
  if ((int)threadIdx.x < H2p) sbb[threadIdx.x] = (bb && (int)threadIdx.x < H2) ? bb[threadIdx.x] : 0.f;
  const int64_t tiles = (B + ML_TS - 1) / ML_TS;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t b0 = t * ML_TS;
    const int ns = (int)min((int64_t)ML_TS, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < ML_TS * K0; i += ML_THREADS) {
      const int s = i / K0, k = i - s * K0;
      sx[s * ML_PK + k] = s < ns ? x[(b0 + s) * xs + k] : 0.f;
    }
    __syncthreads();
    int s0, c0;
    if (ml_block(H1, &s0, &c0)) {
      float acc[4][4] = {};
      ml_gemm_4x4(acc, sx, ML_PK, s0, sWaT, H1p, c0, K0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v;
        v.x = fmaxf(acc[i][0] + sba[c0], 0.f); v.y = fmaxf(acc[i][1] + sba[c0 + 1], 0.f);
        v.z = fmaxf(acc[i][2] + sba[c0 + 2], 0.f); v.w = fmaxf(acc[i][3] + sba[c0 + 3], 0.f);
        *reinterpret_cast<float4*>(sh + (s0 + 16 * i) * ML_P1 + c0) = v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ns * H1; i += ML_THREADS) {
      const int r = i / H1, j = i - r * H1;
      ha[(b0 + r) * has + j] = sh[r * ML_P1 + j];
    }
    if (ml_block(H2, &s0, &c0)) {
      float acc[4][4] = {};
      ml_gemm_4x4(acc, sh, ML_P1, s0, sWbT, H2p, c0, H1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (s0 + 16 * i < ns) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c0 + c < H2) hb[(b0 + s0 + 16 * i) * hbs + c0 + c] = fmaxf(acc[i][c] + sbb[c0 + c], 0.f);
        }
    }
  }
}

extern "C" int tzr_mlp2_fwd(const float* d_x, int64_t x_stride, int64_t B, int K0, const float* d_Wa,
                            const float* d_ba, int H1, const float* d_Wb, const float* d_bb, int H2, float* d_ha,
                            int64_t ha_stride, float* d_hb, int64_t hb_stride, void* stream) {
  if (!d_x || !d_Wa || !d_Wb || !d_ha || !d_hb || B < 0 || K0 <= 0 || H1 <= 0 || H2 <= 0) return TZR_ERR_INVALID;
  if (K0 > ML_K0 || H1 > ML_H1 || H2 > ML_H2) return TZR_ERR_UNSUPPORTED;
  if (B == 0) return TZR_OK;
  // the DLRM-Criteo shape runs on the matrix cores, one wave per 16-sample tile (mlp_mfma.hip)
  if (g_tzr_mlp_mfma >= 0 && K0 <= 16 && H1 == 64 && H2 == 16 && !(ha_stride & 3) && !(reinterpret_cast<uintptr_t>(d_ha) & 15) &&
      !(reinterpret_cast<uintptr_t>(d_Wb) & 15)) {
    tzr_mlp2_fwd16_launch(d_x, x_stride, B, K0, d_Wa, d_ba, d_Wb, d_bb, d_ha, ha_stride, d_hb, hb_stride,
                          static_cast<hipStream_t>(stream));
    TZR_CHECK_LAUNCH();
    return TZR_OK;
  }
  const int64_t tiles = (B + ML_TS - 1) / ML_TS;
  hipLaunchKernelGGL(tzr_mlp2_fwd_kernel, dim3((unsigned)std::min<int64_t>(tiles, 1024)), dim3(ML_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_x, x_stride, B, K0, d_Wa, d_ba, H1, d_Wb, d_bb, H2, d_ha,
                     ha_stride, d_hb, hb_stride);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// finish: out[o] = scale * sum over workgroups (in order) of parts[g][o]; outputs scattered to up to 6 destinations
// ------------------------------------------------------------------------------------------------------------------
struct MlParts {
  float* dst[6];
  int n[6];
};
// 16 outputs per workgroup x 16 slices of the workgroup range: a thread adds its slice's partials (independent
// loads, 8 in flight), thread (o, slice 0) then adds the 16 slice sums in slice order.  (One thread per output walking
// all G partials was a chain of G dependent adds on strided loads: 120 us for G = 512.)
__global__ __launch_bounds__(ML_THREADS) void tzr_mlp_finish_kernel(const float* __restrict__ parts, int G, int P,
                                                                    MlParts out) {
  __shared__ float sl[16][17];
  const int ol = threadIdx.x & 15, sq = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;
  const int per = (G + 15) / 16;
  const int g0 = sq * per, g1 = min(G, g0 + per);
  float v = 0.f;
  if (o < P) {
    for (int g = g0; g < g1; g += 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = g + j < g1 ? parts[(size_t)(g + j) * P + o] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) v += t[j];
    }
  }
  sl[sq][ol] = v;
  __syncthreads();
  if (sq != 0 || o >= P) return;
  v = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) v += sl[q][ol];
  int base = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (o >= base && o < base + out.n[i]) {
      if (out.dst[i]) out.dst[i][o - base] = v;
      return;
    }
    base += out.n[i];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// two-layer ReLU MLP, backward (weight and bias gradients; the input is data: no input gradient)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ML_THREADS) void tzr_mlp2_bwd_kernel(
    const float* __restrict__ dhb, int64_t dhbs, const float* __restrict__ hb, int64_t hbs,
    const float* __restrict__ ha, int64_t has, const float* __restrict__ x, int64_t xs, int64_t B, int K0, int H1,
    int H2, const float* __restrict__ Wb, float* __restrict__ parts, int P) {
  __shared__ __attribute__((aligned(16))) float sWb[ML_H2 * ML_H1];  // [j][H1p]: as stored
  __shared__ __attribute__((aligned(16))) float sx[ML_TS * ML_PK];
  __shared__ __attribute__((aligned(16))) float sha[ML_TS * ML_P1];
  __shared__ __attribute__((aligned(16))) float sga[ML_TS * ML_P1];
  __shared__ __attribute__((aligned(16))) float sgb[ML_TS * ML_P2];
  const int H1p = ml_up4(H1), K0p = ml_up4(K0), H2e = (H2 + 1) & ~1;
  for (int i = threadIdx.x; i < H2 * H1p; i += ML_THREADS) {
    const int j = i / H1p, k = i - j * H1p;
    sWb[i] = k < H1 ? Wb[j * H1 + k] : 0.f;
  }
  float dWb[ML_OB][2][4] = {}, dWa[ML_OB][2][4] = {};
  float dbb = 0.f, dba = 0.f;
  const int64_t tiles = (B + ML_TS - 1) / ML_TS;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t b0 = t * ML_TS;
    const int ns = (int)min((int64_t)ML_TS, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < ML_TS * K0p; i += ML_THREADS) {
      const int s = i / K0p, k = i - s * K0p;
      sx[s * ML_PK + k] = (s < ns && k < K0) ? x[(b0 + s) * xs + k] : 0.f;
    }
    for (int i = threadIdx.x; i < ML_TS * H1p; i += ML_THREADS) {
      const int s = i / H1p, k = i - s * H1p;
      sha[s * ML_P1 + k] = (s < ns && k < H1) ? ha[(b0 + s) * has + k] : 0.f;
    }
    for (int i = threadIdx.x; i < ML_TS * H2e; i += ML_THREADS) {
      const int s = i / H2e, j = i - s * H2e;
      float g = 0.f;
      if (s < ns && j < H2 && hb[(b0 + s) * hbs + j] > 0.f) g = dhb[(b0 + s) * dhbs + j];
      sgb[s * ML_P2 + j] = g;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < ML_OB; ++m) {
      int j0, k0;
      if (ml_oblock(m, H2, H1, &j0, &k0)) ml_outer_2x4(dWb[m], sgb, ML_P2, j0, sha, ML_P1, k0);
    }
    dbb += ml_colsum(sgb, ML_P2, H2);
    {  // ga = (gb Wb) masked by ha > 0
      int s0, c0;
      if (ml_block(H1, &s0, &c0)) {
        float acc[4][4] = {};
        ml_gemm_4x4(acc, sgb, ML_P2, s0, sWb, H1p, c0, H2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 h = *reinterpret_cast<const float4*>(sha + (s0 + 16 * i) * ML_P1 + c0);
          float4 v;
          v.x = h.x > 0.f ? acc[i][0] : 0.f; v.y = h.y > 0.f ? acc[i][1] : 0.f;
          v.z = h.z > 0.f ? acc[i][2] : 0.f; v.w = h.w > 0.f ? acc[i][3] : 0.f;
          *reinterpret_cast<float4*>(sga + (s0 + 16 * i) * ML_P1 + c0) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < ML_OB; ++m) {
      int j0, k0;
      if (ml_oblock(m, H1, K0, &j0, &k0)) ml_outer_2x4(dWa[m], sga, ML_P1, j0, sx, ML_PK, k0);
    }
    dba += ml_colsum(sga, ML_P1, H1);
  }
  // partial row of this workgroup: [dWb (H2*H1) | dbb (H2) | dWa (H1*K0) | dba (H1)]
  float* row = parts + (size_t)blockIdx.x * P;
#pragma unroll
  for (int m = 0; m < ML_OB; ++m) {
    int j0, k0;
    if (ml_oblock(m, H2, H1, &j0, &k0)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (j0 + a < H2 && k0 + b < H1) row[(j0 + a) * H1 + k0 + b] = dWb[m][a][b];
    }
    if (ml_oblock(m, H1, K0, &j0, &k0)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (j0 + a < H1 && k0 + b < K0) row[H2 * H1 + H2 + (j0 + a) * K0 + k0 + b] = dWa[m][a][b];
    }
  }
  if ((int)threadIdx.x < H2) row[H2 * H1 + threadIdx.x] = dbb;
  if ((int)threadIdx.x < H1) row[H2 * H1 + H2 + H1 * K0 + threadIdx.x] = dba;
}

extern "C" size_t tzr_mlp_workspace(void) { return (size_t)ML_MAX_WG * (ML_H2 * ML_H1 + ML_H1 * ML_K0 + 4 * ML_H1) * sizeof(float) + 256; }

// the producing launch of tzr_mlp2_bwd: one partial-sum row per workgroup, [dWb: H2 x H1 | dbb: H2 | dWa: H1 x K0 | dba: H1]
static int mlp2_bwd_partials(const float* d_dhb, int64_t dhb_stride, const float* d_hb, int64_t hb_stride, const float* d_ha,
                             int64_t ha_stride, const float* d_x, int64_t x_stride, int64_t B, int K0, int H1, int H2,
                             const float* d_Wb, void* ws, size_t ws_bytes, int* G_out, int* P_out, void* stream) {
  if (!d_dhb || !d_hb || !d_ha || !d_x || !d_Wb || B <= 0 || K0 <= 0 || H1 <= 0 || H2 <= 0) return TZR_ERR_INVALID;
  if (K0 > ML_K0 || H1 > ML_H1 || H2 > ML_H2) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_mlp_workspace() - 256) return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t tiles = (B + ML_TS - 1) / ML_TS;
  int G = (int)std::min<int64_t>(tiles, ML_MAX_WG);
  const int P = H2 * H1 + H2 + H1 * K0 + H1;
  float* parts = static_cast<float*>(ws);
  if (g_tzr_mlp_mfma >= 0 && K0 <= 16 && H1 == 64 && H2 == 16 && !((ha_stride | hb_stride | dhb_stride) & 3) &&
      !((reinterpret_cast<uintptr_t>(d_ha) | reinterpret_cast<uintptr_t>(d_hb) | reinterpret_cast<uintptr_t>(d_dhb)) & 15))
    G = tzr_mlp2_bwd16_launch(d_dhb, dhb_stride, d_hb, hb_stride, d_ha, ha_stride, d_x, x_stride, B, K0, d_Wb, parts, P, s);
  else
    hipLaunchKernelGGL(tzr_mlp2_bwd_kernel, dim3(G), dim3(ML_THREADS), 0, s, d_dhb, dhb_stride, d_hb, hb_stride, d_ha,
                       ha_stride, d_x, x_stride, B, K0, H1, H2, d_Wb, parts, P);
  TZR_CHECK_LAUNCH();
  *G_out = G;
  *P_out = P;
  return TZR_OK;
}

extern "C" int tzr_mlp2_bwd(const float* d_dhb, int64_t dhb_stride, const float* d_hb, int64_t hb_stride,
                            const float* d_ha, int64_t ha_stride, const float* d_x, int64_t x_stride, int64_t B, int K0,
                            int H1, int H2, const float* d_Wb, float* d_dWa, float* d_dba, float* d_dWb, float* d_dbb,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!d_dWa || !d_dWb) return TZR_ERR_INVALID;
  int G, P;
  const int rc = mlp2_bwd_partials(d_dhb, dhb_stride, d_hb, hb_stride, d_ha, ha_stride, d_x, x_stride, B, K0, H1, H2, d_Wb, ws, ws_bytes,
                                   &G, &P, stream);
  if (rc != TZR_OK) return rc;
  MlParts out;
  for (int i = 0; i < 6; ++i) {
    out.dst[i] = nullptr;
    out.n[i] = 0;
  }
  out.dst[0] = d_dWb; out.n[0] = H2 * H1;
  out.dst[1] = d_dbb; out.n[1] = H2;
  out.dst[2] = d_dWa; out.n[2] = H1 * K0;
  out.dst[3] = d_dba; out.n[3] = H1;
  hipLaunchKernelGGL(tzr_mlp_finish_kernel, dim3((P + 15) / 16), dim3(ML_THREADS), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(ws), G, P, out);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ... without the finish launch: the partial-sum rows stay in `ws` (G rows of P floats: *out_G, *out_P; columns
// [dWb | dbb | dWa | dba]) for tzr_dense_adam_fused, which adds them up on its way to the four tensors' Adam updates.
extern "C" int tzr_mlp2_bwd_parts(const float* d_dhb, int64_t dhb_stride, const float* d_hb, int64_t hb_stride,
                                  const float* d_ha, int64_t ha_stride, const float* d_x, int64_t x_stride, int64_t B, int K0,
                                  int H1, int H2, const float* d_Wb, void* ws, size_t ws_bytes, int* out_G, int* out_P,
                                  void* stream) {
  if (!out_G || !out_P) return TZR_ERR_INVALID;
  return mlp2_bwd_partials(d_dhb, dhb_stride, d_hb, hb_stride, d_ha, ha_stride, d_x, x_stride, B, K0, H1, H2, d_Wb, ws, ws_bytes, out_G,
                           out_P, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// tail of the top MLP: hidden layer, logit, mean BCE-with-logits, and the backward down to the previous layer
// ------------------------------------------------------------------------------------------------------------------
template <typename LabelT>
__global__ __launch_bounds__(ML_THREADS) void tzr_mlp_tail_kernel(
    const float* __restrict__ y1, int64_t y1s, const LabelT* __restrict__ labels, int64_t B, int H1,
    const float* __restrict__ W2, const float* __restrict__ b2, int H2, const float* __restrict__ w3,
    const float* __restrict__ b3, float* __restrict__ logits, float* __restrict__ g1, int64_t g1s,
    float* __restrict__ parts, int P) {
  __shared__ __attribute__((aligned(16))) float sW2T[ML_H1 * ML_H2];  // [k][H2p]: forward
  __shared__ __attribute__((aligned(16))) float sW2[ML_H2 * ML_H1];   // [j][H1p]: back-projection
  __shared__ float sb2[ML_H2], sw3[ML_H2];
  __shared__ __attribute__((aligned(16))) float sy1[ML_TS * ML_P1];
  __shared__ __attribute__((aligned(16))) float sy2[ML_TS * ML_P2];
  __shared__ __attribute__((aligned(16))) float sg2[ML_TS * ML_P2];
  float* const sg1 = sy1;  // g1 overwrites y1 in place: a thread masks with, then replaces, its own 4 x 4 entries
  __shared__ float sdl[ML_TS], sloss[ML_TS];
  const int H1p = ml_up4(H1), H2p = ml_up4(H2);
  for (int i = threadIdx.x; i < H1 * H2p; i += ML_THREADS) {
    const int k = i / H2p, j = i - k * H2p;
    sW2T[i] = j < H2 ? W2[j * H1 + k] : 0.f;
  }
  for (int i = threadIdx.x; i < H2 * H1p; i += ML_THREADS) {
    const int j = i / H1p, k = i - j * H1p;
    sW2[i] = k < H1 ? W2[j * H1 + k] : 0.f;
  }
  if ((int)threadIdx.x < H2p) {
    sb2[threadIdx.x] = (b2 && (int)threadIdx.x < H2) ? b2[threadIdx.x] : 0.f;
    sw3[threadIdx.x] = (int)threadIdx.x < H2 ? w3[threadIdx.x] : 0.f;
  }
  const float bias3 = b3 ? b3[0] : 0.f;
  const float inv = 1.0f / (float)B;
  float dW2[ML_OB][2][4] = {};
  float db2 = 0.f, dw3 = 0.f, db3 = 0.f, db1 = 0.f, loss = 0.f;
  const int64_t tiles = (B + ML_TS - 1) / ML_TS;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t b0 = t * ML_TS;
    const int ns = (int)min((int64_t)ML_TS, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < ML_TS * H1p; i += ML_THREADS) {
      const int s = i / H1p, k = i - s * H1p;
      sy1[s * ML_P1 + k] = (s < ns && k < H1) ? y1[(b0 + s) * y1s + k] : 0.f;
    }
    __syncthreads();
    int s0, c0;
    if (ml_block(H2, &s0, &c0)) {
      float acc[4][4] = {};
      ml_gemm_4x4(acc, sy1, ML_P1, s0, sW2T, H2p, c0, H1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v;
        v.x = fmaxf(acc[i][0] + sb2[c0], 0.f); v.y = fmaxf(acc[i][1] + sb2[c0 + 1], 0.f);
        v.z = fmaxf(acc[i][2] + sb2[c0 + 2], 0.f); v.w = fmaxf(acc[i][3] + sb2[c0 + 3], 0.f);
        if (c0 + 1 >= H2) v.y = 0.f;  // padded columns carry nothing
        if (c0 + 2 >= H2) v.z = 0.f;
        if (c0 + 3 >= H2) v.w = 0.f;
        *reinterpret_cast<float4*>(sy2 + (s0 + 16 * i) * ML_P2 + c0) = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < ML_TS) {  // one sample per thread: logit, loss term, d(loss)/d(logit)
      float z = bias3;
      for (int j = 0; j < H2; ++j) z = fmaf(sy2[threadIdx.x * ML_P2 + j], sw3[j], z);
      float dl = 0.f, li = 0.f;
      if ((int)threadIdx.x < ns) {
        const float y = (float)labels[b0 + threadIdx.x];
        const float e = expf(-fabsf(z));
        li = fmaxf(z, 0.f) - z * y + log1pf(e);  // the formula of tzr_bce_logits (dense_ops.hip)
        const float sig = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        dl = (sig - y) * inv;
        logits[b0 + threadIdx.x] = z;
      }
      sdl[threadIdx.x] = dl;
      sloss[threadIdx.x] = li;
    }
    __syncthreads();
    if ((int)threadIdx.x < H2) {  // dw3[j] += sum_s dl[s] y2[s][j]
      float v = 0.f;
      for (int r = 0; r < ML_TS; ++r) v = fmaf(sdl[r], sy2[r * ML_P2 + threadIdx.x], v);
      dw3 += v;
    }
    if (threadIdx.x == ML_THREADS - 1) {
      float v = 0.f, l = 0.f;
      for (int r = 0; r < ML_TS; ++r) {
        v += sdl[r];
        l += sloss[r];
      }
      db3 += v;
      loss += l;
    }
    for (int i = threadIdx.x; i < ML_TS * H2p; i += ML_THREADS) {
      const int r = i / H2p, j = i - r * H2p;
      sg2[r * ML_P2 + j] = (j < H2 && sy2[r * ML_P2 + j] > 0.f) ? sdl[r] * sw3[j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < ML_OB; ++m) {
      int j0, k0;
      if (ml_oblock(m, H2, H1, &j0, &k0)) ml_outer_2x4(dW2[m], sg2, ML_P2, j0, sy1, ML_P1, k0);
    }
    db2 += ml_colsum(sg2, ML_P2, H2);
    __syncthreads();  // everybody is done reading y1 as an operand: g1 replaces it below
    if (ml_block(H1, &s0, &c0)) {  // g1 = (g2 W2) masked by y1 > 0
      float acc[4][4] = {};
      ml_gemm_4x4(acc, sg2, ML_P2, s0, sW2, H1p, c0, H2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 h = *reinterpret_cast<const float4*>(sy1 + (s0 + 16 * i) * ML_P1 + c0);
        float4 v;
        v.x = h.x > 0.f ? acc[i][0] : 0.f; v.y = h.y > 0.f ? acc[i][1] : 0.f;
        v.z = h.z > 0.f ? acc[i][2] : 0.f; v.w = h.w > 0.f ? acc[i][3] : 0.f;
        *reinterpret_cast<float4*>(sg1 + (s0 + 16 * i) * ML_P1 + c0) = v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ns * H1; i += ML_THREADS) {
      const int r = i / H1, k = i - r * H1;
      g1[(b0 + r) * g1s + k] = sg1[r * ML_P1 + k];
    }
    db1 += ml_colsum(sg1, ML_P1, H1);
  }
  // partial row: [dW2 (H2*H1) | db2 (H2) | dw3 (H2) | db3, loss (2) | db1 (H1)]
  float* row = parts + (size_t)blockIdx.x * P;
#pragma unroll
  for (int m = 0; m < ML_OB; ++m) {
    int j0, k0;
    if (ml_oblock(m, H2, H1, &j0, &k0)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (j0 + a < H2 && k0 + b < H1) row[(j0 + a) * H1 + k0 + b] = dW2[m][a][b];
    }
  }
  if ((int)threadIdx.x < H2) {
    row[H2 * H1 + threadIdx.x] = db2;
    row[H2 * H1 + H2 + threadIdx.x] = dw3;
  }
  if (threadIdx.x == ML_THREADS - 1) {
    row[H2 * H1 + 2 * H2] = db3;
    row[H2 * H1 + 2 * H2 + 1] = loss * inv;
  }
  if ((int)threadIdx.x < H1) row[H2 * H1 + 2 * H2 + 2 + threadIdx.x] = db1;
}

// d_scalars[0] = d(loss)/d(b3), d_scalars[1] = the loss (mean over the batch)
extern "C" int tzr_mlp_tail(const float* d_y1, int64_t y1_stride, const void* d_labels, int labels_itemsize,
                            int labels_are_float, int64_t B, int H1, const float* d_W2, const float* d_b2, int H2,
                            const float* d_w3, const float* d_b3, float* d_logits, float* d_g1, int64_t g1_stride,
                            float* d_dW2, float* d_db2, float* d_dw3, float* d_scalars, float* d_db1, void* ws,
                            size_t ws_bytes, void* stream) {
  if (!d_y1 || !d_labels || !d_W2 || !d_w3 || !d_logits || !d_g1 || !d_dW2 || !d_db2 || !d_dw3 || !d_scalars ||
      !d_db1 || B <= 0 || H1 <= 0 || H2 <= 0)
    return TZR_ERR_INVALID;
  if (H1 > ML_H1 || H2 > ML_H2) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_mlp_workspace() - 256) return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t tiles = (B + ML_TS - 1) / ML_TS;
  int G = (int)std::min<int64_t>(tiles, ML_MAX_WG);
  const int P = H2 * H1 + 2 * H2 + 2 + H1;
  float* parts = static_cast<float*>(ws);
  // the DLRM-Criteo shape runs on the matrix cores, one wave per 16-sample tile (mlp_mfma.hip)
  const bool mfma = g_tzr_mlp_mfma >= 0 && H1 == 64 && H2 == 32 && !((y1_stride | g1_stride) & 3) &&
                    !((reinterpret_cast<uintptr_t>(d_y1) | reinterpret_cast<uintptr_t>(d_g1)) & 15);
  if (mfma) {
    G = tzr_mlp_tail64_launch(d_y1, y1_stride, d_labels, labels_itemsize, labels_are_float, B, d_W2, d_b2, d_w3, d_b3, d_logits,
                              d_g1, g1_stride, parts, s);
    if (G < 0) return TZR_ERR_UNSUPPORTED;
  } else {
#define TZR_TAIL_LAUNCH(T)                                                                                         \
  hipLaunchKernelGGL(tzr_mlp_tail_kernel<T>, dim3(G), dim3(ML_THREADS), 0, s, d_y1, y1_stride,                     \
                     static_cast<const T*>(d_labels), B, H1, d_W2, d_b2, H2, d_w3, d_b3, d_logits, d_g1, g1_stride, \
                     parts, P)
  if (labels_are_float && labels_itemsize == 4) TZR_TAIL_LAUNCH(float);
  else if (!labels_are_float && labels_itemsize == 8) TZR_TAIL_LAUNCH(int64_t);
  else if (!labels_are_float && labels_itemsize == 4) TZR_TAIL_LAUNCH(int32_t);
  else return TZR_ERR_UNSUPPORTED;
#undef TZR_TAIL_LAUNCH
  }
  MlParts out;
  out.dst[0] = d_dW2; out.n[0] = H2 * H1;
  out.dst[1] = d_db2; out.n[1] = H2;
  out.dst[2] = d_dw3; out.n[2] = H2;
  out.dst[3] = d_scalars; out.n[3] = 2;
  out.dst[4] = d_db1; out.n[4] = H1;
  out.dst[5] = nullptr; out.n[5] = 0;
  hipLaunchKernelGGL(tzr_mlp_finish_kernel, dim3((P + 15) / 16), dim3(ML_THREADS), 0, s, parts, G, P,
                     out);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
