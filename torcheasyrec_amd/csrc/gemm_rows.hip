// Products of a TALL activation matrix with a SMALL weight matrix on the matrix cores (gfx950, exact-fp32
// v_mfma_f32_16x16x4_f32): the Linear layers of an MLP that runs on every POSITION of a sequence batch -- the attention MLP of
// DIN on the jagged positions (/root/reference/tzrec/modules/sequence.py:101-128 over /root/reference/tzrec/modules/mlp.py:58-83;
// N = 450 k rows at the Taobao config against weights of 144 x 256 and 256 x 64).
//
//   tzr_linear_rows        out[n, h] = act(sum_k in[n, k] W(k, h) + bias[h])     forward of a layer (W = weight[h][k]) and the
//                                                                                 input gradient (W = weight[k][h], no bias)
//   tzr_linear_rows_wgrad  dW[h, k]  = sum_n g[n, h] x[n, k]                      weight gradient: the contraction is the long side
//
// A GEMM library tiles such a product as if both sides were large: round 5's profile of the DIN step had five Tensile kernels
// at 0.58-0.72 of the fp32 MFMA peak, each re-reading its weight tile from L2 per 16-wide k block.  Here the weight is the
// thing that stays put:
//
// tzr_linear_rows: tzr_linear_bwd_relu's mapping (linear_bwd.hip) generalised.  A workgroup walks 16-row tiles of the input
//   (persistent, grid-strided); wave w owns 16 HB output columns and keeps its block of the weight IN REGISTERS for the whole
//   kernel (K / 4 x HB per lane).  MFMA operand roles swapped (A = weight^T block, B = input tile^T) so that a lane ends up with
//   FOUR CONSECUTIVE output columns of one row: bias, ReLU and the store are 16-byte pieces straight to global memory.  The
//   contraction index is permuted (k(s, q) = 16 (s / 4) + 4 q + s % 4) so that four MFMA steps' input operands are one
//   ds_read_b128; the input tile is the only thing in LDS (double-buffered, ONE barrier per tile, next tile's loads in flight
//   across it).  Per tile and wave: K / 16 LDS reads for K / 4 x HB MFMAs.
// tzr_linear_rows_wgrad: a workgroup of four waves owns the WHOLE [H, K] output (wave w: H / 4 rows x all K columns, up to 36
//   accumulator blocks) and walks 16-row blocks of g and x through LDS (rows = the contraction index: operands are column
//   reads, strides padded to 16 mod 32 banks); its partial sum goes to the workspace once, at the end, and a second launch adds
//   the workgroups' partial sums in workgroup order: no float atomics, bit-reproducible.
#include <tzr_gfx950.h>

#include "tzr_common.h"

#define GR_TS 16
#define GR_MAX_WG 768
#define GR_WG_MAX_WG 512

typedef float gr_f32x4 __attribute__((ext_vector_type(4)));

int g_tzr_gemm_rows_wg = 0;  // tzr_tune("gemm_rows_wg"): > 0 caps the workgroups of both kernels (tests: many tiles per workgroup)

// ---- out = act(in W + bias) ------------------------------------------------------------------------------------------
template <int KS /* K / 4 */, int HB /* 16-column blocks per wave */, int WAVES, int TT /* 16-row tiles per turn */, int WPE, bool RV /* a gathered row vector is added */>
__global__ __launch_bounds__(WAVES* TZR_WAVE) TZR_WAVES_PER_EU(WPE) void tzr_gemm_rows_kernel(
    const float* __restrict__ in, int64_t in_stride, const float* __restrict__ W, int64_t w_stride, int w_out_major,
    const float* __restrict__ bias, const float* __restrict__ rowvec, int64_t rowvec_stride, const int32_t* __restrict__ row_index,
    int relu, int64_t N, float* __restrict__ out, int64_t out_stride) {
  constexpr int K = 4 * KS, P = K + 4, K4 = K / 4, THREADS = WAVES * TZR_WAVE, ROWS = GR_TS * TT;
  constexpr int NST = (ROWS * K4 + THREADS - 1) / THREADS;  // 16-byte pieces of a turn's input rows per thread
  __shared__ __attribute__((aligned(16))) float TA[2][ROWS * P];
  __shared__ int32_t TI[2][ROWS];  // row_index of a turn's rows: staged with the rows (a register carried from turn to turn made
                                   // its consumer wait for the turn's stores)
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  const int cb = wv * 16 * HB;
  // weight blocks: operand A of step s is W(k(s, q), cb + 16 jb + r)
  float Wa[HB][KS];
  if (w_out_major) {  // nn.Linear's weight [H, K]: four consecutive k are one 16-byte load
#pragma unroll
    for (int jb = 0; jb < HB; ++jb)
#pragma unroll
      for (int e = 0; e < KS / 4; ++e) {
        const float4 w4 = tzr_ldg4(W + (int64_t)(cb + 16 * jb + r) * w_stride + 16 * e + 4 * q);
        Wa[jb][4 * e] = w4.x, Wa[jb][4 * e + 1] = w4.y, Wa[jb][4 * e + 2] = w4.z, Wa[jb][4 * e + 3] = w4.w;
      }
  } else {
#pragma unroll
    for (int jb = 0; jb < HB; ++jb)
#pragma unroll
      for (int s = 0; s < KS; ++s) Wa[jb][s] = tzr_ldg(W + (int64_t)(16 * (s >> 2) + 4 * q + (s & 3)) * w_stride + cb + 16 * jb + r);
  }
  gr_f32x4 bv[HB];  // the accumulators START from the bias
#pragma unroll
  for (int jb = 0; jb < HB; ++jb) {
    const float4 b4 = bias ? tzr_ldg4(bias + cb + 16 * jb + 4 * q) : tzr_zero4();
    bv[jb] = gr_f32x4{b4.x, b4.y, b4.z, b4.w};
  }
  // the thread's pieces of a turn's rows (a piece index beyond the tile: the thread repeats its first piece -- no branch in the loop)
  int srow[NST], sc4[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = (int)threadIdx.x + i * THREADS;
    const int id2 = idx < ROWS * K4 ? idx : (int)threadIdx.x % (ROWS * K4);
    srow[i] = id2 / K4;
    sc4[i] = id2 % K4;
  }
  const int64_t nturns = (N + ROWS - 1) / ROWS;
  const int64_t last = N - 1;
  // Addresses = a UNIFORM base of the turn (scalar registers, scalar arithmetic) + a lane offset that never changes (one
  // register per piece): no vector instruction per load -- the fp32 matrix pipe and the vector ALU are one pipe, every VALU
  // instruction of the turn is time taken from the products (variants without any memory access ran no faster: profiles/r06w).
  // Rows beyond N (the last turn only) are read as row N - 1: an output row depends on its own input row only, and theirs go
  // to row N - 1's address with row N - 1's bits.
  const int rows_last = (int)(N - (nturns - 1) * ROWS);  // rows of the last turn, 1 .. ROWS
  uint32_t poff[NST], poff_tail[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    poff[i] = (uint32_t)(srow[i] * (int)in_stride + 4 * sc4[i]);
    poff_tail[i] = (uint32_t)((srow[i] < rows_last ? srow[i] : rows_last - 1) * (int)in_stride + 4 * sc4[i]);
  }
  auto fetch = [&](int64_t turn, int i) -> float4 {
#ifdef GR_NO_LOAD  // (experiment)
    return make_float4((float)turn, 1.f, 2.f, (float)i);
#endif
    const float* const base = in + turn * ROWS * in_stride;  // uniform
    return tzr_ldg4(base + (turn == nturns - 1 ? poff_tail[i] : poff[i]));
  };
  int64_t t = blockIdx.x;
  const int irow = (int)threadIdx.x % ROWS;  // the row whose index this thread stages (threads >= ROWS repeat: same value, no branch)
  if (t < nturns) {
#pragma unroll
    for (int i = 0; i < NST; ++i) tzr_st4(&TA[0][srow[i] * P + 4 * sc4[i]], fetch(t, i));
    if (RV) TI[0][irow] = row_index[t * ROWS + (t == nturns - 1 ? (irow < rows_last ? irow : rows_last - 1) : irow)];
  }
  __syncthreads();
  int buf = 0;
  for (; t < nturns; t += gridDim.x) {
    const int64_t tn = std::min<int64_t>(t + gridDim.x, nturns - 1);  // (the last turn's prefetch repeats a turn: never staged twice into a buffer in use)
    float4 nx[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) nx[i] = fetch(tn, i);
    // the per-row addend: gathered now, added behind the products
    float4 rv[TT][HB];
    int32_t nxi = 0;
    if (RV) {
      nxi = row_index[tn * ROWS + (tn == nturns - 1 ? (irow < rows_last ? irow : rows_last - 1) : irow)];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float* const rp = rowvec + (int64_t)TI[buf][tt * GR_TS + r] * rowvec_stride + cb + 4 * q;
#pragma unroll
        for (int jb = 0; jb < HB; ++jb) rv[tt][jb] = tzr_ldg4(rp + 16 * jb);
      }
    }
    gr_f32x4 acc[TT][HB];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int jb = 0; jb < HB; ++jb) acc[tt][jb] = bv[jb];
    const float* const A = &TA[buf][r * P + 4 * q];
    // operand reads two steps ahead of their products
    constexpr int E = KS / 4;
    float4 pa[TT][2];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
#ifdef GR_NO_LDS  // (experiment: operands from registers)
      pa[tt][0] = pa[tt][1] = make_float4((float)lane, 1.f, (float)t, 3.f);
#else
      pa[tt][0] = tzr_ld4(A + tt * GR_TS * P);
      pa[tt][1] = tzr_ld4(A + tt * GR_TS * P + (E > 1 ? 16 : 0));
#endif
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float a4[TT][4];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float4 av = pa[tt][e & 1];
        a4[tt][0] = av.x, a4[tt][1] = av.y, a4[tt][2] = av.z, a4[tt][3] = av.w;
#ifndef GR_NO_LDS
        if (e + 2 < E) pa[tt][e & 1] = tzr_ld4(A + tt * GR_TS * P + 16 * (e + 2));
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int jb = 0; jb < HB; ++jb)
            acc[tt][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[jb][4 * e + c], a4[tt][c], acc[tt][jb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next turn's rows into the other buffer BEFORE this turn's stores are issued: the wait for those loads must not become
    // a wait for the stores (stores behind a row test: at the join the compiler has to assume none was issued and counts short)
#pragma unroll
    for (int i = 0; i < NST; ++i) tzr_st4(&TA[buf ^ 1][srow[i] * P + 4 * sc4[i]], nx[i]);
    if (RV) TI[buf ^ 1][irow] = nxi;
    // stores WITHOUT a row test: a row beyond N was computed from row N - 1's input (fetch) and goes to row N - 1's address -- the
    // same bits the owner of that row stores.  (Stores inside a branch make every later vector-memory wait count as if none had
    // been issued: the row-index prefetch below would wait for this turn's stores.)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int rl = t == nturns - 1 ? (tt * GR_TS + r < rows_last ? tt * GR_TS + r : rows_last - 1) : tt * GR_TS + r;
      float* op = out + t * ROWS * out_stride + (uint32_t)(rl * (int)out_stride + cb + 4 * q);
#pragma unroll
      for (int jb = 0; jb < HB; ++jb) {
        float4 o = make_float4(acc[tt][jb][0], acc[tt][jb][1], acc[tt][jb][2], acc[tt][jb][3]);
        if (RV) o = tzr_add4(o, rv[tt][jb]);
        if (relu) o = make_float4(tzr_relu(o.x), tzr_relu(o.y), tzr_relu(o.z), tzr_relu(o.w));
#ifdef GR_NO_STORE  // (experiment: what the stores cost -- none is issued, the compiler cannot know)
        if (o.x == 12345.678f)
#endif
#ifdef GR_NT_STORE
        tzr_stg4_nt(op + 16 * jb, o);
#else
        tzr_stg4(op + 16 * jb, o);
#endif
      }
    }
#ifndef GR_NO_BARRIER  // (experiment: wrong results, the turn without its synchronisation)
    tzr_lds_barrier();  // every wave is done with TA[buf]; TA[buf ^ 1] is complete (global loads / stores stay in flight)
#endif
    buf ^= 1;
  }
}

// ---- dW = g^T x ------------------------------------------------------------------------------------------------------
// (v padded so that rows 4 s + q, q = 0..3, of a column read start 16 banks apart: ds_read_b32 serves lanes 0-31 = two rows per cycle)
__host__ __device__ constexpr int gr_pad(int v) { return v + ((16 - v % 32) + 32) % 32; }

template <int HBW /* 16-row blocks of dW per wave */, int KB /* 16-column blocks of dW */, int WAVES, int WPE>
__global__ __launch_bounds__(WAVES* TZR_WAVE) TZR_WAVES_PER_EU(WPE) void tzr_gemm_tn_kernel(
    const float* __restrict__ g, int64_t g_stride, const float* __restrict__ x, int64_t x_stride, int64_t N, float* __restrict__ parts) {
  constexpr int H = 16 * HBW * WAVES, K = 16 * KB, SG = gr_pad(H), SX = gr_pad(K), G4 = H / 4, X4 = K / 4;
  constexpr int THREADS = WAVES * TZR_WAVE, PIECES = GR_TS * (G4 + X4), NST = (PIECES + THREADS - 1) / THREADS;
  __shared__ __attribute__((aligned(16))) float TG[2][GR_TS * SG];
  __shared__ __attribute__((aligned(16))) float TX[2][GR_TS * SX];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int r = lane & 15, q = lane >> 4;
  // the thread's pieces of a block: which matrix, row, 16-byte column
  int skind[NST], srow[NST], sc4[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = (int)threadIdx.x + i * THREADS;
    if (idx < GR_TS * G4) {
      skind[i] = 0, srow[i] = idx / G4, sc4[i] = idx % G4;
    } else if (idx < PIECES) {
      skind[i] = 1, srow[i] = (idx - GR_TS * G4) / X4, sc4[i] = (idx - GR_TS * G4) % X4;
    } else {
      skind[i] = -1, srow[i] = 0, sc4[i] = 0;
    }
  }
  auto fetch = [&](int64_t blk, int i) -> float4 {
    const int64_t row = blk * GR_TS + srow[i];
#ifdef GR_TN_NO_LOAD  // (experiment: the kernel without its global loads)
    if (row >= 0) return make_float4(1.f, 2.f, 3.f, (float)i);
#endif
    if (skind[i] < 0 || row >= N) return tzr_zero4();
    return skind[i] == 0 ? tzr_ldg4(g + row * g_stride + 4 * sc4[i]) : tzr_ldg4(x + row * x_stride + 4 * sc4[i]);
  };
  auto stage = [&](int b, int i, float4 v) {
    if (skind[i] == 0) tzr_st4(&TG[b][srow[i] * SG + 4 * sc4[i]], v);
    else if (skind[i] == 1) tzr_st4(&TX[b][srow[i] * SX + 4 * sc4[i]], v);
  };
  gr_f32x4 acc[HBW][KB];
#pragma unroll
  for (int i = 0; i < HBW; ++i)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[i][kb] = gr_f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t nblocks = (N + GR_TS - 1) / GR_TS;
  int64_t t = blockIdx.x;
  if (t < nblocks) {
#pragma unroll
    for (int i = 0; i < NST; ++i) stage(0, i, fetch(t, i));
  }
  __syncthreads();
  int buf = 0;
  for (; t < nblocks; t += gridDim.x) {
    const int64_t tn = t + gridDim.x;
    float4 nx[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) nx[i] = tn < nblocks ? fetch(tn, i) : tzr_zero4();
    const float* const GA = &TG[buf][q * SG + 16 * wv * HBW + r];
    const float* const XB = &TX[buf][q * SX + r];
    // operands of step s + 1 are read before step s is multiplied (not with 36 accumulator blocks: no registers left for it)
    constexpr int PF = HBW * KB < 36 ? 1 : 0;
    float a[2][HBW], b[2][KB];
#pragma unroll
    for (int i = 0; i < HBW; ++i) a[0][i] = GA[16 * i];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) b[0][kb] = XB[16 * kb];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (PF ? s < 3 : s > 0) {
        const int sl = PF ? s + 1 : s;  // the step whose operands are read now
#pragma unroll
        for (int i = 0; i < HBW; ++i) a[PF ? sl & 1 : 0][i] = GA[4 * sl * SG + 16 * i];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[PF ? sl & 1 : 0][kb] = XB[4 * sl * SX + 16 * kb];
      }
      if (PF) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HBW; ++i)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          acc[i][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[PF ? s & 1 : 0][i], b[PF ? s & 1 : 0][kb], acc[i][kb], 0, 0, 0);
      if (PF) __builtin_amdgcn_sched_barrier(0);
    }
    if (tn < nblocks) {
#pragma unroll
      for (int i = 0; i < NST; ++i) stage(buf ^ 1, i, nx[i]);
    }
    tzr_lds_barrier();
    buf ^= 1;
  }
  // the workgroup's partial sum: parts[wg][h][k], h = 16 (wv HBW + i) + 4 q + j, k = 16 kb + r
  float* const mine = parts + (size_t)blockIdx.x * H * K;
#pragma unroll
  for (int i = 0; i < HBW; ++i)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) tzr_stg(mine + (size_t)(16 * (wv * HBW + i) + 4 * q + j) * K + 16 * kb + r, acc[i][kb][j]);
}

// dW[e] = sum over the workgroups of parts[.][e], in workgroup order (16 slices summed concurrently with 8 loads in flight each,
// combined in slice order -- tzr_linear_bwd_finish_kernel's arrangement)
#define GR_FIN_THREADS 1024
__global__ __launch_bounds__(GR_FIN_THREADS) void tzr_gemm_tn_finish_kernel(const float* __restrict__ parts, int n_wg, int H, int K,
                                                                            float* __restrict__ dw, int64_t dw_stride, int accumulate) {
  __shared__ float red[GR_FIN_THREADS];
  const int E = H * K;
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  float t = 0.f;
  if (e < E) {
    for (int k0 = slice; k0 < n_wg; k0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 16 * u;
        v[u] = k < n_wg ? tzr_ldg(parts + (size_t)k * E + e) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (slice == 0 && e < E) {
    float s = 0.f;
    for (int sl = 0; sl < 16; ++sl) s += red[sl * 64 + (threadIdx.x & 63)];
    float* const o = dw + (int64_t)(e / K) * dw_stride + e % K;
    *o = accumulate ? *o + s : s;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
namespace {
struct RowsCfg { int K, H, HB, WAVES, TT, WPE, RV; };
// (weight registers per lane = K / 4 x HB; a single accumulator chain (HB = 1) takes two tiles per turn; WPE = waves per SIMD the
// kernel is compiled for, checked with scripts/r05/kres.py: no scratch; RV = the gathered-row-vector form is built too)
#define GR_ROWS_CONFIGS(X) \
  X(16, 64, 1, 4, 2, 3, 1) \
  X(16, 128, 2, 4, 1, 3, 1) \
  X(16, 256, 4, 4, 1, 3, 1) \
  X(32, 64, 1, 4, 2, 3, 1) \
  X(32, 128, 2, 4, 1, 3, 1) \
  X(32, 256, 4, 4, 1, 3, 1) \
  X(64, 16, 1, 1, 1, 3, 0) \
  X(128, 16, 1, 1, 1, 3, 0) \
  X(256, 16, 1, 1, 1, 1, 0) \
  X(64, 32, 1, 2, 1, 3, 0) \
  X(128, 32, 1, 2, 1, 3, 0) \
  X(256, 32, 1, 2, 1, 2, 0) \
  X(48, 64, 1, 4, 2, 3, 1) \
  X(48, 128, 2, 4, 1, 3, 1) \
  X(48, 256, 4, 4, 1, 3, 1) \
  X(64, 64, 1, 4, 2, 3, 1) \
  X(64, 128, 2, 4, 1, 3, 1) \
  X(64, 256, 4, 4, 1, 3, 1) \
  X(96, 64, 1, 4, 2, 3, 1) \
  X(96, 128, 2, 4, 1, 3, 1) \
  X(96, 256, 4, 4, 1, 2, 1) \
  X(128, 64, 1, 4, 2, 3, 1) \
  X(128, 128, 2, 4, 1, 3, 1) \
  X(128, 256, 4, 4, 1, 2, 1) \
  X(144, 64, 1, 4, 2, 3, 1) \
  X(144, 128, 2, 4, 1, 3, 1) \
  X(144, 256, 4, 4, 1, 2, 0) \
  X(192, 64, 1, 4, 2, 3, 0) \
  X(192, 128, 2, 4, 1, 2, 1) \
  X(192, 256, 2, 8, 1, 3, 0) \
  X(256, 64, 1, 4, 2, 2, 1) \
  X(256, 128, 2, 4, 1, 2, 1) \
  X(256, 256, 2, 8, 1, 2, 1) \
  X(64, 48, 1, 3, 2, 3, 0) \
  X(64, 96, 1, 6, 1, 3, 0) \
  X(64, 144, 3, 3, 1, 3, 0) \
  X(64, 192, 3, 4, 1, 3, 0) \
  X(128, 48, 1, 3, 2, 3, 0) \
  X(128, 96, 1, 6, 1, 3, 0) \
  X(128, 144, 3, 3, 1, 2, 0) \
  X(128, 192, 3, 4, 1, 2, 0) \
  X(256, 48, 1, 3, 2, 2, 0) \
  X(256, 96, 1, 6, 1, 3, 0) \
  X(256, 144, 1, 9, 1, 3, 0) \
  X(256, 192, 2, 6, 1, 2, 0)
constexpr RowsCfg kRows[] = {
#define GR_ROW(K_, H_, HB_, W_, TT_, WPE_, RV_) {K_, H_, HB_, W_, TT_, WPE_, RV_},
    GR_ROWS_CONFIGS(GR_ROW)
#undef GR_ROW
};
const RowsCfg* rows_cfg(int K, int H) {
  for (const RowsCfg& c : kRows)
    if (c.K == K && c.H == H) return &c;
  return nullptr;
}
}  // namespace

extern "C" int tzr_linear_rows_supported(int K, int H) {  // 1: without a row vector only; 3: with one too
  const RowsCfg* c = rows_cfg(K, H);
  return c ? (c->RV ? 3 : 1) : 0;
}

extern "C" int tzr_linear_rows(const float* d_in, int64_t in_stride, const float* d_w, int64_t w_stride, int w_out_major,
                               const float* d_bias, const float* d_rowvec, int64_t rowvec_stride, const int32_t* d_row_index,
                               int relu, int64_t N, int K, int H, float* d_out, int64_t out_stride, void* stream) {
  if (!d_in || !d_w || !d_out || N <= 0 || K <= 0 || H <= 0 || (d_rowvec && !d_row_index)) return TZR_ERR_INVALID;
  const RowsCfg* c = rows_cfg(K, H);
  if (!c || (d_rowvec && !c->RV)) return TZR_ERR_UNSUPPORTED;
  if ((in_stride & 3) || (out_stride & 3) || in_stride < K || out_stride < H || w_stride < (w_out_major ? K : H)) return TZR_ERR_UNSUPPORTED;
  if ((w_out_major && (w_stride & 3)) || (d_rowvec && ((rowvec_stride & 3) || rowvec_stride < H))) return TZR_ERR_UNSUPPORTED;
  if (((uintptr_t)d_in | (uintptr_t)d_out | (uintptr_t)d_bias | (uintptr_t)d_rowvec | (uintptr_t)(w_out_major ? d_w : d_in)) & 15)
    return TZR_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t nturns = (N + GR_TS * c->TT - 1) / (GR_TS * c->TT);
  const int lds = 2 * GR_TS * c->TT * (K + 4) * (int)sizeof(float);
  const int per_cu = std::min(c->WPE * 4 / c->WAVES, 160 * 1024 / lds);  // resident workgroups per CU
  int cap = std::max(per_cu, 1) * 256;
  if (g_tzr_gemm_rows_wg > 0) cap = std::min(cap, g_tzr_gemm_rows_wg);
  const int grid = (int)std::min<int64_t>(nturns, cap);
#define GR_CASE(K_, H_, HB_, W_, TT_, WPE_, RV_)                                                                                      \
  if (K == K_ && H == H_) {                                                                                                           \
    if (d_rowvec)                                                                                                                     \
      hipLaunchKernelGGL((tzr_gemm_rows_kernel<K_ / 4, HB_, W_, TT_, WPE_, RV_ != 0>), dim3((unsigned)grid), dim3(W_ * TZR_WAVE), 0, s, d_in, \
                         in_stride, d_w, w_stride, w_out_major, d_bias, d_rowvec, rowvec_stride, d_row_index, relu, N, d_out, out_stride); \
    else                                                                                                                              \
      hipLaunchKernelGGL((tzr_gemm_rows_kernel<K_ / 4, HB_, W_, TT_, WPE_, false>), dim3((unsigned)grid), dim3(W_ * TZR_WAVE), 0, s, d_in,  \
                         in_stride, d_w, w_stride, w_out_major, d_bias, d_rowvec, rowvec_stride, d_row_index, relu, N, d_out, out_stride); \
  } else
  GR_ROWS_CONFIGS(GR_CASE) { return TZR_ERR_UNSUPPORTED; }
#undef GR_CASE
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

namespace {
struct TnCfg { int H, K, HBW, KB, WAVES; };
// dW [H, K]: wave w owns H / WAVES rows x all K columns; HBW x KB accumulator blocks (<= 36: 144 registers)
constexpr TnCfg kTn[] = {
    {64, 16, 1, 1, 4},   {64, 32, 1, 2, 4},   {128, 16, 2, 1, 4},  {128, 32, 2, 2, 4},  {256, 16, 4, 1, 4},  {256, 32, 4, 2, 4},
    {64, 48, 1, 3, 4},   {64, 64, 1, 4, 4},   {64, 96, 1, 6, 4},   {64, 128, 1, 8, 4},   {64, 144, 1, 9, 4},  {64, 192, 1, 12, 4},
    {64, 256, 1, 16, 4}, {128, 48, 2, 3, 4},  {128, 64, 2, 4, 4},  {128, 96, 2, 6, 4},   {128, 128, 2, 8, 4}, {128, 144, 2, 9, 4},
    {128, 192, 2, 12, 4}, {128, 256, 2, 16, 4}, {256, 48, 4, 3, 4}, {256, 64, 4, 4, 4},  {256, 96, 4, 6, 4},  {256, 128, 4, 8, 4},
    {256, 144, 4, 9, 4},
};
const TnCfg* tn_cfg(int H, int K) {
  for (const TnCfg& c : kTn)
    if (c.H == H && c.K == K) return &c;
  return nullptr;
}
int tn_grid(int64_t N) {
  const int64_t nblocks = (N + GR_TS - 1) / GR_TS;
  int cap = GR_WG_MAX_WG;
  if (g_tzr_gemm_rows_wg > 0) cap = std::min(cap, g_tzr_gemm_rows_wg);
  // (a workgroup's partial sum costs H K floats written and read again: at least 8 blocks of rows each)
  return (int)std::max<int64_t>(1, std::min<int64_t>((nblocks + 7) / 8, cap));
}
}  // namespace

extern "C" int tzr_linear_rows_wgrad_supported(int H, int K) { return tn_cfg(H, K) ? 1 : 0; }

extern "C" size_t tzr_linear_rows_wgrad_workspace(int64_t N, int H, int K) {
  return (size_t)tn_grid(N) * (size_t)H * (size_t)K * sizeof(float) + 256;
}

template <int HBW, int KB, int WAVES>
static void tn_launch(hipStream_t s, int grid, const float* g, int64_t gs, const float* x, int64_t xs, int64_t N, float* parts) {
  constexpr int WPE = HBW * KB >= 24 ? 2 : 3;
  hipLaunchKernelGGL((tzr_gemm_tn_kernel<HBW, KB, WAVES, WPE>), dim3((unsigned)grid), dim3(WAVES * TZR_WAVE), 0, s, g, gs, x, xs, N, parts);
}

extern "C" int tzr_linear_rows_wgrad(const float* d_g, int64_t g_stride, const float* d_x, int64_t x_stride, int64_t N, int H, int K,
                                     float* d_dw, int64_t dw_stride, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!d_g || !d_x || !d_dw || N <= 0 || H <= 0 || K <= 0) return TZR_ERR_INVALID;
  if (!tn_cfg(H, K)) return TZR_ERR_UNSUPPORTED;
  if ((g_stride & 3) || (x_stride & 3) || g_stride < H || x_stride < K || dw_stride < K) return TZR_ERR_UNSUPPORTED;
  if (((uintptr_t)d_g | (uintptr_t)d_x) & 15) return TZR_ERR_UNSUPPORTED;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < tzr_linear_rows_wgrad_workspace(N, H, K) - 256) return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = tn_grid(N);
  float* parts = static_cast<float*>(ws);
#define TN_CASE(H_, K_, HBW_, KB_, W_)                                  \
  if (H == H_ && K == K_) {                                             \
    tn_launch<HBW_, KB_, W_>(s, grid, d_g, g_stride, d_x, x_stride, N, parts); \
  } else
  TN_CASE(64, 16, 1, 1, 4) TN_CASE(64, 32, 1, 2, 4) TN_CASE(128, 16, 2, 1, 4) TN_CASE(128, 32, 2, 2, 4) TN_CASE(256, 16, 4, 1, 4)
  TN_CASE(256, 32, 4, 2, 4) TN_CASE(64, 48, 1, 3, 4) TN_CASE(64, 64, 1, 4, 4) TN_CASE(64, 96, 1, 6, 4) TN_CASE(64, 128, 1, 8, 4) TN_CASE(64, 144, 1, 9, 4)
  TN_CASE(64, 192, 1, 12, 4) TN_CASE(64, 256, 1, 16, 4) TN_CASE(128, 48, 2, 3, 4) TN_CASE(128, 64, 2, 4, 4) TN_CASE(128, 96, 2, 6, 4)
  TN_CASE(128, 128, 2, 8, 4) TN_CASE(128, 144, 2, 9, 4) TN_CASE(128, 192, 2, 12, 4) TN_CASE(128, 256, 2, 16, 4) TN_CASE(256, 48, 4, 3, 4)
  TN_CASE(256, 64, 4, 4, 4) TN_CASE(256, 96, 4, 6, 4) TN_CASE(256, 128, 4, 8, 4) TN_CASE(256, 144, 4, 9, 4) { return TZR_ERR_UNSUPPORTED; }
#undef TN_CASE
  hipLaunchKernelGGL(tzr_gemm_tn_finish_kernel, dim3((unsigned)((H * K + 63) / 64)), dim3(GR_FIN_THREADS), 0, s, parts, grid, H, K, d_dw,
                     dw_stride, accumulate);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
