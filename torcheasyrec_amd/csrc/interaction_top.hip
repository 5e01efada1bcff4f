// K9b: the DLRM dot interaction FUSED with the first layer of the MLP behind it (gfx950).
//
// In DLRM.predict (/root/reference/tzrec/models/dlrm.py:123-135) the interaction output
//   z[b] = [ strict upper triangle of X_b X_b^T | dense row | sparse rows ]          (P + 16 n floats, 783 for Criteo)
// feeds `final_mlp`, whose first Linear is [P + 16 n -> 64].  Unfused, the 205 MB z row (B = 65 536) is written by the
// interaction forward, read by a GEMM, written again as dz by another GEMM and read by the interaction backward.  Here
//
//   tzr_dot_interaction_top_fwd:  y1 = act(z W1^T + b1) with the z tile produced in LDS (and written to HBM only when
//                                 the caller wants it: training keeps it for the weight gradient g1^T z);
//   tzr_dot_interaction_top_bwd:  dz tile = g1 tile [16 x 64] . W1 [64 x (P + 16 n)] scattered straight into the
//                                 per-sample symmetric S matrices / pass-through rows that the interaction backward
//                                 contracts with X (dX = (G + G^T) X + pass-through): dz never exists in HBM.
//
// Both are exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 6.6 GFLOP at B = 65 536 = 42 us at the 157 TFLOP/s peak) with W1
// (200 KB: more than the LDS) resident in REGISTERS for the whole launch: one persistent workgroup of 16 waves per CU, the
// columns of W1 regrouped into "virtual" 16-column blocks (pair blocks: pair index 16 v + r, padded; then one block per
// X row), every wave owning a slice of them.  A tile is 16 samples (the rows of one MFMA block), one sample per wave for
// the per-sample work.  Four waves per SIMD is the point of the shape: each wave's LDS / memory latencies are covered
// by the MFMAs of the other three (an 8-wave version with twice the work per wave ran at half the MFMA rate, and
// interleaving the instruction streams by hand inside a wave did not help).
#include "tzr_common.h"
#include <type_traits>
#include <tzr_gfx950.h>

#define IT_THREADS 1024
#define IT_WAVES (IT_THREADS / TZR_WAVE)
#define IT_TS 16                     // samples per tile = rows of one MFMA block = waves
#define IT_H 64                      // width of the layer behind the interaction
#define IT_KS (IT_H / 4)             // MFMA k-steps of the g1 . W1 product
#define IT_D 16
#define IT_MAXBLK 64                 // virtual column blocks: ceil(P / 16) + n <= 64, i.e. n <= 32
#define IT_BPW (IT_MAXBLK / IT_WAVES)  // backward: column blocks per wave (4)
#define IT_KB (IT_MAXBLK / 4)        // forward: blocks per K-group (16)
#define IT_MAX_BATCH (int64_t(1) << 30)  // samples per call: the kernels count tiles and samples in 32 bits
#define IT_N_CRITEO 27               // DLRM-Criteo: 26 tables + the dense vector (351 pairs: 22 + 27 = 49 column blocks)
#define IT_SROW 33                   // row pitch of one S matrix (odd: conflict-free column reads)
#define IT_SS (32 * IT_SROW + 4)     // floats per sample in S (1060: the four sample groups of an accumulator scatter land 16 banks apart)
#define IT_PS (32 * IT_D)            // floats per sample of pass-through gradients
#define IT_XS (32 * (IT_D + 1))      // floats of one X image
#define IT_ZP (16 * IT_MAXBLK + 4)   // floats per sample of the LDS z tile (virtual columns; +4: banks of the b128 A-operand reads)
#define IT_YP (IT_H + 1)

typedef float it_f32x4 __attribute__((ext_vector_type(4)));

// -DIT_PROF (scripts/bench_interaction_top.py --prof builds a second library with it): every wave sums the shader
// clocks it spends in each phase of the tile loop into a global table; not compiled into the product.
#ifdef IT_PROF
#define IT_PROF_DECL uint64_t it_tl = __builtin_amdgcn_s_memtime(), it_ts[6] = {0, 0, 0, 0, 0, 0}
#define IT_PROF_MARK(i) do { const uint64_t it_now = __builtin_amdgcn_s_memtime(); it_ts[i] += it_now - it_tl; it_tl = it_now; } while (0)
#define IT_PROF_DUMP(tab) do { if (tab && lane == 0) for (int i_ = 0; i_ < 6; ++i_) (tab)[((size_t)blockIdx.x * IT_WAVES + wv) * 6 + i_] = it_ts[i_]; } while (0)
extern "C" uint64_t* g_tzr_it_prof;
uint64_t* g_tzr_it_prof = nullptr;
extern "C" void tzr_it_prof_table(uint64_t* d_table) { g_tzr_it_prof = d_table; }
#else
#define IT_PROF_DECL
#define IT_PROF_MARK(i)
#define IT_PROF_DUMP(tab)
#endif

struct __attribute__((packed, aligned(4))) it_f4u {
  float x, y, z, w;
};
__device__ __forceinline__ void it_st4_a4(float* p, float4 v) {  // 16-byte store to a 4-byte aligned address
  it_f4u u;
  u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
  *reinterpret_cast<it_f4u*>(p) = u;
}

__device__ __forceinline__ const float* it_row(const float* dense, int64_t dense_stride, const float* sparse,
                                               int64_t sparse_stride, int64_t b, int i, int hd) {
  return (hd && i == 0) ? dense + b * dense_stride : sparse + b * sparse_stride + (int64_t)(i - hd) * IT_D;
}

// X rows min(r, n - 1) / min(16 + r, n - 1), columns 4 q .. 4 q + 3, of sample b (clamped into the batch: a sample behind
// it reads the last one, nothing of it is stored)
struct ItX {
  float4 lo, hi;
};
__device__ __forceinline__ ItX it_fetch_x(const float* dense, int64_t dense_stride, const float* sparse, int64_t sparse_stride,
                                          int64_t b, int64_t B, int n, int hd, int r, int q) {
  const int rr0 = r < n ? r : n - 1, rr1 = 16 + r < n ? 16 + r : n - 1;
  b = b < B ? b : B - 1;
  ItX x;
  x.lo = tzr_ld4(it_row(dense, dense_stride, sparse, sparse_stride, b, rr0, hd) + 4 * q);
  x.hi = tzr_ld4(it_row(dense, dense_stride, sparse, sparse_stride, b, rr1, hd) + 4 * q);
  return x;
}

// virtual column c (pairs padded to a multiple of 16, then the X rows) -> column of z / W1, or -1 for a pad slot
__device__ __forceinline__ int it_real_col(int c, int P, int npb, int nblk) {
  if (c < P) return c;
  if (c < 16 * npb || c >= 16 * nblk) return -1;
  return c - (16 * npb - P);
}

// ---- backward ------------------------------------------------------------------------------------------------------
// wave w owns column blocks w, w + 16, ... (at most 4: 4 x 16 B-operand registers per lane).  Per tile: every wave
// multiplies the tile's g1 fragment with its blocks, scatters the accumulators into LDS, and behind a barrier contracts
// ONE sample of the tile the way tzr_dot_interaction_bwd_kernel does.  HBM traffic: X in, dX out, g1 in (243 MB at
// B = 65 536) instead of 243 + 2 x 205 MB.
struct ItBwdArgs {
  const float *dense, *sparse, *g1, *W1, *scale;
  float *gdense, *gsparse;
  int64_t dense_stride, sparse_stride, g1_stride, ldw, gdense_stride, gsparse_stride, B;
  int n, hd, stagger;
  uint64_t* prof;
};

#define IT_GP (IT_H + 4)              // row pitch of the g1 tile in LDS (68: the b128 A-operand reads spread over the banks)

// NB column blocks per wave with their W1 fragments in registers; XL: wave 0 owns one more (block 16 NB: the 49th of
// DLRM-Criteo) whose fragment lives in LDS -- a fourth block in registers is 64 of the 128 a lane has and spills.
//
// The dz tile lies in LDS exactly as a GEMM would write it -- [16 samples][virtual columns], pairs first -- and the
// contraction picks G[i][j] out of it by pair index (S = G + G^T is never materialised: one LDS write per element, no offset
// table).  Two such tiles (when they fit: 16 nblk + 4 <= IT_ZPD) make ONE barrier per tile enough: the product of tile
// t + G goes into the other buffer while the samples of tile t are still being contracted.
#define IT_ZPD 884  // pitch up to which two dz tiles fit beside the X images (n <= 29)

#define IT_TP 34  // row pitch of the pair-offset table (uint16)

template <int NB, bool XL>
__device__ __forceinline__ void it_bwd_loop(const ItBwdArgs& a, float* __restrict__ Z, float* __restrict__ xs, float* __restrict__ Gs,
                                            float* __restrict__ Wx, uint16_t* __restrict__ Tab, int lane, int wv) {
  const int r = lane & 15, q = lane >> 4;
  // (XL is DLRM-Criteo's 49 blocks, i.e. 27 vectors: the row pitch, the block bounds and the lane masks are constants there)
  const int n = XL ? IT_N_CRITEO : a.n;
  const int P = n * (n - 1) / 2;
  const int npb = (P + 15) >> 4, nblk = npb + n;
  const int zp = 16 * nblk + 4;           // pitch of a sample's dz row
  const bool dbl = zp <= IT_ZPD;          // two dz tiles fit
  const int zt = IT_TS * zp;              // floats per tile
  // ---- W1 fragments (B operand: lane supplies W[k = 16 q + ks][column of (block, r)]); blocks behind the wave's last stay
  // zero.  Staged through LDS in slabs of 16 blocks x 64 rows so the global loads are coalesced runs: lanes asking for
  // their own elements straight from L2 (4 to 16 lines per load, every CU the same lines at once) took 11 - 24 us.
  float Wf[NB][IT_KS];
  {
    const float sc = a.scale ? *a.scale : 1.f;
    float* Wl = Z;  // [64 rows][256 + 1]: the slab
    const int c = threadIdx.x & 255, k0 = threadIdx.x >> 8;  // thread -> slab column, rows k0, k0 + 4, ...
    // (every wave walks all slabs: the loads are the workgroup's; the next slab's loads fly while this one is handed out)
    float w[16];
    int col = it_real_col(c, P, npb, nblk);
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = a.W1[(int64_t)(k0 + 4 * i) * a.ldw + (col >= 0 ? col : 0)];
#pragma unroll
    for (int m = 0; m < IT_BPW; ++m) {
      float wn[16];
      const int coln = m + 1 < IT_BPW ? it_real_col(256 * (m + 1) + c, P, npb, nblk) : -1;
      if (m + 1 < IT_BPW) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wn[i] = a.W1[(int64_t)(k0 + 4 * i) * a.ldw + (coln >= 0 ? coln : 0)];
      }
      __syncthreads();  // the previous slab has been read
#pragma unroll
      for (int i = 0; i < 16; ++i) Wl[(k0 + 4 * i) * 257 + c] = col >= 0 ? sc * w[i] : 0.f;
      __syncthreads();
      if (m < NB) {
#pragma unroll
        for (int ks = 0; ks < IT_KS; ++ks) Wf[m][ks] = Wl[(16 * q + ks) * 257 + 16 * wv + r];
      } else if (XL && m == NB && wv == 0) {
#pragma unroll
        for (int ks = 0; ks < IT_KS; ++ks) Wx[ks * TZR_WAVE + lane] = Wl[(16 * q + ks) * 257 + r];
      }
      if (m + 1 < IT_BPW) {
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = wn[i];
        col = coln;
      }
    }
    __syncthreads();
  }
  // ---- where pair (k, c) lies in a sample's dz row (byte offset), or the zero slot: the first pad float behind the row
  {
    const int e = threadIdx.x, k = e >> 5, c = e & 31;
    const int i = k < c ? k : c, j = k < c ? c : k;
    const bool ok = i != j && j < n;
    Tab[k * IT_TP + c] = (uint16_t)(4 * (ok ? (i * (2 * n - 1 - i)) / 2 + j - i - 1 : 16 * nblk));
    if (e < (dbl ? 2 : 1) * IT_TS) Z[(e >> 4) * zt + (e & 15) * zp + 16 * nblk] = 0.f;  // the zero slot of every sample of the tile(s) (the W1 slabs lay here)
  }
  __syncthreads();
  const uint16_t* tabl = Tab + q * IT_TP + r;   // lane constants of the contraction (k = 4 ks + q, c = 16 h + r)
  const int xoff = q * (IT_D + 1) + r;
  // (tile and sample counters in 32 bits -- the launcher checks B: their compares are scalar instructions then, 64-bit
  // ones are VALU work; the sample offsets are widened where they are multiplied with a stride)
  const int Bn = (int)a.B;
  const int ntiles = (Bn + IT_TS - 1) / IT_TS;
  const int G = gridDim.x;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // the g1 tile goes through LDS (one element per thread, double-buffered): every wave needs all of it as its A operand,
  // and sixteen copies from L2 would cost more than the HBM traffic of the whole kernel
  const int gs = wv, gh = lane;  // (the sample is the wave's: a scalar row base, the lane is the column)
  auto g1_elem = [&](int tt) {
    const int row = tt * IT_TS + gs;
    const bool ok = tt < ntiles && row < Bn;
    const float v = (a.g1 + (int64_t)(ok ? row : 0) * a.g1_stride)[gh];
    return ok ? v : 0.f;
  };
  // dz[:, blocks of this wave] of the g1 tile in Gs[gbuf] -> dz tile zbuf.  A operand: lane (i = r, q) reads g1[i][16 q ..
  // 16 q + 15], k-step ks uses element 16 q + ks (the contraction order is free as long as the W1 fragment agrees);
  // accumulator reg j of lane (r, q) = dz[sample 4 q + j][column r of the block]
  auto product = [&](int gbuf, int zbuf) {
    const float* gp = Gs + gbuf * (IT_TS * IT_GP) + r * IT_GP + 16 * q;
    float* zo = Z + zbuf * zt + (4 * q) * zp + r;
    it_f32x4 acc[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = it_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const float4 av = tzr_ld4(gp + 4 * k4);
      const float ag[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[e], Wf[m][4 * k4 + e], acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < NB; ++m)
      if (wv + IT_WAVES * m < nblk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) zo[j * zp + 16 * (wv + IT_WAVES * m)] = acc[m][j];
      }
    if (XL && wv == 0) {
      it_f32x4 accx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const float4 av = tzr_ld4(gp + 4 * k4);
        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, Wx[(4 * k4 + 0) * TZR_WAVE + lane], accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, Wx[(4 * k4 + 1) * TZR_WAVE + lane], accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, Wx[(4 * k4 + 2) * TZR_WAVE + lane], accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, Wx[(4 * k4 + 3) * TZR_WAVE + lane], accx, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // (four fragment reads at a time: sixteen at once are sixteen registers)
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) zo[j * zp + 16 * (IT_WAVES * NB)] = accx[j];
    }
  };
  Gs[gs * IT_GP + gh] = g1_elem(t);
  Gs[IT_TS * IT_GP + gs * IT_GP + gh] = g1_elem(t + G);
  // ---- the wave's sample of a tile.  General shapes: X rows in the layout of the loads (16 bytes of a row per lane), turned
  // into the contraction's operand layout through an image in LDS.  XL (27 vectors): loaded in the operand layout in the
  // first place -- lane (r, q), k-step ks: X[row 4 ks + q][column r] = element 64 ks + lane of the sample's rows, seven
  // coalesced dword loads -- and rows 28 .. 31 (k-step 7) are not multiplied at all.  No image, no LDS round trip.
  ItX X;
  float xv[XL ? 7 : 1];
  const unsigned l6 = q == 3 ? lane - IT_D : lane;  // (k-step 6 of q = 3 is row 27: it repeats row 26 and is zeroed at its use)
  // XL: the operand of k-step ks of the wave's sample of tile tt
  auto fetch_k = [&](int tt, int ks) {
    const int bi = tt * IT_TS + wv;
    const int64_t b = bi < Bn ? bi : Bn - 1;
    const float* sp = a.sparse + b * a.sparse_stride - IT_D * a.hd;  // row k >= hd of X at sp + 16 k
    if (ks == 0) {  // (two masked loads off scalar bases, not one load through a selected 64-bit pointer: no VALU)
      float x;
      if (a.hd && q == 0) x = (a.dense + b * a.dense_stride)[(unsigned)lane];
      else x = sp[(unsigned)lane];
      return x;
    }
    return (sp + 64 * ks)[ks == 6 ? l6 : (unsigned)lane];  // (base + constant, then the lane: an immediate offset of the load)
  };
  auto fetch = [&](int tt) {
    if (XL) {
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) xv[XL ? ks : 0] = fetch_k(tt, ks);
    } else {
      X = it_fetch_x(a.dense, a.dense_stride, a.sparse, a.sparse_stride, (int64_t)tt * IT_TS + wv, a.B, n, a.hd, r, q);
    }
  };
  // XL: where S[4 ks + q][r] and S[4 ks + q][16 + r] lie -- a table of this WAVE's own (in the space of the X image it does
  // not write): byte offsets from Z of ITS row of dz buffer 0 (< 2^16), so that an entry is the address of the read as it
  // stands -- the other buffer, and Z itself, are constants of the instruction; the shared table's row offsets want an add
  // per element (14 VALU instructions per tile and wave)
  uint16_t* const tw = reinterpret_cast<uint16_t*>(xs) + lane;  // [2 ks + h][64 lanes]
  if (XL) {
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 4 * ks + q, c = 16 * h + r;
        const int i = k < c ? k : c, jj = k < c ? c : k;
        const bool ok = i != jj && jj < n;
        tw[(2 * ks + h) * TZR_WAVE] = (uint16_t)(4 * (wv * zp + (ok ? (i * (2 * n - 1 - i)) / 2 + jj - i - 1 : 16 * nblk)));
      }
  }
  fetch(t);
  float gnext = g1_elem(t + 2 * G);
  __syncthreads();
  product(0, 0);
  // Half of the waves (two of the four on every SIMD) run the product of the NEXT tile before they contract their sample
  // of this one, the other half behind it: in lockstep all sixteen would sit in the LDS-latency-bound contraction at
  // once with the MFMA pipe mostly idle, then all in the product.
  const bool product_first_wave = a.stagger == 2 ? false : (((wv >> 2) & 1) != (a.stagger == 1));
  const bool v0 = r < n, v1 = 16 + r < n;
  IT_PROF_DECL;
  // tile t, its dz and g1 tiles in buffer CUR (XL: a constant of the code -- the loop is unrolled by two, and there is one loop
  // per order of a wave's turn: the buffers are immediate offsets, and nothing changes registers between turns)
  auto tile = [&](auto CUR, auto PF) {
    const int cur = CUR;
    const bool product_first = PF;
    tzr_lds_barrier();  // the dz tile of t is complete; every wave is done with tile t - G (its dz tile, its g1 tile)
    IT_PROF_MARK(0);  // wait
    const bool more = t + G < ntiles;
    const int zcur = dbl ? cur : 0;
    // ---- general shapes: the X image of this wave's sample; then the loads of the tiles ahead take off (the sample of t + G
    // into the registers just read, the g1 element of t + 2 G).  Order matters: hipcc waits for a loop-carried load with
    // s_waitcnt vmcnt(0), i.e. for EVERYTHING in flight -- nothing may be issued shortly before such a wait (profiles/r03ak).
    if (!XL) {
      const float4 x0 = v0 ? X.lo : tzr_zero4(), x1 = v1 ? X.hi : tzr_zero4();
      float* p0 = xs + r * (IT_D + 1) + 4 * q;
      float* p1 = xs + (16 + r) * (IT_D + 1) + 4 * q;
      p0[0] = x0.x; p0[1] = x0.y; p0[2] = x0.z; p0[3] = x0.w;
      p1[0] = x1.x; p1[1] = x1.y; p1[2] = x1.z; p1[3] = x1.w;
    }
    if (!XL) fetch(more ? t + G : t);  // (XL: every operand register is refilled right behind the MFMAs that read it)
    __builtin_amdgcn_wave_barrier();  // the X image is private to this wave
    IT_PROF_MARK(1);  // X image, prefetch issue
    if (product_first && dbl && more) product(cur ^ 1, cur ^ 1);
    IT_PROF_MARK(2);  // product (first half of the waves)
    // ---- dX = (G + G^T) X + pass-through of sample wv of the tile (transposed product, see interaction.hip): S[k][i] is
    // the dz element of pair (min, max) of (k, i), zero on the diagonal and beyond n.  Where that element lies in the dz row
    // comes from a table (byte offsets; the invalid pairs point at a pad float that stays zero): worked out per element it
    // was ~15 VALU instructions x 16 elements per tile and wave -- and VALU instructions share the fp32 MFMAs' pipe
    // (283 -> ~60 VALU instructions per tile and wave, profiles/r04al).  The table is in LDS, XL's in registers.
    const char* zs = reinterpret_cast<const char*>(Z + zcur * zt + wv * zp);
    it_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
    if (XL) {
      const char* zb = reinterpret_cast<const char*>(Z + cur * zt);
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        float x = xv[XL ? ks : 0];
        if (ks == 6) x = q == 3 ? 0.f : x;
        const unsigned o0 = tw[(2 * ks) * TZR_WAVE], o1 = tw[(2 * ks + 1) * TZR_WAVE];
        const float s0 = *reinterpret_cast<const float*>(zb + o0), s1 = *reinterpret_cast<const float*>(zb + o1);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, s0, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, s1, d1, 0, 0, 0);
        xv[XL ? ks : 0] = fetch_k(more ? t + G : t, ks);
      }
    } else {
      // (all eight k-steps, no branch on n: rows of the X image beyond n are zero and pairs beyond n read the zero slot)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float x = xs[(4 * ks) * (IT_D + 1) + xoff];
        const unsigned o0 = tabl[(4 * ks) * IT_TP], o1 = tabl[(4 * ks) * IT_TP + 16];
        const float s0 = *reinterpret_cast<const float*>(zs + o0), s1 = *reinterpret_cast<const float*>(zs + o1);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, s0, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, s1, d1, 0, 0, 0);
      }
    }
    IT_PROF_MARK(3);  // contraction
    const int bi = t * IT_TS + wv;
    const int64_t b = bi;
    int rq = r * IT_D + 4 * q;
    TZR_OPAQUE(rq);  // (recomputed per tile, not hoisted and spilled)
    const float* pt = reinterpret_cast<const float*>(zs) + 16 * npb;
    if (bi < Bn) {
      // (b is wave-uniform: scalar row bases, 32-bit lane offsets)
      float* const gsrow = a.gsparse + b * a.gsparse_stride - IT_D * a.hd;  // row k >= hd of the gradient at gsrow + 16 k
      if (v0) {
        const float4 p = tzr_ld4(pt + rq);
        const float4 v = make_float4(d0[0] + p.x, d0[1] + p.y, d0[2] + p.z, d0[3] + p.w);
        if (a.hd && r == 0) tzr_st4(a.gdense + b * a.gdense_stride + (unsigned)rq, v);
        else tzr_st4(gsrow + (unsigned)rq, v);
      }
      if (v1) {
        const float4 p = tzr_ld4(pt + 16 * IT_D + rq);
        const float4 v = make_float4(d1[0] + p.x, d1[1] + p.y, d1[2] + p.z, d1[3] + p.w);
        tzr_st4(gsrow + IT_D * 16 + (unsigned)rq, v);
      }
    }
    IT_PROF_MARK(4);  // pass-through + stores
    if (!dbl) tzr_lds_barrier();  // one dz tile only: every wave must be done with it before the next product lands
    if (more && !(dbl && product_first)) product(cur ^ 1, dbl ? cur ^ 1 : 0);
    // the g1 tile of t + 2 G takes the buffer the product of tile t read (a tile ago: every wave is past it); its element
    // was loaded a whole turn ago, the one of t + 3 G takes off now -- loaded at the top of the turn it was waited for in the
    // middle of the contraction (hipcc's counts leave only the newest loads in flight: profiles/r05bi)
    Gs[cur * (IT_TS * IT_GP) + gs * IT_GP + gh] = gnext;
    gnext = g1_elem(t + 3 * G);
    IT_PROF_MARK(5);  // product (second half), g1 hand-over
  };
  auto run = [&](auto PF) {
    for (;;) {
      tile(std::integral_constant<int, 0>(), PF);
      t += G;
      if (t >= ntiles) break;
      tile(std::integral_constant<int, 1>(), PF);
      t += G;
      if (t >= ntiles) break;
    }
  };
  if (XL) {
    if (product_first_wave) run(std::true_type());
    else run(std::false_type());
  } else {  // (the general shapes keep one loop: twice unrolled, and once per order, they spill)
    for (int cur = 0; t < ntiles; t += G, cur ^= 1) tile(cur, product_first_wave);
  }
  IT_PROF_DUMP(a.prof);
}

__global__ __launch_bounds__(IT_THREADS) void tzr_ia_top_bwd_kernel(ItBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float Z[2 * IT_TS * IT_ZPD];  // the dz tile(s); 2 x 16 x 884 >= 16 x 1028
  __shared__ float Xs[IT_WAVES][IT_XS];     // per wave: the X image of its sample
  __shared__ __attribute__((aligned(16))) float Gs[2 * IT_TS * IT_GP];   // the g1 tile, double-buffered
  __shared__ float Wx[IT_KS * TZR_WAVE];    // W1 fragment of wave 0's extra block (it_bwd_loop<NB, true>)
  __shared__ uint16_t Tab[32 * IT_TP];      // byte offset of pair (k, c) in a sample's dz row
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TZR_WAVE));  // (scalar: per-wave bases stay in SGPRs)
  const int n = a.n;
  const int P = n * (n - 1) / 2;
  const int npb = (P + 15) >> 4, nblk = npb + n;
  // DLRM-Criteo: 49 blocks = 3 per wave and one more for wave 0 (fragment in LDS); up to 48: 3 per wave, zero-weighted;
  // more: 4 per wave in registers (spills, correct)
  if (nblk <= 3 * IT_WAVES) it_bwd_loop<3, false>(a, Z, &Xs[wv][0], Gs, Wx, Tab, lane, wv);
  else if (n == IT_N_CRITEO) it_bwd_loop<3, true>(a, Z, &Xs[wv][0], Gs, Wx, Tab, lane, wv);  // (the one n with 3 x 16 + 1 blocks)
  else it_bwd_loop<4, false>(a, Z, &Xs[wv][0], Gs, Wx, Tab, lane, wv);
}

// ---- forward -------------------------------------------------------------------------------------------------------
// The virtual column blocks are the CONTRACTION index here.  Wave w = (K-group kg = w >> 2, output block hb = w & 3):
// it multiplies the z columns of the blocks of its K-group (a quarter of them: 13 / 12 / 12 / 12 for Criteo) with the
// matching W1 columns of its 16 outputs (B operand W1[h = 16 hb + r][column 4 q + kk of block v]: 4 registers per block;
// A operand: one ds_read_b128 per block) into a partial [16 samples x 16 outputs]; the four partials of an output are
// summed through LDS in fixed order (deterministic), bias and ReLU applied, y1 written as one contiguous 4 KB run per
// tile.  The z tile is double-buffered: the row of tile t + G (pairwise products by MFMA, LDS scatter, the stores of z to
// HBM) is produced right behind the product of tile t, with one barrier pair per tile.
struct ItFwdArgs {
  const float *dense, *sparse, *W1, *bias;
  float *z, *y1;
  int64_t dense_stride, sparse_stride, ldw, z_stride, y1_stride, B;
  int n, hd, relu, stagger;
  uint64_t* prof;
};

// one sample's row of the z tile: pairwise products by MFMA scattered into `zs`, X rows behind them; `o` != null: the
// same row out to HBM.  Lanes without a valid target write to a per-lane trash slot or repeat a neighbour's store (same
// address, same value) instead of branching.
__device__ __forceinline__ void it_fwd_row(const ItX X, float* __restrict__ zs, float* __restrict__ trash, float* __restrict__ o,
                                           int n, int P, int pt0, int lane) {
  TZR_OPAQUE(lane);  // (the lane arithmetic below is redone per row: hoisted out of the tile loop it is a dozen registers, spilled)
  const int r = lane & 15, q = lane >> 4;
  const bool v0 = r < n, v1 = 16 + r < n;
  const int rr0 = v0 ? r : n - 1, rr1 = v1 ? 16 + r : n - 1;
  const float4 a0 = v0 ? X.lo : tzr_zero4(), a1 = v1 ? X.hi : tzr_zero4();
  it_f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c11 = c00;
  const float x0[4] = {a0.x, a0.y, a0.z, a0.w};
  const float x1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], x0[e], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], x1[e], c01, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], x1[e], c11, 0, 0, 0);
  }
  // strict upper triangle, row-major (i < j): idx(i, j) = i (2 n - i - 1) / 2 + j - i - 1; from row i to row i + 1 the
  // offset of column j moves by n - i - 2
  int t0 = (4 * q) * (2 * n - 4 * q - 1) / 2 - 4 * q - 1, t1 = (16 + 4 * q) * (2 * n - 4 * q - 17) / 2 - 4 * q - 17;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int i0 = 4 * q + reg, i1 = 16 + i0;
    const int j0 = r, j1 = 16 + r;
    *((i0 < j0 && j0 < n) ? zs + t0 + j0 : trash) = c00[reg];
    *((j1 < n) ? zs + t0 + j1 : trash) = c01[reg];
    *((i1 < j1 && j1 < n) ? zs + t1 + j1 : trash) = c11[reg];
    t0 += n - i0 - 2;
    t1 += n - i1 - 2;
  }
  tzr_st4(zs + pt0 + IT_D * rr0 + 4 * q, X.lo);  // (lanes of a row >= n repeat row n - 1: same address, same value)
  tzr_st4(zs + pt0 + IT_D * rr1 + 4 * q, X.hi);
  if (o) {
    __builtin_amdgcn_wave_barrier();  // the row was written by this wave only
    it_st4_a4(o + P + IT_D * rr0 + 4 * q, X.lo);
    it_st4_a4(o + P + IT_D * rr1 + 4 * q, X.hi);
  }
}

// ... and the pairs of that row out to HBM (apart from it_fwd_row so that the caller can put the next prefetch between):
// 16 bytes per lane (the row starts 4-byte aligned only: unaligned dwordx4 stores), the last P % 4 elements one by one
__device__ __forceinline__ void it_fwd_row_pairs_out(const float* __restrict__ zs, float* __restrict__ o, int P, int lane) {
  if (o) {
    const int n4 = P >> 2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // P <= 496: at most 124 groups of four
      int i4 = lane + TZR_WAVE * k;
      i4 = i4 < n4 ? i4 : n4 - 1;  // (a lane behind the end repeats the last group's store)
      if (n4 > 0) it_st4_a4(o + 4 * i4, tzr_ld4(zs + 4 * i4));
    }
    const int rest = 4 * n4 + (lane & 3);
    if (rest < P && lane < 4) o[rest] = zs[rest];
  }
}

// W1 fragments [block of the group][kk] for a wave's 16 outputs (pad columns and blocks behind the group are zero) and for
// k-step kg of the left-over blocks.  Staged through LDS (the z tiles' space, [32 rows][ZP]) in two halves of 32 rows so
// the global loads are coalesced runs; leaves both z tiles zeroed (the pad slots behind the last pair stay zero).
template <int NB, int REM, int ZP>
__device__ __forceinline__ void it_fwd_stage_w(const ItFwdArgs& a, float* __restrict__ Zs, float (&Wf)[NB][4], float (&Wx)[REM > 0 ? REM : 1],
                                               int n, int P, int npb, int base, int rem, int vfirst, int vlast, int r, int q,
                                               int kg, int hb) {
  const int nblk = npb + n;
  float* Wl = Zs;
  const int c = threadIdx.x;  // thread -> virtual column (16 nblk <= 1024 of them)
  const int col = c < 16 * nblk ? it_real_col(c, P, npb, nblk) : -1;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    float w[2][16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) w[g][i] = a.W1[(int64_t)(32 * half + 16 * g + i) * a.ldw + (col >= 0 ? col : 0)];
    __syncthreads();  // the previous half has been read
    if (c < ZP) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) Wl[(16 * g + i) * ZP + c] = col >= 0 ? w[g][i] : 0.f;
    }
    __syncthreads();
    if ((hb >> 1) == half) {
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        const int v = vfirst + m;
        const float4 w = tzr_ld4(Wl + (16 * (hb & 1) + r) * ZP + 16 * (v < vlast ? v : vfirst) + 4 * q);
        const float keep = v < vlast ? 1.f : 0.f;
        Wf[m][0] = keep * w.x; Wf[m][1] = keep * w.y; Wf[m][2] = keep * w.z; Wf[m][3] = keep * w.w;
      }
#pragma unroll
      for (int j = 0; j < REM; ++j)
        Wx[j] = j < rem ? Wl[(16 * (hb & 1) + r) * ZP + 16 * (4 * base + j) + 4 * q + kg] : 0.f;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * IT_TS * ZP; k += IT_THREADS) Zs[k] = 0.f;
  __syncthreads();
}

// NB whole blocks [vfirst, vfirst + base) per wave (`base` of them real, the rest zero weights) and one k-step of each of
// the REM left-over blocks 4 base + j (k-step kg of it): every wave runs the same 4 NB + REM MFMAs per tile.
// NFIX > 0: the number of vectors is this constant (the pair offsets of the scatter, the clamps of the row loads and the
// block counts fold into immediates: registers and VALU work of the row builder)
template <int NB, int REM, bool ZOUT, int NFIX>
__device__ __forceinline__ void it_fwd_loop(const ItFwdArgs& a, float* __restrict__ Zs, float* __restrict__ Ys,
                                            float* __restrict__ trash, int lane, int wv) {
  // pitch of a sample's row in the z tile: the blocks of THIS n when it is a constant (LDS left for a second Ys)
  constexpr int ZP = NFIX > 0 ? 16 * ((NFIX * (NFIX - 1) / 2 + 15) / 16 + NFIX) + 4 : IT_ZP;
  const int r = lane & 15, q = lane >> 4;
  const int kg = wv >> 2, hb = wv & 3;
  const int n = NFIX > 0 ? NFIX : a.n;
  const int P = n * (n - 1) / 2;
  const int npb = (P + 15) >> 4;
  // K-group kg: whole blocks [kg base, kg base + base), base = nblk / 4, and one k-step of each of the nblk % 4 blocks left
  const int base = (npb + n) >> 2, rem = (npb + n) & 3;
  const int pt0 = 16 * npb;  // virtual column of X row 0
  const int vfirst = kg * base, vlast = vfirst + base;
  float Wf[NB][4];
  float Wx[REM > 0 ? REM : 1];  // ... and for k-step kg of the left-over blocks
  it_fwd_stage_w<NB, REM, ZP>(a, Zs, Wf, Wx, n, P, npb, base, rem, vfirst, vlast, r, q, kg, hb);
  const int64_t ntiles = (a.B + IT_TS - 1) / IT_TS;
  const int64_t G = gridDim.x;
  int64_t t = blockIdx.x;
  if (t >= ntiles) return;
  {
    const int64_t b = t * IT_TS + wv;
    const ItX X = it_fetch_x(a.dense, a.dense_stride, a.sparse, a.sparse_stride, b, a.B, n, a.hd, r, q);
    float* o = (ZOUT && a.z && b < a.B) ? a.z + b * a.z_stride : nullptr;
    it_fwd_row(X, Zs + wv * ZP, trash, o, n, P, pt0, lane);
    it_fwd_row_pairs_out(Zs + wv * ZP, o, P, lane);
  }
  ItX X = it_fetch_x(a.dense, a.dense_stride, a.sparse, a.sparse_stride, (t + G < ntiles ? t + G : t) * IT_TS + wv, a.B, n, a.hd, r, q);
  int cur = 0;
  // (the bias of this thread's two outputs, read ONCE: a load inside the loop is waited for with vmcnt(0), prefetches and all)
  float bias0 = a.bias ? a.bias[2 * (lane & 31)] : 0.f, bias1 = a.bias ? a.bias[2 * (lane & 31) + 1] : 0.f;
  TZR_OPAQUE(bias0);  // (forces the wait for these loads HERE, not -- as vmcnt(0) -- at their first use inside the loop)
  TZR_OPAQUE(bias1);
  IT_PROF_DECL;
  for (; t < ntiles; t += G, cur ^= 1) {
    const float* zb = Zs + cur * (IT_TS * ZP);
    tzr_lds_barrier();  // tile t complete in zb; the other buffer and Ys free
    IT_PROF_MARK(0);  // wait 1
    // ---- partial y1[:, 16 hb ..] over the z columns of this wave's K-group
    // Half of the waves (two of the four on every SIMD) produce the next tile's row BEFORE the product, the other half
    // behind it: in lockstep all sixteen would be in their LDS / store phase at once and the MFMA pipe would idle.
    const bool half = (wv >> 2) & 1;
    // (1 = every wave builds its row first, 2 = every wave behind the product; z not written: always behind it)
    const bool row_first = ZOUT && (a.stagger == 0 ? half : a.stagger == 1);
    auto next_row = [&]() {
    // ---- the row of sample wv of tile t + G into the other buffer (and out to HBM); then tile t + 2 G's X rows take off
      if (t + G < ntiles) {
        const int64_t b = (t + G) * IT_TS + wv;
        float* zs = Zs + (cur ^ 1) * (IT_TS * ZP) + wv * ZP;
        if (!ZOUT) {
          // every wave is behind the product here: the rows fetched now are used a product and two barriers later, so
          // they take off BEHIND the row and need no second set of registers (which spilled W1 fragments)
          it_fwd_row(X, zs, trash, nullptr, n, P, pt0, lane);
          X = it_fetch_x(a.dense, a.dense_stride, a.sparse, a.sparse_stride, (t + 2 * G < ntiles ? t + 2 * G : t) * IT_TS + wv,
                         a.B, n, a.hd, r, q);
          return;
        }
        float* o = (a.z && b < a.B) ? a.z + b * a.z_stride : nullptr;
        // the X rows of the tile after that take off FIRST (into a second set of registers): a row-first wave is back
        // here ~3 k clocks after its last row, less than an HBM round trip under this load -- it stood 5 k clocks per tile
        // waiting for rows fetched at the end of the previous row (profiles/r03an)
        const ItX Xn = it_fetch_x(a.dense, a.dense_stride, a.sparse, a.sparse_stride, (t + 2 * G < ntiles ? t + 2 * G : t) * IT_TS + wv,
                                  a.B, n, a.hd, r, q);
        it_fwd_row(X, zs, trash, o, n, P, pt0, lane);
        it_fwd_row_pairs_out(zs, o, P, lane);
        X = Xn;
      }
    };
    if (row_first) next_row();
    IT_PROF_MARK(1);  // row (first half of the waves)
    it_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
      // A operand of block m, k-steps 0..3 = columns 4 q + kk: one ds_read_b128, issued two blocks ahead of its MFMAs and
      // pinned there (left alone, hipcc hoists all NB reads to the top: 52 registers, and spills W1 fragments)
      const float* zr = zb + r * ZP + 4 * q;
      auto rd = [&](int m) {
        const int v = vfirst + m;
        return tzr_ld4(zr + 16 * (v < vlast ? v : vfirst));  // (a block behind the group: zero weights on valid data)
      };
      constexpr int RING = ZOUT ? 3 : 2;  // (z not written: one block ahead; the second costs W1 fragments their registers)
      float4 av[RING];
      av[0] = rd(0);
      if (RING > 2) av[1] = rd(1 < NB ? 1 : 0);
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (m + RING - 1 < NB) av[(m + RING - 1) % RING] = rd(m + RING - 1);
        const float4 c = av[m % RING];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, Wf[m][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, Wf[m][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, Wf[m][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, Wf[m][3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < REM; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zr[16 * (4 * base + (j < rem ? j : 0)) + kg], Wx[j], acc, 0, 0, 0);
    }
    // accumulator reg j of lane (r, q) = partial y1[sample 4 q + j][h = 16 hb + r]
#pragma unroll
    for (int j = 0; j < 4; ++j) Ys[kg * (IT_TS * IT_YP) + (4 * q + j) * IT_YP + 16 * hb + r] = acc[j];
    IT_PROF_MARK(2);  // product
    if (!row_first) next_row();
    IT_PROF_MARK(3);  // row (second half)
    tzr_lds_barrier();
    IT_PROF_MARK(4);  // wait 2
    // ---- sum of the four partials in K-group order, bias, activation -- by the waves that produced their row BEHIND the
    // product only (thread -> sample, two outputs): hipcc waits for the prefetched X rows with s_waitcnt vmcnt(0), and a
    // y1 store issued here would be in flight when a row-first wave reaches that wait right behind the barrier (it stood
    // there 5 k clocks per tile, profiles/r03ak)
    if (!half) {
      const int t8 = ((wv >> 3) << 2) | (wv & 3);  // 0..7 among the row-second waves
      const int s = 2 * t8 + (lane >> 5), h = 2 * (lane & 31);
      float v0 = bias0, v1 = bias1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v0 += Ys[g * (IT_TS * IT_YP) + s * IT_YP + h];
        v1 += Ys[g * (IT_TS * IT_YP) + s * IT_YP + h + 1];
      }
      if (a.relu) {
        v0 = v0 > 0.f ? v0 : 0.f;
        v1 = v1 > 0.f ? v1 : 0.f;
      }
      const int64_t b = t * IT_TS + s;
      if (b < a.B) {
        a.y1[b * a.y1_stride + h] = v0;
        a.y1[b * a.y1_stride + h + 1] = v1;
      }
    }
    IT_PROF_MARK(5);  // sum of the partials
  }
  IT_PROF_DUMP(a.prof);
}

// ---- DLRM-Criteo's forward of the training step: 27 vectors, z not written -------------------------------------------
// Same tiles, same MFMAs, same order of every sum as it_fwd_loop<12, 1, ., 27>; what differs is what the tile loop does NOT
// do -- VALU instructions share the fp32 MFMAs' pipe (NOTES.md), and the general loop spends ~140 of them per tile and
// wave on values that never change:
//   * every LDS address is a lane constant worked out once (the kernel has ~30 registers to spare on this path): the twelve
//     scatter targets of the pair products, the two X-row stores, the A-operand reads, the partial sums.  The tile loop is
//     unrolled by two so that the buffer in turn is an immediate offset of those addresses;
//   * a lane without a valid scatter target writes to a slot of its own BEHIND its wave's row (the row pitch carries 64
//     floats for that), so that the target is an address like any other -- no compare, no select;
//   * the operands of the pair products are the X rows as loaded: a row beyond n (a repeat of row n - 1) only reaches
//     entries that land in those slots;
//   * ONE barrier per tile: the partial sums are double-buffered and the sum of tile t - G is taken at the top of tile t's
//     turn by all sixteen waves, one output per thread (before: behind a second barrier, by eight waves, two outputs each).
#define IT_C_P (IT_N_CRITEO * (IT_N_CRITEO - 1) / 2)         // 351 pairs
#define IT_C_NPB ((IT_C_P + 15) / 16)                         // 22 blocks of them
#define IT_C_NBLK (IT_C_NPB + IT_N_CRITEO)                    // 49 column blocks
#define IT_C_ZP (16 * IT_C_NBLK + 4 + TZR_WAVE)               // row pitch: the columns, the bank pad, a slot per lane (852)
#define IT_C_ZT (IT_TS * IT_C_ZP)                             // floats per z tile
#define IT_C_YT (4 * IT_TS * IT_YP)                           // floats per set of partial sums
static_assert(IT_C_NBLK == 4 * 12 + 1, "twelve whole blocks per K-group and one k-step of the 49th");
static_assert(2 * IT_C_ZT + 2 * IT_C_YT <= 2 * IT_TS * IT_ZP + IT_C_YT, "fits the forward's LDS");
static_assert(IT_C_ZT * 4 < 65536 && IT_C_YT * 4 < 65536, "the other buffer is an immediate offset of a DS instruction");

__device__ __forceinline__ void it_fwd_criteo(const ItFwdArgs& a, float* __restrict__ Zs, float* __restrict__ Ys, int lane, int wv) {
  constexpr int n = IT_N_CRITEO, P = IT_C_P, npb = IT_C_NPB, NB = 12;
  const int r = lane & 15, q = lane >> 4;
  const int kg = wv >> 2, hb = wv & 3;
  const int vfirst = kg * NB;
  float Wf[NB][4];
  float Wx[1];
  it_fwd_stage_w<NB, 1, IT_C_ZP>(a, Zs, Wf, Wx, n, P, npb, NB, 1, vfirst, vfirst + NB, r, q, kg, hb);
  const int Bn = (int)a.B;  // (32-bit tile and sample counters, checked by the launcher: scalar compares)
  const int ntiles = (Bn + IT_TS - 1) / IT_TS;
  const int G = gridDim.x;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // ---- lane constants (addresses in buffer 0)
  float* const zrow = Zs + wv * IT_C_ZP;  // this wave's row of a tile: sample wv
  float* tg[12];                          // where the pair products go: [reg][c00, c01, c11]
  {
    float* const slot = zrow + 16 * IT_C_NBLK + 4 + lane;
    // strict upper triangle, row-major (i < j): idx(i, j) = i (2 n - i - 1) / 2 + j - i - 1
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i0 = 4 * q + reg, i1 = 16 + i0, j0 = r, j1 = 16 + r;
      const int b0 = i0 * (2 * n - i0 - 1) / 2 - i0 - 1, b1 = i1 * (2 * n - i1 - 1) / 2 - i1 - 1;
      tg[3 * reg + 0] = (i0 < j0) ? zrow + b0 + j0 : slot;
      tg[3 * reg + 1] = (j1 < n) ? zrow + b0 + j1 : slot;
      tg[3 * reg + 2] = (i1 < j1 && j1 < n) ? zrow + b1 + j1 : slot;
    }
  }
  const int rr1 = 16 + r < n ? 16 + r : n - 1;
  float* const xs0 = zrow + 16 * npb + IT_D * r + 4 * q;    // X row r behind the pairs (r < n always)
  float* const xs1 = zrow + 16 * npb + IT_D * rr1 + 4 * q;  // (lanes of a row >= n repeat row n - 1: same address, same value)
  const float* const zr = Zs + r * IT_C_ZP + 4 * q + 16 * vfirst;          // A operand: sample r, this K-group's blocks
  const float* const zx = Zs + r * IT_C_ZP + 4 * q + 16 * (4 * NB) + kg;   // ... and k-step kg of the 49th block
  float* const yw = Ys + kg * (IT_TS * IT_YP) + (4 * q) * IT_YP + 16 * hb + r;  // accumulator reg j = partial y1[sample 4 q + j][16 hb + r]
  const float* const yr = Ys + wv * IT_YP + lane;                            // the sum: sample wv, output h = lane
  // X rows r and min(16 + r, n - 1), columns 4 q .. 4 q + 3 of a sample: 32-bit lane offsets from the sample's rows (the
  // sample is wave-uniform: scalar bases)
  const bool lo_dense = a.hd && r == 0;
  const unsigned lo_off = (unsigned)((r > a.hd ? r - a.hd : 0) * IT_D + 4 * q), hi_off = (unsigned)((rr1 - a.hd) * IT_D + 4 * q);
  auto fetch = [&](int tt) {
    const int bi = tt * IT_TS + wv;
    const int64_t b = bi < Bn ? bi : Bn - 1;
    const float* sp = a.sparse + b * a.sparse_stride;
    ItX x;
    x.lo = tzr_ld4(lo_dense ? a.dense + b * a.dense_stride + 4 * q : sp + lo_off);
    x.hi = tzr_ld4(sp + hi_off);
    return x;
  };
  // one sample's row of a z tile (buffer offset `zo`): pairwise products by MFMA to their targets, X rows behind them
  auto row = [&](const ItX& X, const int zo) {
    it_f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c11 = c00;
    const float x0[4] = {X.lo.x, X.lo.y, X.lo.z, X.lo.w};
    const float x1[4] = {X.hi.x, X.hi.y, X.hi.z, X.hi.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], x0[e], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], x1[e], c01, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], x1[e], c11, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      tg[3 * reg + 0][zo] = c00[reg];
      tg[3 * reg + 1][zo] = c01[reg];
      tg[3 * reg + 2][zo] = c11[reg];
    }
    tzr_st4(xs0 + zo, X.lo);
    tzr_st4(xs1 + zo, X.hi);
  };
  row(fetch(t), 0);
  ItX X = fetch(t + G < ntiles ? t + G : t);
  float bias = a.bias ? a.bias[lane] : 0.f;
  TZR_OPAQUE(bias);  // (forces the wait for this load HERE, not -- as vmcnt(0) -- at its first use inside the loop)
  // y1[sample wv of tile tt][h = lane] = the four partials in K-group order + bias, activation; in two steps so that the LDS
  // reads can take off a product ahead of the sum
  struct Part { float p[4]; };
  auto sum_load = [&](const int yo) {
    Part s;
#pragma unroll
    for (int g = 0; g < 4; ++g) s.p[g] = yr[yo + g * (IT_TS * IT_YP)];
    return s;
  };
  auto sum_store = [&](int tt, const Part& s) {
    float v = bias;
#pragma unroll
    for (int g = 0; g < 4; ++g) v += s.p[g];
    if (a.relu) v = v > 0.f ? v : 0.f;
    const int b = tt * IT_TS + wv;
    if (b < Bn) a.y1[(int64_t)b * a.y1_stride + lane] = v;
  };
  // Order of a wave's turn (a.stagger: 0 half of the waves -- two of the four on every SIMD -- each way, 1 / 2 every wave the
  // first / the second way):
  //   row first:  row of tile t + G, fetch, sum of tile t - G, product of tile t -- starts on X out of registers, so the MFMA
  //               pipe has work right behind the barrier while the others wait for their first LDS reads;
  //   row second: product of tile t (the partials of tile t - G read ahead of it, summed behind it), row, fetch.
  // (A y1 store must not be in flight right before a wait for X -- hipcc waits with vmcnt(0) -- hence the places of the sum.)
  // (one loop per order, chosen once: with both orders in one loop body X ends a turn in different registers and is copied
  // -- and waited for -- at the top of the next)
  const bool row_first_wave = a.stagger == 0 ? ((wv >> 2) & 1) != 0 : a.stagger == 1;
  IT_PROF_DECL;
  // tile t out of buffer CUR; `prev`: tile t - G was this workgroup's too (its partials are in the other set)
  auto tile = [&](auto CUR, auto RF, bool prev) {
    constexpr int cur = decltype(CUR)::value;
    constexpr bool row_first = decltype(RF)::value;
    constexpr int zo = cur * IT_C_ZT, zn = (cur ^ 1) * IT_C_ZT, yo = cur * IT_C_YT, yp = (cur ^ 1) * IT_C_YT;
    auto product = [&]() {
      it_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // A operand of block m, k-steps 0..3 = columns 4 q + kk: one ds_read_b128, issued a block ahead of its MFMAs and pinned
      // there (left alone, hipcc hoists all twelve reads to the top: 48 registers)
      float4 av[2];
      av[0] = tzr_ld4(zr + zo);
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (m + 1 < NB) av[(m + 1) & 1] = tzr_ld4(zr + zo + 16 * (m + 1));
        const float4 c = av[m & 1];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, Wf[m][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, Wf[m][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, Wf[m][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, Wf[m][3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zx[zo], Wx[0], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) yw[yo + j * IT_YP] = acc[j];
    };
    // the row of sample wv of tile t + G into the other buffer; then tile t + 2 G's X rows take off (unconditionally: the
    // loads land in X's registers -- no copy, and no wait, at the top of the next turn)
    auto next_row = [&]() {
      if (t + G < ntiles) row(X, zn);
      X = fetch(t + 2 * G < ntiles ? t + 2 * G : t);
    };
    tzr_lds_barrier();  // tile t complete in buffer cur; the other buffer and the partials of tile t - 2 G free
    IT_PROF_MARK(0);  // wait
    if (row_first) {
      next_row();
      IT_PROF_MARK(1);  // row (first)
      if (prev) sum_store(t - G, sum_load(yp));
      IT_PROF_MARK(5);  // sum of the previous tile's partials
      product();
      IT_PROF_MARK(2);  // product
    } else {
      Part s;
      if (prev) s = sum_load(yp);
      __builtin_amdgcn_sched_barrier(0);
      product();
      IT_PROF_MARK(2);  // product
      if (prev) sum_store(t - G, s);
      IT_PROF_MARK(5);
      next_row();
      IT_PROF_MARK(3);  // row (second)
    }
  };
  int last = 0;
  auto run = [&](auto RF) {
    bool prev = false;
    for (;;) {
      tile(std::integral_constant<int, 0>(), RF, prev);
      t += G;
      last = 0;
      if (t >= ntiles) break;
      tile(std::integral_constant<int, 1>(), RF, true);
      t += G;
      last = 1;
      if (t >= ntiles) break;
      prev = true;
    }
  };
  if (row_first_wave) run(std::true_type());
  else run(std::false_type());
  tzr_lds_barrier();  // the last tile's partials
  sum_store(t - G, sum_load(last * IT_C_YT));
  IT_PROF_DUMP(a.prof);
}

template <bool ZOUT>
__device__ __forceinline__ void it_fwd_body(const ItFwdArgs& a) {
  // two z tiles at the pitch of the body + the partial sums (it_fwd_criteo: two sets of them behind its smaller tiles)
  constexpr int kZCriteo = 2 * IT_TS * (16 * 49 + 4), kYs = 4 * IT_TS * IT_YP;
  __shared__ __attribute__((aligned(16))) float Sm[2 * IT_TS * IT_ZP + kYs];
  __shared__ float trash[IT_THREADS];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TZR_WAVE));  // (scalar: per-wave bases stay in SGPRs)
  float* tr = trash + threadIdx.x;
  // Criteo (27 vectors: 22 pair blocks + 27 = 49): 12 blocks + 1 k-step per wave (49 MFMAs, 49 registers of W1); other
  // shapes run the longest body, zero-weighted
  if (a.n == IT_N_CRITEO) {
    if (ZOUT) it_fwd_loop<12, 1, ZOUT, IT_N_CRITEO>(a, Sm, Sm + kZCriteo, tr, lane, wv);
    else it_fwd_criteo(a, Sm, Sm + 2 * IT_C_ZT, lane, wv);
  } else {
    it_fwd_loop<IT_KB, 3, ZOUT, 0>(a, Sm, Sm + 2 * IT_TS * IT_ZP, tr, lane, wv);
  }
}

// two kernels, not one branch: the registers of the z-writing body (store addresses, a second set of X rows) would be the
// budget of the other one too, and W1 fragments of the training step's forward (z = null) would be spilled
__global__ __launch_bounds__(IT_THREADS) void tzr_ia_top_fwd_kernel(ItFwdArgs a) { it_fwd_body<false>(a); }
__global__ __launch_bounds__(IT_THREADS) void tzr_ia_top_fwd_z_kernel(ItFwdArgs a) { it_fwd_body<true>(a); }

int g_tzr_it_stagger = 0;  // tzr_tune("it_stagger"): which half of the waves runs the next product first (0 / 1), 2 = none (experiments)
// tzr_tune("it_fwd_stagger"): forward, order of row building and product in a wave's turn: 0 / 1 = half of the waves each way
// (two of the four on every SIMD), 2 / 3 = every wave row-first / row-second.  Half and half is the default for both
// kernels: with the z stores it hides them behind the other half's product (111.6 vs 116-118 us, profiles/r04ap); without
// them (it_fwd_criteo) the row-first waves give the MFMA pipe work right behind the barrier (84.5 vs 85.9 us, profiles/r05ba)
int g_tzr_it_fwd_stagger = 0;
int g_tzr_it_wgs = 0;  // tzr_tune("it_wgs"): workgroups of the fused kernels (0 = one per CU)

static unsigned it_grid(int64_t B) {
  const int64_t cus = g_tzr_it_wgs > 0 ? g_tzr_it_wgs : 256;  // MI355X: 256 CUs, one persistent workgroup each
  const int64_t tiles = (B + IT_TS - 1) / IT_TS;
  return (unsigned)(tiles < cus ? (tiles < 1 ? 1 : tiles) : cus);
}

extern "C" int tzr_dot_interaction_top_supported(int F, int D, int has_dense, int H) {
  const int n = F + (has_dense ? 1 : 0);
  if (D != IT_D || H != IT_H || n < 2 || n > 32) return 0;
  return ((n * (n - 1) / 2 + 15) / 16 + n) <= IT_MAXBLK ? 1 : 0;
}

extern "C" int tzr_dot_interaction_top_bwd(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                           int64_t sparse_stride, int F, int D, int64_t B, const float* d_g1,
                                           int64_t g1_stride, int H, const float* d_W1, int64_t ldw,
                                           const float* d_scale, float* d_grad_dense, int64_t grad_dense_stride,
                                           float* d_grad_sparse, int64_t grad_sparse_stride, void* stream) {
  const int hd = d_dense ? 1 : 0;
  const int n = F + hd;
  if (!d_sparse || !d_g1 || !d_W1 || !d_grad_sparse || F <= 0 || B < 0) return TZR_ERR_INVALID;
  if (B > IT_MAX_BATCH) return TZR_ERR_UNSUPPORTED;
  if (hd && !d_grad_dense) return TZR_ERR_INVALID;
  if (!tzr_dot_interaction_top_supported(F, D, hd, H)) return TZR_ERR_UNSUPPORTED;
  if (ldw < n * (n - 1) / 2 + IT_D * n) return TZR_ERR_INVALID;
  if ((sparse_stride & 3) || (grad_sparse_stride & 3) || (g1_stride & 3) || (hd && ((dense_stride | grad_dense_stride) & 3)) ||
      ((reinterpret_cast<uintptr_t>(d_sparse) | reinterpret_cast<uintptr_t>(d_grad_sparse) | reinterpret_cast<uintptr_t>(d_g1) |
        reinterpret_cast<uintptr_t>(d_dense) | reinterpret_cast<uintptr_t>(d_grad_dense)) & 15))
    return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ItBwdArgs a;
  a.dense = d_dense; a.sparse = d_sparse; a.g1 = d_g1; a.W1 = d_W1; a.scale = d_scale; a.gdense = d_grad_dense; a.gsparse = d_grad_sparse;
  a.dense_stride = dense_stride; a.sparse_stride = sparse_stride; a.g1_stride = g1_stride; a.ldw = ldw;
  a.gdense_stride = grad_dense_stride; a.gsparse_stride = grad_sparse_stride; a.B = B; a.n = n; a.hd = hd;
  a.stagger = g_tzr_it_stagger;
#ifdef IT_PROF
  a.prof = g_tzr_it_prof;
#else
  a.prof = nullptr;
#endif
  hipLaunchKernelGGL(tzr_ia_top_bwd_kernel, dim3(it_grid(B)), dim3(IT_THREADS), 0, st, a);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_dot_interaction_top_fwd(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                           int64_t sparse_stride, int F, int D, int64_t B, const float* d_W1,
                                           int64_t ldw, const float* d_bias, int H, int relu, float* d_z,
                                           int64_t z_stride, float* d_y1, int64_t y1_stride, void* stream) {
  const int hd = d_dense ? 1 : 0;
  const int n = F + hd;
  if (!d_sparse || !d_W1 || !d_y1 || F <= 0 || B < 0) return TZR_ERR_INVALID;
  if (B > IT_MAX_BATCH) return TZR_ERR_UNSUPPORTED;
  if (!tzr_dot_interaction_top_supported(F, D, hd, H)) return TZR_ERR_UNSUPPORTED;
  const int width = n * (n - 1) / 2 + IT_D * n;
  if (ldw < width || (d_z && z_stride < width)) return TZR_ERR_INVALID;
  if ((sparse_stride & 3) || (hd && (dense_stride & 3)) ||
      ((reinterpret_cast<uintptr_t>(d_sparse) | reinterpret_cast<uintptr_t>(d_dense)) & 15) ||
      (reinterpret_cast<uintptr_t>(d_y1) & 3) || (reinterpret_cast<uintptr_t>(d_z) & 3))
    return TZR_ERR_INVALID;
  if (B == 0) return TZR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ItFwdArgs a;
  a.dense = d_dense; a.sparse = d_sparse; a.W1 = d_W1; a.bias = d_bias; a.z = d_z; a.y1 = d_y1;
  a.dense_stride = dense_stride; a.sparse_stride = sparse_stride; a.ldw = ldw; a.z_stride = z_stride; a.y1_stride = y1_stride;
  a.B = B; a.n = n; a.hd = hd; a.relu = relu;
  a.stagger = g_tzr_it_fwd_stagger > 0 ? g_tzr_it_fwd_stagger - 1 : 0;
#ifdef IT_PROF
  a.prof = g_tzr_it_prof;
#else
  a.prof = nullptr;
#endif
  if (d_z) hipLaunchKernelGGL(tzr_ia_top_fwd_z_kernel, dim3(it_grid(B)), dim3(IT_THREADS), 0, st, a);
  else hipLaunchKernelGGL(tzr_ia_top_fwd_kernel, dim3(it_grid(B)), dim3(IT_THREADS), 0, st, a);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
