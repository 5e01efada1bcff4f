// K9c: the weight gradient of the Linear behind the DLRM dot interaction, dW1 = g1^T z, WITHOUT z in HBM (gfx950).
//
// DLRM.predict (/root/reference/tzrec/models/dlrm.py:123-135) feeds z[b] = [pairs of X_b X_b^T | dense row | sparse rows]
// (P + 16 n floats) into `final_mlp` (/root/reference/tzrec/modules/mlp.py:58-83); autograd's weight gradient of its first
// Linear is g1^T z, a [64 x B] . [B x (P + 16 n)] product over the stored z (205 MB at B = 65 536, written by the forward
// only for this).  Here the z columns are rebuilt from X on the chip:
//
//   * the P + 16 n columns are cut into FOUR column groups, at most one block of the pair matrix each -- c00 = pairs (i < j <
//     16), c01 = (i < 16 <= j), c11 = (16 <= i < j) -- plus a share of the X rows that evens out the MFMA work; a workgroup =
//     (batch slice, column group).  With 256 workgroups that is 64 batch slices: 64 x 200 KB of partial sums instead of
//     256 x 200 KB for workgroups that each own the full width (the partials are the only HBM writes of this kernel);
//     the four groups of one slice sit on the same XCD (blockIdx -> XCD round-robin), so its L2 serves the X rows they share.
//   * a tile is 32 samples in LDS, z[sample][column] and g1[sample][h], exactly as the loaders hold them: an operand register
//     (16 bytes of one X row of one sample) is stored as it is -- it IS the group's X-row columns -- and a pair block as dwords;
//     the samples are the contraction index of v_mfma_f32_16x16x4_f32, the multipliers read one dword per k-step.
//   * eight waves multiply (wave = output block of the 64 layer outputs x column half: up to eight 16 x 16 accumulators,
//     64 MFMAs per tile), eight load and build the next tile meanwhile; ONE barrier per tile (two tiles in LDS).
//   * tzr_ia_wgrad_reduce_kernel sums the slices' partials in fixed order (deterministic), applies the loss scale and puts
//     the columns back in z's order.
// Exact fp32 MFMA throughout.  6.6 GFLOP + 1.6 for the pair blocks at B = 65 536.
//
// Measured (B = 65 536, profiles/r04x .. r04ak): 103 us + 6.5 us for the reduce kernel alone, the same as the library's split-K
// GEMM plus the z store it needs inside the step (0.5693 vs 0.5668 ms per step; ahead of it at 8 192 and 32 768).  The
// multipliers run at the MFMA pipe's pace (~4 500 clocks per tile against 4 096 of MFMA issue); what is left is the loaders'
// chain per tile -- and the finding that cost the most time here: fp32 MFMAs and VALU instructions share ONE pipe on this
// chip, so every v_mov / address instruction of a loader is taken from the product's time, and a loader's own 16 MFMAs queue
// behind the multipliers'.  Versions on the way: all sixteen waves doing everything in turn 131 us (phases additive) ->
// specialised waves 129 -> loads two tiles ahead 122 -> 1 KB-per-instruction operand loads 118 -> 32-bit sample offsets (700 ->
// 250 instructions per tile and loader) 109 -> X rows out of the pair-operand registers 113 (fewer loads, more v_mov) -> tiles in
// the loaders' own layout instead of transposed (no v_mov, dword operand reads) 109.
// HBM traffic: X and g1 once (131 MB; the column groups' re-reads hit the L2, profiles/r04ad).
#include <cstring>

#include "tzr_common.h"
#include <tzr_gfx950.h>

#include "wgrad_reduce.h"

#define WG_THREADS 1024
#define WG_WAVES (WG_THREADS / TZR_WAVE)
#define WG_S 32            // samples per tile
#define WG_P 336           // floats of a sample's row of a tile in LDS (>= 16 npb + 20 xn, = 16 mod 32: the dword operand reads of lanes (r, q) and (r, q + 1) fall on different bank halves)
#define WG_PG 80           // ... of its g1 row (64 + 16)
#define WG_MAXB 16         // column blocks (of 16) per group

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

// -DIT_PROF (scripts/build_prof_lib.sh): per wave, the clocks spent in each phase of the tile loop, into the table of
// tzr_it_prof_table (interaction_top.hip); not compiled into the product.
#ifdef IT_PROF
extern "C" uint64_t* g_tzr_it_prof;
#define WG_PROF_DECL uint64_t wg_tl = __builtin_amdgcn_s_memtime(), wg_ts[6] = {0, 0, 0, 0, 0, 0}
#define WG_PROF_MARK(i) do { const uint64_t wg_now = __builtin_amdgcn_s_memtime(); wg_ts[i] += wg_now - wg_tl; wg_tl = wg_now; } while (0)
#define WG_PROF_DUMP(tab) do { if (tab && lane == 0) for (int i_ = 0; i_ < 6; ++i_) (tab)[((size_t)blockIdx.x * WG_WAVES + wv) * 6 + i_] = wg_ts[i_]; } while (0)
#define WG_PROF_TOUCH(x) do { float wg_tmp = (x); TZR_OPAQUE(wg_tmp); } while (0)
#define WG_BUILD !(a.debug & 2)
#define WG_LOAD !(a.debug & 4)
#else
#define WG_PROF_DECL
#define WG_PROF_MARK(i)
#define WG_PROF_DUMP(tab)
#define WG_PROF_TOUCH(x)
// (the phase-skipping bits 2 and 4 of tzr_tune("wg_debug") exist in the IT_PROF build only: as run-time conditions around the
// loaders' stores and loads they cost the product its prefetch depth -- with a path on which a set of loads is never
// consumed hipcc waits for EVERY load in flight before it refills the set (s_waitcnt vmcnt(0) once per tile instead of
// vmcnt(12): profiles/r05bl))
#define WG_BUILD true
#define WG_LOAD true
#endif
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

struct __attribute__((packed, aligned(4))) wg_f4u {
  float x, y, z, w;
};
__device__ __forceinline__ float4 wg_ld4_a4(const float* p) {  // 16-byte load from a 4-byte aligned address
  const wg_f4u u = *reinterpret_cast<const wg_f4u*>(p);
  return make_float4(u.x, u.y, u.z, u.w);
}


struct WgArgs {
  const float *dense, *sparse, *g1;
  float* part;  // [slices][64][vw]
  int64_t dense_stride, sparse_stride, g1_stride, B;
  int n, hd, slices, vw, debug;
  uint64_t* prof;
  WgGroup g[WG_NG];
};

// Column groups for n rows (n0 = min(n, 16) low rows, n1 = n - 16 high rows): the pair blocks go to groups 0 / 1 / 2, the X
// rows to whichever group has the least MFMA work so far (a pair block costs 16 production MFMAs per 4 samples on top of
// its columns).  Every group ends with at most 16 blocks: c00 / c11 have <= 8, c01 <= 16, and 64 blocks hold everything.
// A loader reads whole 16-row blocks of X (the low rows, the high rows or both), which serve its pair block AND its X rows:
// the row ranges go out in the order group 0 (c00: low rows), 3, 1, 2 (c11: high rows), so that the groups that need one
// block for their pairs mostly find their X rows in it.
static void wg_plan(int n, WgGroup g[WG_NG], int* vw) {
  const int n0 = n < 16 ? n : 16, n1 = n > 16 ? n - 16 : 0;
  const int np[WG_NG] = {n0 * (n0 - 1) / 2, n0 * n1, n1 * (n1 - 1) / 2, 0};
  int cost[WG_NG];
  for (int k = 0; k < WG_NG; ++k) {
    g[k].src = np[k] > 0 ? k + 1 : 0;
    g[k].npb = (np[k] + 15) / 16;
    g[k].xn = 0;
    cost[k] = (g[k].src ? 64 : 0) + 16 * g[k].npb;
  }
  for (int row = 0; row < n; ++row) {
    int best = -1;
    for (int k = 0; k < WG_NG; ++k)
      if (g[k].npb + g[k].xn < WG_MAXB && (best < 0 || cost[k] < cost[best])) best = k;
    g[best].xn += 1;
    cost[best] += 16;
  }
  const int order[WG_NG] = {0, 3, 1, 2};
  int x0 = 0;
  for (int o = 0; o < WG_NG; ++o) {
    const int k = order[o];
    g[k].x0 = x0;
    x0 += g[k].xn;
  }
  int vb = 0;
  for (int k = 0; k < WG_NG; ++k) {
    g[k].nb = g[k].npb + g[k].xn;
    g[k].vbase = vb;
    vb += 16 * g[k].nb;
  }
  *vw = vb;
}

// X row i of sample b (row 0 is the dense row when there is one)
__device__ __forceinline__ const float* wg_row(const WgArgs& a, int64_t b, int i) {
  return (a.hd && i == 0) ? a.dense + b * a.dense_stride : a.sparse + b * a.sparse_stride + (int64_t)(i - a.hd) * WG_D;
}

// The sixteen waves of a workgroup are SPECIALISED (first version: every wave loaded, built and multiplied in turn, the
// three phases simply added up -- 27 + 48 + 57 us, profiles/r04x; a wave whose loads wait for a slot in the memory pipe
// cannot issue its MFMAs either):
//   waves 0 .. 7   load and build: loader l owns samples 4 l .. 4 l + 3 of every tile.  It reads 16-row blocks of X the way
//                  the MFMA wants them -- lane (r, q): 16 bytes of row r at column 4 q, one instruction = 1 KB in one piece
//                  -- builds the pair block of each sample (4 MFMAs) and stores, per entry, the four samples side by side;
//                  the same registers are the group's X-row columns (a 4 x 4 transpose that costs nothing: component c of
//                  the four samples' registers = one 16-byte store); one g1 item.  It runs two tiles ahead.
//                  (X rows loaded one float per lane, four samples per item: 24 more load instructions per tile and loader,
//                  and the CU's one memory pipe took ~2 000 clocks per tile to take them all, profiles/r04ae.)
//   waves 8 .. 15  multiply: wave 8 + w = (output block hb = w & 3, column half ch = w >> 2), blocks ch, ch + 2, ... (<= 8
//                  accumulators); two of them on every SIMD keep its MFMA pipe fed.
// One barrier per tile hands tile t + 1 over.

// A tile in LDS is `z[sample][column]`, the way the loaders hold it: a loaded operand register (16 bytes of one X row of one
// sample) is stored as it is, a pair-block result as four dwords -- no register shuffling on the VALU, which shares its pipe
// with the multipliers' MFMAs (kept transposed, `zT[column][sample]`, every 16-byte store wanted four v_mov: 45 per tile).
// The multipliers read their operands a dword per k-step instead of 16 bytes per four.  Column of block b in a sample's
// row: the pair blocks 16 apart, the X-row blocks 20 apart (a lane per ROW stores 16 bytes into them: 16 floats apart the
// eight lanes of a store group would share two bank quads).
#define WG_ZT (WG_S * WG_P)  // floats per tile buffer
__device__ __forceinline__ int wg_block_base(int b, int npb) { return b < npb ? 16 * b : 16 * npb + 20 * (b - npb); }

// Registers a loader thread carries from the loads of a tile to its LDS stores.  Every load lands in the register it is
// used from: nothing touches a loaded value (no select, no mask, no copy) before the stores -- that would be a wait right
// behind the loads.  fetch() is branch-free (rows and samples clamped): hipcc then knows how many loads are younger than the
// ones it waits for and leaves them in flight (s_waitcnt vmcnt(N), N > 0).
template <bool LO, bool HI>
struct WgRegs {
  float4 lo[LO ? 4 : 1];  // (row r, columns 4 q .. 4 q + 3) of the wave's four samples
  float4 hi[HI ? 4 : 1];  // (row 16 + r, ...)
  float4 gv;              // g1 item: outputs 4 (lane & 15) .. + 3 of sample lane >> 4
};

// SRC = the group's block of the pair matrix (0: X rows only); LO / HI: which 16-row blocks of X the loaders read (what the
// pair block needs and what the group's X rows lie in).
template <int SRC, bool LO, bool HI>
__device__ __forceinline__ void wg_body(const WgArgs& a, const WgGroup& G, float* __restrict__ zT, float* __restrict__ gT, int slice) {
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TZR_WAVE));
  const int r = lane & 15, q = lane >> 4;
  const int n = a.n;
  const int n0 = n < 16 ? n : 16, n1 = n > 16 ? n - 16 : 0;
  // (32-bit arithmetic throughout: the launcher checks that B * stride fits 32 bits for the three inputs -- with 64-bit
  // sample offsets the loaders ran ~700 instructions per tile, most of them address arithmetic, profiles/r04ab)
  const int B = (int)a.B;
  const int ntiles = (B + WG_S - 1) / WG_S;
  int t = slice;

  const bool is_loader = wv < 8;
  // The multipliers are the younger half of the workgroup and lose every arbitration by age against the loaders' VALU and
  // memory instructions; their MFMA stream is what the tile time is made of.  Static priority, set once: 107.0 -> 104.2 us
  // (profiles/r05be; the loaders at the higher priority instead: no change).
  if (!is_loader) __builtin_amdgcn_s_setprio(1);
  if (is_loader) {
    // ================================================= loaders
    const int lw = wv;  // the quad of samples of this wave
    const unsigned dstride = (unsigned)a.dense_stride, sstride = (unsigned)a.sparse_stride, gstride = (unsigned)a.g1_stride;
    typedef WgRegs<LO, HI> Regs;
    // lane constants (a few registers, kept across the loop)
    const bool lo_dense = a.hd && r == 0;  // this lane's low row is the dense row
    const int rlo = r < n ? r : n - 1, rhi = 16 + r < n ? 16 + r : n - 1;
    const unsigned lo_off = (unsigned)((rlo > a.hd ? rlo - a.hd : 0) * WG_D + 4 * q);
    const unsigned hi_off = (unsigned)((rhi - a.hd) * WG_D + 4 * q);  // (a high row is never the dense row)
    // loads of tile tt into registers (samples behind the batch read the last one; their g1 is zeroed at the store); the
    // sample is wave-uniform: scalar row bases, 32-bit lane offsets
    const unsigned g_off = (unsigned)(lane >> 4) * gstride + 4 * (lane & 15);
    auto fetch = [&](Regs& R, int tt) {
      const int s0 = tt * WG_S + 4 * lw;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned se = (unsigned)(s0 + e < B ? s0 + e : B - 1);
        const float* sp = a.sparse + se * sstride;  // (scalar)
        if (LO) R.lo[e] = tzr_ld4(lo_dense ? a.dense + se * dstride + 4 * q : sp + lo_off);
        if (HI) R.hi[e] = tzr_ld4(sp + hi_off);
      }
      // g1 of the four samples as ONE instruction: lane -> (sample lane >> 4, outputs 4 (lane & 15) ..), 1 KB in one piece
      // like the X rows (it was a dword per lane and sample: four instructions and four LDS stores per tile and loader)
      if (s0 + 3 < B) {  // (scalar branch: a scalar row base and a lane constant; only the batch's last tile clamps per lane)
        R.gv = wg_ld4_a4(a.g1 + (unsigned)s0 * gstride + g_off);
      } else {
        const unsigned sg = (unsigned)(s0 + (lane >> 4) < B ? s0 + (lane >> 4) : B - 1);
        R.gv = wg_ld4_a4(a.g1 + sg * gstride + 4 * (lane & 15));
      }
    };
    WG_PROF_DECL;
    // The registers (loaded from tile tt) into tile buffer `buf`, and the loads of tile tn into the same registers, in the
    // order that keeps the chain short: the X-row / g1 stores first (they wait for nothing but the loads), then the pair
    // MFMAs -- which queue behind the multipliers' -- then the NEW loads (the operand registers are free once the MFMAs have
    // issued; the 1 KB-per-instruction loads take the CU's memory pipe ~16 clocks each), the pair stores last.
    auto step = [&](int buf, Regs& R, int tt, int tn, bool build, bool load) {
      float* zb = zT + buf * WG_ZT + (4 * lw) * WG_P;  // rows of this wave's four samples
      wg_f32x4 c[4];
      if (build) {
        // the group's X rows out of the registers: row -> block npb + (row - x0), this lane's 16 bytes of it, per sample
        float* xb = zb + 16 * G.npb + 4 * q;
        if (LO) {
          const int xl = r - G.x0;
          if (xl >= 0 && xl < G.xn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tzr_st4(xb + 20 * xl + e * WG_P, R.lo[e]);
          }
        }
        if (HI) {
          const int xl = 16 + r - G.x0;
          if (xl >= 0 && xl < G.xn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tzr_st4(xb + 20 * xl + e * WG_P, R.hi[e]);
          }
        }
        const int s = tt * WG_S + 4 * lw;
        const float4 g = R.gv;
        float* go = gT + buf * (WG_S * WG_PG) + (4 * lw + (lane >> 4)) * WG_PG + 4 * (lane & 15);
        if (s + 3 < B) tzr_st4(go, g);  // (scalar branch: only the batch's last tile masks)
        else tzr_st4(go, s + (lane >> 4) < B ? g : tzr_zero4());
        if (SRC != 0) {
          // per sample a 16 x 16 block of X X^T by four v_mfma_f32_16x16x4_f32 (lane (r, q) supplies column 4 q + k-step of
          // row r); result register e of lane (r, q) = entry (i = 4 q + e, j = r)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float4 av = SRC == 3 ? R.hi[HI ? e : 0] : R.lo[LO ? e : 0], bv = SRC == 1 ? R.lo[LO ? e : 0] : R.hi[HI ? e : 0];
            c[e] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
            c[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, c[e], 0, 0, 0);
            c[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, c[e], 0, 0, 0);
            c[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, c[e], 0, 0, 0);
            c[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, c[e], 0, 0, 0);
          }
        }
      }
      WG_PROF_MARK(1);  // X / g1 stores, pair MFMAs issued
      if (load) fetch(R, tn);
      WG_PROF_MARK(2);  // load issue
      if (build && SRC != 0) {
        // entry (i, j) -> column of the group: c00 / c11 the strict upper triangle of their rows, row-major; c01 all of i x
        // n1; sample e's block into sample e's row
        const int nn = SRC == 1 ? n0 : n1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = 4 * q + k, j = r;
          const bool ok = SRC == 2 ? j < n1 : (i < j && j < nn);
          const int col = SRC == 2 ? i * n1 + j : (i * (2 * nn - i - 1)) / 2 + j - i - 1;
          if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) zb[e * WG_P + col] = c[e][k];
          }
        }
      }
    };
    if (t >= ntiles) return;
    const int S = a.slices;
    // R0 / R1 alternate: the set stored in an iteration is refilled with the tile three ahead -- the loads run TWO tiles ahead
    // of the stores (one tile of multiplying does not cover a load under this traffic)
    Regs R0, R1;
    fetch(R0, t);
    fetch(R1, t + S);
    step(0, R0, t, t + 2 * S, true, true);
    tzr_lds_barrier();
    // One turn per tile of this slice, the two register sets in turn.  Counted, with the odd turn behind the loop -- NOT
    // `for (;;) { A; if (done) break; B; if (done) break; }`: hipcc routes both exits through the loop's latch, its wait
    // counts then allow for the path "A, latch, A" on which A's own loads are the youngest in flight, and every A waits for
    // everything (s_waitcnt vmcnt(0): the loads B issued a moment ago).  This way both turns leave the other set's twelve
    // loads in flight (vmcnt(12)): profiles/r05bm.
    auto turn_a = [&]() {
      WG_PROF_TOUCH(R1.gv.w);
      WG_PROF_MARK(0);  // loads of the tile about to be stored
      step(1, R1, t + S, t + 3 * S, WG_BUILD, WG_LOAD);  // (buffer 1.  Behind the last tile: samples >= B with g1 = 0 into a buffer nobody multiplies)
      WG_PROF_MARK(4);  // pair stores
      tzr_lds_barrier();  // tile t + S complete, the multipliers done with tile t
      WG_PROF_MARK(3);  // barrier
      t += S;
    };
    auto turn_b = [&]() {
      WG_PROF_TOUCH(R0.gv.w);
      WG_PROF_MARK(0);
      step(0, R0, t + S, t + 3 * S, WG_BUILD, WG_LOAD);
      WG_PROF_MARK(4);
      tzr_lds_barrier();
      WG_PROF_MARK(3);
      t += S;
    };
    const int turns = (ntiles - t + S - 1) / S;
    for (int k = 0; k < (turns >> 1); ++k) {
      turn_a();
      turn_b();
    }
    if (turns & 1) turn_a();
    WG_PROF_DUMP(a.prof);
    return;
  }
  // =================================================== multipliers
  // dW[16 hb .., blocks of this wave] += g1^T . z over the 32 samples of a buffer.  Blocks ch, ch + 2, .., ch + 10 always (a
  // block behind the group's last reads columns nobody writes -- zeros -- and is not stored), ch + 12 / ch + 14 behind scalar
  // branches.
  const int hb = wv & 3, ch = (wv & 7) >> 2;
  const int nbw = G.nb > ch ? (G.nb - ch + 1) >> 1 : 0;  // blocks ch, ch + 2, ... of this wave
  wg_f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
  int bb[8];  // LDS offsets of this wave's blocks (scalar)
#pragma unroll
  for (int m = 0; m < 8; ++m) bb[m] = wg_block_base(ch + 2 * m < WG_MAXB ? ch + 2 * m : WG_MAXB - 1, G.npb);
  // lane (r, q), k-step t of a tile: sample 4 t + q, one dword of either operand (k index q)
  auto product = [&](int buf) {
    const float* gp = gT + buf * (WG_S * WG_PG) + q * WG_PG + 16 * hb + r;
    const float* zp = zT + buf * WG_ZT + q * WG_P + r;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {  // (half a tile's operand reads in flight before its first MFMA)
      float av[4], bv[6][4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int tt = 4 * h2 + t4;
        av[t4] = gp[tt * 4 * WG_PG];
#pragma unroll
        for (int m = 0; m < 6; ++m) bv[m][t4] = zp[tt * 4 * WG_P + bb[m]];
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int m = 0; m < 6; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], bv[m][t4], acc[m], 0, 0, 0);
      if (nbw > 6) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
          acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], zp[(4 * h2 + t4) * 4 * WG_P + bb[6]], acc[6], 0, 0, 0);
      }
      if (nbw > 7) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
          acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], zp[(4 * h2 + t4) * 4 * WG_P + bb[7]], acc[7], 0, 0, 0);
      }
    }
  };
  if (t < ntiles) {
    tzr_lds_barrier();  // tile t is in buffer 0
    WG_PROF_DECL;
    for (;;) {  // (two tiles per trip: the buffer is a constant of each half)
      if (!(a.debug & 1)) product(0);
      WG_PROF_MARK(4);  // product
      tzr_lds_barrier();
      WG_PROF_MARK(5);  // barrier
      t += a.slices;
      if (t >= ntiles) break;
      if (!(a.debug & 1)) product(1);
      WG_PROF_MARK(4);
      tzr_lds_barrier();
      WG_PROF_MARK(5);
      t += a.slices;
      if (t >= ntiles) break;
    }
    WG_PROF_DUMP(a.prof);
  }
  // ---- partial sums of this slice: accumulator register e of lane (r, q) = dW[h = 16 hb + 4 q + e][column r of the block]
  int le = lane;
  TZR_OPAQUE(le);  // (the store addresses are built here, not before the tile loop and carried through it)
  float* po = a.part + ((int64_t)slice * WG_H + 16 * hb + 4 * (le >> 4)) * a.vw + G.vbase + (le & 15);
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (m < nbw) {
#pragma unroll
      for (int e = 0; e < 4; ++e) po[(int64_t)e * a.vw + 16 * (ch + 2 * m)] = acc[m][e];
    }
}

__global__ __launch_bounds__(WG_THREADS) void tzr_ia_wgrad_kernel(WgArgs a) {
  __shared__ __attribute__((aligned(16))) float zT[2 * WG_ZT];  // [tile][sample][column]
  __shared__ __attribute__((aligned(16))) float gT[2 * WG_S * WG_PG];  // [tile][sample][h]
  // workgroup -> (slice, group): the four groups of a slice on one XCD
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int gi = kx & 3, slice = (kx >> 2) * 8 + xcd;
  const WgGroup G = a.g[gi];
  for (int k = threadIdx.x; k < 2 * WG_ZT; k += WG_THREADS) zT[k] = 0.f;  // pad columns of the last pair block stay zero
  __syncthreads();
  // which 16-row blocks of X the loaders of this group read: what its pair block needs, and where its X rows lie
  const bool xlo = G.xn > 0 && G.x0 < 16, xhi = G.xn > 0 && G.x0 + G.xn > 16;
  const bool lo = xlo || G.src == 1 || G.src == 2, hi = xhi || G.src == 2 || G.src == 3;
  if (G.src == 0) {
    if (lo && hi) wg_body<0, true, true>(a, G, zT, gT, slice);
    else if (hi) wg_body<0, false, true>(a, G, zT, gT, slice);
    else wg_body<0, true, false>(a, G, zT, gT, slice);
  } else if (G.src == 1) {
    if (hi) wg_body<1, true, true>(a, G, zT, gT, slice);
    else wg_body<1, true, false>(a, G, zT, gT, slice);
  } else if (G.src == 2) {
    wg_body<2, true, true>(a, G, zT, gT, slice);
  } else {
    if (lo) wg_body<3, true, true>(a, G, zT, gT, slice);
    else wg_body<3, false, true>(a, G, zT, gT, slice);
  }
}

__global__ __launch_bounds__(256) void tzr_ia_wgrad_reduce_kernel(WgReduceArgs a) {
  const int lane = threadIdx.x & 63;
  const int o = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
  int h, c;
  bool live;
  const float v = wg_reduce_output(a, o, lane, &h, &c, &live);
  if (live && (lane >> 4) == 0) a.dW[(int64_t)h * a.ldw + c] = v;
}

int g_tzr_wg_debug = 0;  // tzr_tune("wg_debug"): phase-skipping bits for timing experiments (1 no product, 2 no tile build, 4 no loads, 8 no main kernel, 16 no reduce kernel): wrong results

static int wg_slices(int64_t B) {
  const int64_t ntiles = (B + WG_S - 1) / WG_S;
  int64_t s8 = (ntiles + 31) / 32;  // about four tiles per slice before a slice is added
  s8 = s8 < 1 ? 1 : (s8 > WG_MAXSLICES / 8 ? WG_MAXSLICES / 8 : s8);
  return (int)(8 * s8);
}

extern "C" int64_t tzr_dot_interaction_top_wgrad_workspace(int F, int D, int has_dense, int H) {
  if (!tzr_dot_interaction_top_supported(F, D, has_dense, H)) return 0;
  WgGroup g[WG_NG];
  int vw;
  wg_plan(F + (has_dense ? 1 : 0), g, &vw);
  return (int64_t)WG_MAXSLICES * WG_H * vw * (int64_t)sizeof(float);
}

// The weight-gradient kernel alone: the batch slices' partial sums stay in d_ws and `ra` says how to add them up
// (wg_reduce_output) -- for tzr_dot_interaction_top_wgrad's own reduce launch, or for a consumer that takes them as they are
// (tzr_dense_adam_fused).
static int wgrad_partials(const float* d_dense, int64_t dense_stride, const float* d_sparse, int64_t sparse_stride, int F, int D,
                          int64_t B, const float* d_g1, int64_t g1_stride, int H, const float* d_scale, void* d_ws, int64_t ws_bytes,
                          WgReduceArgs* ra, void* stream) {
  const int hd = d_dense ? 1 : 0;
  const int n = F + hd;
  if (F <= 0 || B < 0 || (B > 0 && (!d_g1 || !d_sparse))) return TZR_ERR_INVALID;
  if (!tzr_dot_interaction_top_supported(F, D, hd, H)) return TZR_ERR_UNSUPPORTED;
  if ((sparse_stride & 3) || (hd && (dense_stride & 3)) ||
      ((reinterpret_cast<uintptr_t>(d_sparse) | reinterpret_cast<uintptr_t>(d_dense)) & 15) || (reinterpret_cast<uintptr_t>(d_g1) & 3))
    return TZR_ERR_INVALID;
  // (the kernel's sample offsets are 32-bit)
  if (B >= (int64_t)1 << 31 || B * sparse_stride >= (int64_t)1 << 32 || B * g1_stride >= (int64_t)1 << 32 ||
      (hd && B * dense_stride >= (int64_t)1 << 32) || sparse_stride < 0 || g1_stride < 0 || dense_stride < 0)
    return TZR_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int vw;
  wg_plan(n, ra->g, &vw);
  const int slices = B > 0 ? wg_slices(B) : 0;
  if (B > 0) {
    if (!d_ws || ws_bytes < (int64_t)slices * WG_H * vw * (int64_t)sizeof(float) || (reinterpret_cast<uintptr_t>(d_ws) & 15))
      return TZR_ERR_INVALID;
    WgArgs a;
    a.dense = d_dense; a.sparse = d_sparse; a.g1 = d_g1; a.part = static_cast<float*>(d_ws);
    a.dense_stride = dense_stride; a.sparse_stride = sparse_stride; a.g1_stride = g1_stride; a.B = B;
    a.n = n; a.hd = hd; a.slices = slices; a.vw = vw; a.debug = g_tzr_wg_debug;
#ifdef IT_PROF
    a.prof = g_tzr_it_prof;
#else
    a.prof = nullptr;
#endif
    for (int k = 0; k < WG_NG; ++k) a.g[k] = ra->g[k];
    if (!(g_tzr_wg_debug & 8)) hipLaunchKernelGGL(tzr_ia_wgrad_kernel, dim3(WG_NG * slices), dim3(WG_THREADS), 0, st, a);
    TZR_CHECK_LAUNCH();
  }
  ra->part = static_cast<const float*>(d_ws); ra->scale = d_scale; ra->dW = nullptr; ra->ldw = 0; ra->n = n; ra->slices = slices; ra->vw = vw;
  return TZR_OK;
}

extern "C" int tzr_dot_interaction_top_wgrad(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                             int64_t sparse_stride, int F, int D, int64_t B, const float* d_g1,
                                             int64_t g1_stride, int H, const float* d_scale, float* d_dW1, int64_t ldw,
                                             void* d_ws, int64_t ws_bytes, void* stream) {
  const int n = F + (d_dense ? 1 : 0);
  const int width = n * (n - 1) / 2 + WG_D * n;
  if (!d_dW1 || ldw < width || (reinterpret_cast<uintptr_t>(d_dW1) & 3)) return TZR_ERR_INVALID;
  WgReduceArgs ra;
  const int rc = wgrad_partials(d_dense, dense_stride, d_sparse, sparse_stride, F, D, B, d_g1, g1_stride, H, d_scale, d_ws, ws_bytes, &ra,
                                stream);
  if (rc != TZR_OK) return rc;
  ra.dW = d_dW1;
  ra.ldw = ldw;
  if (!(g_tzr_wg_debug & 16))
    hipLaunchKernelGGL(tzr_ia_wgrad_reduce_kernel, dim3((WG_H * width + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), ra);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// ... without the reduce launch: h_out (TzrWgradParts, include/tzrec_hip.h) describes the partial sums for tzr_dense_adam_fused,
// which adds them up on its way to the Adam update of W1 (B > 0; d_ws stays the caller's until that call has run).
extern "C" int tzr_dot_interaction_top_wgrad_parts(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                                   int64_t sparse_stride, int F, int D, int64_t B, const float* d_g1,
                                                   int64_t g1_stride, int H, const float* d_scale, void* d_ws, int64_t ws_bytes,
                                                   TzrWgradParts* h_out, void* stream) {
  static_assert(sizeof(WgReduceArgs) <= sizeof(TzrWgradParts), "the opaque blob of the C ABI holds the reduce arguments");
  if (!h_out || B <= 0) return TZR_ERR_INVALID;
  WgReduceArgs ra;
  const int rc = wgrad_partials(d_dense, dense_stride, d_sparse, sparse_stride, F, D, B, d_g1, g1_stride, H, d_scale, d_ws, ws_bytes, &ra,
                                stream);
  if (rc != TZR_OK) return rc;
  std::memset(h_out, 0, sizeof(*h_out));
  std::memcpy(h_out, &ra, sizeof(ra));
  return TZR_OK;
}
