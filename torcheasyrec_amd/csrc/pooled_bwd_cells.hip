// K6 / K7 in the "cells" form: a one-launch index plan and the apply that reads it (see pooled_bwd_cells.h for the design).
//
// Replaces, like pooled_bwd.hip / pooled_bwd_apply.hip, fbgemm's transpose_embedding_input + split_embedding_backward_codegen_*
// behind autograd of self.ebc(kjt) (/root/reference/tzrec/modules/embedding.py:930) with the optimizer fused by
// apply_optimizer_in_backward (/root/reference/tzrec/main.py:774-781; optim/optimizer_builder.py:30-97): per distinct (table, row)
// the gradient rows of its lookups are added in LOOKUP-POSITION order and the row is updated once.  The order of the additions
// is the four-launch plan's; the GROUPING of the partial sums follows each plan's own unit / tile boundaries, so the two agree to
// fp32 rounding, and each is bit-reproducible: a function of the ids alone (tests/test_pooled_parity.py: bwd_path "cells").
#include <tzr_gfx950.h>

#include "pooled_fwd_u1.h"

#include <algorithm>
#include <cstring>
#include <vector>

#ifdef IT_PROF  // scripts/build_prof_lib.sh: wall-clock stamps (100 MHz) of every workgroup of the cells apply, read back by tzr_cells_prof_dump
#define CELLS_PROF_WGS 4096
__device__ uint64_t g_cells_prof[CELLS_PROF_WGS * 16];
// slots: 0 start | 1 bounds known | 2 lookups in registers | 3 sorted, in LDS | 4 .. 7 a wave's tiles done | 8 end | 9 n | 10 split
#define CELLS_MARK(i) do { if (blockIdx.x < CELLS_PROF_WGS && threadIdx.x == 0) g_cells_prof[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#define CELLS_NOTE(i, v) do { if (blockIdx.x < CELLS_PROF_WGS && threadIdx.x == 0) g_cells_prof[blockIdx.x * 16 + (i)] = (uint64_t)(v); } while (0)
#define BWD_PROF_MARK(i) do { if ((i) == 2 && blockIdx.x < CELLS_PROF_WGS && (threadIdx.x & (TZR_WAVE - 1)) == 0) \
    g_cells_prof[blockIdx.x * 16 + 4 + threadIdx.x / TZR_WAVE] = wall_clock64(); } while (0)
extern "C" int tzr_cells_prof_dump(uint64_t* h_out, int n_wg) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_cells_prof), (size_t)std::min(n_wg, CELLS_PROF_WGS) * 16 * sizeof(uint64_t)) == hipSuccess ? 0 : -1;
}
#else
#define CELLS_MARK(i)
#define CELLS_NOTE(i, v)
#endif

#include "pooled_bwd_apply.h"
#include "pooled_bwd_cells.h"
#include "pooled_bwd_sort.h"

// ------------------------------------------------------------------------------------------------------------------------
// geometry (host)
// ------------------------------------------------------------------------------------------------------------------------
// (Measured and dropped in round 6 -- profiles/r06g, r06h: rows of small tables summed in streaming passes instead of gathered and
// sorted, and units sized by expected traffic under a budget of resident workgroups (more, smaller units for the 40 M-row tables):
// 92.6 - 100.9 us against 88 -- the tile loops run at the memory system's rate for the launch's TOTAL traffic whatever the
// partition, smaller units only add staging, and the streaming code cost the gathered units' path registers.)
namespace {

struct HostGeo {
  BwdCellsGeo g;
  std::vector<BwdCellChunk> chunks;
  std::vector<BwdCellUnit> units;
  std::vector<uint32_t> fstart;
  std::vector<int32_t> fkey, fbo;
  std::vector<uint16_t> bnd;  // boundary list: bucket numbers, table after table
};

inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

// bwd_bucket_params, host side
inline void bucket_params_host(int64_t rows, int* nb, uint64_t* mult) {
  if (rows <= BWD_NB) {
    *nb = rows < 1 ? 1 : (int)rows;
    *mult = 1ull << 32;
  } else {
    *nb = BWD_NB;
    *mult = ((uint64_t)BWD_NB << 32) / (uint64_t)rows;
  }
}

// TZR_OK, or TZR_ERR_UNSUPPORTED when this (tables, batch size) is not a case for the cells plan
int build_geometry(const TzrTable* tabs, int T, const TzrFeature* feats, int F, int64_t B, int max_dim, HostGeo& H) {
  if (T <= 0 || F <= 0 || B <= 0 || max_dim <= 0 || max_dim > BWD_MAXDIM || (max_dim & 3)) return TZR_ERR_INVALID;
  H.fstart.assign(F + 1, 0);
  H.fkey.assign(F, 0);
  H.fbo.assign(F, 0);
  std::vector<int> seen(F, 0);
  for (int f = 0; f < F; ++f) {
    const TzrFeature& ft = feats[f];
    if (ft.order < 0 || ft.order >= F || seen[ft.order]) return TZR_ERR_INVALID;
    seen[ft.order] = 1;
    H.fstart[ft.order] = ft.table < 0 ? 0u : (uint32_t)B;  // one id per bag
    H.fkey[ft.order] = ft.key;
    H.fbo[ft.order] = f;
  }
  {
    uint64_t run = 0;
    for (int o = 0; o < F; ++o) {
      const uint32_t n = H.fstart[o];
      H.fstart[o] = (uint32_t)run;
      run += n;
    }
    if (run >= (1ull << 32)) return TZR_ERR_UNSUPPORTED;
    H.fstart[F] = (uint32_t)run;
  }
  const int64_t N = H.fstart[F];
  if (N == 0) return TZR_ERR_UNSUPPORTED;
  const int ch = bwd_pick_ch(N);
  int64_t n_chunks = 0;
  std::vector<int64_t> cfirst(T + 1, 0), nt(T, 0);
  for (int t = 0; t < T; ++t) {
    const TzrTable& tb = tabs[t];
    if (tb.n_feats < 0 || (tb.n_feats > 0 && (tb.first_order < 0 || tb.first_order + tb.n_feats > F))) return TZR_ERR_INVALID;
    const int64_t n = tb.n_feats > 0 ? (int64_t)H.fstart[tb.first_order + tb.n_feats] - (int64_t)H.fstart[tb.first_order] : 0;
    const int64_t C = (n + ch - 1) / ch;
    if (C > BWD_CELLS_MAXC) return TZR_ERR_UNSUPPORTED;
    if (n > 0 && (tb.rows <= 0 || tb.rows > (1LL << 32))) return TZR_ERR_UNSUPPORTED;
    nt[t] = n;
    cfirst[t] = n_chunks;
    n_chunks += C;
  }
  cfirst[T] = n_chunks;
  if (n_chunks > bwd_max_chunks(N, T, ch)) return TZR_ERR_UNSUPPORTED;

  // ---- units of a bucketed table: ranges of whole buckets, ceil(512 / U) of them at most -- the smallest U whose largest range
  // still expects <= target lookups (evenly drawn ids: 6 standard deviations under the LDS capacity) ----
  const int64_t target = BWD_CELLS_TARGET;
  std::vector<int64_t> U(T, 0);
  for (int t = 0; t < T; ++t) {
    if (nt[t] == 0 || tabs[t].rows <= BWD_NB) continue;
    const int64_t per = (target * BWD_NB) / nt[t];  // buckets a unit may hold
    if (per < 1) return TZR_ERR_UNSUPPORTED;        // (one bucket already expects more than a unit: the table is too big for this batch size)
    U[t] = (BWD_NB + per - 1) / per;
  }

  int64_t n_recs = 0, n_counters = 0;
  for (int t = 0; t < T; ++t) {
    const TzrTable& tb = tabs[t];
    const int64_t C = cfirst[t + 1] - cfirst[t];
    if (C == 0) continue;
    const int64_t ts = H.fstart[tb.first_order], te = H.fstart[tb.first_order + tb.n_feats];
    const int64_t n = te - ts;
    int nb;
    uint64_t mult;
    bucket_params_host(tb.rows, &nb, &mult);
    const int64_t fbase = tb.n_feats == 1 ? (int64_t)H.fkey[tb.first_order] * B : -1;
    const int32_t bnd0 = (int32_t)H.bnd.size();
    BwdCellUnit u;
    std::memset(&u, 0, sizeof(u));
    u.tb = tb;
    u.feat = H.fbo[tb.first_order];
    u.ts = (uint32_t)ts;
    u.rec = u.rec0 = u.counter = -1;
    u.crel0 = 0;
    u.ncell = (int32_t)C;
    if (tb.rows > BWD_NB) {
      // bucketed table: U units, each a range of the 512 buckets over all chunks
      const int64_t Ut = U[t];
      for (int64_t j = 0; j <= Ut; ++j) H.bnd.push_back((uint16_t)(j * nb / Ut));
      for (int64_t j = 0; j < Ut; ++j) {
        u.b0 = (int32_t)(j * nb / Ut);
        u.i0 = bnd0 + (int32_t)j;
        u.i1 = u.i0 + 1;
        H.units.push_back(u);
      }
    } else {
      const int64_t e = (n + tb.rows - 1) / tb.rows;
      if (e > target) {
        // gathered units of ONE row each, the row split over K chunk ranges: every row is a boundary
        for (int64_t r = 0; r <= tb.rows; ++r) H.bnd.push_back((uint16_t)r);
        const int64_t m = std::max<int64_t>(1, (target * tb.rows * C) / n);
        const int64_t K = std::min<int64_t>(C, (C + m - 1) / m);
        for (int64_t r = 0; r < tb.rows; ++r) {
          for (int64_t k = 0; k < K; ++k) {
            u.crel0 = (int32_t)(k * C / K);
            u.ncell = (int32_t)((k + 1) * C / K) - u.crel0;
            u.b0 = (int32_t)r;
            u.nrows = 0;
            u.i0 = bnd0 + (int32_t)r;
            u.i1 = u.i0 + 1;
            u.split = K > 1 ? (int32_t)K : 0;
            u.rec = K > 1 ? (int32_t)(n_recs + k) : -1;
            u.rec0 = K > 1 ? (int32_t)n_recs : -1;
            u.counter = K > 1 ? (int32_t)n_counters : -1;
            H.units.push_back(u);
          }
          if (K > 1) {
            n_recs += K;
            n_counters += 1;
          }
        }
      } else {
        // few lookups per row: whole rows grouped up to the target, gathered and sorted like a bucket range
        const int64_t rpu = std::max<int64_t>(1, (target * tb.rows) / n);
        int32_t g = 0;
        for (int64_t r = 0; r < tb.rows; r += rpu, ++g) {
          H.bnd.push_back((uint16_t)r);
          u.b0 = (int32_t)r;
          u.i0 = bnd0 + g;
          u.i1 = u.i0 + 1;
          H.units.push_back(u);
        }
        H.bnd.push_back((uint16_t)tb.rows);
      }
    }
    const int32_t nbnd = (int32_t)H.bnd.size() - bnd0;
    for (int64_t c = 0; c < C; ++c) {
      BwdCellChunk cd;
      std::memset(&cd, 0, sizeof(cd));
      cd.t = t;
      cd.nb = nb;
      cd.s = ts + c * ch;
      cd.e = std::min(te, cd.s + ch);
      cd.ts = ts;
      cd.mult = mult;
      cd.rows = tb.rows;
      cd.fbase = fbase;
      cd.bnd0 = bnd0;
      cd.nbnd = nbnd;
      H.chunks.push_back(cd);
    }
  }
  BwdCellsGeo& g = H.g;
  std::memset(&g, 0, sizeof(g));
  g.n_chunks = n_chunks;
  g.n_units = (int64_t)H.units.size();
  g.n_recs = n_recs;
  g.n_counters = n_counters;
  g.n_feats = F;
  g.max_dim = max_dim;
  g.ch = ch;
  g.n_positions = N;
  g.n_bnd = (int64_t)H.bnd.size();
  // (the boundary table lives in the plan's histogram area: max_chunks rows of BWD_NB uint32)
  if (g.n_bnd * BWD_CELLS_MAXC * 2 > bwd_max_chunks(N, T, ch) * (int64_t)BWD_NB * 4) return TZR_ERR_UNSUPPORTED;
  int64_t off = align256(sizeof(BwdCellsGeo));
  g.off_chunks = off;
  off = align256(off + n_chunks * (int64_t)sizeof(BwdCellChunk));
  g.off_units = off;
  off = align256(off + g.n_units * (int64_t)sizeof(BwdCellUnit));
  g.off_fstart = off;
  off = align256(off + (F + 1) * 4);
  g.off_fkey = off;
  off = align256(off + F * 4);
  g.off_fbo = off;
  off = align256(off + F * 4);
  g.off_bnd = off;
  off = align256(off + g.n_bnd * 2);
  g.off_recs = off;
  off = align256(off + std::max<int64_t>(1, n_recs) * max_dim * 4);
  g.off_rcount = off;
  off = align256(off + std::max<int64_t>(1, n_recs) * 4);
  g.off_counters = off;
  off = align256(off + std::max<int64_t>(1, n_counters) * 4);
  g.off_overflow = off;
  off = align256(off + 4 * (BWD_CELLS_OVF_LIST + 2 * g.n_units));
  g.bytes = off;
  return TZR_OK;
}

}  // namespace

// Host image of the geometry buffer for (h_tables, h_feats, B): every bag holds exactly one id.  h_out == NULL: only the size.
// out_info[0] = bytes, [1] = chunks (= workgroups of the plan launch), [2] = units (= workgroups of the apply launch),
// [3] = table-major positions, [4] = positions per chunk, [5] = partial-sum records, [6] = byte offset of the overflow word.
// TZR_ERR_UNSUPPORTED: not a case for the cells plan (a table with more than BWD_CELLS_MAXC chunks, no lookups, ...).
extern "C" int tzr_bwd_cells_geometry(const TzrTable* h_tables, int n_tables, const TzrFeature* h_feats, int n_feats, int64_t B,
                                      int max_dim, void* h_out, size_t out_bytes, int64_t* out_info8) {
  if (!h_tables || !h_feats || !out_info8) return TZR_ERR_INVALID;
  HostGeo H;
  const int rc = build_geometry(h_tables, n_tables, h_feats, n_feats, B, max_dim, H);
  if (rc != TZR_OK) return rc;
  const BwdCellsGeo& g = H.g;
  out_info8[0] = g.bytes;
  out_info8[1] = g.n_chunks;
  out_info8[2] = g.n_units;
  out_info8[3] = g.n_positions;
  out_info8[4] = g.ch;
  out_info8[5] = g.n_recs;
  out_info8[6] = g.off_overflow;
  out_info8[7] = 0;
  if (!h_out) return TZR_OK;
  if ((int64_t)out_bytes < g.bytes) return TZR_ERR_WORKSPACE;
  char* b = static_cast<char*>(h_out);
  std::memset(b, 0, (size_t)g.bytes);
  std::memcpy(b, &g, sizeof(g));
  std::memcpy(b + g.off_chunks, H.chunks.data(), H.chunks.size() * sizeof(BwdCellChunk));
  std::memcpy(b + g.off_units, H.units.data(), H.units.size() * sizeof(BwdCellUnit));
  std::memcpy(b + g.off_fstart, H.fstart.data(), H.fstart.size() * 4);
  std::memcpy(b + g.off_fkey, H.fkey.data(), H.fkey.size() * 4);
  std::memcpy(b + g.off_fbo, H.fbo.data(), H.fbo.size() * 4);
  std::memcpy(b + g.off_bnd, H.bnd.data(), H.bnd.size() * 2);
  return TZR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// partition: every chunk ordered by bucket in place + its bucket starts.  No workgroup talks to another.
// ------------------------------------------------------------------------------------------------------------------------
struct BwdPartLds {  // 13.3 KB
  BwdRankLds<BWD_NB> L;
  uint2 stage[BWD_CH];
};

__device__ __forceinline__ void bwd_cells_partition_body(const BwdCellsView& V, const TzrTable* __restrict__ tables, const BwdSrcArgs& A,
                                                         uint2* __restrict__ slab, uint16_t* __restrict__ bnd, int ch, unsigned chunk,
                                                         BwdPartLds& PL) {
  auto& L = PL.L;
  auto& stage = PL.stage;
  const BwdCellChunk cd = V.chunks[chunk];
  if (cd.t < 0) return;
  const int n = (int)(cd.e - cd.s);
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  constexpr int kRounds = BWD_CH / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds], dig[kRounds], dest[kRounds];
  uint32_t vmask = 0;
  if (cd.fbase >= 0) {  // (workgroup-uniform) one key reads the table: every id load unconditional, position clamped (bwd_elem_one)
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int lp = wv * pw + r * TZR_WAVE + lane;
      const int lc = lp < n ? lp : n - 1;
      const int64_t i = cd.fbase + (cd.s - cd.ts) + lc;
      int64_t id = A.values[i];
      if ((uint64_t)id >= (uint64_t)cd.rows) id = 0;  // memory safety; K4 reports / clamps
      kreg[r] = (uint32_t)id;
      sreg[r] = (uint32_t)i;
      dig[r] = bwd_bucket(kreg[r], cd.mult);
      vmask |= (r < rounds && lp < n) ? 1u << r : 0u;
    }
  } else {
    const TzrTable tb = tables[cd.t];
    BwdGeo G;
    G.fstart = V.fstart;
    G.fkey = V.fkey;
    G.tchunk = nullptr;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int lp = wv * pw + r * TZR_WAVE + lane;
      kreg[r] = sreg[r] = dig[r] = 0u;
      if (r < rounds && lp < n) {
        vmask |= 1u << r;
        int64_t kk;
        bwd_elem0(G, tb, A, cd.s + lp, &kreg[r], &sreg[r], &kk);
        dig[r] = bwd_bucket(kreg[r], cd.mult);
      }
    }
  }
  bwd_rank_tile<BWD_NB, kRounds>(dig, vmask, rounds, bwd_bits((uint32_t)cd.nb - 1u), L, dest);
#pragma unroll
  for (int r = 0; r < kRounds; ++r)
    if ((vmask >> r) & 1u) stage[dest[r]] = make_uint2(kreg[r], sreg[r]);
  // this chunk's column of the boundary table: where each unit boundary of the table (a bucket number) starts inside the chunk
  // (buckets at or above nb hold nothing: their start is n)
  {
    const int col = (int)((cd.s - cd.ts) / (cd.e > cd.s ? (int64_t)ch : 1));
    for (int i = threadIdx.x; i < cd.nbnd; i += BWD_THREADS)
      bnd[(size_t)(cd.bnd0 + i) * BWD_CELLS_MAXC + col] = L.lstart[V.bnd_bucket[cd.bnd0 + i]];
  }
  __syncthreads();
  uint2* out = slab + cd.s;
  for (int i = threadIdx.x; i < n; i += BWD_THREADS) out[i] = stage[i];  // one coalesced 8-byte store per lookup
}

__global__ __launch_bounds__(BWD_THREADS) void tzr_bwd_cells_partition_kernel(
    BwdCellsView V, const TzrTable* __restrict__ tables, BwdSrcArgs A, uint2* __restrict__ slab, uint16_t* __restrict__ bnd, int ch) {
  __shared__ BwdPartLds PL;
  bwd_cells_partition_body(V, tables, A, slab, bnd, ch, blockIdx.x, PL);
}

// The plan in the FORWARD's launch.  The partition needs the ids and nothing else, and what it does with them is LDS and ALU work
// (rank 1 024 lookups by bucket, 16 bytes of HBM traffic per lookup) while the one-id forward next to it is bound by the rate of its
// 64-byte row requests: as a launch of its own the plan is 12 us of the step in which HBM idles; as a second stream inside the step's
// graph it costs more than it hides (fork + join: +10 us, scripts/r06/gpu_async_plan.sh).  Here the two kinds of workgroup share
// one grid, chosen by the workgroup index alone.  Measured at B = 65 536 (step, driver flags, profiles/r06au): the plan's
// workgroups BEHIND the forward's -3.5 us (they fill the slots the forward's last workgroups leave), in front of them -2,
// alternating with them +13 -- the forward is bound by its resident waves (98 registers: four workgroups per CU), and every slot a
// partition workgroup holds is one fewer gather stream in flight.
static_assert(FWD_THREADS == BWD_THREADS, "one workgroup size for both kinds");
struct FwdPlanArgs {
  const TzrTable* ftables;
  const TzrFeature* ffeats;
  const TzrSlot* slots;
  const int64_t* values;
  int64_t B;
  int32_t n_slots, tile_b;
  uint32_t n_fwd, n_part;
  const TzrTable* btables;
  uint2* slab;
  uint16_t* bnd;
  int32_t ch, order;
};

// Residency: the two kinds of workgroup share ONE LDS area (the larger: 22.8 KB) and the forward keeps 4 gathers in flight per
// thread instead of 8 (52 registers), so that SEVEN workgroups fit a CU instead of four.  The forward alone does not care (49.6-54.9
// us at every residency, profiles/r06j); the pair does: forward + plan 58.4-59.4 us at 4 workgroups / 8 gathers, 56.0-56.5 at 5 / 8,
// 54.3-55.3 at 6 or 7 / 4 (profiles/r06aw).  Tiles of 64 samples (1 024 forward workgroups: all resident at once, the plan's
// workgroups start in the slots beside them): 52.0-53.1 against 55.2 at 32; 72 / 78 / 96 / 128: 56-60.
#ifndef FWD_PLAN_WAVES
#define FWD_PLAN_WAVES 7
#endif
#ifndef FWD_PLAN_UNROLL
#define FWD_PLAN_UNROLL 4
#endif
union FwdPlanLds {  // (one kind of workgroup or the other: the larger of the two, so that LDS does not bound residency)
  Fwd1Lds f;
  BwdPartLds p;
};

__global__ __launch_bounds__(BWD_THREADS) TZR_WAVES_PER_EU(FWD_PLAN_WAVES) void tzr_pooled_fwd_u1_cells_plan_kernel(FwdPlanArgs a, FwdDsts dsts,
                                                                                                          BwdCellsView V, BwdSrcArgs A) {
  __shared__ FwdPlanLds S;
  const uint32_t i = blockIdx.x;
  const uint32_t both = 2u * min(a.n_fwd, a.n_part);
  bool part;
  uint32_t idx;
  if (a.order == 1) {  // the plan's workgroups first
    part = i < a.n_part;
    idx = part ? i : i - a.n_part;
  } else if (a.order == 2) {  // ... last
    part = i >= a.n_fwd;
    idx = part ? i - a.n_fwd : i;
  } else if (i < both) {
    part = (i & 1u) != 0u;
    idx = i >> 1;
  } else {
    part = a.n_part > a.n_fwd;
    idx = i - (both >> 1);
  }
  if (part)  // (workgroup-uniform)
    bwd_cells_partition_body(V, a.btables, A, a.slab, a.bnd, a.ch, idx, S.p);
  else
    fwd_u1_body<FWD_PLAN_UNROLL>(a.ftables, a.ffeats, a.slots, a.n_slots, a.values, a.B, a.tile_b, dsts, idx, 0u, S.f);
}

// ------------------------------------------------------------------------------------------------------------------------
// apply
// ------------------------------------------------------------------------------------------------------------------------
union BwdCellsLds {
  BwdSortLds S;
  BwdUnitLds U;
};

// lookup `i` (0 .. n) of a unit whose cells' exclusive counts / slab starts lie in LDS
__device__ __forceinline__ uint2 bwd_cells_elem(const uint2* __restrict__ slab, const uint32_t* cpre, const uint32_t* cbase,
                                                int ncell, uint32_t i) {
  int lo = 0;
#pragma unroll
  for (int step = BWD_CELLS_MAXC / 2; step > 0; step >>= 1) {  // largest cell with cpre[cell] <= i (empty cells are skipped)
    const int cand = lo + step;
    if (cand < ncell && cpre[cand] <= i) lo = cand;
  }
  return slab[cbase[lo] + (i - cpre[lo])];
}

// Partial sum of one unit of a split row -> its record; the last of the row's units to arrive adds the records in unit order
// and updates the row.  Wave 0; `sum` valid in the lanes < dim / 4.
template <bool ADAM>
__device__ __forceinline__ void bwd_cells_combine(const BwdCellUnit& u, const BwdCellsView& V, const BwdOpt& opt, float lr,
                                                  int max_dim, float4 sum, uint32_t count, int lane) {
  const int lg = u.tb.dim >> 2;
  if (lane < lg) bwd_publish4(V.recs + (size_t)u.rec * max_dim + 4 * lane, sum);
  if (lane == 0) tzr_publish_u32(V.rcount + u.rec, count);
  tzr_drain_stores();
  int last = 0;
  if (lane == 0) last = tzr_arrive(V.counters + u.counter) == (uint32_t)(u.split - 1) ? 1 : 0;
  last = __shfl(last, 0, TZR_WAVE);
  if (!last) return;
  if (lane == 0) tzr_publish_u32(V.counters + u.counter, 0u);  // the counter is zero again when the launch ends
  // the row's records a wave's worth at a time -- 64 / (D / 4) records per round trip -- and added in unit order by shuffles (one
  // record after the other this walk was a chain of cache-bypassing loads: ~1.5 us each, 21 of them for a row of the 3-row
  // Criteo table at B = 65 536, in the workgroup that is the last of its row to arrive: the lesson of bwd_stitch_unit)
  const int gw = TZR_WAVE / lg;
  const int gi = lane / lg, c = lane - gi * lg;
  float4 tot = tzr_zero4();
  uint32_t cnt = 0;
  for (int k0 = 0; k0 < u.split; k0 += gw) {
    const int k = k0 + gi;
    const bool in = gi < gw && k < u.split;
    float4 v = tzr_zero4();
    uint32_t cn = 0;
    if (in) {
      v = bwd_consume4(V.recs + (size_t)(u.rec0 + k) * max_dim + 4 * c);
      cn = tzr_consume_u32(V.rcount + u.rec0 + k);
    }
    const int nk = min(gw, u.split - k0);
    for (int g = 0; g < nk; ++g) {  // (wave-uniform)
      const float4 piece = bwd_shfl4(v, g * lg + (lane < lg ? lane : 0));
      if (lane < lg) tot = tzr_add4(tot, piece);
      cnt += (uint32_t)__shfl((int)cn, g * lg, TZR_WAVE);
    }
  }
  if (cnt) bwd_apply_row_wave<ADAM>(u.tb, opt, lr, (uint32_t)u.b0, tot, lane);  // (a row nobody looked up is not touched)
}

// Sum of the gradient rows of the lookups of ONE row held by the cells in LDS (cpre / cbase, n lookups; `row`: only lookups of
// that row count -- pass BWD_SENT when the cells hold nothing else), in lookup order: lane group q takes lookups q, q + G,
// q + 2 G, ... in order (UF of them in flight together: their {row, position} pairs first, then their gradient rows), the groups'
// sums are added in group order -- a function of the ids alone.  All threads call; the result (and the number of lookups) in
// wave 0, lanes < dim / 4.  `red`: 4 KB of LDS, `sm`: 2 * BWD_WAVES + 64 words.
// FAST (compile time): the table is fp32-gradient "one key, one buffer" (bwd_reduce_unit's FK != 0 case): the gradient row of
// lookup position i is fgp + (i - fkb) * fgs -- no descriptor walk, no branch around a load.
template <bool FAST>
__device__ __forceinline__ float4 bwd_cells_stream_row(
    const BwdCellUnit& u, const BwdCellsView& V, const uint2* __restrict__ slab, const uint32_t* cpre, const uint32_t* cbase,
    int ncell, int n, uint32_t row, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B,
    int grad_mode, const TzrDst* sG, float* red, uint32_t* sm, uint32_t* count_out) {
  constexpr int UF = 4;
  const TzrTable& tb = u.tb;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int lg = tb.dim >> 2, gw = TZR_WAVE / lg;
  const int gi = lane / lg, c = lane - gi * lg;
  const bool lane_on = gi < gw;
  const int groups = gw * BWD_WAVES, q = wv * gw + gi;
  const bool single = tb.n_feats == 1;
  const BwdSrc one = bwd_resolve(feats + V.feat_by_order[tb.first_order], sG);
  const float* const fgp = grad_mode == 1 ? reinterpret_cast<const float*>(sG[0].ptr) : one.gp0;
  const int64_t fgs = grad_mode == 1 ? sG[0].stride : one.gs0;
  const uint32_t fkb = grad_mode == 0 ? (uint32_t)feats[V.feat_by_order[tb.first_order]].key * (uint32_t)B : 0u;
  float4 acc = tzr_zero4();
  uint32_t cnt = 0;
  const int trips = (n + groups * UF - 1) / (groups * UF);
  // the {row, position} pairs of a trip are fetched a trip AHEAD of their gradient rows: one round trip per trip, not two
  uint2 e[UF];
#pragma unroll
  for (int k = 0; k < UF; ++k) {
    const int i = k * groups + q;
    e[k] = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)(i < n ? i : n - 1));  // (clamped, unconditional)
  }
  for (int j = 0; j < trips; ++j) {
    float4 g[UF];
#pragma unroll
    for (int k = 0; k < UF; ++k) {
      if constexpr (FAST) g[k] = tzr_ldg4(fgp + (int64_t)(e[k].y - fkb) * fgs + 4 * c);
      else g[k] = bwd_lookup_grad(feats, tb, V.feat_by_order, sG, one, single, grad_mode, nullptr, weights, nullptr, B, 1, e[k].y, c);
    }
    uint2 en[UF];
#pragma unroll
    for (int k = 0; k < UF; ++k) {
      const int i = ((j + 1) * UF + k) * groups + q;
      en[k] = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)(i < n ? i : n - 1));
    }
#pragma unroll
    for (int k = 0; k < UF; ++k) {
      const int i = (j * UF + k) * groups + q;
      if (lane_on && i < n && (row == BWD_SENT || e[k].x == row)) {
        acc = tzr_add4(acc, g[k]);
        cnt += 1;
      }
      e[k] = en[k];
    }
  }
  __syncthreads();  // (red / sm may still be read from an earlier call)
  if (lane_on) {
    float* r = red + (size_t)(q * lg + c) * 4;
    r[0] = acc.x; r[1] = acc.y; r[2] = acc.z; r[3] = acc.w;
    if (c == 0) sm[2 * BWD_WAVES + q] = cnt;
  }
  __syncthreads();
  float4 tot = tzr_zero4();
  uint32_t ct = 0;
  if (wv == 0) {
    for (int g2 = 0; g2 < groups; ++g2) {
      if (lane < lg) {
        const float* r = red + (size_t)(g2 * lg + lane) * 4;
        tot = tzr_add4(tot, make_float4(r[0], r[1], r[2], r[3]));
      }
      ct += sm[2 * BWD_WAVES + g2];
    }
  }
  *count_out = ct;
  return tot;
}

// A unit with more lookups than the LDS unit holds (skewed ids: the geometry expects evenly filled buckets) is done piece by
// piece, the way the one-launch backward walks a row range that does not fit (pooled_bwd_direct.hip): a histogram of the unit's
// lookups over 256 sub-ranges of the rows still to do gives the longest prefix of sub-ranges that fits one LDS unit; those
// lookups are compacted into the exchange buffer in arrival order and take the ordinary sort + reduction (the SAME call sites as
// an ordinary unit's, in a workgroup of its own role: bwd_cells_worker); a single row with more lookups than a unit is summed in a streaming
// pass.  Every pass reads the unit's cells again (L2).  Correct for any ids, slower than the exact plan's tile-parallel heavy
// buckets -- the caller sees the overflow word move and sends this id distribution there.
//
// bwd_cells_next_piece: advances `cur` over the unit's rows [cur, khi) until a piece is staged -- L.S.pk / ps[0 .. np) hold its
// lookups in arrival order, returns np > 0 -- or the rows are exhausted (0).  Rows streamed on the way are updated here.
template <bool ADAM>
__device__ __forceinline__ int bwd_cells_next_piece(
    const BwdCellUnit& u, const BwdCellsView& V, const uint2* __restrict__ slab, const uint32_t* cpre, const uint32_t* cbase,
    int ncell, int n, uint32_t& cur, uint32_t khi, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B,
    int grad_mode, const BwdOpt& opt, float lr, BwdCellsLds& L, const TzrDst* sG) {
  const TzrTable& tb = u.tb;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  float* const red = reinterpret_cast<float*>(&L.S.L);  // 4 KB (the ranking rows: free outside the sort)
  uint32_t* const hist = L.S.gstart;
  uint32_t* const sm = L.S.gstart + 264;                // (behind the 257 histogram words)
  constexpr int SB = BWD_THREADS;  // sub-ranges per pass: one per thread
  while (cur < khi) {
    uint32_t lim = khi;
    for (;;) {
      const uint32_t span = lim - cur;
      const uint64_t m2 = span <= (uint32_t)SB ? (1ull << 32) : (((uint64_t)SB << 32) / span);
      __syncthreads();
      hist[threadIdx.x] = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += BWD_THREADS) {
        const uint32_t k = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)i).x;
        if (k >= cur && k < lim) atomicAdd(&hist[(uint32_t)(((uint64_t)(k - cur) * m2) >> 32)], 1u);
      }
      __syncthreads();
      bwd_block_scan(hist, SB, L.S.wtot);  // exclusive starts; hist[SB] = lookups in [cur, lim)
      const uint32_t in_range = hist[SB];
      // p = leading sub-ranges that fit one unit together: hist is monotone, p = #{ j in 1 .. SB : hist[j] <= UMAX }
      uint32_t fit = hist[threadIdx.x + 1] <= (uint32_t)BWD_UMAX ? 1u : 0u;
      for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) fit += (uint32_t)__shfl_xor((int)fit, m, TZR_WAVE);
      if (lane == 0) sm[wv] = fit;
      __syncthreads();
      uint32_t p = 0;
#pragma unroll
      for (int w = 0; w < BWD_WAVES; ++w) p += sm[w];
      if (in_range == 0) {
        cur = lim;
        break;
      }
      if (p >= 1) {
        const uint32_t end = p >= (uint32_t)SB ? lim : min(lim, cur + (uint32_t)((((uint64_t)p << 32) + m2 - 1) / m2));
        // the piece's lookups compacted in arrival order into the exchange buffer
        __syncthreads();  // (hist / sm are read above)
        uint32_t total = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int base = 0; base < n; base += BWD_THREADS) {
          const int i = base + (int)threadIdx.x;
          const uint2 e = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)(i < n ? i : n - 1));
          const bool m = i < n && e.x >= cur && e.x < end;
          const unsigned long long bal = __ballot(m);
          if (lane == 0) sm[BWD_WAVES + wv] = (uint32_t)__popcll(bal);
          __syncthreads();
          uint32_t pre = total, all = 0;
#pragma unroll
          for (int w = 0; w < BWD_WAVES; ++w) {
            const uint32_t cw = sm[BWD_WAVES + w];
            if (w < wv) pre += cw;
            all += cw;
          }
          const uint32_t at = pre + (uint32_t)__popcll(bal & lt);
          if (m && at < (uint32_t)BWD_UMAX) {
            L.S.pk[at] = e.x;
            L.S.ps[at] = e.y;
          }
          total += all;
          __syncthreads();
        }
        cur = end;
        if (total == 0) break;  // (leading sub-ranges without a lookup in front of one that does not fit: nothing to reduce)
        return (int)min(total, (uint32_t)BWD_UMAX);
      }
      // the first sub-range alone does not fit
      const uint32_t end0 = min(lim, cur + (uint32_t)((((uint64_t)1 << 32) + m2 - 1) / m2));
      if (end0 - cur <= 1u) {  // one row with more lookups than a unit: streamed
        uint32_t cnt;
        const float4 sum = bwd_cells_stream_row<false>(u, V, slab, cpre, cbase, ncell, n, cur, feats, weights, B, grad_mode, sG, red, sm, &cnt);
        if (wv == 0 && cnt) bwd_apply_row_wave<ADAM>(tb, opt, lr, cur, sum, lane);
        cur += 1;
        break;
      }
      lim = end0;  // narrow
    }
  }
  return 0;
}

// The bounds of the cells of boundary rows [i0, i1) x the unit's chunks into LDS (cpre: exclusive counts, cpre[ncell] = the
// lookups; cbase: slab position of a cell's first lookup): two CONTIGUOUS 2-byte-per-chunk reads of the boundary table.  All
// threads call; ends with a barrier.
__device__ __forceinline__ int bwd_cells_bounds(const BwdCellUnit& u, int i0, int i1, const uint16_t* __restrict__ bnd, int ch,
                                                uint32_t* cpre, uint32_t* cbase, uint32_t* wtot) {
  const int ncell = u.ncell;
  if ((int)threadIdx.x < ncell) {
    const int col = u.crel0 + (int)threadIdx.x;
    const uint32_t a = bnd[(size_t)i0 * BWD_CELLS_MAXC + col], b = bnd[(size_t)i1 * BWD_CELLS_MAXC + col];
    cpre[threadIdx.x] = b - a;
    cbase[threadIdx.x] = u.ts + (uint32_t)col * (uint32_t)ch + a;
  }
  __syncthreads();
  bwd_block_scan(cpre, ncell, wtot);
  return (int)cpre[ncell];
}

// The ordinary unit from its lookups in registers: sort by (row id, arrival) in LDS, reduction, the split rows' records.  `np`
// lookups, element r of a lane = arrival position wv * pw + r * 64 + lane (bwd_stage_unit's fused form).
template <bool ADAM, int FK, int NT, int MAXR>
__device__ __forceinline__ void bwd_cells_sort_reduce(
    const BwdCellUnit& u, const BwdCellsView& V, uint32_t (&kreg)[MAXR], uint32_t (&sreg)[MAXR], uint32_t vmask, uint32_t kmin,
    uint32_t kmax, int np, int ft_dst, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B,
    int grad_mode, const BwdOpt& opt, float lr, int max_dim, BwdCellsLds& L, const TzrDst* sG) {
  const TzrTable& tb = u.tb;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int pw = bwd_wave_span(np);
  const int rounds = pw / TZR_WAVE;
  uint32_t dest[MAXR];
  for (int m = TZR_WAVE >> 1; m > 0; m >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, m, TZR_WAVE));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, m, TZR_WAVE));
  }
  if (lane == 0) {
    L.S.smm[wv] = kmin;
    L.S.smm[BWD_WAVES + wv] = kmax;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    kmin = min(kmin, L.S.smm[w]);
    kmax = max(kmax, L.S.smm[BWD_WAVES + w]);
  }
  __syncthreads();  // smm is reused by the core
#ifdef CELLS_ALWAYS_GROUPED  // (timing experiment: the sort as bwd_stage_unit runs it)
  const bool one_row = false, grouped = true;
#else
  // ONE row -- a slice of a split row, a hot row alone in its unit: arrival order is the order.  Grouping by the low bits of the
  // row id wants many distinct rows: a unit of an exact table (a handful of rows, hundreds of lookups each) would send every
  // lookup to one of a few LDS counters; it takes the counting pass(es) over its few bits
  const bool one_row = kmin == kmax, grouped = tb.rows > BWD_NB;  // (workgroup-uniform)
#endif
  if (one_row) {
#pragma unroll
    for (int r = 0; r < MAXR; ++r) dest[r] = (uint32_t)(wv * pw + r * TZR_WAVE + lane);
  } else {
    bwd_sort_core<MAXR>(kreg, sreg, vmask, pw, rounds, kmin, max(1, bwd_bits(kmax - kmin)), grouped, L.S, dest);
  }
  __syncthreads();  // the sort's LDS is dead: the unit's arrays take its place
#pragma unroll
  for (int r = 0; r < MAXR; ++r)
    if ((vmask >> r) & 1u) {
      L.U.sK[dest[r] + 1] = kreg[r];
      L.U.sS[dest[r]] = sreg[r];
    }
  if (threadIdx.x == 0) {
    // a unit (a piece) owns its rows -- no run continues outside it -- except one slice of a split row: open on both sides,
    // so that the reduction leaves the slice's whole sum as its leading piece and updates nothing
    const uint32_t edge = u.split > 0 ? (uint32_t)u.b0 : BWD_SENT;
    L.U.sK[0] = edge;
    L.U.sK[np + 1] = edge;
  }
  __syncthreads();
  CELLS_MARK(3);
  auto tail = [&](unsigned cf, uint32_t okey, const float4& clead, const float4& osum) {  // wave 0
    (void)cf;
    (void)okey;
    (void)osum;
    if (u.split > 0) bwd_cells_combine<ADAM>(u, V, opt, lr, max_dim, clead, (uint32_t)np, lane);
  };
  if constexpr (FK != 0) {
    const bool fast = tb.w_dtype == TZR_DT_F32 && (grad_mode == 1 || (tb.n_feats == 1 && ft_dst == 1));  // (workgroup-uniform)
    if (fast)
      bwd_reduce_unit<false, NT, FK>(tb, feats, V.feat_by_order, nullptr, nullptr, weights, B, 1, grad_mode, opt, L.U, sG, np, tail);
    else
      bwd_reduce_unit<false, 1, 0>(tb, feats, V.feat_by_order, nullptr, nullptr, weights, B, 1, grad_mode, opt, L.U, sG, np, tail);
  } else {
    bwd_reduce_unit<ADAM, 1, 0>(tb, feats, V.feat_by_order, nullptr, nullptr, weights, B, 1, grad_mode, opt, L.U, sG, np, tail);
  }
}

// Worker workgroups (the last BWD_CELLS_WORKERS of the apply launch's grid): wait until every unit has looked at its size, then
// take the units that did not fit off the list, piece by piece.  The last worker to finish resets the launch's counters.
template <bool ADAM>
__device__ __forceinline__ void bwd_cells_worker(
    const BwdCellsView& V, int n_units, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B,
    int grad_mode, const BwdOpt& opt, int max_dim, const uint2* __restrict__ slab, const uint16_t* __restrict__ bnd, int ch,
    BwdCellsLds& L, const TzrDst* sG, uint32_t* xcell) {
  __shared__ uint32_t s_item;
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  uint32_t* const ov = V.overflow;
  {
    const uint32_t want = tzr_consume_u32(ov + BWD_CELLS_OVF_EPOCH) + 1u;  // (moved only by the last worker of a launch, at its end)
    const uint32_t* flags = ov + BWD_CELLS_OVF_LIST + n_units;
    for (int i = threadIdx.x; i < n_units; i += BWD_THREADS)
      while (tzr_consume_u32(flags + i) != want) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  const uint32_t len = tzr_consume_u32(ov + BWD_CELLS_OVF_LEN);
  const float lr = *opt.lr;
  uint32_t* const cpre = xcell;
  uint32_t* const cbase = xcell + BWD_CELLS_MAXC + 8;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = len ? atomicAdd(ov + BWD_CELLS_OVF_CURSOR, 1u) : 0u;
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= len) break;
    const BwdCellUnit u = V.units[tzr_consume_u32(ov + BWD_CELLS_OVF_LIST + item)];
    const TzrTable& tb = u.tb;
    const int ncell = u.ncell;
    const int n = bwd_cells_bounds(u, u.i0, u.i1, bnd, ch, cpre, cbase, L.S.wtot);
    const int ft_dst = feats[u.feat].n_dst;
    if (u.split > 0) {  // one slice of ONE row: its sum is the record
      uint32_t cnt;
      const float4 sum = bwd_cells_stream_row<false>(u, V, slab, cpre, cbase, ncell, n, BWD_SENT, feats, weights, B, grad_mode, sG,
                                                     reinterpret_cast<float*>(&L.S.L), L.S.gstart + 264, &cnt);
      if (wv == 0) bwd_cells_combine<ADAM>(u, V, opt, lr, max_dim, sum, cnt, lane);
      continue;
    }
    int nb;
    uint64_t mult;
    bwd_bucket_params(tb.rows, &nb, &mult);
    const uint32_t bend = V.bnd_bucket[u.i1];  // the bucket behind the unit's last
    uint32_t cur = (uint32_t)((((uint64_t)u.b0 << 32) + mult - 1) / mult);  // rows of the unit's buckets: [cur, khi)
    uint64_t khi64 = (((uint64_t)bend << 32) + mult - 1) / mult;
    if (khi64 > (uint64_t)tb.rows) khi64 = (uint64_t)tb.rows;
    const uint32_t khi = (uint32_t)khi64;
    for (;;) {
      const int np = bwd_cells_next_piece<ADAM>(u, V, slab, cpre, cbase, ncell, n, cur, khi, feats, weights, B, grad_mode, opt, lr, L, sG);
      if (np == 0) break;
      constexpr int kRounds = BWD_UMAX / BWD_THREADS;
      const int pw = bwd_wave_span(np);
      const int rounds = pw / TZR_WAVE;
      uint32_t kreg[kRounds], sreg[kRounds];
      uint32_t vmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
        const int lp = wv * pw + r * TZR_WAVE + lane;
        const bool in = r < rounds && lp < np;
        kreg[r] = in ? L.S.pk[lp] : 0u;
        sreg[r] = in ? L.S.ps[lp] : 0u;
        if (in) {
          vmask |= 1u << r;
          kmin = min(kmin, kreg[r]);
          kmax = max(kmax, kreg[r]);
        }
      }
      __syncthreads();
      bwd_cells_sort_reduce<ADAM, 0, 1, kRounds>(u, V, kreg, sreg, vmask, kmin, kmax, np, ft_dst, feats, weights, B, grad_mode, opt, lr,
                                                 max_dim, L, sG);
      __syncthreads();  // (waves 1.. left the reduction before wave 0's stitch: everyone is here before the next piece's passes)
    }
  }
  // the launch's counters back to zero by the last worker out (every worker has read `len` and its last cursor value by then)
  if (threadIdx.x == 0 && tzr_arrive(ov + BWD_CELLS_OVF_DONE) == (uint32_t)BWD_CELLS_WORKERS - 1u) {
    if (len) atomicAdd(ov + BWD_CELLS_OVF_TOTAL, len);
    tzr_publish_u32(ov + BWD_CELLS_OVF_LEN, 0u);
    tzr_publish_u32(ov + BWD_CELLS_OVF_CURSOR, 0u);
    tzr_publish_u32(ov + BWD_CELLS_OVF_DONE, 0u);
    tzr_publish_u32(ov + BWD_CELLS_OVF_EPOCH, tzr_consume_u32(ov + BWD_CELLS_OVF_EPOCH) + 1u);  // the next launch's flag value
  }
}

template <bool ADAM, int FK, int NT>
__device__ __forceinline__ void bwd_cells_apply_body(
    BwdCellsView V, int n_units, const TzrFeature* __restrict__ feats, const float* __restrict__ weights, int64_t B, int grad_mode,
    const BwdGrads& G, const BwdOpt& opt, int max_dim, const uint2* __restrict__ slab, const uint16_t* __restrict__ bnd, int ch) {
  __shared__ BwdCellsLds L;
  __shared__ TzrDst sG[TZR_MAX_DST];
  __shared__ uint32_t xcell[2 * BWD_CELLS_MAXC + 8];  // the unit's cells: exclusive counts [ncell + 1], slab starts [ncell]
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < TZR_MAX_DST; ++i) sG[i] = G.d[i];  // static indices: straight from kernarg
  }
  if ((int)blockIdx.x >= n_units) {  // (workgroup-uniform; nothing of the unit path below is live here)
    __syncthreads();
#ifndef CELLS_NO_WORKER  // (timing experiment: the units' code alone; units that do not fit are then simply dropped)
    bwd_cells_worker<ADAM>(V, n_units, feats, weights, B, grad_mode, opt, max_dim, slab, bnd, ch, L, sG, xcell);
#endif
    return;
  }
  CELLS_MARK(0);
#ifndef CELLS_NO_PRIO
  // The seven units a CU holds start together and would stay in lock step (their lookups, their LDS sorts, their gathers all at the
  // same moments: profiles/r06g); wave priorities by residency slot -- workgroup i shares its CU with i +- 256 k -- let them drift
  // apart, so that one unit's gathers are in flight while another sorts (apply 90.9 -> 86.8 us, same box: profiles/r06av)
  tzr_prio_by_slot(blockIdx.x);
  // ... and the slots 4 .. 6, which repeat the priorities 0 .. 2, start 64 x 64 clocks (~1.7 us) late: 87.2 -> 86.2 us (five same-box
  // pairs, profiles/r06av); 0.85 us: nothing; every slot k x 0.4 / 0.2 us late: 88.9 / 90.3.
  if ((blockIdx.x >> 8) >= 4u) __builtin_amdgcn_s_sleep(64);
#endif
  const BwdCellUnit u = V.units[blockIdx.x];
  const uint32_t epoch = V.overflow[BWD_CELLS_OVF_EPOCH];  // (constant during a launch; loaded with the unit)
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  const int ncell = u.ncell;
  const int ft_dst = feats[u.feat].n_dst;  // the table's first key: read together with the cells' bounds (no round trip of its own)
  uint32_t* const cpre = xcell;
  uint32_t* const cbase = xcell + BWD_CELLS_MAXC + 8;
  const float lr = *opt.lr;
  const int n = bwd_cells_bounds(u, u.i0, u.i1, bnd, ch, cpre, cbase, L.S.wtot);
  CELLS_MARK(1);
  CELLS_NOTE(9, n);
  if (threadIdx.x == 0) {  // "I have looked at my size" (+ my index on the list when it does not fit): what the workers wait for
    if (n > BWD_UMAX) {
      const uint32_t at = atomicAdd(V.overflow + BWD_CELLS_OVF_LEN, 1u);
      tzr_publish_u32(V.overflow + BWD_CELLS_OVF_LIST + at, (uint32_t)blockIdx.x);
      tzr_drain_stores();
    }
    tzr_publish_u32(V.overflow + BWD_CELLS_OVF_LIST + n_units + blockIdx.x, epoch + 1u);  // (a store: nothing waits for it)
  }
  if (n > BWD_UMAX || n == 0) return;  // skewed ids: a worker takes this unit
  // ---- gather the cells' lookups into registers (arrival order = chunk-major = lookup-position order inside a row) ----
  constexpr int kRounds = BWD_UMAX / BWD_THREADS;
  const int pw = bwd_wave_span(n);
  const int rounds = pw / TZR_WAVE;
  uint32_t kreg[kRounds], sreg[kRounds];
  uint32_t vmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int lp = wv * pw + r * TZR_WAVE + lane;
    const bool in = r < rounds && lp < n;
#ifdef CELLS_EXP_CONTIG  // (timing experiment: the same number of lookups of the same table from ONE contiguous run -- wrong sums)
    const uint2 v = slab[cbase[0] + (uint32_t)(lp < n ? lp : n - 1)];
#else
    const uint2 v = bwd_cells_elem(slab, cpre, cbase, ncell, (uint32_t)(lp < n ? lp : n - 1));  // (clamped, unconditional)
#endif
    kreg[r] = in ? v.x : 0u;
    sreg[r] = in ? v.y : 0u;
    if (in) {
      vmask |= 1u << r;
      kmin = min(kmin, v.x);
      kmax = max(kmax, v.x);
    }
  }
  CELLS_MARK(2);
  bwd_cells_sort_reduce<ADAM, FK, NT, kRounds>(u, V, kreg, sreg, vmask, kmin, kmax, n, ft_dst, feats, weights, B, grad_mode, opt, lr,
                                               max_dim, L, sG);
  CELLS_MARK(8);
}

#define TZR_CELLS_APPLY_KERNEL(NAME, ADAM_, FK_, ATTR)                                                                         \
  __global__ __launch_bounds__(BWD_THREADS) ATTR void NAME(BwdCellsView V, int n_units, const TzrFeature* __restrict__ feats,  \
                                                          const float* __restrict__ weights, int64_t B, int grad_mode,         \
                                                          BwdGrads G, BwdOpt opt, int max_dim, const uint2* __restrict__ slab, \
                                                          const uint16_t* __restrict__ bnd, int ch) {                          \
    bwd_cells_apply_body<ADAM_, FK_, 1>(V, n_units, feats, weights, B, grad_mode, G, opt, max_dim, slab, bnd, ch);             \
  }
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_adagrad_kernel, false, TZR_OPT_ADAGRAD, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_rowwise_kernel, false, TZR_OPT_ROWWISE_ADAGRAD, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_sgd_kernel, false, TZR_OPT_SGD, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_general_kernel, false, 0, TZR_WAVES_PER_EU(7))
TZR_CELLS_APPLY_KERNEL(tzr_bwd_cells_apply_adam_kernel, true, 0, )

// ------------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------------
static int cells_common(const void* h_geo, void* d_geo, void* ws, size_t ws_bytes, int64_t n_values, int n_feats, int n_tables,
                        int max_dim, BwdCellsGeo* g, BwdCellsView* V, BwdPlan* P) {
  if (!h_geo || !d_geo || (reinterpret_cast<uintptr_t>(d_geo) & 255)) return TZR_ERR_INVALID;
  std::memcpy(g, h_geo, sizeof(BwdCellsGeo));
  if (g->n_feats != n_feats || g->max_dim != max_dim || g->n_chunks <= 0 || g->n_units <= 0) return TZR_ERR_INVALID;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TZR_ERR_WORKSPACE;
  if (bwd_layout(P, ws, n_values, g->n_positions, n_feats, n_tables, max_dim) > ws_bytes) return TZR_ERR_WORKSPACE;
  if (P->ch != (int)g->ch || g->n_chunks > P->max_chunks) return TZR_ERR_INVALID;  // (a geometry built for another batch size)
  *V = bwd_cells_view(d_geo, *g);
  return TZR_OK;
}

// The cells plan of one batch of ONE id per bag.  `h_geo` / `d_geo`: the geometry image tzr_bwd_cells_geometry made for these
// tables and this B, on the host and (a copy the caller uploaded, 256-byte aligned) on the device.  `ws`: the workspace of
// tzr_pooled_bwd_workspace (the same call sizes it for either plan).  One launch, asynchronous on `stream`.
extern "C" int tzr_pooled_bwd_cells_plan(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats, int n_feats,
                                         int max_dim, const int64_t* d_values, int64_t n_values, int64_t B, const void* h_geo,
                                         void* d_geo, void* ws, size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B <= 0) return TZR_ERR_INVALID;
  BwdCellsGeo g;
  BwdCellsView V;
  BwdPlan P;
  const int rc = cells_common(h_geo, d_geo, ws, ws_bytes, n_values, n_feats, n_tables, max_dim, &g, &V, &P);
  if (rc != TZR_OK) return rc;
  if (!d_values) return TZR_ERR_INVALID;
  if (n_values >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = nullptr;
  A.B = B;
  A.uniform = 1;
  hipLaunchKernelGGL(tzr_bwd_cells_partition_kernel, dim3((unsigned)g.n_chunks), dim3(BWD_THREADS), 0, static_cast<hipStream_t>(stream), V,
                     d_tables, A, P.ks[1], reinterpret_cast<uint16_t*>(P.hist), (int)g.ch);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// tzr_pooled_fwd (one id per bag, fp32 tables, unweighted: what tzr_pooled_fwd_ex sends to its LDS-ids kernel) and
// tzr_pooled_bwd_cells_plan of the same batch as ONE launch.  TZR_ERR_UNSUPPORTED: not that case (the caller makes the two calls).
extern int g_tzr_fwd_tile_b;
int g_tzr_fwd_plan_order = 2;  // tzr_tune("fwd_plan_order"): 0 = alternating, 1 = the plan's workgroups first, 2 = last (measured: see the kernel)
int g_tzr_fwd_plan = 1;  // tzr_tune("fwd_plan"): 0 = never (tzr_pooled_fwd_cells_plan_supported says no), 2 = at any batch size (tests)

extern "C" int tzr_pooled_fwd_cells_plan_supported(int n_slots, int64_t B) {
  // (batch sizes at which tzr_pooled_fwd_ex takes its LDS-ids kernel: smaller ones run the general kernel, and mostly the one-launch backward)
  return (g_tzr_fwd_plan != 0 && n_slots > 0 && n_slots <= FWD1_SLOTS && B > 0 && (B >= 32768 || g_tzr_fwd_plan == 2)) ? 1 : 0;
}

extern "C" int tzr_pooled_fwd_cells_plan(const TzrTable* d_ftables, const TzrFeature* d_ffeats, int n_ffeats, const TzrSlot* d_slots,
                                         int n_slots, const TzrDst* h_dsts, int n_dst, const TzrTable* d_tables, int n_tables,
                                         const TzrFeature* d_feats, int n_feats, int max_dim, const int64_t* d_values, int64_t n_values,
                                         int64_t B, const void* h_geo, void* d_geo, void* ws, size_t ws_bytes, void* stream) {
  if (!d_ftables || !d_ffeats || !d_slots || !h_dsts || n_ffeats <= 0 || n_slots <= 0 || n_dst <= 0 || n_dst > TZR_MAX_DST ||
      !d_tables || !d_feats || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B <= 0)
    return TZR_ERR_INVALID;
  if (!tzr_pooled_fwd_cells_plan_supported(n_slots, B)) return TZR_ERR_UNSUPPORTED;
  BwdCellsGeo g;
  BwdCellsView V;
  BwdPlan P;
  const int rc = cells_common(h_geo, d_geo, ws, ws_bytes, n_values, n_feats, n_tables, max_dim, &g, &V, &P);
  if (rc != TZR_OK) return rc;
  if (!d_values) return TZR_ERR_INVALID;
  if (n_values >= (1LL << 32)) return TZR_ERR_UNSUPPORTED;
  FwdDsts dsts;
  for (int i = 0; i < TZR_MAX_DST; ++i) {
    dsts.d[i].ptr = 0;
    dsts.d[i].stride = 0;
  }
  for (int i = 0; i < n_dst; ++i) {
    if (!h_dsts[i].ptr || (h_dsts[i].stride & 3) || (h_dsts[i].ptr & 15)) return TZR_ERR_INVALID;
    if (h_dsts[i].stride > 0x7fffffffLL) return TZR_ERR_UNSUPPORTED;
    dsts.d[i] = h_dsts[i];
  }
  BwdSrcArgs A;
  A.feats = d_feats;
  A.values = d_values;
  A.offsets = nullptr;
  A.B = B;
  A.uniform = 1;
  FwdPlanArgs a;
  a.ftables = d_ftables;
  a.ffeats = d_ffeats;
  a.slots = d_slots;
  a.values = d_values;
  a.B = B;
  a.n_slots = n_slots;
  a.tile_b = g_tzr_fwd_tile_b > 0 ? g_tzr_fwd_tile_b : (B >= 32768 ? 64 : (B >= 8192 ? 16 : 8));  // (64: see the kernel)
  a.n_fwd = (uint32_t)((B + a.tile_b - 1) / a.tile_b);
  a.n_part = (uint32_t)g.n_chunks;
  a.btables = d_tables;
  a.slab = P.ks[1];
  a.bnd = reinterpret_cast<uint16_t*>(P.hist);
  a.ch = (int)g.ch;
  a.order = g_tzr_fwd_plan_order;
  hipLaunchKernelGGL(tzr_pooled_fwd_u1_cells_plan_kernel, dim3(a.n_fwd + a.n_part), dim3(BWD_THREADS), 0, static_cast<hipStream_t>(stream),
                     a, dsts, V, A);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

// The apply of a cells plan: same meaning of grad_mode / h_grads / h_optim as tzr_pooled_bwd_apply, same sums in the same order.
extern "C" int tzr_pooled_bwd_cells_apply(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats, int n_tables, int max_dim,
                                          const float* d_weights, int64_t n_values, int64_t B, int grad_mode, const TzrDst* h_grads,
                                          int n_dst, const TzrSparseOptim* h_optim, const void* h_geo, void* d_geo, void* ws,
                                          size_t ws_bytes, void* stream) {
  if (!d_tables || !d_feats || !h_grads || !h_optim || n_tables <= 0 || n_feats <= 0 || n_values < 0 || B <= 0 || n_dst <= 0 ||
      n_dst > TZR_MAX_DST || max_dim <= 0 || max_dim > BWD_MAXDIM || (max_dim & 3) || (grad_mode != 0 && grad_mode != 1))
    return TZR_ERR_INVALID;
  if (!h_optim->d_lr) return TZR_ERR_INVALID;
  if (h_optim->kind != TZR_OPT_SGD && h_optim->kind != TZR_OPT_ADAGRAD && h_optim->kind != TZR_OPT_ROWWISE_ADAGRAD &&
      h_optim->kind != TZR_OPT_ACCUMULATE && h_optim->kind != TZR_OPT_ADAM)
    return TZR_ERR_UNSUPPORTED;
  if (h_optim->kind == TZR_OPT_ADAM && !h_optim->d_adam) return TZR_ERR_INVALID;
  BwdCellsGeo g;
  BwdCellsView V;
  BwdPlan P;
  const int rc = cells_common(h_geo, d_geo, ws, ws_bytes, n_values, n_feats, n_tables, max_dim, &g, &V, &P);
  if (rc != TZR_OK) return rc;
  BwdGrads G;
  for (int i = 0; i < TZR_MAX_DST; ++i) {
    G.d[i].ptr = 0;
    G.d[i].stride = 0;
  }
  for (int i = 0; i < n_dst; ++i) {
    if (!h_grads[i].ptr || (h_grads[i].stride & 3) || (h_grads[i].ptr & 15)) return TZR_ERR_INVALID;
    G.d[i] = h_grads[i];
  }
  BwdOpt opt;
  opt.kind = h_optim->kind;
  opt.wd_mode = h_optim->weight_decay_mode;
  opt.clip = h_optim->gradient_clipping;
  opt.lr = reinterpret_cast<const float*>(h_optim->d_lr);
  opt.eps = h_optim->eps;
  opt.wd = h_optim->weight_decay;
  opt.max_grad = h_optim->max_gradient;
  opt.beta1 = h_optim->beta1;
  opt.beta2 = h_optim->beta2;
  opt.adam = reinterpret_cast<const float*>(h_optim->d_adam);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define TZR_CELLS_LAUNCH(K)                                                                                               \
  hipLaunchKernelGGL(K, dim3((unsigned)g.n_units + BWD_CELLS_WORKERS), dim3(BWD_THREADS), 0, s, V, (int)g.n_units, d_feats, d_weights, \
                     B, grad_mode, G, opt, max_dim, (const uint2*)P.ks[1], (const uint16_t*)reinterpret_cast<uint16_t*>(P.hist), (int)g.ch)
  const bool fast_shape = !d_weights && (opt.kind == TZR_OPT_ADAGRAD || opt.kind == TZR_OPT_ROWWISE_ADAGRAD || opt.kind == TZR_OPT_SGD);
  if (opt.kind == TZR_OPT_ADAM) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_adam_kernel);
  } else if (!fast_shape) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_general_kernel);
  } else if (opt.kind == TZR_OPT_ADAGRAD) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_adagrad_kernel);
  } else if (opt.kind == TZR_OPT_ROWWISE_ADAGRAD) {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_rowwise_kernel);
  } else {
    TZR_CELLS_LAUNCH(tzr_bwd_cells_apply_sgd_kernel);
  }
#undef TZR_CELLS_LAUNCH
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
