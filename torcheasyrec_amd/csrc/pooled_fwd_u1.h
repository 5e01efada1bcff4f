// The one-id-per-bag pooled forward (K5) as a device function: the body of tzr_pooled_fwd_u1_kernel (pooled_fwd.hip) and of the
// forward workgroups of the launch that carries the backward's index plan beside them (pooled_bwd_cells.hip:
// tzr_pooled_fwd_u1_cells_plan_kernel).  /root/reference/tzrec/modules/embedding.py:930, 972-976.
#pragma once
#include "tzr_common.h"

#define FWD_THREADS 256
#define FWD_UNROLL 4
#define FWD_MAX_SLOTS 512

struct FwdDsts {
  TzrDst d[TZR_MAX_DST];
};

// ---- one id per bag, fp32 tables, no per-sample weights (Criteo-shaped batches): ids staged in LDS ----
// The general kernel above runs this case as `id load -> row load -> store` per group of 4 elements
// with ~8 waves per CU: two dependent memory round trips per group, ~13 groups per thread in series
// (measured 49 us at B = 65536, where the traffic alone needs ~30).  Here a workgroup first pulls
// the ids of its tile -- one coalesced pass, one round trip -- into LDS, so every row gather of the
// tile is an independent load: FWD1_UNROLL of them in flight per thread, and tiles of 32 samples make
// 2048 workgroups at B = 65536 (~13 KB of LDS each: the register file, not LDS, bounds residency).
#ifndef FWD1_UNROLL
#define FWD1_UNROLL 8
#endif
#define FWD1_SLOTS 128    // slots per workgroup row (blockIdx.y); DLRM-Criteo: 104
#define FWD1_MAX_IDS 2048 // ids of a (sub-)tile held in LDS: groups x samples

struct Fwd1Slot {  // 24 bytes
  const float* w;  // table row 0 + the slot's float4 column group
  float* dst;      // destination buffer + col
  int32_t dst_stride;
  int32_t w_stride;
};

#ifndef FWD_PROF_MARK
#define FWD_PROF_MARK(i)
#endif

struct Fwd1Lds {  // 22.8 KB
  Fwd1Slot rs[FWD1_SLOTS];
  int64_t srows[FWD1_SLOTS];   // per slot, then compacted per id group
  int32_t sfeat[FWD1_SLOTS];   // KJT key index, same
  int64_t grows[FWD1_SLOTS];
  int32_t gfeat[FWD1_SLOTS];
  uint16_t gid[FWD1_SLOTS];    // id group of a slot (consecutive slots of one key share it)
  int64_t sid[FWD1_MAX_IDS];
  uint32_t wsum[FWD_THREADS / TZR_WAVE];
};

// workgroup (bx, by) of the grid (tiles of `tile_b` samples, rows of FWD1_SLOTS slots); UNR gathers in flight per thread
template <int UNR>
__device__ __forceinline__ void fwd_u1_body(
    const TzrTable* __restrict__ tables, const TzrFeature* __restrict__ feats,
    const TzrSlot* __restrict__ slots, int n_slots, const int64_t* __restrict__ values, int64_t B,
    int tile_b, const FwdDsts& dsts, unsigned bx, unsigned by, Fwd1Lds& S) {
  auto& rs = S.rs;
  auto& srows = S.srows;
  auto& sfeat = S.sfeat;
  auto& grows = S.grows;
  auto& gfeat = S.gfeat;
  auto& gid = S.gid;
  auto& sid = S.sid;
  auto& wsum = S.wsum;
  static_assert(FWD1_SLOTS <= FWD_THREADS, "one slot per thread in the prologue");
  const int s0 = by * FWD1_SLOTS;
  const int ns = min(FWD1_SLOTS, n_slots - s0);
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wv = threadIdx.x / TZR_WAVE;
  FWD_PROF_MARK(0);
  {
    const int s = threadIdx.x;
    if (s < ns) {
      const TzrSlot sl = slots[s0 + s];
      const TzrFeature ft = feats[sl.feature];
      const TzrTable tb = tables[ft.table];
      Fwd1Slot r;
      r.w = reinterpret_cast<const float*>(tb.w) + (size_t)sl.chunk * 4;
      r.dst = reinterpret_cast<float*>(dsts.d[sl.dst].ptr) + sl.col;
      r.dst_stride = (int32_t)dsts.d[sl.dst].stride;
      r.w_stride = tb.w_stride;
      rs[s] = r;
      srows[s] = tb.rows;
      sfeat[s] = ft.key;
    }
  }
  __syncthreads();
  // id groups: slot s opens a group unless it reads the key (and row count) of slot s - 1
  int ngroups;
  {
    const int s = threadIdx.x;
    const bool open = s < ns && (s == 0 || sfeat[s] != sfeat[s - 1] || srows[s] != srows[s - 1]);
    const unsigned long long bm = __ballot(open);
    const uint32_t before = (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = (uint32_t)__popcll(bm);
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FWD_THREADS / TZR_WAVE; ++w) {
      if (w < wv) pre += wsum[w];
      tot += wsum[w];
    }
    if (s < ns) {
      const uint32_t g = pre + before + (open ? 1u : 0u) - 1u;
      gid[s] = (uint16_t)g;
      if (open) {
        grows[g] = srows[s];
        gfeat[g] = sfeat[s];
      }
    }
    ngroups = (int)tot;
  }
  __syncthreads();
  FWD_PROF_MARK(1);  // slots resolved, id groups formed
  const int64_t b0 = (int64_t)bx * tile_b;
  const int nb = (int)min((int64_t)tile_b, B - b0);
  const int sub = min(nb, FWD1_MAX_IDS / ngroups);  // samples per LDS pass (ngroups <= FWD1_SLOTS: >= 8)
  // k / ns by multiplication: the quotient is at most one short (ns = 1: 2^32 does not fit, 2^32 - 1 is one short too)
  const uint32_t magic = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (0x100000000ull + (uint64_t)ns - 1) / (uint64_t)ns);
  for (int sb = 0; sb < nb; sb += sub) {  // one pass when the tile's ids fit (the launcher sees to that for Criteo shapes)
    const int cnt = min(sub, nb - sb);
    if (sb) __syncthreads();
    // ids of the sub-tile: group-major, `sub` consecutive samples per group (coalesced 8-byte loads)
    for (int i = threadIdx.x; i < ngroups * sub; i += FWD_THREADS) {
      const int g = i / sub;
      const int bl = i - g * sub;
      int64_t id = 0;
      if (bl < cnt) id = values[(int64_t)gfeat[g] * B + b0 + sb + bl];
      if ((uint64_t)id >= (uint64_t)grows[g]) id = 0;  // memory safety; K4 counts / reports them
      sid[i] = id;
    }
    __syncthreads();
    FWD_PROF_MARK(2);  // the tile's ids in LDS
    const int total = cnt * ns;
    for (int k0 = threadIdx.x; k0 < total; k0 += FWD_THREADS * UNR) {
      // No lane-dependent condition around the LDS reads and the row loads (an element behind the tile's last repeats it
      // and is not stored): `if (ok) acc = load(...)` is a branch per element to hipcc, and the slot / id reads inside it were
      // waited for one element at a time -- eight "independent" gathers issued as a chain.
      float* dp[UNR];
      const float* wp[UNR];
      float4 acc[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int k = k0 + u * FWD_THREADS;
        k = k < total ? k : total - 1;
        int bl = (int)__umulhi((uint32_t)k, magic);
        if ((bl + 1) * ns <= k) ++bl;  // the magic quotient is at most one short
        const int s = k - bl * ns;
        const Fwd1Slot r = rs[s];
        const int64_t id = sid[(int)gid[s] * sub + bl];
        dp[u] = r.dst + (b0 + sb + bl) * (int64_t)r.dst_stride;
        wp[u] = r.w + id * (int64_t)r.w_stride;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc[u] = tzr_ldg4(wp[u]);
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (k0 + u * FWD_THREADS < total) tzr_stg4(dp[u], acc[u]);
    }
  }
  FWD_PROF_MARK(3);  // (thread 0's last store issued)
}
