// Shared between the backward plan (pooled_bwd.hip, K6) and apply (pooled_bwd_apply.hip, K7):
// the workspace layout that carries the plan from one C-ABI call to the other, the bucket map of
// the one global partition pass, and the in-LDS stable counting pass both sides sort with.
#pragma once
#include "tzr_common.h"

#define BWD_THREADS 256
#define BWD_WAVES (BWD_THREADS / TZR_WAVE)
#define BWD_CH 1024  // MAX positions per input chunk of the partition pass = block of the unit grid
                     // (the launchers pick 256 .. 1024 from the number of lookups: BwdPlan.ch)
#define BWD_RB 9
#define BWD_NB 512   // buckets per table of the global partition pass (1 << BWD_RB)
#define BWD_TH 256   // a bucket with more lookups is "heavy": sorted by the heavy kernel, cut at blocks
#define BWD_UMAX (BWD_CH + BWD_TH)  // capacity of one unit of the apply: < BWD_CH + BWD_TH lookups
#define BWD_HT 1024  // tile of the heavy-bucket sort (4 rounds per wave: every role of the sort kernel within 64 VGPRs)
#define BWD_GEO 1024 // lookups / tables up to which every hist workgroup derives the geometry itself
#define BWD_MAXDIM 256
#define BWD_SENT 0xFFFFFFFFu  // never a row id

// boundary record flags (per wave range in LDS, per unit in the workspace)
#define BWD_LEAD 1u        // the first run continues a run that started before this span
#define BWD_LEAD_WHOLE 2u  // ... and does not end inside it
#define BWD_TRAIL 4u       // the last run continues past the end of this span

// Everything a workgroup needs to know about its chunk, written once by the hist kernel: one
// load instead of a binary search over the chunk map plus dependent table lookups at the head of
// every scatter / reduce workgroup (these kernels are latency-, not bandwidth-bound).
// Chunk c of a table is (a) the c-th block of BWD_CH table-major INPUT positions for hist / scatter
// and (b) the c-th UNIT of sorted positions [ucut[c], ucut[c+1]) for reduce / stitch.
struct BwdChunkDesc {
  int32_t t;           // table, -1 for surplus chunks
  int32_t nb;          // buckets of the table's partition pass (<= BWD_NB)
  int32_t exact;       // every bucket is one row id: the partition pass alone is the sort
  int32_t last_chunk;  // first chunk of the NEXT table (stitch walks up to it)
  int64_t s, e;        // input positions [s, e) of the chunk
  int64_t ts, te;      // positions of the whole table
  uint64_t mult;       // bucket of row id k = (k * mult) >> 32 (monotone in k)
};

struct BwdHeavy {  // work item of the sort kernel: heavy bucket `bin` of table t = positions [start, end)
  int32_t t;
  uint32_t bin, start, end;
  int32_t tile;  // >= 0: BWD_HT-tile of a bucket one counting pass sorts (<= BWD_NB row ids);
                 // -1: the whole bucket, several passes, one workgroup
  int32_t pad[3];
};

struct BwdPlan {  // pointers into the caller workspace
  uint32_t* feat_start;    // [F+1] start of each lookup (by order) in table-major position space
  int32_t* feat_key;       // [F] KJT key of the lookup with that order
  int32_t* feat_by_order;  // [F]
  int32_t* tab_chunk;      // [T+1] first chunk of each table
  uint2* ks[3];            // [N] {local row id, original lookup position}: ks[1] holds the bucket-
                           //     partitioned lookups (final for exact tables), ks[0] the sorted ones
                           //     of every other table; one 8-byte element = ONE store per move.
                           //     ks[2]: ping-pong partner of ks[0] in the multi-pass (serial) heavy path --
                           //     ks[1] is read-only for the whole sort launch (other workgroups walk it)
  uint32_t* bag_of;        // [NV] bag index key*B+b of every lookup (only when bags are jagged)
  uint32_t* hist;          // [max_chunks * BWD_NB] chunk-exclusive bucket counts
  uint32_t* binbase;       // [T * (BWD_NB+1)] global start of every (table, bucket), end of the last
  uint32_t* ucut;          // [max_chunks + 1] first sorted position of every unit
  uint32_t* uflag;         // [max_chunks] 1 = nothing for the unit sort (exact table / inside one heavy bucket)
  uint32_t* hbits;         // [T * BWD_NB/32] bitmap of the heavy buckets of every table
  uint32_t* hcount;        // [1] work items listed
  uint32_t* tab_stitch;    // [T] 1 = the table holds sorted buckets: runs may cross unit boundaries
  uint32_t* sexp;          // [T * BWD_NB] units overlapping the (sorted) bucket when > 1, else 0
  uint32_t* sarr;          // [T * BWD_NB] ... of which have published their boundary record (apply)
  BwdHeavy* hlist;         // [max_heavy]
  uint32_t* cflags;        // [max_chunks] boundary record of every unit
  uint32_t* clkey;         // [max_chunks] key of the unit's leading open run
  uint32_t* ctkey;         // [max_chunks] key of the unit's trailing open run
  float* clead;            // [max_chunks * max_dim]
  float* ctrail;           // [max_chunks * max_dim]
  BwdChunkDesc* cdesc;     // [max_chunks]
  uint32_t* stot;          // [T * nslices * BWD_NB] scan slices: bucket counts of a slice of a table's chunks, then
                           //     (in place) the slice's base = counts of the slices before it (only when nslices > 1)
  uint32_t* scnt;          // [T] slices of the table that have arrived (zeroed by the hist launch)
  uint32_t* umix;          // [max_chunks] 1 = the unit's table holds heavy buckets (scan launch): its light lookups are sorted by the
                           //     sort launch; the units of every other bucketed table are sorted by the APPLY itself in LDS when
                           //     the plan was built fused (hcount[1]; pooled_bwd_apply.hip: bwd_stage_unit)
  int32_t fuse;            // host side of hcount[1] (plan launch only)
  int32_t nslices;         // workgroups per table of the scan launch
  int64_t max_chunks;
  int64_t max_heavy;
  int32_t ch;  // positions per chunk (multiple of 256, <= BWD_CH)
};

// Positions per chunk.  The plan / apply kernels are latency-bound: the time of a launch is the
// time of ONE workgroup unless the chip is oversubscribed, so a small problem wants small chunks
// (B = 8192 Criteo: 213k lookups = 208 chunks of 1024 on 256 CUs, each wave walking 16 dependent
// tiles; 832 chunks of 256 do 4).  g_tzr_bwd_ch (tzr_tune "bwd_ch") overrides.
extern int g_tzr_bwd_ch;
static inline int bwd_pick_ch(int64_t N) {
  if (g_tzr_bwd_ch >= 256 && g_tzr_bwd_ch <= BWD_CH && g_tzr_bwd_ch % 256 == 0) return g_tzr_bwd_ch;
  if (N <= 512 * 1024) return 256;
  if (N <= 1024 * 1024) return 512;
  return BWD_CH;
}
static inline int64_t bwd_max_chunks(int64_t N, int T, int ch) { return N / ch + T + 1; }
// One workgroup per table walks that table's chunks in the scan launch (per bucket: a prefix over the chunks, 4 KB of
// counters per chunk, read and written, through a single CU): 1 628 chunks of 256 lookups -- the one item table of the
// sequence path, 417 k ids -- made that launch 128 us (profiles/r03bp).  When the average table has more than 128 chunks
// the walk is cut into slices of ~128 chunks, a workgroup each; the last to arrive finishes the table.
#define BWD_MAXSL 16
extern int g_tzr_bwd_scan_slices;  // tzr_tune("bwd_scan_slices"): > 0 forces the workgroups per table of the scan launch (<= BWD_MAXSL)
static inline int bwd_pick_slices(int64_t N, int T, int ch) {
  if (g_tzr_bwd_scan_slices > 0) return g_tzr_bwd_scan_slices > BWD_MAXSL ? BWD_MAXSL : g_tzr_bwd_scan_slices;
  const int64_t avg = N / ((int64_t)(T > 0 ? T : 1) * ch);
  if (avg <= 128) return 1;
  const int64_t s = (avg + 127) / 128;
  return (int)(s > BWD_MAXSL ? BWD_MAXSL : s);
}

// NV = ids in the KJT values array; N = capacity of the table-major position space (sum over
// lookups of their key length: a key read through two tables is sorted twice).
static inline size_t bwd_layout(BwdPlan* p, void* ws, int64_t NV, int64_t N, int F, int T,
                                int max_dim) {
  TzrCarver c(ws);
  BwdPlan q;
  q.ch = bwd_pick_ch(N);
  q.max_chunks = bwd_max_chunks(N, T, q.ch);
  q.max_heavy = N / (BWD_TH + 1) + N / BWD_HT + 1;
  q.feat_start = c.take<uint32_t>(F + 1);
  q.feat_key = c.take<int32_t>(F);
  q.feat_by_order = c.take<int32_t>(F);
  q.tab_chunk = c.take<int32_t>(T + 1);
  for (int i = 0; i < 3; ++i) q.ks[i] = c.take<uint2>(N);
  q.bag_of = c.take<uint32_t>(NV);
  q.hist = c.take<uint32_t>(q.max_chunks * BWD_NB);
  q.binbase = c.take<uint32_t>((size_t)T * (BWD_NB + 1));
  q.ucut = c.take<uint32_t>(q.max_chunks + 1);
  q.uflag = c.take<uint32_t>(q.max_chunks);
  q.hbits = c.take<uint32_t>((size_t)T * (BWD_NB / 32));
  q.hcount = c.take<uint32_t>(4);
  q.tab_stitch = c.take<uint32_t>(T);
  q.sexp = c.take<uint32_t>((size_t)T * BWD_NB);
  q.sarr = c.take<uint32_t>((size_t)T * BWD_NB);
  q.hlist = c.take<BwdHeavy>(q.max_heavy);
  q.cflags = c.take<uint32_t>(q.max_chunks);
  q.clkey = c.take<uint32_t>(q.max_chunks);
  q.ctkey = c.take<uint32_t>(q.max_chunks);
  q.clead = c.take<float>((size_t)q.max_chunks * max_dim);
  q.ctrail = c.take<float>((size_t)q.max_chunks * max_dim);
  q.cdesc = c.take<BwdChunkDesc>(q.max_chunks);
  q.nslices = bwd_pick_slices(N, T, q.ch);
  q.stot = c.take<uint32_t>(q.nslices > 1 ? (size_t)T * q.nslices * BWD_NB : 1);
  q.scnt = c.take<uint32_t>(T > 0 ? T : 1);
  q.umix = c.take<uint32_t>(q.max_chunks);
  q.fuse = 0;
  if (p) *p = q;
  return c.off;
}

// chunk id -> its descriptor; false for surplus workgroups.
__device__ __forceinline__ bool bwd_chunk(const BwdPlan& P, int chunk, BwdChunkDesc* d) {
  if (chunk >= P.max_chunks) return false;
  *d = P.cdesc[chunk];
  return d->t >= 0;
}

// ---- bucket map of the partition pass ---------------------------------------------------------
// rows <= BWD_NB: bucket = row id (the pass is the whole sort).  Otherwise BWD_NB buckets of
// ~rows/BWD_NB consecutive row ids each: uniform ids fill them evenly whatever the row count
// (top-bit digits would use 306 of 512 buckets for a 40M-row table).
__device__ __forceinline__ void bwd_bucket_params(int64_t rows, int* nb, uint64_t* mult) {
  if (rows <= BWD_NB) {
    *nb = rows < 1 ? 1 : (int)rows;
    *mult = 1ull << 32;
  } else {
    *nb = BWD_NB;
    *mult = ((uint64_t)BWD_NB << 32) / (uint64_t)rows;
  }
}
__device__ __forceinline__ uint32_t bwd_bucket(uint32_t key, uint64_t mult) {
  return (uint32_t)(((uint64_t)key * mult) >> 32);
}
__device__ __forceinline__ int bwd_bits(uint32_t max_value) {  // bits needed to hold max_value
  return max_value == 0 ? 0 : 32 - __clz((int)max_value);
}

// ---- one stable counting pass over a tile held in registers -------------------------------------
// Ownership is wave-contiguous: wave w holds local positions [w*pw, (w+1)*pw), lane l position
// w*pw + r*64 + l in round r (rounds = pw/64, the same for all waves).  A wave ranks its own
// elements round by round with match-any ballots against a wave-private count row -- no workgroup
// barrier inside the loop -- and the waves are then ordered by one scan: three barriers per pass
// whatever the tile size.
template <int NB_>
struct BwdRankLds {
  uint16_t wcnt[BWD_WAVES][NB_];  // per wave: count, then exclusive-over-waves offset, per digit
  uint16_t lstart[NB_ + 1];       // tile-local start of each digit (digit-major order), total
  uint32_t wtot[BWD_WAVES];
};

// dest[r] = index of the element in the stable digit-sorted order of the tile; on return
// L.lstart / L.wcnt stay valid (rank among the tile's elements of digit d = dest - L.lstart[d]).
// `wbits` = number of significant digit bits.  All threads of the workgroup call.
template <int NB_, int MAXR>
__device__ __forceinline__ void bwd_rank_tile(const uint32_t (&dig)[MAXR], uint32_t vmask, int rounds,
                                              int wbits, BwdRankLds<NB_>& L, uint32_t (&dest)[MAXR]) {
  const int tid = threadIdx.x;
  const int lane = tid & (TZR_WAVE - 1);
  const int wv = tid / TZR_WAVE;
  for (int i = tid; i < BWD_WAVES * NB_; i += BWD_THREADS) (&L.wcnt[0][0])[i] = 0;
  __syncthreads();
  volatile TZR_LDS_AS uint16_t* wrow = (volatile TZR_LDS_AS uint16_t*)L.wcnt[wv];  // (LDS address space kept: ds_read_u16 / ds_write_b16, not FLAT)
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t loc[MAXR];
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    loc[r] = 0;
    if (r < rounds) {  // workgroup-uniform
      const bool v = (vmask >> r) & 1u;
      const uint32_t d = dig[r];
      unsigned long long peers = __ballot(v);
      for (int bit = 0; bit < wbits; ++bit) {
        const int on = (d >> bit) & 1;
        const unsigned long long bm = __ballot(on);
        peers &= on ? bm : ~bm;
      }
      const uint32_t rank = (uint32_t)__popcll(peers & lt);
      const uint32_t pre = v ? (uint32_t)wrow[d] : 0u;
      __builtin_amdgcn_wave_barrier();
      if (v && rank == 0) wrow[d] = (uint16_t)(pre + (uint32_t)__popcll(peers));
      __builtin_amdgcn_wave_barrier();
      loc[r] = pre + rank;
    }
  }
  __syncthreads();
  constexpr int DPT = NB_ / BWD_THREADS;
  static_assert(DPT >= 1 && DPT * BWD_THREADS == NB_, "digits are dealt to threads evenly");
  uint32_t tsum[DPT];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    const int d = tid * DPT + j;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < BWD_WAVES; ++w) {
      const uint32_t c = L.wcnt[w][d];
      L.wcnt[w][d] = (uint16_t)run;
      run += c;
    }
    tsum[j] = run;
    mine += run;
  }
  uint32_t incl = mine;
  for (int dd = 1; dd < TZR_WAVE; dd <<= 1) {
    const uint32_t o = __shfl_up(incl, dd, TZR_WAVE);
    if (lane >= dd) incl += o;
  }
  if (lane == TZR_WAVE - 1) L.wtot[wv] = incl;
  __syncthreads();
  uint32_t excl = incl - mine;
  for (int w = 0; w < wv; ++w) excl += L.wtot[w];
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    L.lstart[tid * DPT + j] = (uint16_t)excl;
    excl += tsum[j];
  }
  if (tid == BWD_THREADS - 1) L.lstart[NB_] = (uint16_t)excl;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    dest[r] = 0;
    if (r < rounds && ((vmask >> r) & 1u))
      dest[r] = (uint32_t)L.lstart[dig[r]] + (uint32_t)L.wcnt[wv][dig[r]] + loc[r];
  }
}

// positions owned per wave for a tile of n elements: ceil(n / waves) rounded up to whole rounds
__device__ __forceinline__ int bwd_wave_span(int n) {
  return (((n + BWD_WAVES - 1) / BWD_WAVES) + TZR_WAVE - 1) & ~(TZR_WAVE - 1);
}
