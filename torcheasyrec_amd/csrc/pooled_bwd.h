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
#define BWD_GEO 256  // lookups / tables up to which every partition workgroup derives the geometry itself
#define BWD_SUB 1024  // lookups ranked at a time by a partition workgroup (one sub-tile of its chunk)
#define BWD_PK 4      // a partition chunk = up to BWD_PK unit blocks: 4x fewer workgroups pay the pass's fixed
                      // costs (geometry, publish, arrival) and a table scan reads 4x fewer rows
#define BWD_LROW (BWD_NB + 4)  // uint16 entries per chunk row of bucket starts (NB + 1 used; 8-byte stores)
#define BWD_SEGB 256           // chunks per batch when a bucket range is gathered from the chunk slabs
#define BWD_MAXDIM 256
#define BWD_SENT 0xFFFFFFFFu  // never a row id

// boundary record flags (per wave range in LDS, per unit in the workspace)
#define BWD_LEAD 1u        // the first run continues a run that started before this span
#define BWD_LEAD_WHOLE 2u  // ... and does not end inside it
#define BWD_TRAIL 4u       // the last run continues past the end of this span

// Everything a workgroup needs to know about its chunk, written once by the partition kernel: one
// load instead of a binary search over the chunk map plus dependent table lookups at the head of
// every scatter / reduce workgroup (these kernels are latency-, not bandwidth-bound).
// Chunk c of a table is (a) the c-th block of BWD_CH table-major INPUT positions for the partition
// kernel (its slab) and (b) the c-th UNIT of sorted positions [ucut[c], ucut[c+1]) for sort / reduce.
struct BwdChunkDesc {
  int32_t t;           // table, -1 for surplus chunks
  int32_t nb;          // buckets of the table's partition pass (<= BWD_NB)
  int32_t exact;       // every bucket is one row id
  int32_t last_chunk;  // first chunk of the NEXT table (stitch walks up to it)
  int32_t first_chunk; // first chunk of this table
  int32_t first_pchunk; // first PARTITION chunk of this table
  int64_t s, e;        // input positions [s, e) of the chunk
  int64_t ts, te;      // positions of the whole table
  uint64_t mult;       // bucket of row id k = (k * mult) >> 32 (monotone in k)
};

// Work item of the sort kernel: heavy bucket `bin` of table t = sorted positions [start, binbase[bin+1]).
// Its lookups sit in the chunk slabs of the table (bucket `bin` of every chunk, in chunk order =
// table-major order); a TILE is the lookups of chunks [c_begin, c_end) (relative to the table's first
// chunk).  Tiles are cut by a rule that needs no per-chunk data (bwd_tile_chunks: about BWD_HT / 2 lookups
// when the bucket is spread evenly); a tile worker takes whatever its chunks hold, BWD_HT lookups at a time.
#define BWD_HK_ONEPASS 0  // bucket of <= BWD_NB row ids: one counting pass, tile-parallel
#define BWD_HK_HOT 1      // wide bucket with more than one tile: split around its hot row, tile-parallel
#define BWD_HK_SERIAL 2   // the whole bucket by one workgroup (c_begin = 0, c_end = chunks of the table)
#define BWD_HK_COPY 3     // bucket = one row (exact table): the tile is copied, chunk order IS the order
struct BwdHeavy {
  int32_t t;
  uint32_t bin, start;
  int32_t c_begin, c_end;
  int32_t kind;
  int32_t pad[2];
};
// chunks per tile of a heavy bucket of `run` lookups in a table of C chunks
static inline __host__ __device__ uint32_t bwd_tile_chunks(uint32_t run, uint32_t C) {
  const uint64_t g = ((uint64_t)(BWD_HT / 2) * C) / run;
  return g < 1 ? 1u : (uint32_t)g;
}
// The work items of table t (sorted positions from ts, index t) are listed from hlist[bwd_hbase(ts, t)]:
// a heavy bucket of `run` lookups makes at most 4 * run / BWD_HT + 1 tiles and there are at most
// n_t / (BWD_TH + 1) heavy buckets, so the regions of consecutive tables never overlap.
static inline __host__ __device__ uint32_t bwd_hbase(uint32_t ts, uint32_t t) {
  return 4u * (ts / BWD_HT) + ts / (BWD_TH + 1) + 8u * t;
}

struct BwdPlan {  // pointers into the caller workspace
  uint32_t* feat_start;    // [F+1] start of each lookup (by order) in table-major position space
  int32_t* feat_key;       // [F] KJT key of the lookup with that order
  int32_t* feat_by_order;  // [F]
  int32_t* tab_chunk;      // [T+1] first chunk (unit) of each table
  int32_t* tab_pchunk;     // [T+1] first partition chunk of each table
  uint2* ks[3];            // [N] {local row id, original lookup position}: ks[1] holds the chunk SLABS
                           //     (partition chunk q = pch input positions of its table, stably ordered by
                           //     bucket inside the chunk), ks[0] the sorted lookups of every table, ks[2] is the
                           //     ping-pong scratch of the serial heavy path; 8-byte elements = ONE store per move
  uint32_t* bag_of;        // [NV] bag index key*B+b of every lookup (only when bags are jagged)
  uint16_t* lst;           // [max_pchunks * BWD_LROW] slab-local start of every bucket of every partition chunk (+ total)
  uint32_t* binbase;       // [T * (BWD_NB+1)] global start of every (table, bucket), end of the last
  uint32_t* ucut;          // [max_chunks + 1] first sorted position of every unit
  uint32_t* ub0;           // [max_chunks + 1] first bucket a unit may hold light lookups of
  uint32_t* uflag;         // [max_chunks] 1 = nothing for the unit sort (the unit lies inside one heavy bucket)
  uint32_t* hbits;         // [T * BWD_NB/32] bitmap of the heavy buckets of every table
  uint32_t* tarr;          // [T] chunks of the table that have published their slab (zeroed before the launch)
  uint32_t* tcount;        // [T] work items of the table (heavy tiles), listed from hlist[bwd_hbase(ts, t)]
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
  uint64_t* prof;          // [max_chunks * 8] phase timestamps of the partition pass (tzr_tune("bwd_prof"); else unused)
  int64_t max_chunks;
  int64_t max_pchunks;
  int64_t max_heavy;
  int32_t ch;   // positions per unit block (multiple of 256, <= BWD_CH)
  int32_t pch;  // positions per partition chunk (ch * 1 .. BWD_PK)
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
// Unit blocks per partition chunk: as many (up to BWD_PK) as still leave ~400 partition workgroups.
// g_tzr_bwd_pk (tzr_tune "bwd_pk") overrides.
extern int g_tzr_bwd_pk;
static inline int bwd_pick_pk(int64_t N, int ch) {
  if (g_tzr_bwd_pk >= 1 && g_tzr_bwd_pk <= BWD_PK) return g_tzr_bwd_pk;
  const int64_t k = N / ((int64_t)ch * 400);
  return k < 1 ? 1 : (k > BWD_PK ? BWD_PK : (int)k);
}

// NV = ids in the KJT values array; N = capacity of the table-major position space (sum over
// lookups of their key length: a key read through two tables is sorted twice).
static inline size_t bwd_layout(BwdPlan* p, void* ws, int64_t NV, int64_t N, int F, int T,
                                int max_dim) {
  TzrCarver c(ws);
  BwdPlan q;
  q.ch = bwd_pick_ch(N);
  q.pch = q.ch * bwd_pick_pk(N, q.ch);
  q.max_chunks = bwd_max_chunks(N, T, q.ch);
  q.max_pchunks = bwd_max_chunks(N, T, q.pch);
  q.max_heavy = 4 * (N / BWD_HT) + N / (BWD_TH + 1) + 8 * (int64_t)T + 8;  // bwd_hbase(N, T)
  q.feat_start = c.take<uint32_t>(F + 1);
  q.feat_key = c.take<int32_t>(F);
  q.feat_by_order = c.take<int32_t>(F);
  q.tab_chunk = c.take<int32_t>(T + 1);
  q.tab_pchunk = c.take<int32_t>(T + 1);
  for (int i = 0; i < 3; ++i) q.ks[i] = c.take<uint2>(N);
  q.bag_of = c.take<uint32_t>(NV);
  q.lst = c.take<uint16_t>((size_t)q.max_pchunks * BWD_LROW);
  q.binbase = c.take<uint32_t>((size_t)T * (BWD_NB + 1));
  q.ucut = c.take<uint32_t>(q.max_chunks + 1);
  q.ub0 = c.take<uint32_t>(q.max_chunks + 1);
  q.uflag = c.take<uint32_t>(q.max_chunks);
  q.hbits = c.take<uint32_t>((size_t)T * (BWD_NB / 32));
  q.tarr = c.take<uint32_t>((size_t)T + 4);
  q.tcount = c.take<uint32_t>((size_t)T + 4);
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
  q.prof = c.take<uint64_t>((size_t)q.max_pchunks * 8);
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
  // (accessed as LDS, never through a generic pointer: a `volatile uint16_t*` here made this hipcc emit
  // an illegal compare against src_shared_base once the sort kernel grew; the asm statements keep the
  // compiler from moving the read past the write, the LDS keeps a wave's accesses in order)
  uint16_t* wrow = L.wcnt[wv];
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
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      if (v && rank == 0) wrow[d] = (uint16_t)(pre + (uint32_t)__popcll(peers));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
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
