// Shared between the backward plan (pooled_bwd.hip, K6) and apply (pooled_bwd_apply.hip, K7):
// the workspace layout that carries the plan from one C-ABI call to the other.
#pragma once
#include "tzr_common.h"

#define BWD_THREADS 256
#define BWD_WAVES (BWD_THREADS / TZR_WAVE)
#define BWD_CH 1024  // sorted positions per chunk (= per workgroup in hist / scatter / reduce)
#define BWD_RB 9     // max radix digit width
#define BWD_NB 512   // bins per chunk histogram row (1 << BWD_RB)
#define BWD_RANGE (BWD_CH / BWD_WAVES)  // sorted positions reduced by one wave
#define BWD_MAXDIM 256
#define BWD_SENT 0xFFFFFFFFu  // never a row id

// boundary record flags (per wave range in LDS, per chunk in the workspace)
#define BWD_LEAD 1u        // the first run continues a run that started before this span
#define BWD_LEAD_WHOLE 2u  // ... and does not end inside it
#define BWD_TRAIL 4u       // the last run continues past the end of this span

// Everything a workgroup needs to know about its chunk, written once by the prep kernel: one
// 48-byte load instead of a binary search over the chunk map plus three dependent table lookups at
// the head of every hist / scatter / reduce workgroup (these kernels are latency-, not
// bandwidth-bound: ~7 workgroups per CU, one pass each).
struct BwdChunkDesc {
  int32_t t;      // table, -1 for surplus chunks
  int32_t width;  // digit width of the table's sort
  int32_t npass;
  int32_t last_chunk;  // first chunk of the NEXT table (stitch walks up to it)
  int64_t s, e;        // positions [s, e) of the chunk
  int64_t ts, te;      // positions of the whole table
};

struct BwdPlan {  // pointers into the caller workspace
  int64_t* feat_start;     // [F+1] start of each lookup (by order) in table-major position space
  int32_t* feat_by_order;  // [F]
  int64_t* tab_start;      // [T+1]
  int32_t* tab_chunk;      // [T+1] first chunk of each table
  int32_t* tab_width;      // [T] digit width (0 = nothing to sort)
  int32_t* tab_npass;      // [T]
  uint2* ks[2];            // [N] {local row id, original lookup position} (ping-pong); one 8-byte
                           //     element so a scattered element is ONE store, not two
  uint32_t* bag_of;        // [NV] bag index key*B+b of every lookup (only when bags are jagged)
  uint32_t* hist;          // [max_chunks * BWD_NB] chunk-exclusive digit counts
  uint32_t* binbase;       // [T * BWD_NB] global start of every (table, digit)
  uint32_t* cflags;        // [max_chunks] boundary record of every chunk
  uint32_t* clkey;         // [max_chunks] key of the chunk's leading open run
  uint32_t* ctkey;         // [max_chunks] key of the chunk's trailing open run
  float* clead;            // [max_chunks * max_dim]
  float* ctrail;           // [max_chunks * max_dim]
  BwdChunkDesc* cdesc;     // [max_chunks]
  int64_t max_chunks;
};

static inline int64_t bwd_max_chunks(int64_t N, int T) { return N / BWD_CH + T + 1; }

// NV = ids in the KJT values array; N = capacity of the table-major position space (sum over
// lookups of their key length: a key read through two tables is sorted twice).
static inline size_t bwd_layout(BwdPlan* p, void* ws, int64_t NV, int64_t N, int F, int T,
                                int max_dim) {
  TzrCarver c(ws);
  BwdPlan q;
  q.max_chunks = bwd_max_chunks(N, T);
  q.feat_start = c.take<int64_t>(F + 1);
  q.feat_by_order = c.take<int32_t>(F);
  q.tab_start = c.take<int64_t>(T + 1);
  q.tab_chunk = c.take<int32_t>(T + 1);
  q.tab_width = c.take<int32_t>(T);
  q.tab_npass = c.take<int32_t>(T);
  for (int i = 0; i < 2; ++i) q.ks[i] = c.take<uint2>(N);
  q.bag_of = c.take<uint32_t>(NV);
  q.hist = c.take<uint32_t>(q.max_chunks * BWD_NB);
  q.binbase = c.take<uint32_t>((size_t)T * BWD_NB);
  q.cflags = c.take<uint32_t>(q.max_chunks);
  q.clkey = c.take<uint32_t>(q.max_chunks);
  q.ctkey = c.take<uint32_t>(q.max_chunks);
  q.clead = c.take<float>((size_t)q.max_chunks * max_dim);
  q.ctrail = c.take<float>((size_t)q.max_chunks * max_dim);
  q.cdesc = c.take<BwdChunkDesc>(q.max_chunks);
  if (p) *p = q;
  return c.off;
}

// chunk id -> its descriptor; false for surplus workgroups.
__device__ __forceinline__ bool bwd_chunk(const BwdPlan& P, int chunk, BwdChunkDesc* d) {
  if (chunk >= P.max_chunks) return false;
  *d = P.cdesc[chunk];
  return d->t >= 0;
}
