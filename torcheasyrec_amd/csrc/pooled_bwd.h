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

struct BwdPlan {  // pointers into the caller workspace
  int64_t* feat_start;     // [F+1] start of each lookup (by order) in table-major position space
  int32_t* feat_by_order;  // [F]
  int64_t* tab_start;      // [T+1]
  int32_t* tab_chunk;      // [T+1] first chunk of each table
  int32_t* tab_width;      // [T] digit width (0 = nothing to sort)
  int32_t* tab_npass;      // [T]
  uint32_t* key[2];        // [N] local row id (ping-pong)
  uint32_t* src[2];        // [N] original lookup position
  uint32_t* bag_of;        // [NV] bag index key*B+b of every lookup (only when bags are jagged)
  uint32_t* hist;          // [max_chunks * BWD_NB] chunk-exclusive digit counts
  uint32_t* binbase;       // [T * BWD_NB] global start of every (table, digit)
  uint32_t* cflags;        // [max_chunks] boundary record of every chunk
  uint32_t* clkey;         // [max_chunks] key of the chunk's leading open run
  uint32_t* ctkey;         // [max_chunks] key of the chunk's trailing open run
  float* clead;            // [max_chunks * max_dim]
  float* ctrail;           // [max_chunks * max_dim]
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
  for (int i = 0; i < 2; ++i) {
    q.key[i] = c.take<uint32_t>(N);
    q.src[i] = c.take<uint32_t>(N);
  }
  q.bag_of = c.take<uint32_t>(NV);
  q.hist = c.take<uint32_t>(q.max_chunks * BWD_NB);
  q.binbase = c.take<uint32_t>((size_t)T * BWD_NB);
  q.cflags = c.take<uint32_t>(q.max_chunks);
  q.clkey = c.take<uint32_t>(q.max_chunks);
  q.ctkey = c.take<uint32_t>(q.max_chunks);
  q.clead = c.take<float>((size_t)q.max_chunks * max_dim);
  q.ctrail = c.take<float>((size_t)q.max_chunks * max_dim);
  if (p) *p = q;
  return c.off;
}

// chunk id -> (table, first position, end position, table span); false for surplus workgroups.
__device__ __forceinline__ bool bwd_chunk(const BwdPlan& P, const TzrTable* tables, int T,
                                          int chunk, int* t_out, int64_t* s_out, int64_t* e_out,
                                          int64_t* tab_s_out, int64_t* tab_e_out) {
  if (chunk >= P.tab_chunk[T]) return false;
  int lo = 0, hi = T;  // last t with tab_chunk[t] <= chunk (the non-empty table holding it)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (P.tab_chunk[mid] <= chunk) lo = mid; else hi = mid;
  }
  const int t = lo;
  const TzrTable tb = tables[t];
  const int64_t ts = P.tab_start[t];
  const int64_t te = tb.n_feats > 0 ? P.feat_start[tb.first_order + tb.n_feats] : ts;
  const int64_t s = ts + (int64_t)(chunk - P.tab_chunk[t]) * BWD_CH;
  *t_out = t;
  *s_out = s;
  *e_out = min(te, s + BWD_CH);
  *tab_s_out = ts;
  *tab_e_out = te;
  return true;
}
