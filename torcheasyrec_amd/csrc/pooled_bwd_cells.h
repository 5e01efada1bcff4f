// The "cells" form of the backward index plan (K6) for batches of one id per bag: ONE launch instead of four.
//
// The four-launch plan (pooled_bwd.hip) orders a table's lookups globally: a histogram per chunk, a scan over the chunks and
// buckets, a partition pass to global offsets -- 40 us at B = 65 536 for no algorithmic byte, most of it launch boundaries and a
// scan that 26 workgroups run on a 256-CU chip (profiles/r05ao).  The global order is not needed since the apply sorts its
// unit itself (round 5): a unit only has to be able to FIND its lookups.  So:
//
//   partition  (one launch, a workgroup per chunk of <= 1024 table-major positions, no communication between workgroups):
//              the chunk's lookups ordered by bucket IN PLACE -- `slab` positions [s, e) hold chunk c's {row id, lookup
//              position} sorted by (bucket, position) -- and, for every unit boundary of its table, where that bucket starts
//              inside the chunk (a column of the boundary table below).
//   apply      a unit = the cells of a bucket range x a chunk range, known at launch time (the geometry below is a function
//              of the tables and the batch size alone, built on the host once and kept on the device by the caller): it
//              reads its cells' bounds (one round trip), gathers them into LDS (one more), sorts by (row id, position) and
//              reduces as before.  Units own DISJOINT row ranges: no run crosses a unit, no stitching -- except the rows of a
//              table with so few rows that one row holds more lookups than a unit: such a row is split over chunk ranges
//              whose units leave partial sums, added in unit order by the last of them to arrive (the tiny tables' mechanism
//              of pooled_bwd_direct.hip).
//
// Unit sizes are EXPECTATIONS (uniformly drawn ids fill the buckets evenly: ~1024 +- 32 lookups against a capacity of
// BWD_UMAX = 1280).  A unit that holds more -- skewed ids -- leaves itself on a list; BWD_CELLS_WORKERS workgroups at the END of
// the same launch's grid wait until every unit has looked at its size, then take the listed units piece by piece
// (pooled_bwd_cells.hip: bwd_cells_worker).  Still one launch, still correct for any ids, and the ordinary units' code holds
// nothing of it.  The geometry buffer's overflow word counts such units: how the caller learns that this id distribution belongs
// on the exact plan, whose heavy-bucket machinery is made for it.
#pragma once
#include "pooled_bwd.h"

#define BWD_CELLS_MAXC 256   // chunks of one table (cells a unit may gather from): 262 144 lookups per table at 1024 per chunk
#define BWD_CELLS_TARGET 1076  // largest EXPECTED size of a unit: BWD_UMAX - 6 sqrt(BWD_UMAX)
#ifndef BWD_CELLS_WORKERS
#define BWD_CELLS_WORKERS 32   // workgroups behind the units of the apply launch that take the units that did not fit
#endif
// words of the overflow area (BwdCellsView::overflow)
#define BWD_CELLS_OVF_TOTAL 0    // units that did not fit since the buffer was made (never reset by the kernels: the caller's signal)
#define BWD_CELLS_OVF_LEN 1      // ... of this launch (reset by the last worker)
#define BWD_CELLS_OVF_EPOCH 2    // launches finished on this buffer: a unit that has looked at its size says so by writing EPOCH + 1
                                 // into its flag word (a plain store -- a shared arrival counter cost every unit a contended atomic)
#define BWD_CELLS_OVF_CURSOR 3   // next list entry to hand out
#define BWD_CELLS_OVF_DONE 4     // workers that have finished
#define BWD_CELLS_OVF_LIST 8     // first of the n_units list words; the n_units flag words follow them

struct BwdCellChunk {  // 64 bytes: everything a partition workgroup needs, in ONE load
  int32_t t;           // table, -1: surplus
  int32_t nb;          // buckets of the table's map
  int64_t s, e;        // table-major positions [s, e) of the chunk
  int64_t ts;          // first position of the table
  uint64_t mult;       // bucket of row id k = (k * mult) >> 32
  int64_t rows;        // (ids are clamped against it)
  int64_t fbase;       // single-key table: index of position ts in the KJT's values (key * B); -1: several keys read the table
  int32_t bnd0, nbnd;  // the table's unit boundaries: entries [bnd0, bnd0 + nbnd) of the boundary list (bucket numbers) = rows of
                       // the boundary table; this chunk's column in it is (chunk - first chunk of the table)
};
static_assert(sizeof(BwdCellChunk) == 64, "one 64-byte load");

// Boundary table (device, in the plan's workspace): uint16 bnd[boundary row][BWD_CELLS_MAXC]: bnd[g][c] = start, inside chunk c of
// the table, of bucket `boundary list[g]` (chunk c's lookups are ordered by bucket).  Cell (unit, chunk c) = the chunk's
// lookups [bnd[i0][c], bnd[i1][c]) -- two CONTIGUOUS 2-byte-per-chunk reads per unit (the partition writes a column of it:
// one 2-byte store per boundary of its table).

struct BwdCellUnit {  // 96 bytes: the unit's table (a copy: no dependent load) + its cells
  TzrTable tb;
  int32_t split;       // > 0: one of `split` units of ONE row (b0) of an exact table: they leave partial sums
  int32_t crel0;       // first chunk of the unit, relative to the table's first chunk
  int32_t ncell;       // chunks of the unit
  int32_t i0, i1;      // boundary rows of the unit's first bucket and of the bucket behind its last
  int32_t b0;          // first bucket (= row id for an exact table)
  int32_t nrows;       // (0; reserved)
  int32_t rec;         // split: index of this unit's partial-sum record
  int32_t rec0;        // split: record of the row's first unit (the row's records are consecutive)
  int32_t counter;     // split: index of the row's arrival counter
  int32_t feat;        // index (in the TzrFeature array) of the table's first key
  uint32_t ts;         // first table-major position of the table
};
static_assert(sizeof(BwdCellUnit) == 96, "two 48-byte halves");

// Geometry buffer (caller-owned device memory, built by tzr_bwd_cells_geometry into a host image the caller uploads once per
// (tables, batch size); the kernels only write its tail: records, counters, the overflow area).
struct BwdCellsGeo {
  int64_t n_chunks, n_units, n_recs, n_counters, n_feats, max_dim, ch, n_positions, n_bnd;
  // byte offsets from the start of the buffer
  int64_t off_chunks, off_units, off_fstart, off_fkey, off_fbo, off_bnd, off_recs, off_rcount, off_counters, off_overflow, bytes;
};

struct BwdCellsView {  // device pointers into the geometry buffer
  const BwdCellChunk* chunks;
  const BwdCellUnit* units;
  const uint32_t* fstart;     // [F + 1]
  const int32_t* fkey;        // [F]
  const int32_t* feat_by_order;  // [F]
  const uint16_t* bnd_bucket; // [n_bnd] the boundary list: bucket number of every boundary row
  float* recs;                // [n_recs * max_dim] partial sums of split units
  uint32_t* rcount;           // [n_recs] lookups behind each partial sum
  uint32_t* counters;         // [n_counters] arrivals per split row (zero between launches)
  uint32_t* overflow;         // [BWD_CELLS_OVF_LIST + 2 n_units]: counters, the list of this launch's units that did not fit, flags
};

static inline BwdCellsView bwd_cells_view(void* base, const BwdCellsGeo& g) {
  char* b = static_cast<char*>(base);
  BwdCellsView v;
  v.chunks = reinterpret_cast<const BwdCellChunk*>(b + g.off_chunks);
  v.units = reinterpret_cast<const BwdCellUnit*>(b + g.off_units);
  v.fstart = reinterpret_cast<const uint32_t*>(b + g.off_fstart);
  v.fkey = reinterpret_cast<const int32_t*>(b + g.off_fkey);
  v.feat_by_order = reinterpret_cast<const int32_t*>(b + g.off_fbo);
  v.bnd_bucket = reinterpret_cast<const uint16_t*>(b + g.off_bnd);
  v.recs = reinterpret_cast<float*>(b + g.off_recs);
  v.rcount = reinterpret_cast<uint32_t*>(b + g.off_rcount);
  v.counters = reinterpret_cast<uint32_t*>(b + g.off_counters);
  v.overflow = reinterpret_cast<uint32_t*>(b + g.off_overflow);
  return v;
}
