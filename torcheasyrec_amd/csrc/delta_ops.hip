// Delta-embedding tracker for gfx950: which rows of which table were looked up since the last dump.
//
// Replaces the id store behind tzrec's ModelDeltaTracker
// (/root/reference/tzrec/utils/delta_embedding_dump.py:352-641): torchrec's DeltaStoreTrec.append keeps
// one id tensor per (batch, table) and `compact` / `get_unique` run torch.cat(...).unique() over them
// (:478-513, :565-609) -- memory grows with the dump interval and every compaction is a device sort.
// Here a table's touched set is a BITMAP in HBM, one bit per local row (40 M rows = 5 MB; the whole
// DLRM-Criteo model 25.5 MB of 288 GB): recording a lookup is one pass over the ids (8 B read per id,
// a 4-byte atomic OR only for rows not marked yet), memory is constant, and "sorted unique ids" is a
// popcount scan of the bitmap -- no sort anywhere.
//
//   tzr_delta_mark      ids of the lookup segments -> bits           (every step, HBM/L2-bound)
//   tzr_delta_count     bitmap -> number of touched rows             (dump time)
//   tzr_delta_collect   bitmap -> ascending row ids (+ clear)        (dump time)
#include "tzr_common.h"

#define DL_THREADS 256
#define DL_TILES 16
#define DL_WORDS_PER_BLOCK (DL_THREADS * DL_TILES)  // 4096 words = 131 072 rows per workgroup

__global__ __launch_bounds__(DL_THREADS) void tzr_delta_mark_kernel(
    const TzrDeltaSeg* __restrict__ segs, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ key_offsets, int64_t key_stride, int64_t uniform_len,
    unsigned long long* __restrict__ oob) {
  const TzrDeltaSeg S = segs[blockIdx.y];
  if (!S.bitmap) return;
  const int64_t k = S.key;
  const int64_t s = key_offsets ? key_offsets[k * key_stride] : k * key_stride * uniform_len;
  const int64_t e = key_offsets ? key_offsets[(k + 1) * key_stride] : (k + 1) * key_stride * uniform_len;
  unsigned bad = 0;
  for (int64_t i = s + (int64_t)blockIdx.x * DL_THREADS + threadIdx.x; i < e;
       i += (int64_t)gridDim.x * DL_THREADS) {
    const int64_t id = ids[i];
    if ((uint64_t)id >= (uint64_t)S.rows) {
      ++bad;
      continue;
    }
    uint32_t* w = S.bitmap + (id >> 5);
    const uint32_t bit = 1u << (id & 31);
    // bits only ever go 0 -> 1 between two collects: a stale read costs one redundant atomic, a
    // fresh one saves it (hot rows of small tables are marked by the first lookup of the interval)
    if (!(*reinterpret_cast<volatile uint32_t*>(w) & bit)) atomicOr(w, bit);
  }
  if (bad && oob) atomicAdd(oob, (unsigned long long)bad);
}

// Sum over the workgroup; result valid in every thread.  `red`: DL_THREADS / TZR_WAVE slots of LDS.
__device__ __forceinline__ unsigned dl_block_sum(unsigned v, unsigned* red) {
  for (int d = TZR_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
  const int wave = threadIdx.x / TZR_WAVE;
  __syncthreads();  // red may still be read by the previous call
  if ((threadIdx.x & (TZR_WAVE - 1)) == 0) red[wave] = v;
  __syncthreads();
  unsigned t = 0;
  for (int w = 0; w < DL_THREADS / TZR_WAVE; ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(DL_THREADS) void tzr_delta_count_kernel(
    const uint32_t* __restrict__ bitmap, int64_t n_words, uint32_t* __restrict__ block_counts,
    unsigned long long* __restrict__ total) {
  __shared__ unsigned red[DL_THREADS / TZR_WAVE];
  const int64_t base = (int64_t)blockIdx.x * DL_WORDS_PER_BLOCK;
  unsigned c = 0;
  for (int j = 0; j < DL_TILES; ++j) {
    const int64_t idx = base + (int64_t)j * DL_THREADS + threadIdx.x;
    if (idx < n_words) c += __popc(bitmap[idx]);
  }
  const unsigned t = dl_block_sum(c, red);
  if (threadIdx.x == 0) {
    block_counts[blockIdx.x] = t;
    if (total && t) atomicAdd(total, (unsigned long long)t);
  }
}

// Exclusive scan of the per-workgroup counts (one workgroup; the list is rows / 131 072 long).
__global__ __launch_bounds__(DL_THREADS) void tzr_delta_scan_kernel(
    const uint32_t* __restrict__ block_counts, int64_t n_blocks, int64_t* __restrict__ block_start) {
  __shared__ int64_t wave_tot[DL_THREADS / TZR_WAVE];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wave = threadIdx.x / TZR_WAVE;
  int64_t carry = 0;
  for (int64_t t0 = 0; t0 < n_blocks; t0 += DL_THREADS) {
    const int64_t i = t0 + threadIdx.x;
    const int64_t c = i < n_blocks ? (int64_t)block_counts[i] : 0;
    int64_t inc = c;  // inclusive scan inside the wave
    for (int d = 1; d < TZR_WAVE; d <<= 1) {
      const int64_t up = __shfl_up(inc, d);
      if (lane >= d) inc += up;
    }
    __syncthreads();
    if (lane == TZR_WAVE - 1) wave_tot[wave] = inc;
    __syncthreads();
    int64_t before = 0, all = 0;
    for (int w = 0; w < DL_THREADS / TZR_WAVE; ++w) {
      if (w < wave) before += wave_tot[w];
      all += wave_tot[w];
    }
    if (i < n_blocks) block_start[i] = carry + before + inc - c;
    carry += all;
  }
}

__global__ __launch_bounds__(DL_THREADS) void tzr_delta_emit_kernel(
    uint32_t* __restrict__ bitmap, int64_t n_words, const int64_t* __restrict__ block_start,
    int64_t id_base, int clear, int64_t* __restrict__ out, int64_t capacity) {
  __shared__ unsigned wave_tot[DL_THREADS / TZR_WAVE];
  const int lane = threadIdx.x & (TZR_WAVE - 1);
  const int wave = threadIdx.x / TZR_WAVE;
  const int64_t base = (int64_t)blockIdx.x * DL_WORDS_PER_BLOCK;
  int64_t pos0 = block_start[blockIdx.x];
  for (int j = 0; j < DL_TILES; ++j) {
    const int64_t idx = base + (int64_t)j * DL_THREADS + threadIdx.x;
    uint32_t w = idx < n_words ? bitmap[idx] : 0u;
    const unsigned c = __popc(w);
    unsigned inc = c;
    for (int d = 1; d < TZR_WAVE; d <<= 1) {
      const unsigned up = __shfl_up(inc, d);
      if (lane >= d) inc += up;
    }
    __syncthreads();
    if (lane == TZR_WAVE - 1) wave_tot[wave] = inc;
    __syncthreads();
    unsigned before = 0, all = 0;
    for (int q = 0; q < DL_THREADS / TZR_WAVE; ++q) {
      if (q < wave) before += wave_tot[q];
      all += wave_tot[q];
    }
    if (w) {
      int64_t pos = pos0 + before + inc - c;
      const int64_t first = id_base + idx * 32;
      if (clear) bitmap[idx] = 0u;
      while (w) {
        const int b = __ffs((int)w) - 1;
        if (pos < capacity) out[pos] = first + b;
        ++pos;
        w &= w - 1;
      }
    }
    pos0 += all;
  }
}

static inline int64_t dl_words(int64_t rows) { return (rows + 31) / 32; }
static inline int64_t dl_blocks(int64_t rows) {
  return std::max<int64_t>(1, (dl_words(rows) + DL_WORDS_PER_BLOCK - 1) / DL_WORDS_PER_BLOCK);
}

extern "C" int tzr_delta_mark(const TzrDeltaSeg* d_segs, int n_segs, const int64_t* d_ids,
                              const int64_t* d_key_offsets, int64_t key_stride, int64_t uniform_len,
                              int64_t n_ids, int64_t* d_oob, void* stream) {
  if (n_segs < 0 || n_ids < 0 || key_stride < 0 || uniform_len < 0) return TZR_ERR_INVALID;
  if (!d_key_offsets && uniform_len == 0 && n_ids > 0) return TZR_ERR_INVALID;
  if (n_segs == 0 || n_ids == 0) return TZR_OK;
  if (!d_segs || !d_ids) return TZR_ERR_INVALID;
  if (n_segs > 65535) return TZR_ERR_UNSUPPORTED;
  const int64_t per_seg = (n_ids + n_segs - 1) / n_segs;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (per_seg + DL_THREADS - 1) / DL_THREADS));
  hipLaunchKernelGGL(tzr_delta_mark_kernel, dim3(gx, (unsigned)n_segs), dim3(DL_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_segs, d_ids, d_key_offsets, key_stride,
                     uniform_len, reinterpret_cast<unsigned long long*>(d_oob));
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" size_t tzr_delta_collect_workspace(int64_t rows) {
  if (rows < 0) return 0;
  const int64_t nb = dl_blocks(rows);
  return tzr_align_up((size_t)nb * sizeof(uint32_t)) + tzr_align_up((size_t)nb * sizeof(int64_t)) + 256;
}

extern "C" int tzr_delta_count(const uint32_t* d_bitmap, int64_t rows, int64_t* d_total, void* ws,
                               size_t ws_bytes, void* stream) {
  if (rows < 0 || !d_total) return TZR_ERR_INVALID;
  if (rows == 0) return TZR_OK;  // *d_total is an accumulator: the caller zeroes it
  if (!d_bitmap || !ws) return TZR_ERR_INVALID;
  if (ws_bytes < tzr_delta_collect_workspace(rows)) return TZR_ERR_WORKSPACE;
  TzrCarver carve(ws);
  const int64_t nb = dl_blocks(rows);
  uint32_t* counts = carve.take<uint32_t>((size_t)nb);
  hipLaunchKernelGGL(tzr_delta_count_kernel, dim3((unsigned)nb), dim3(DL_THREADS), 0,
                     static_cast<hipStream_t>(stream), d_bitmap, dl_words(rows), counts,
                     reinterpret_cast<unsigned long long*>(d_total));
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_delta_collect(uint32_t* d_bitmap, int64_t rows, int64_t id_base, int clear,
                                 int64_t* d_out_ids, int64_t capacity, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (rows < 0 || capacity < 0) return TZR_ERR_INVALID;
  if (rows == 0) return TZR_OK;
  if (!d_bitmap || !ws || (capacity > 0 && !d_out_ids)) return TZR_ERR_INVALID;
  if (ws_bytes < tzr_delta_collect_workspace(rows)) return TZR_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  TzrCarver carve(ws);
  const int64_t nb = dl_blocks(rows), nw = dl_words(rows);
  uint32_t* counts = carve.take<uint32_t>((size_t)nb);
  int64_t* start = carve.take<int64_t>((size_t)nb);
  hipLaunchKernelGGL(tzr_delta_count_kernel, dim3((unsigned)nb), dim3(DL_THREADS), 0, s, d_bitmap, nw,
                     counts, static_cast<unsigned long long*>(nullptr));
  hipLaunchKernelGGL(tzr_delta_scan_kernel, dim3(1), dim3(DL_THREADS), 0, s, counts, nb, start);
  hipLaunchKernelGGL(tzr_delta_emit_kernel, dim3((unsigned)nb), dim3(DL_THREADS), 0, s, d_bitmap, nw,
                     start, id_base, clear, d_out_ids, capacity);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
