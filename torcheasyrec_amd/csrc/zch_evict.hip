// Admission / eviction round of a zero-collision-hash table as a SELECTION, not a sort.
//
// The round (torchrec MCHManagedCollisionModule._update_and_evict [upstream 1.7.0], configured by
// /root/reference/tzrec/features/feature.py:693-736: LFU / LRU / DistanceLFU eviction policies;
// semantics restated in oracle/zch_oracle.py) ranks the R residents and the n new candidates together --
// score descending, residents before candidates, raw id ascending -- and keeps the first Z' = zch_size - 1.
// Sorting R + n entries three times (what zch.py did with torch.sort) moves tens of GB for a 200 M-row table.
// But at most n entries can be dropped (R <= Z'), and WHICH ones is a rank query: the D = R + n - Z' smallest
// in the reverse order, the "drop key"
//     E = ( score ascending            64 bits: order-preserving image of the double
//         , resident ? 1 : 0            1 bit : at equal score candidates go first
//         , ~(raw id, biased)           64 bits: at equal score and kind the LARGER id goes first )
// Keys are unique (a candidate never is a resident), so the D-th smallest E is a threshold T and
// dropped = { E <= T }.  T is found by MSB-first radix selection: one pass per 11-bit digit, each pass one
// streaming read of the table's three per-row arrays (24 B per row) into a 2 048-bin LDS histogram; the host
// reads the bins, narrows the prefix, stops early when a bin is taken whole.  Worst case 13 passes (6 + 1 + 6);
// then ONE pass marks the residents that stay.  No temporaries beyond the 16 KB of bins and the two mark arrays.
#include "tzr_common.h"

#include <math.h>

#define ZE_THREADS 256
#define ZE_BITS 11
#define ZE_BINS (1 << ZE_BITS)

struct ZchRound {  // one round's view of a module + the candidates
  const int64_t* row_ids;    // [zch_size] raw id per row, TZR_ZCH_EMPTY when free
  const int64_t* counts;     // [zch_size]
  const int64_t* last_iter;  // [zch_size]
  int64_t n_rows;            // zch_size - 1 (the last row is the shared fallback row)
  const int64_t* new_ids;    // [n_new] distinct candidates
  const int64_t* new_cnt;    // [n_new] their lookups since the last round
  int64_t n_new;
  int64_t cur_iter;
  int32_t policy;  // 0 lfu, 1 lru, 2 distance lfu
  int32_t pad;
  double decay_exponent;
};

struct ZchCut {  // what is already known of the threshold, and the digit being counted
  uint64_t t1;      // field 0: the bits of the score image above shift + ZE_BITS; fields 1, 2: the whole image
  uint64_t t3;      // field 2: the bits of the id image above shift + ZE_BITS
  int32_t field;    // 0: digit of the score image; 1: the kind bit (bins 0 / 1); 2: digit of the id image
  int32_t shift;    // position of the digit inside its 64-bit field
  int32_t t2;       // field 2: the kind that holds the threshold
  int32_t bits;     // width of the digit
};

__device__ __forceinline__ uint64_t ze_score_image(const ZchRound& R, int64_t cnt, int64_t last) {
  const int64_t d = R.cur_iter - last;
  const double dist = (double)(d < 1 ? 1 : d);
  double sc;
  if (R.policy == 0) {
    sc = (double)cnt;
  } else {
    const double age = R.decay_exponent == 1.0 ? dist : pow(dist, R.decay_exponent);
    sc = R.policy == 1 ? 1.0 / age : (double)cnt / age;
  }
  uint64_t u;
  __builtin_memcpy(&u, &sc, sizeof(u));
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // monotone: smaller double -> smaller image
}

__device__ __forceinline__ uint64_t ze_id_image(int64_t id) {
  return ~((uint64_t)id ^ 0x8000000000000000ull);  // larger id -> smaller image
}

// entry e of the round: e < n_rows -> resident of row e (if any), else candidate e - n_rows
__device__ __forceinline__ bool ze_entry(const ZchRound& R, int64_t e, uint64_t* k1, int* k2, uint64_t* k3) {
  if (e < R.n_rows) {
    const int64_t id = R.row_ids[e];
    if (id == TZR_ZCH_EMPTY) return false;
    *k1 = ze_score_image(R, R.counts[e], R.last_iter[e]);
    *k2 = 1;
    *k3 = ze_id_image(id);
    return true;
  }
  const int64_t j = e - R.n_rows;
  *k1 = ze_score_image(R, R.new_cnt[j], R.cur_iter);
  *k2 = 0;
  *k3 = ze_id_image(R.new_ids[j]);
  return true;
}

__global__ __launch_bounds__(ZE_THREADS) void tzr_zch_select_hist_kernel(ZchRound R, ZchCut C,
                                                                         unsigned long long* __restrict__ bins) {
  __shared__ unsigned int h[ZE_BINS];
  for (int i = threadIdx.x; i < ZE_BINS; i += ZE_THREADS) h[i] = 0;
  __syncthreads();
  const int64_t total = R.n_rows + R.n_new;
  const unsigned mask = (1u << C.bits) - 1u;
  for (int64_t e = (int64_t)blockIdx.x * ZE_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * ZE_THREADS) {
    uint64_t k1, k3;
    int k2;
    if (!ze_entry(R, e, &k1, &k2, &k3)) continue;
    unsigned d;
    if (C.field == 0) {
      const int hi = C.shift + C.bits;
      if (hi < 64 && (k1 >> hi) != C.t1) continue;
      d = (unsigned)(k1 >> C.shift) & mask;
    } else if (C.field == 1) {
      if (k1 != C.t1) continue;
      d = (unsigned)k2;
    } else {
      if (k1 != C.t1 || k2 != C.t2) continue;
      const int hi = C.shift + C.bits;
      if (hi < 64 && (k3 >> hi) != C.t3) continue;
      d = (unsigned)(k3 >> C.shift) & mask;
    }
    atomicAdd(&h[d], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ZE_BINS; i += ZE_THREADS)
    if (h[i]) atomicAdd(&bins[i], (unsigned long long)h[i]);
}

// kept[e] = 1 for every entry whose drop key is ABOVE the threshold (t1, t2, t3), residents in
// row_kept[n_rows] (empty rows: 0), candidates in new_kept[n_new]
__global__ __launch_bounds__(ZE_THREADS) void tzr_zch_select_mark_kernel(ZchRound R, uint64_t t1, int t2, uint64_t t3,
                                                                         int drop_none, uint8_t* __restrict__ row_kept,
                                                                         uint8_t* __restrict__ new_kept) {
  const int64_t total = R.n_rows + R.n_new;
  for (int64_t e = (int64_t)blockIdx.x * ZE_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * ZE_THREADS) {
    uint64_t k1, k3;
    int k2;
    const bool live = ze_entry(R, e, &k1, &k2, &k3);
    bool keep = live;
    if (live && !drop_none) {
      const bool le = k1 < t1 || (k1 == t1 && (k2 < t2 || (k2 == t2 && k3 <= t3)));
      keep = !le;
    }
    if (e < R.n_rows) row_kept[e] = keep ? 1 : 0;
    else new_kept[e - R.n_rows] = keep ? 1 : 0;
  }
}

static int ze_round(const TzrZchModule* h_module, const int64_t* d_row_ids, const int64_t* d_new_ids,
                    const int64_t* d_new_cnt, int64_t n_new, int64_t cur_iter, int policy, double decay_exponent,
                    ZchRound* R) {
  if (!h_module || !d_row_ids || !h_module->counts || !h_module->last_iter || h_module->zch_size < 2 || n_new < 0 ||
      policy < 0 || policy > 2)
    return TZR_ERR_INVALID;
  if (n_new > 0 && (!d_new_ids || !d_new_cnt)) return TZR_ERR_INVALID;
  R->row_ids = d_row_ids;
  R->counts = h_module->counts;
  R->last_iter = h_module->last_iter;
  R->n_rows = h_module->zch_size - 1;
  R->new_ids = d_new_ids;
  R->new_cnt = d_new_cnt;
  R->n_new = n_new;
  R->cur_iter = cur_iter;
  R->policy = policy;
  R->pad = 0;
  R->decay_exponent = decay_exponent;
  return TZR_OK;
}

static unsigned ze_grid(int64_t total) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(256 * 16, (total + ZE_THREADS - 1) / ZE_THREADS));
}

extern "C" int tzr_zch_select_hist(const TzrZchModule* h_module, const int64_t* d_row_ids, const int64_t* d_new_ids,
                                   const int64_t* d_new_cnt, int64_t n_new, int64_t cur_iter, int policy,
                                   double decay_exponent, int field, int shift, int bits, uint64_t t1, int t2,
                                   uint64_t t3, uint64_t* d_bins, void* stream) {
  ZchRound R;
  const int rc = ze_round(h_module, d_row_ids, d_new_ids, d_new_cnt, n_new, cur_iter, policy, decay_exponent, &R);
  if (rc != TZR_OK) return rc;
  if (!d_bins || field < 0 || field > 2 || bits < 1 || bits > ZE_BITS || shift < 0 || shift + bits > 64 ||
      (field == 1 && (shift != 0 || bits != 1)))
    return TZR_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(d_bins, 0, ZE_BINS * sizeof(uint64_t), s) != hipSuccess) return TZR_ERR_LAUNCH;
  ZchCut C;
  C.t1 = t1;
  C.t3 = t3;
  C.field = field;
  C.shift = shift;
  C.t2 = t2;
  C.bits = bits;
  hipLaunchKernelGGL(tzr_zch_select_hist_kernel, dim3(ze_grid(R.n_rows + R.n_new)), dim3(ZE_THREADS), 0, s, R, C,
                     reinterpret_cast<unsigned long long*>(d_bins));
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}

extern "C" int tzr_zch_select_mark(const TzrZchModule* h_module, const int64_t* d_row_ids, const int64_t* d_new_ids,
                                   const int64_t* d_new_cnt, int64_t n_new, int64_t cur_iter, int policy,
                                   double decay_exponent, int drop_none, uint64_t t1, int t2, uint64_t t3,
                                   uint8_t* d_row_kept, uint8_t* d_new_kept, void* stream) {
  ZchRound R;
  const int rc = ze_round(h_module, d_row_ids, d_new_ids, d_new_cnt, n_new, cur_iter, policy, decay_exponent, &R);
  if (rc != TZR_OK) return rc;
  if (!d_row_kept || (n_new > 0 && !d_new_kept)) return TZR_ERR_INVALID;
  hipLaunchKernelGGL(tzr_zch_select_mark_kernel, dim3(ze_grid(R.n_rows + R.n_new)), dim3(ZE_THREADS), 0,
                     static_cast<hipStream_t>(stream), R, t1, t2, t3, drop_none, d_row_kept, d_new_kept);
  TZR_CHECK_LAUNCH();
  return TZR_OK;
}
