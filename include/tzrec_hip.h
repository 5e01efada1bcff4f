/*
 * tzrec_hip.h -- C ABI of libtzrec_hip.so, the MI355X (gfx950) sharded-embedding hot path
 * for TorchEasyRec (tzrec).
 *
 * This header is the drop-in boundary ("kernel-level seam").  Every entry
 * point replaces one call the reference makes into an un-vendored native wheel (fbgemm-gpu 1.7.0 /
 * torchrec 1.7.0, pinned in /root/reference/requirements/runtime.txt) or one pure-PyTorch module of
 * the reference.  The reference call site that reaches each op is cited next to it
 * (paths relative to /root/reference).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every `d_*` pointer is DEVICE memory; `h_*` is host memory read during the call only.
 *   - `stream` is a hipStream_t passed as void*; every launch is asynchronous on it.
 *   - no allocation inside: scratch comes from the caller (`ws`, sized by the matching
 *     `*_workspace` query, 256-byte aligned).
 *   - return 0 (TZR_OK) or a negative TZR_ERR_* code; never throws, never syncs the device.
 *   - thread-compatible: one caller thread per device (the reference runs one Python thread per
 *     rank, tzrec/utils/dist_util.py:57-75).
 *
 * Jagged input contract = torchrec KeyedJaggedTensor as tzrec builds it
 * (tzrec/datasets/data_parser.py:576-585): `values` int64[N] concatenated key-major then
 * sample-major, `lengths` per (key, sample) key-major, stride = B, `offsets` = [0] + cumsum(lengths)
 * (int64[F*B+1]), optional per-id float `weights`.
 */
#ifndef TZREC_HIP_H_
#define TZREC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TZR_DT_F32 0
#define TZR_DT_F16 1

#define TZR_OK 0
#define TZR_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, unsupported dim) */
#define TZR_ERR_LAUNCH (-2)      /* hipLaunch / hipMemsetAsync reported an error */
#define TZR_ERR_WORKSPACE (-3)   /* workspace too small or misaligned */
#define TZR_ERR_UNSUPPORTED (-4) /* valid request this build has no kernel for */

#define TZR_MAX_DST 8      /* destination (feature-group) buffers per pooled lookup */
#define TZR_MAX_FEAT_DST 4 /* feature groups one feature may be copied into */

#define TZR_POOL_SUM 0
#define TZR_POOL_MEAN 1

#define TZR_OPT_SGD 0
#define TZR_OPT_ADAGRAD 1         /* elementwise state  [rows, D]  (torchrec.optim Adagrad) */
#define TZR_OPT_ROWWISE_ADAGRAD 2 /* one scalar per row [rows]     (RowWiseAdagrad)         */
#define TZR_OPT_ADAM 4            /* elementwise exp_avg | exp_avg_sq [rows, 2 D] (torchrec.optim Adam):
                                     state row = [m(D) | v(D)], m_stride >= 2 D             */
#define TZR_OPT_ACCUMULATE 3      /* no update: the summed gradient of every touched row is
                                     written to TzrTable.m (dense float [rows, dim]); used for
                                     replicated (data_parallel) tables before the all-reduce  */

#define TZR_WD_NONE 0
#define TZR_WD_L2 1
#define TZR_WD_DECOUPLE 2

#define TZR_BOUNDS_FATAL 0   /* count out-of-range ids, leave them (caller raises)   */
#define TZR_BOUNDS_WARNING 1 /* count and clamp to row 0 (fbgemm default [upstream]) */
#define TZR_BOUNDS_IGNORE 2  /* clamp silently                                        */

/* One embedding table shard resident in HBM.  Replaces the fbgemm TBE weight/optimizer-state
 * buffers (tzrec/optim/optimizer.py:106-255).  `w`/`m` are device addresses of row 0; a row is
 * `dim` floats, rows are `*_stride` floats apart (stride 2*dim with m = w + dim gives the
 * interleaved [w|m] 128-byte row used for Adagrad at dim 16). */
typedef struct TzrTable {
  uint64_t w;       /* device address of weights row 0: float* or, w_dtype = TZR_DT_F16, half*  */
  uint64_t m;       /* float* device address of optimizer state row 0 (0 for SGD)             */
  int64_t rows;     /* rows held by this shard                                                */
  int32_t dim;      /* embedding dim D (multiple of 4, <= 256)                                */
  int32_t w_stride; /* ELEMENTS (floats / halves) between consecutive weight rows               */
  int32_t m_stride; /* floats between consecutive state rows (rowwise adagrad: 1)             */
  int32_t first_order; /* smallest TzrFeature.order among the keys reading this table         */
  int32_t n_feats;  /* number of KJT keys reading this table (their orders are consecutive)   */
  int32_t w_dtype;  /* TZR_DT_F32 | TZR_DT_F16 (tzrec feature config `data_type`,
                       tzrec/features/feature.py:346-356,626): half weights are widened to fp32
                       on read; the fused optimizer computes in fp32 and rounds to nearest even
                       on write; optimizer state is always fp32                              */
} TzrTable; /* 48 bytes */

/* One lookup = (KJT key -> table).  Usually one per key; a key read through two tables (DeepFM:
 * `cat_0` feeds `cat_0_emb` for the fm/deep groups and `cat_0_emb_wide` for the wide group,
 * tzrec/modules/embedding.py:744-745,777-786) has one TzrFeature per table.  `dst`/`col` say
 * where its pooled [B, dim] block lands: feature groups sharing a lookup (DeepFM `fm` and `deep`,
 * embedding.py:972-976) each get a copy, and the backward sums the groups' gradients. */
typedef struct TzrFeature {
  int32_t table;   /* index into the TzrTable array                                           */
  int32_t key;     /* KJT key index: bag (key, b) is offsets[key*B + b] .. offsets[key*B+b+1]   */
  int32_t pooling; /* TZR_POOL_SUM / TZR_POOL_MEAN (tzrec/features/feature.py:586-591)         */
  int32_t n_dst;   /* 1..TZR_MAX_FEAT_DST                                                      */
  int32_t dst[TZR_MAX_FEAT_DST]; /* destination buffer index                                   */
  int32_t col[TZR_MAX_FEAT_DST]; /* first float column inside that buffer (multiple of 4)      */
  int32_t order;   /* rank of this lookup when lookups are sorted by (table, index): the
                      table-major order the backward plan groups ids in.  A table is read at
                      most once per key.                                                       */
  int32_t reserved[3];
} TzrFeature; /* 64 bytes */

/* One float4 of one destination row: the forward kernel walks slots so that consecutive lanes
 * write consecutive float4s of a destination row (fuses fbgemm permute_pooled_embs /
 * KeyedTensor.regroup_as_dict, tzrec/modules/embedding.py:972-976). */
typedef struct TzrSlot {
  int32_t feature; /* KJT key index                                                            */
  int32_t chunk;   /* float4 index inside the embedding row                                    */
  int32_t dst;     /* destination buffer index                                                 */
  int32_t col;     /* float column in the destination buffer                                   */
} TzrSlot; /* 16 bytes */

/* Destination / gradient buffer of one feature group: float [B, stride]. */
typedef struct TzrDst {
  uint64_t ptr;   /* float* device address                                                     */
  int64_t stride; /* floats between consecutive samples (multiple of 4)                        */
} TzrDst;

/* Fused sparse optimizer applied inside the backward (tzrec/main.py:774-781,
 * tzrec/optim/optimizer_builder.py:30-97, protos/optimizer.proto:76-139). */
typedef struct TzrSparseOptim {
  int32_t kind;              /* TZR_OPT_*                                                       */
  int32_t weight_decay_mode; /* TZR_WD_* (rowwise adagrad only)                                 */
  uint64_t d_lr;             /* const float* DEVICE scalar: schedulers mutate it per step
                                (tzrec/main.py:877-879); graph-replay safe                      */
  float eps;                 /* fbgemm default 1e-8 [upstream]; not configurable from tzrec     */
  float weight_decay;
  float max_gradient;        /* used when gradient_clipping != 0                                */
  int32_t gradient_clipping;
  float beta1;               /* TZR_OPT_ADAM (protos/optimizer.proto:89-96)                     */
  float beta2;
  uint64_t d_adam;           /* TZR_OPT_ADAM: float[4] DEVICE state {step, 1 - beta1^step,
                                1 - beta2^step, -}, advanced once per training step by
                                tzr_sparse_adam_tick (graph-replay safe)                        */
} TzrSparseOptim; /* 48 bytes */

/* Sparse Adam step counter: d_adam[0] += 1, d_adam[1] = 1 - beta1^step, d_adam[2] = 1 - beta2^step.
 * Call once per training step BEFORE the step's tzr_pooled_bwd_apply / tzr_dense_rows_update calls
 * (fbgemm's `iter`, incremented once per step [upstream]). */
int tzr_sparse_adam_tick(float* d_adam, float beta1, float beta2, void* stream);

/* ---- library identity ------------------------------------------------------------------- */
const char* tzr_backend(void); /* "hip-gfx950" (the CPU lane emulator used by tests says "emu") */
int tzr_abi_version(void);

/* ---- index stage (integer, bit-exact) ---------------------------------------------------- */

/* K3: lengths -> offsets (exclusive scan, offsets[n] = total).  Replaces fbgemm
 * asynchronous_complete_cumsum reached from KeyedJaggedTensor.offsets()
 * (tzrec/modules/embedding.py:930).  lengths_itemsize is 4 (int32) or 8 (int64). */
size_t tzr_lengths_to_offsets_workspace(int64_t n);
int tzr_lengths_to_offsets(const void* d_lengths, int lengths_itemsize, int64_t n,
                           int64_t* d_offsets, void* ws, size_t ws_bytes, void* stream);

/* K4: bounds check of every id against its table's row count.  Replaces fbgemm
 * bounds_check_indices (mode from ParameterConstraints.bounds_check_mode,
 * tzrec/utils/plan_util.py:580,594).  d_oob_count (int64[1]) is incremented per bad id;
 * WARNING/IGNORE clamp the id to 0 in place. */
int tzr_bounds_check(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                     int64_t* d_values, const int64_t* d_offsets, int64_t B, int mode,
                     int64_t* d_oob_count, void* stream);

/* K1: KJT permute.  Replaces fbgemm permute_2D_sparse_data reached from the sharded
 * input_dist (tzrec/modules/embedding.py:930 -> torchrec KJT.permute [upstream]).
 * out key t takes input key permute[t] (T output keys, F input keys, repeats allowed).
 * d_in_offsets is int64[F*B+1]; d_out_offsets (int64[T*B+1]) and d_out_lengths are written;
 * d_out_values / d_out_weights must hold the permuted total (<= caller's bound n_out_max). */
size_t tzr_kjt_permute_workspace(int64_t T, int64_t B);
int tzr_kjt_permute(const int32_t* d_permute, int T, int F, int64_t B,
                    const void* d_in_lengths, int lengths_itemsize, const int64_t* d_in_offsets,
                    const int64_t* d_in_values, const float* d_in_weights,
                    void* d_out_lengths, int64_t* d_out_offsets, int64_t* d_out_values,
                    float* d_out_weights, int64_t n_out_max, void* ws, size_t ws_bytes,
                    void* stream);

/* K2: row-wise bucketize.  Replaces fbgemm block_bucketize_sparse_features (row-wise sharded
 * tables; RW geometry = contiguous ceil(rows/W) blocks, tzrec/utils/plan_util.py:
 * 1049-1060 -> torchrec calculate_shard_sizes_and_offsets [upstream]).  For id x of key f:
 * dst = x / block_size[f] (clamped to W-1), local = x - dst*block_size[f].  Output is
 * rank-major: new_lengths[(r*F + f)*B + b]; ids keep their relative order inside each
 * (r, f, b) bag.  d_unbucketize_permute[i] = output position of input id i (nullable).
 * d_rank_offsets (nullable, int32[F]) rotates the owner: dst = (offset[f] + x / block_size[f]) mod W,
 * which spreads tables with fewer rows than ranks (and table-wise placement: block_size = rows,
 * offset = owner) over the node instead of piling them on rank 0.
 * block_size[f] == 0 selects hash routing for key f: dst = splitmix64(x) mod W and the id travels
 * unchanged (zero-collision-hash tables, whose owners map raw ids to rows themselves). */
size_t tzr_block_bucketize_workspace(int64_t F, int64_t B, int W);
int tzr_block_bucketize(const int64_t* d_block_sizes, const int32_t* d_rank_offsets, int F,
                        int64_t B, int W,
                        const int64_t* d_offsets, const int64_t* d_values, const float* d_weights,
                        int64_t n_values, void* d_new_lengths, int lengths_itemsize,
                        int64_t* d_new_offsets, int64_t* d_new_values, float* d_new_weights,
                        int64_t* d_unbucketize_permute, void* ws, size_t ws_bytes, void* stream);

/* Requester side of the sharded exchange for a KJT whose bags all hold `bag_len` ids (Criteo: 1):
 * K1 + K2 restricted to what the id-granularity exchange consumes, in 3 launches instead of 13.
 * For the n_sel selected keys (d_sel[f'] = KJT key index; block size / rank offset per selected key
 * as in tzr_block_bucketize, block size 0 = hash routing):
 *   d_out_ids[N']       local ids grouped by (owner rank, selected key), lookup order inside a group
 *   d_unbucketize[N']   position in d_out_ids of lookup j of selected key f' at index f'*B*bag_len + j
 *   d_counts[W*n_sel]   ids per (rank, key), rank-major (what the counts all-to-all sends)
 * with N' = n_sel * B * bag_len.  W <= 64 and W * n_sel <= 256 in this build.  Same results as
 * tzr_kjt_permute + tzr_block_bucketize (tests compare them). */
size_t tzr_exchange_bucketize_workspace(int n_sel, int64_t n_per_key, int W);
int tzr_exchange_bucketize(const int32_t* d_sel, int n_sel, const int64_t* d_block_sizes,
                           const int32_t* d_rank_offsets, int64_t B, int bag_len, int W,
                           const int64_t* d_values, int64_t* d_out_ids, int64_t* d_unbucketize,
                           int64_t* d_counts, void* ws, size_t ws_bytes, void* stream);

/* Capacity-bounded form of the same (the reference's input dist is torchrec's KJTAllToAll reached
 * from TrainPipelineSparseDist, tzrec/utils/dist_util.py:221-303, whose split sizes cross the host every step):
 * every destination rank owns a FIXED slice of `d_message`, so the ids all-to-all has equal splits known
 * without reading anything back, and the whole sharded step keeps static shapes.
 *   S = tzr_exchange_message_stride(n_sel, capacity) = n_sel + 1 + capacity   (int64 words per destination)
 *   d_message[d*S + f']          ids of selected key f' sent to rank d (clamped to what fits)
 *   d_message[d*S + n_sel]       1 if this rank had to DROP ids for any destination, else 0
 *   d_message[d*S + n_sel+1 ..]  the ids, grouped by key, lookup order inside a key
 *   d_unbucketize[N']            position (in words from the start of d_message) of every lookup's id; a
 *                                dropped lookup points at its destination's first id slot
 * A step whose overflow word is set anywhere must be redone through the exact exchange (the caller checks the
 * flag tzr_exchange_owner_segments returns before it uses anything computed from the message). */
int64_t tzr_exchange_message_stride(int n_sel, int64_t capacity);
int tzr_exchange_bucketize_capped(const int32_t* d_sel, int n_sel, const int64_t* d_block_sizes,
                                  const int32_t* d_rank_offsets, int64_t B, int bag_len, int W,
                                  const int64_t* d_values, int64_t capacity, int64_t* d_message,
                                  int64_t* d_unbucketize, void* ws, size_t ws_bytes, void* stream);
/* The same layout from a DENSE bucketize result (ragged / weighted bags: tzr_block_bucketize): d_counts[W*n_sel] ids per
 * (rank, key), d_ids[n_ids] rank-major, d_unbucketize[n_ids] = dense position of every lookup -> d_message as above and
 * d_unbucketize_out[n_ids] = message position of every lookup (may alias d_unbucketize). */
int tzr_exchange_pad(const int64_t* d_counts, int W, int n_sel, int64_t capacity, const int64_t* d_ids,
                     const int64_t* d_unbucketize, int64_t n_ids, int64_t* d_message,
                     int64_t* d_unbucketize_out, void* stream);
/* Owner side: the received message (W slices of S words, source-rank major) -> key segments over its W*S
 * positions for tzr_rows_gather / tzr_pooled_bwd_plan / _apply (ids pointer = d_message itself):
 *   key s*(n_sel+1)          dead (the words before rank s's ids: unused capacity of rank s-1 + the header)
 *   key s*(n_sel+1) + 1 + f' the ids of selected key f' from rank s
 *   key W*(n_sel+1)          dead (unused capacity of the last rank)
 * d_key_start: int64[W*(n_sel+1) + 2].  *d_overflow = 1 when any source rank dropped ids. */
int tzr_exchange_owner_segments(const int64_t* d_message, int W, int n_sel, int64_t capacity,
                                int64_t* d_key_start, int64_t* d_overflow, void* stream);

/* ---- pooled embedding lookup ------------------------------------------------------------- */

/* K5 (+K8 fused): pooled gather forward.  Replaces torchrec EmbeddingBagCollection.forward ->
 * fbgemm split_embedding_codegen_forward_{un,}weighted (self.ebc(kjt),
 * tzrec/modules/embedding.py:930) and the regroup copy (:972-976):
 *   dst[g][b, col_f : col_f+D_f] = sum_{i in bag(f,b)} weight_i * W_t[id_i, :]   (mean: / len)
 * Empty bag -> zeros.  h_dsts: host array of n_dst destination buffers.  d_slots enumerates every
 * float4 of every destination row exactly once, in destination-then-column order. */
int tzr_pooled_fwd(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                   const TzrSlot* d_slots, int n_slots, const int64_t* d_values,
                   const int64_t* d_offsets, const float* d_weights, int64_t B,
                   const TzrDst* h_dsts, int n_dst, int uniform_bag_len, void* stream);
/* Same, with flags.  TZR_FWD_MIXED_DTYPE: some table holds fp16 rows (TzrTable.w_dtype); without
 * the flag (and in tzr_pooled_fwd) every table is read as fp32. */
#define TZR_FWD_MIXED_DTYPE 1
int tzr_pooled_fwd_ex(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                      const TzrSlot* d_slots, int n_slots, const int64_t* d_values,
                      const int64_t* d_offsets, const float* d_weights, int64_t B,
                      const TzrDst* h_dsts, int n_dst, int uniform_bag_len, int flags, void* stream);

/* K6: backward index plan = group the N lookups by (table, row), duplicates adjacent, original
 * order kept inside a row (stable segmented radix sort).  Replaces fbgemm
 * transpose_embedding_input (linearize + cub radix sort + run-length) reached from the autograd
 * of self.ebc(kjt).  Depends only on the ids, so it can run ahead of the forward on another
 * stream.  The plan lives in `ws` and is consumed by tzr_pooled_bwd_apply. */
size_t tzr_pooled_bwd_workspace(int64_t n_values, int64_t n_positions, int n_feats, int n_tables,
                                int64_t B,
                                int max_dim);
int tzr_pooled_bwd_plan(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats,
                        int n_feats, int n_keys, int64_t max_rows, int max_dim, const int64_t* d_values,
                        const int64_t* d_offsets, int64_t n_values, int64_t n_positions,
                        int64_t B,
                        int uniform_bag_len, void* ws, size_t ws_bytes, void* stream);

/* K6 + K7 in ONE launch, no index plan: the small-batch form of the fused backward (csrc/pooled_bwd_direct.hip).
 * Same semantics and argument meaning as tzr_pooled_bwd_plan followed by tzr_pooled_bwd_apply -- per distinct (table, row)
 * of the batch ONE update with the sum of its lookups' gradients, summation order a function of the ids alone -- for
 * batches whose per-table id lists are small enough to be read by every workgroup of the table (a rank's 8 192-sample
 * share of examples/dlrm_criteo.config: 213 k lookups).  Workgroup (t, j) owns the j-th of k equal row ranges of table t,
 * reads all ids of t, keeps its own in LDS, sorts, reduces, applies; the rows of a tiny table are split over several
 * workgroups by position instead.  Cost grows with (lookups per table)^2 / 256: tzr_pooled_bwd_direct_supported says
 * whether a batch should take this entry -- 0 for shapes it refuses (TZR_ERR_UNSUPPORTED: more than 256 lookups or tables,
 * ragged pooled bags = grad_mode 0 with uniform_bag_len != 1) and for more than 16 384 lookups per table on average
 * (tzr_tune "bwd_direct": 1 = any size, -1 = never); callers take the planned pair then.  Replaces the same fbgemm pieces
 * as the pair (transpose_embedding_input + split_embedding_backward_codegen_*_exact, tzrec/modules/embedding.py:930,
 * tzrec/main.py:774-781).
 * grad_mode | TZR_GRAD_HOT_ROWS: the caller expects a row with more lookups than fit a workgroup's LDS (1 280) -- the shared
 * row of a zero-collision hash's unseen ids (tzrec/features/feature.py:693-736: row zch_size - 1), a default id: such a row,
 * when it is the mode of the table's first 16 ids, is summed by ALL of the table's workgroups (a slice of the positions each,
 * partial sums added in slice order) instead of by its range's workgroup alone (140 -> 45 us at 8 192 lookups, 95 % on one
 * row).  Without the flag, or when the sample misses, the row is still handled correctly, only slower.
 * `ws`: tzr_pooled_bwd_direct_workspace bytes, ZERO-FILLED by the caller before its first use (arrival counters of the
 * workgroups that share a row of a tiny table; every launch leaves them zero, so the buffer can be kept and reused --
 * one buffer per stream of launches). */
#define TZR_GRAD_HOT_ROWS 0x100
int tzr_pooled_bwd_direct_supported(int64_t n_positions, int n_feats, int n_tables, int uniform_bag_len,
                                    int grad_mode);
size_t tzr_pooled_bwd_direct_workspace(int64_t n_positions, int n_tables, int max_dim);
int tzr_pooled_bwd_direct(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats, int n_feats,
                          int64_t max_rows, int max_dim, const int64_t* d_values, const int64_t* d_offsets,
                          const float* d_weights, int64_t n_values, int64_t n_positions, int64_t B,
                          int uniform_bag_len, int grad_mode, const TzrDst* h_grads, int n_dst,
                          const TzrSparseOptim* h_optim, void* ws, size_t ws_bytes, void* stream);

/* Inspection of a finished plan (tests / debugging): byte offsets into `ws` of out8[0] = the sorted
 * {row, lookup position} pairs (uint32 x 2 per table-major position) of every table with more than
 * 512 rows, out8[1] = the bucket-partitioned pairs (final for tables of <= 512 rows), out8[2] = the
 * table-major start of every lookup by order (uint32[n_feats + 1]); out8[3..7] internal.  After a
 * plan, the pairs of one table hold every lookup of the table exactly once with equal rows adjacent
 * and, inside a row, ascending lookup positions. */
int tzr_pooled_bwd_plan_view(int64_t n_values, int64_t n_positions, int n_feats, int n_tables,
                             int max_dim, int64_t* out8);

/* K7: fused backward + sparse optimizer.  Replaces fbgemm
 * split_embedding_backward_codegen_{sgd,adagrad,rowwise_adagrad}_*_exact (optimizer fused into
 * backward by apply_optimizer_in_backward, tzrec/main.py:774-781): per distinct (table,row)
 * g = sum over its duplicate lookups (in original lookup order) of weight_i * dL/d(pooled bag),
 * summed over the feature groups the feature was copied into; then ONE update of the row:
 *   adagrad          m += g*g;            w -= lr * g / (sqrt(m) + eps)
 *   rowwise adagrad  m += mean_d(g*g);    w -= lr * g / (sqrt(m) + eps)
 *   sgd              w -= lr * g
 * grad_mode 0: h_grads mirrors the forward's h_dsts (same buffer indices / strides).
 * grad_mode 1: h_grads[0] is float [n_values, stride], ONE gradient row per id (indexed by the id's
 * position in `values`): the owner side of the sharded exchange. */
int tzr_pooled_bwd_apply(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats,
                         int n_tables, int max_dim, const int64_t* d_offsets,
                         const float* d_weights, int64_t n_values, int64_t n_positions, int64_t B,
                         int uniform_bag_len, int grad_mode, const TzrDst* h_grads, int n_dst,
                         const TzrSparseOptim* h_optim, void* ws, size_t ws_bytes, void* stream);

/* K6 / K7 in the "cells" form (csrc/pooled_bwd_cells.hip): the index plan as ONE launch for batches of exactly one id per
 * bag -- the shape of examples/dlrm_criteo.config -- and the apply that reads it.  Same reference pieces replaced as
 * tzr_pooled_bwd_plan / _apply (fbgemm transpose_embedding_input + split_embedding_backward_codegen_*_exact behind
 * tzrec/modules/embedding.py:930, tzrec/main.py:774-781), same semantics: per distinct (table, row) the gradient rows of
 * its lookups added in lookup-position order, one update (the grouping of the partial sums follows each plan's own unit
 * boundaries: the two plans agree to fp32 rounding, each is bit-reproducible).  The plan only orders every chunk of <= 1024 lookups by
 * bucket in place and notes the chunk's bucket starts (no histogram, no scan, no communication between workgroups); a unit of
 * the apply gathers the cells of its bucket range from the chunks and sorts them in LDS.  Which unit reads which cells is a
 * function of the tables and B alone: tzr_bwd_cells_geometry builds that description ON THE HOST, once; the caller keeps a
 * device copy (256-byte aligned; the kernels write its tail: partial-sum records, arrival counters, the overflow word) and
 * hands both to the two calls.
 *   tzr_bwd_cells_geometry: h_tables / h_feats = HOST copies of the arrays the device calls get.  h_out == NULL: sizes only.
 *     out_info8: [0] bytes of the image, [1] chunks, [2] units, [3] table-major positions, [4] positions per chunk,
 *     [5] partial-sum records, [6] byte offset of the overflow word (uint32) inside the image.  TZR_ERR_UNSUPPORTED: not a
 *     case for this plan (a table with more than 256 chunks of lookups, an empty batch): take tzr_pooled_bwd_plan.
 *   Unit sizes are expectations for evenly drawn ids (~1024 of a capacity of 1 280).  A unit that holds more is still updated
 *     correctly -- row by row, without the LDS sort, slowly -- and counted in the overflow word, never reset by the library: a
 *     caller that sees it move sends this id distribution to tzr_pooled_bwd_plan, whose heavy-bucket machinery is made for it.
 *   `ws`: tzr_pooled_bwd_workspace bytes (one size serves either plan).  d_weights: per-id weights or NULL. */
int tzr_bwd_cells_geometry(const TzrTable* h_tables, int n_tables, const TzrFeature* h_feats, int n_feats, int64_t B,
                           int max_dim, void* h_out, size_t out_bytes, int64_t* out_info8);
int tzr_pooled_bwd_cells_plan(const TzrTable* d_tables, int n_tables, const TzrFeature* d_feats, int n_feats, int max_dim,
                              const int64_t* d_values, int64_t n_values, int64_t B, const void* h_geo, void* d_geo, void* ws,
                              size_t ws_bytes, void* stream);
/* The forward and the cells plan of the SAME batch as one launch: tzr_pooled_fwd (one id per bag, fp32 tables, no per-sample
 * weights; d_ftables / d_ffeats / d_slots / h_dsts as there) + tzr_pooled_bwd_cells_plan (the other arguments as there).  The
 * plan needs only the ids, and its work is LDS / ALU work while the forward is bound by its 64-byte row requests: the plan's
 * workgroups follow the forward's in ONE grid and run in the slots those leave (seven workgroups of either kind per CU), which
 * hides all but ~2 us of the plan's 15; results are those of the two calls, bit for bit (each workgroup runs the same code on
 * the same data).  tzr_pooled_fwd_cells_plan_supported == 0 / TZR_ERR_UNSUPPORTED: not a case for it (more than
 * 128 slots, B < 32 768, tzr_tune "fwd_plan" 0) -- make the two calls.  In the reference both halves sit behind self.ebc(kjt)
 * and its autograd (tzrec/modules/embedding.py:930; fbgemm's transpose_embedding_input runs at backward time). */
int tzr_pooled_fwd_cells_plan_supported(int n_slots, int64_t B);
int tzr_pooled_fwd_cells_plan(const TzrTable* d_ftables, const TzrFeature* d_ffeats, int n_ffeats, const TzrSlot* d_slots,
                              int n_slots, const TzrDst* h_dsts, int n_dst, const TzrTable* d_tables, int n_tables,
                              const TzrFeature* d_feats, int n_feats, int max_dim, const int64_t* d_values, int64_t n_values,
                              int64_t B, const void* h_geo, void* d_geo, void* ws, size_t ws_bytes, void* stream);
int tzr_pooled_bwd_cells_apply(const TzrTable* d_tables, const TzrFeature* d_feats, int n_feats, int n_tables, int max_dim,
                               const float* d_weights, int64_t n_values, int64_t B, int grad_mode, const TzrDst* h_grads,
                               int n_dst, const TzrSparseOptim* h_optim, const void* h_geo, void* d_geo, void* ws,
                               size_t ws_bytes, void* stream);

/* Dense update of replicated (data_parallel) tables after their accumulated row gradients were
 * all-reduced: for every row r of every table t with a non-zero gradient
 * g = d_acc[(d_row_start[t] + r) * dim ...], apply h_optim exactly like K7 does for one row.  Rows
 * whose gradient is all zero are left untouched (a sparse update never visits them). */
int tzr_dense_rows_update(const TzrTable* d_tables, int n_tables, const int64_t* d_row_start,
                          int64_t total_rows, const float* d_acc, int dim,
                          const TzrSparseOptim* h_optim, void* stream);
/* Same, and the rows of d_acc it applied are zero afterwards: the accumulation buffer is ready for the next step's
 * TZR_OPT_ACCUMULATE pass (which writes only the rows it touches) without a memset launch in front of it. */
int tzr_dense_rows_update_clear(const TzrTable* d_tables, int n_tables, const int64_t* d_row_start,
                                int64_t total_rows, float* d_acc, int dim, const TzrSparseOptim* h_optim,
                                void* stream);

/* ---- row-wise sharded exchange (one process per GPU, RCCL all-to-all between the stages) ---- */

/* Owner side forward: out[j, 0:dim] = W_{table of key(j)}[ids[j], :] for j in [0, n_ids); ids
 * arrive grouped by key, key k owning positions d_key_start[k] .. d_key_start[k+1] (device,
 * int64[n_keys+1]); d_key_table[k] indexes d_tables (< 0: a dead key, its output rows are left
 * untouched).  Replaces the per-shard TBE lookup of
 * torchrec's row-wise sharded EBC [upstream]; one row per id instead of a pooled partial per bag. */
int tzr_rows_gather(const TzrTable* d_tables, const int32_t* d_key_table,
                    const int64_t* d_key_start, int n_keys, const int64_t* d_ids, int64_t n_ids,
                    float* d_out, int64_t out_stride, int dim, void* stream);

/* Requester side backward: d_out[pos(i), 0:dim] = weight_i (/ len for mean) * sum over the feature
 * groups of key f of grad[g][b, col_g(f) : +dim] for every id i of bag (f, b); pos(i) =
 * d_positions[i] (the unbucketize permute) or i when null. */
int tzr_lookup_grads(const TzrFeature* d_feats, int n_feats, const int64_t* d_offsets,
                     const float* d_weights, int64_t B, int uniform_bag_len,
                     const int64_t* d_positions, const TzrDst* h_grads, int n_dst, float* d_out,
                     int64_t out_stride, int dim, void* stream);

/* ---- dense glue of the training step -------------------------------------------------------- */

/* BCEWithLogitsLoss(reduction="mean") and its gradient in two launches (elementwise + partial
 * sums, fixed-order finish).  Replaces the
 * torch.nn.BCEWithLogitsLoss forward+backward built by tzrec/models/rank_model.py:190-191,233-240
 * (with sample weights: mean(loss_i * w_i), rank_model.py:260-273, w_i already normalised by the
 * caller).  *d_loss = mean_i w_i*(max(x,0) - x*y + log1p(exp(-|x|))); d_grad_logits[i] =
 * w_i*(sigmoid(x_i) - y_i)/B.  Labels: float32 (labels_are_float=1) or int32/int64. */
size_t tzr_bce_logits_workspace(int64_t B);
int tzr_bce_logits(const float* d_logits, const void* d_labels, int labels_itemsize,
                   int labels_are_float, const float* d_sample_weight, int64_t B, float* d_loss,
                   float* d_grad_logits, void* ws, size_t ws_bytes, void* stream);

/* Backward of a Linear+ReLU layer up to the GEMMs: d_grad[b,n] = d_grad_y[b,n] * (d_y[b,n] > 0)
 * and d_colsum[n] = sum_b d_grad[b,n] (the bias gradient), one pass.  Replaces
 * threshold_backward + sum(0) in the autograd of tzrec/modules/mlp.py:37-177 (Perceptron =
 * Linear + ReLU).  N multiple of 4, <= 1024; strides in floats, multiples of 4. */
size_t tzr_relu_bwd_colsum_workspace(int64_t B, int N);
int tzr_relu_bwd_colsum(const float* d_grad_y, int64_t grad_y_stride, const float* d_y,
                        int64_t y_stride, int64_t B, int N, float* d_grad, int64_t grad_stride,
                        float* d_colsum, void* ws, size_t ws_bytes, void* stream);
/* The same without its finishing launch: the column sums stay *out_G rows of N partial sums at the head of `ws` (which remains the
 * caller's until they are consumed) -- a source of kind TZR_ADAM_SRC_ROWS (G = *out_G, P = N, col = 0) for tzr_dense_adam_fused:
 * the bias gradient of a Linear + ReLU layer goes into the optimizer's launch without ever being a tensor. */
int tzr_relu_bwd_colsum_parts(const float* d_grad_y, int64_t grad_y_stride, const float* d_y, int64_t y_stride, int64_t B, int N,
                              float* d_grad, int64_t grad_stride, void* ws, size_t ws_bytes, int* out_G, void* stream);

/* Backward of the one-unit logits layer y = x w^T + b (tzrec/models/rank_model.py output layer
 * with num_class 1): d_grad_x[b,:] = gy[b] * w (nullable), d_grad_wb[0:N] = sum_b gy[b] * x[b,:],
 * d_grad_wb[N] = sum_b gy[b] (d_grad_wb holds N + 4 floats).  One pass over x. */
size_t tzr_head_bwd_workspace(int64_t B, int N);
int tzr_head_bwd(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                 const float* d_w, int64_t B, int N, float* d_grad_x, int64_t grad_x_stride,
                 float* d_grad_wb, void* ws, size_t ws_bytes, void* stream);

/* tzr_head_bwd when x is the output of a ReLU (the last hidden layer of the attention MLP in front of DIN's one-unit score
 * layer, tzrec/modules/sequence.py:101-128): d_grad[b,:] = gy[b] * w * (x[b,:] > 0) -- the gradient at the hidden layer's
 * pre-activation --, d_sums = [sum_b gy[b] x[b,:] (N) | sum_b gy[b], 0, 0, 0 | sum_b d_grad[b,:] (N)]: the score layer's weight
 * and bias gradient and the hidden layer's bias gradient from one pass over x.  Replaces tzr_head_bwd + tzr_relu_bwd_colsum. */
size_t tzr_head_bwd_relu_workspace(int64_t B, int N);
int tzr_head_bwd_relu(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                      const float* d_w, int64_t B, int N, float* d_grad, int64_t grad_stride,
                      float* d_sums, void* ws, size_t ws_bytes, void* stream);

/* The mixing step of a multi-gate mixture of experts (tzrec/modules/mmoe.py:63-76) for ALL tasks in one launch:
 *   out_t[b,:] = sum_e softmax(logits_t[b,:])_e * expert_e[b,:]
 * forward: reads logits / expert, writes out and probs (the softmax, [B, n_experts] contiguous per task, kept for the backward);
 * backward: reads grad_out / expert / probs, writes d_expert_e[b,:] = sum_t probs_t[b,e] * grad_out_t[b,:] (the sum over the
 * tasks included) and d_logits_t (softmax backward of <grad_out_t[b,:], expert_e[b,:]>).  Nothing is stacked: every expert is its
 * own [B, H] buffer.  H multiple of 4; float pointers 16-byte aligned, row strides multiples of 4 (logits / d_logits: any). */
#define TZR_MOE_MAX_EXPERTS 8
#define TZR_MOE_MAX_TASKS 4
typedef struct TzrMoeMix {
  int64_t B;
  int32_t H, n_experts, n_tasks, reserved;
  uint64_t expert[TZR_MOE_MAX_EXPERTS];          /* const float* [B, H]                       */
  int64_t expert_stride[TZR_MOE_MAX_EXPERTS];
  uint64_t d_expert[TZR_MOE_MAX_EXPERTS];        /* float* [B, H]            (backward)       */
  int64_t d_expert_stride[TZR_MOE_MAX_EXPERTS];
  uint64_t logits[TZR_MOE_MAX_TASKS];            /* const float* [B, n_experts]   (forward)   */
  int64_t logits_stride[TZR_MOE_MAX_TASKS];
  uint64_t probs[TZR_MOE_MAX_TASKS];             /* float* [B, n_experts] contiguous          */
  uint64_t out[TZR_MOE_MAX_TASKS];               /* float* [B, H]                 (forward)   */
  int64_t out_stride[TZR_MOE_MAX_TASKS];
  uint64_t grad_out[TZR_MOE_MAX_TASKS];          /* const float* [B, H]           (backward)  */
  int64_t grad_out_stride[TZR_MOE_MAX_TASKS];
  uint64_t d_logits[TZR_MOE_MAX_TASKS];          /* float* [B, n_experts]         (backward)  */
  int64_t d_logits_stride[TZR_MOE_MAX_TASKS];
} TzrMoeMix;
int tzr_moe_mix_fwd(const TzrMoeMix* h_mix, void* stream);
int tzr_moe_mix_bwd(const TzrMoeMix* h_mix, void* stream);

/* Linear layers with a handful of output units, n_out <= 8: the logits layer of the rank models
 * (tzrec/models/rank_model.py:190-191) and the gates of MMoE (tzrec/modules/mmoe.py: Linear(in, num_expert)).
 * forward: d_y[b, j] = sum_k d_x[b, k] * d_w[j, k] + d_bias[j] (d_bias nullable); one pass over x.
 * backward: d_grad_x[b, :] = sum_j gy[b, j] * w[j, :] (nullable), d_grad_wb = [gy^T x (n_out x K row-major) | sum_b gy[b, :]
 * (n_out, padded to a multiple of 4)]; one pass over x + a fixed-order sum of the workgroups' rows.  K multiple of 4, <= 1024;
 * x / w / grad_x strides multiples of 4 floats.  Replace torch's addmm + three backward products (one output tile wide). */
int tzr_skinny_linear_fwd(const float* d_x, int64_t x_stride, const float* d_w, int64_t w_stride, const float* d_bias,
                          int64_t B, int K, int n_out, float* d_y, int64_t y_stride, void* stream);
size_t tzr_skinny_linear_bwd_workspace(int64_t B, int K, int n_out);
int tzr_skinny_linear_bwd(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                          const float* d_w, int64_t w_stride, int64_t B, int K, int n_out, float* d_grad_x,
                          int64_t grad_x_stride, float* d_grad_wb, void* ws, size_t ws_bytes, void* stream);
/* ... without its finishing launch: [weight gradient n_out x K | bias gradient n_out, padded to 4] stay *out_G rows of *out_P
 * partial sums at the head of `ws`: two sources of kind TZR_ADAM_SRC_ROWS (col 0 and col n_out K) for tzr_dense_adam_fused. */
int tzr_skinny_linear_bwd_parts(const float* d_grad_y, int64_t grad_y_stride, const float* d_x, int64_t x_stride,
                                const float* d_w, int64_t w_stride, int64_t B, int K, int n_out, float* d_grad_x,
                                int64_t grad_x_stride, void* ws, size_t ws_bytes, int* out_G, int* out_P, void* stream);

/* Input gradient of a Linear layer chained with the ReLU mask and bias gradient of the layer below it (autograd of
 * Linear -> ReLU -> Linear, tzrec/modules/mlp.py:58-83), one launch on the matrix cores (exact fp32):
 *   d_grad_out[n,h] = (sum_k d_grad_in[n,k] * d_w[k,h]) * (d_y[n,h] > 0),  d_colsum[h] = sum_n d_grad_out[n,h].
 * d_w = the upper layer's weight [K, H] row-major (nn.Linear's own layout: out_features x in_features), d_y = the lower
 * layer's ReLU output.  d(loss)/d(y) is never written.  Shapes: tzr_linear_bwd_relu_supported(K, H) (K in {16, 32, 64},
 * H in {64, 128, 256}); strides in floats, multiples of 4; 16-byte aligned pointers.  Deterministic. */
int tzr_linear_bwd_relu_supported(int K, int H);
size_t tzr_linear_bwd_relu_workspace(int64_t N, int H);
int tzr_linear_bwd_relu(const float* d_grad_in, int64_t grad_in_stride, const float* d_w, int64_t w_stride,
                        const float* d_y, int64_t y_stride, int64_t N, int K, int H, float* d_grad_out,
                        int64_t grad_out_stride, float* d_colsum, void* ws, size_t ws_bytes, void* stream);

/* Linear layers over a TALL input (every position of a sequence batch: the attention MLP of DIN on the jagged positions,
 * tzrec/modules/sequence.py:101-128 over tzrec/modules/mlp.py:58-83) with the small weight matrix resident in registers, exact
 * fp32 on the matrix cores (csrc/gemm_rows.hip).
 *   tzr_linear_rows:  d_out[n, h] = act(sum_k d_in[n, k] * W(k, h) + d_bias[h] + d_rowvec[d_row_index[n], h])
 *     w_out_major != 0: W(k, h) = d_w[h * w_stride + k]  (nn.Linear's weight [H, K]: the layer's forward, torch's addmm + ReLU);
 *     w_out_major == 0: W(k, h) = d_w[k * w_stride + h]  (the input gradient g @ weight of a layer whose weight is [K, H]).
 *     d_bias / d_rowvec may be NULL; d_rowvec [R, H] is a per-row addend gathered through d_row_index (int32 [N]) -- the part of a
 *     layer over positions that depends on the SAMPLE only (DIN's query terms).  relu != 0: max(., 0).
 *   tzr_linear_rows_wgrad:  d_dw[h, k] (+)= sum_n d_g[n, h] * d_x[n, k]  (autograd's weight gradient g^T x), partial sums of
 *     <= 512 workgroups added in workgroup order by a second launch (deterministic; workspace from _wgrad_workspace).
 * Shapes: *_supported (K, H multiples of 16 out of a fixed list up to 256; tzr_linear_rows_supported returns 3 where d_rowvec is
 * taken too, 1 where only the plain form is built); strides in floats, multiples of 4; 16-byte aligned
 * pointers.  TZR_ERR_UNSUPPORTED otherwise: the caller keeps its GEMM library call. */
int tzr_linear_rows_supported(int K, int H);
int tzr_linear_rows(const float* d_in, int64_t in_stride, const float* d_w, int64_t w_stride, int w_out_major,
                    const float* d_bias, const float* d_rowvec, int64_t rowvec_stride, const int32_t* d_row_index, int relu,
                    int64_t N, int K, int H, float* d_out, int64_t out_stride, void* stream);
int tzr_linear_rows_wgrad_supported(int H, int K);
size_t tzr_linear_rows_wgrad_workspace(int64_t N, int H, int K);
int tzr_linear_rows_wgrad(const float* d_g, int64_t g_stride, const float* d_x, int64_t x_stride, int64_t N, int H, int K,
                          float* d_dw, int64_t dw_stride, int accumulate, void* ws, size_t ws_bytes, void* stream);

#define TZR_ADAM_MAX_TENSORS 32
typedef struct TzrAdamTensor { /* one dense parameter tensor, device addresses, float32 */
  uint64_t param, grad, exp_avg, exp_avg_sq;
  uint64_t state; /* float[3], zero-initialised by the caller: [0] step count of THIS tensor
                     (torch counts steps per parameter), [1], [2] bias corrections */
  int64_t numel;
} TzrAdamTensor; /* 48 bytes */

/* Adam step over every dense parameter tensor (torch.optim.Adam semantics, amsgrad off, L2
 * weight decay): the dense optimizer TZRecOptimizer.step() runs after the fused sparse backward
 * (tzrec/optim/optimizer.py:56-68; adam_optimizer of tzrec/optim/optimizer_builder.py).
 * Only tensors that have a gradient this step are passed; their step counts are incremented
 * here.  The learning rate is read from d_lr when non-null (so a captured hipGraph follows a
 * scheduler), else `lr`.  h_tensors is a HOST array; 32 tensors per launch pair. */
int tzr_dense_adam(const TzrAdamTensor* h_tensors, int n_tensors, const float* d_lr, float lr,
                   float beta1, float beta2, float eps, float weight_decay, void* stream);

/* The same Adam step in ONE launch, with gradients taken as they lie (csrc/adam_fused.hip): a finished tensor
 * (TZR_ADAM_SRC_TENSOR: TzrAdamTensor.grad), rows of per-workgroup partial sums left by tzr_mlp2_bwd_parts (TZR_ADAM_SRC_ROWS:
 * element i of the tensor = sum over the G rows of parts[g * P + col + i], added in tzr_mlp2_bwd's own finish order), or the
 * batch slices of tzr_dot_interaction_top_wgrad_parts (TZR_ADAM_SRC_WGRAD, at most one per call: *h_wgrad).  The column-sum
 * finish launches and the slice reduction of the DLRM step (tzrec/models/dlrm.py:101-135 behind tzrec/optim/optimizer.py:56-68)
 * disappear into the optimizer's launch; sums and updates are bit-identical to the separate launches.
 * TzrAdamTensor.state here: float[TZR_ADAM_FUSED_STATE]: [0] step count, [1 ..] arrival counters (ZERO between launches: the
 * tensor's workgroups arrive in groups of 32, the last one moves the step on and clears them) -- not interchangeable with
 * tzr_dense_adam's float[3] on the same tensor.
 * A tensor with param == 0 only gets its finished gradient stored into `grad` (for a caller that needs the tensor after all).
 * TZR_ADAM_SRC_TENSOR with `parts` != 0: the finished gradient is read from `parts` instead of `grad` -- with param == 0 a copy,
 * which makes ONE store-only call the packing of a model's gradients (finished or not) into a flat buffer for a collective.
 * h_sources NULL: every gradient is a finished tensor. */
#define TZR_ADAM_FUSED_STATE 40 /* 1 step + 1 + 32 counters (<= 1024 workgroups per tensor), padded */
#define TZR_ADAM_SRC_TENSOR 0
#define TZR_ADAM_SRC_ROWS 1
#define TZR_ADAM_SRC_WGRAD 2
typedef struct TzrAdamSource {
  int32_t kind, G, P, col;
  uint64_t parts; /* const float* device address of the partial-sum rows (TZR_ADAM_SRC_ROWS) */
  uint64_t reserved;
} TzrAdamSource; /* 32 bytes */
typedef struct TzrWgradParts { uint64_t opaque[24]; } TzrWgradParts; /* filled by tzr_dot_interaction_top_wgrad_parts */
int tzr_dense_adam_fused(const TzrAdamTensor* h_tensors, const TzrAdamSource* h_sources, int n_tensors,
                         const TzrWgradParts* h_wgrad, const float* d_lr, float lr, float beta1, float beta2, float eps,
                         float weight_decay, void* stream);
/* tzr_mlp2_bwd / tzr_dot_interaction_top_wgrad without their finishing launch (same arguments, same workspace, which stays the
 * caller's until tzr_dense_adam_fused has consumed it): *out_G rows of *out_P floats, columns [dWb: H2 x H1 | dbb: H2 |
 * dWa: H1 x K0 | dba: H1]; *h_out for TZR_ADAM_SRC_WGRAD (B > 0). */
int tzr_mlp2_bwd_parts(const float* d_dhb, int64_t dhb_stride, const float* d_hb, int64_t hb_stride, const float* d_ha,
                       int64_t ha_stride, const float* d_x, int64_t x_stride, int64_t B, int K0, int H1, int H2,
                       const float* d_Wb, void* ws, size_t ws_bytes, int* out_G, int* out_P, void* stream);
int tzr_dot_interaction_top_wgrad_parts(const float* d_dense, int64_t dense_stride, const float* d_sparse, int64_t sparse_stride,
                                        int F, int D, int64_t B, const float* d_g1, int64_t g1_stride, int H, const float* d_scale,
                                        void* d_ws, int64_t ws_bytes, TzrWgradParts* h_out, void* stream);

/* ---- zero-collision hash ---------------------------------------- */

#define TZR_ZCH_EMPTY INT64_MAX /* unoccupied cell / row (tzrec/utils/zch_util.py:29 ZCH_EMPTY_SLOT) */
#define TZR_ZCH_TOMB (INT64_MAX - 1) /* cell whose id was evicted (tzr_zch_update); as a raw id: served from the shared row, never admitted */

/* One managed-collision table: an open-addressing map raw id -> row plus per-row eviction metadata. */
typedef struct TzrZchModule {
  int64_t* keys;      /* [capacity] raw id per cell, TZR_ZCH_EMPTY when free                    */
  int32_t* rows;      /* [capacity] remapped row of the cell's id                               */
  int64_t* counts;    /* [zch_size] accesses per row           (written when profile != 0)      */
  int64_t* last_iter; /* [zch_size] iteration of the last access                                */
  int64_t capacity;   /* power of two, >= 2 * occupied rows                                     */
  int64_t zch_size;   /* rows of the embedding table; ids not in the map go to row zch_size-1   */
  int64_t reserved[2];
} TzrZchModule; /* 64 bytes */

/* K13: remap the ids of every KJT key through its module (d_key_module[key] = module index, -1 =
 * pass through).  Replaces torchrec MCHManagedCollisionModule.remap + .profile [upstream], built by
 * BaseFeature.mc_module (tzrec/features/feature.py:693-736) and applied in front of the lookup by
 * ManagedCollisionEmbeddingBagCollection (tzrec/modules/embedding.py:856-864).  profile != 0 also
 * bumps counts/last_iter of the rows hit and fills d_candidates (int64[n_values], positional):
 * the raw id where it has no row yet (an admission candidate), TZR_ZCH_EMPTY elsewhere.
 * d_out_values may alias d_values. */
int tzr_zch_remap(const TzrZchModule* d_modules, const int32_t* d_key_module, int n_keys,
                  const int64_t* d_values, const int64_t* d_offsets, int64_t B, int uniform_bag_len,
                  int64_t n_values, int64_t iter, int profile, int64_t* d_out_values,
                  int64_t* d_candidates, void* stream);

/* The same for a step that is replayed from a hipGraph: uniform bags only; the iteration number is read from DEVICE memory
 * (*d_iter: the caller bumps it inside the graph), and with profile != 0 the candidates of the step go to slot
 * `*d_iter % ring_slots` of d_cand_ring, int64 [ring_slots][n_cand_keys][B * uniform_bag_len]: one block per key that has
 * a module (d_key_cand[key] = its index among those keys, -1 for the others), the raw id where it has no row yet,
 * TZR_ZCH_EMPTY elsewhere.  Which step filled which slot is a function of the counter alone. */
int tzr_zch_remap_ring(const TzrZchModule* d_modules, const int32_t* d_key_module, int n_keys,
                       const int64_t* d_values, int64_t B, int uniform_bag_len, int64_t n_values,
                       const int64_t* d_iter, int profile, int64_t* d_out_values, const int32_t* d_key_cand,
                       int n_cand_keys, int64_t* d_cand_ring, int64_t ring_slots, void* stream);

/* Rebuild a module's map from n distinct (raw id, row) pairs (after admission / eviction, or when
 * restoring a checkpoint).  h_module is a HOST struct holding device pointers. */
int tzr_zch_build(const TzrZchModule* h_module, const int64_t* d_ids, const int32_t* d_rows,
                  int64_t n, void* stream);

/* The same after an admission / eviction round, touching only what changed: the n ids d_old_ids[i] (TZR_ZCH_EMPTY: the
 * row was free) leave the map, the n distinct ids d_new_ids[i] enter it with rows d_rows[i].  Evicted cells become
 * tombstones that later insertions reuse; the caller rebuilds (tzr_zch_build) when too many have piled up. */
int tzr_zch_update(const TzrZchModule* h_module, const int64_t* d_old_ids, const int64_t* d_new_ids,
                   const int32_t* d_rows, int64_t n, void* stream);

/* Admission / eviction round (torchrec MCHManagedCollisionModule eviction policies LFU / LRU / DistanceLFU
 * [upstream 1.7.0] as tzrec configures them, tzrec/features/feature.py:693-736, docs/source/feature/zch.md):
 * residents and candidates are ranked together (score descending, residents first, raw id ascending) and the first
 * zch_size - 1 stay.  The entries to drop are found by radix SELECTION on the reversed order instead of sorting:
 * tzr_zch_select_hist counts, over the residents (d_row_ids[r] != TZR_ZCH_EMPTY, r < zch_size - 1) and the n_new
 * candidates that match what is known of the threshold, one digit of the drop key
 *   field 0: bits [shift, shift + bits) of the score image, given its higher bits t1
 *   field 1: the kind (bin 0 candidates, bin 1 residents) of the entries whose score image is t1
 *   field 2: bits [shift, shift + bits) of the id image of the entries with score image t1 and kind t2, given its
 *            higher bits t3
 * into d_bins[2048] (uint64, zeroed by the call); the caller reads them, extends the threshold and calls again
 * (torcheasyrec_amd/zch.py: at most 13 passes, each ONE streaming read of the three per-row arrays).
 * policy: 0 LFU (count), 1 LRU (1 / age), 2 DistanceLFU (count / age), age = max(cur_iter - last, 1) ^ decay_exponent.
 * tzr_zch_select_mark then writes d_row_kept[zch_size - 1] / d_new_kept[n_new] = 1 for every entry ABOVE the threshold
 * (t1, t2, t3) (drop_none != 0: everything live stays).  h_module is a HOST struct holding device pointers. */
int tzr_zch_select_hist(const TzrZchModule* h_module, const int64_t* d_row_ids, const int64_t* d_new_ids,
                        const int64_t* d_new_cnt, int64_t n_new, int64_t cur_iter, int policy,
                        double decay_exponent, int field, int shift, int bits, uint64_t t1, int t2,
                        uint64_t t3, uint64_t* d_bins, void* stream);
int tzr_zch_select_mark(const TzrZchModule* h_module, const int64_t* d_row_ids, const int64_t* d_new_ids,
                        const int64_t* d_new_cnt, int64_t n_new, int64_t cur_iter, int policy,
                        double decay_exponent, int drop_none, uint64_t t1, int t2, uint64_t t3,
                        uint8_t* d_row_kept, uint8_t* d_new_kept, void* stream);

/* ---- small dense layer stacks (tzrec.modules.mlp.MLP, models/dlrm.py:101-135) ----------------
 * The MLPs stay on PyTorch / hipBLASLt in general; the three calls below take over the layer stacks whose GEMMs are
 * too skinny to be worth a launch each (N = 64 / 32 / 16 / 1): all fp32, row-major, `*_stride` in floats.
 * tzr_mlp2_fwd: ha = relu(x Wa^T + ba) [B, H1], hb = relu(ha Wb^T + bb) [B, H2]; Wa [H1, K0], Wb [H2, H1] as
 * nn.Linear stores them; K0 <= 32, H1 <= 64, H2 <= 32 (TZR_ERR_UNSUPPORTED beyond).
 * tzr_mlp2_bwd: from dhb = d(loss)/d(hb): the four parameter gradients (no input gradient: x is data).
 * tzr_mlp_tail: y2 = relu(y1 W2^T + b2), logit = y2 . w3 + b3, loss = mean BCE-with-logits(logit, label) -- and its
 * whole backward: g1 = d(loss)/d(pre-activation of y1) (y1 is a ReLU output: masked where y1 == 0), db1 = column sums
 * of g1 (the bias gradient of the layer that produced y1), dW2, db2, dw3, d_scalars = {d(loss)/d(b3), loss}.
 * ws: tzr_mlp_workspace() bytes.  Deterministic (fixed-order reductions). */
size_t tzr_mlp_workspace(void);
int tzr_mlp2_fwd(const float* d_x, int64_t x_stride, int64_t B, int K0, const float* d_Wa, const float* d_ba, int H1,
                 const float* d_Wb, const float* d_bb, int H2, float* d_ha, int64_t ha_stride, float* d_hb,
                 int64_t hb_stride, void* stream);
int tzr_mlp2_bwd(const float* d_dhb, int64_t dhb_stride, const float* d_hb, int64_t hb_stride, const float* d_ha,
                 int64_t ha_stride, const float* d_x, int64_t x_stride, int64_t B, int K0, int H1, int H2,
                 const float* d_Wb, float* d_dWa, float* d_dba, float* d_dWb, float* d_dbb, void* ws, size_t ws_bytes,
                 void* stream);
int tzr_mlp_tail(const float* d_y1, int64_t y1_stride, const void* d_labels, int labels_itemsize, int labels_are_float,
                 int64_t B, int H1, const float* d_W2, const float* d_b2, int H2, const float* d_w3, const float* d_b3,
                 float* d_logits, float* d_g1, int64_t g1_stride, float* d_dW2, float* d_db2, float* d_dw3,
                 float* d_scalars, float* d_db1, void* ws, size_t ws_bytes, void* stream);

/* ---- sequence path --------------------------------------------- */

/* K12: jagged [N, dim] (+ offsets int64[B+1]) -> dense [B, max_len, dim]; positions past a
 * sample's length get padding_value, sequences longer than max_len are truncated.  Replaces fbgemm
 * jagged_to_padded_dense reached from JaggedTensor.to_padded_dense
 * (tzrec/modules/embedding.py:1429,1480).  The unpooled lookup itself is tzr_rows_gather with one
 * key segment per KJT key, its backward tzr_pooled_bwd_plan/apply with grad_mode 1. */
int tzr_jagged_to_padded_dense(const float* d_values, int64_t values_stride,
                               const int64_t* d_offsets, int64_t B, int64_t max_len, int dim,
                               float padding_value, float* d_out, void* stream);
/* backward of the above: d_values[offsets[b] + l, :] = d_dense[b, l, :] for l < min(len, max_len),
 * zero for truncated positions. */
int tzr_padded_dense_to_jagged(const float* d_dense, const int64_t* d_offsets, int64_t B,
                               int64_t max_len, int dim, float* d_values, int64_t values_stride,
                               void* stream);

/* Multi-valued sequence steps: replaces torch.segment_reduce(jt.values(), pooling,
 * lengths=key_lengths) + nan_to_num in SequenceEmbeddingGroupImpl
 * (tzrec/modules/embedding.py:1353-1366).  Segment s = rows [offsets[s], offsets[s+1]) of
 * d_values [N, dim]; mode 0 = sum, 1 = mean (an empty segment gives a zero row).
 * out [S, dim].  The backward writes every row of d_grad_values covered by a segment. */
int tzr_segment_reduce_fwd(const float* d_values, int64_t values_stride, const int64_t* d_offsets,
                           int64_t S, int dim, int mode, float* d_out, int64_t out_stride,
                           void* stream);
int tzr_segment_reduce_bwd(const float* d_grad_out, int64_t grad_out_stride,
                           const int64_t* d_offsets, int64_t S, int dim, int mode,
                           float* d_grad_values, int64_t grad_values_stride, void* stream);

/* DIN target attention over JAGGED positions: replaces DINEncoder.forward on padded [B, L, .] tensors
 * (tzrec/modules/sequence.py:101-128; the padding is SequenceEmbeddingGroupImpl's to_padded_dense,
 * tzrec/modules/embedding.py:1429-1480).  A position is a row of the unpooled lookup's output [N, D]; sample b owns rows
 * [offsets[b], offsets[b+1]); rows at index >= max_len inside a sample do not exist for the encoder (the padded length /
 * `max_seq_length`).  csrc/din_attention.hip states the masking semantics.  D, H multiples of 4, strides in floats
 * (multiples of 4), max_len <= 2048.
 *   tzr_jagged_segment_ids  d_seg[n] = sample of row n (B for rows at or behind offsets[B])
 *   tzr_din_assemble_fwd    X[n] = [ k_n | q_b * k_n | q_b ]  ([N, 3 D]): the input of the attention MLP whose first layer
 *                           W [q, k, q - k, q * k] the caller folds into [Wb - Wc | Wd | Wa + Wc]
 *   tzr_din_assemble_bwd    d_dkv[n] (+)= dX_n[0:D] + q_b * dX_n[D:2D];  d_dq[b] = sum_n (k_n * dX_n[D:2D] + dX_n[2D:3D])
 *   tzr_din_attn_fwd        s_n = h_n . w + bias[0]; p = softmax of s over a sample's rows; out[b] = sum_n p_n k_n
 *   tzr_din_attn_bwd        d_ds[n] = p_n (g_b . k_n - sum_m p_m g_b . k_m);  d_dkv[n] = p_n g_b
 * (the backward of s = h . w + bias is tzr_head_bwd on d_ds) */
int tzr_jagged_segment_ids(const int64_t* d_offsets, int64_t B, int64_t N, int32_t* d_seg, void* stream);
int tzr_din_assemble_fwd(const float* d_kv, int64_t kv_stride, const float* d_q, int64_t q_stride,
                         const int32_t* d_seg, int64_t B, int64_t N, int D, float* d_X, int64_t x_stride,
                         void* stream);
int tzr_din_assemble_bwd(const float* d_dX, int64_t x_stride, const float* d_kv, int64_t kv_stride,
                         const float* d_q, int64_t q_stride, const int32_t* d_seg, const int64_t* d_offsets,
                         int64_t B, int64_t N, int D, float* d_dkv, int64_t dkv_stride, int accumulate_dkv,
                         float* d_dq, int64_t dq_stride, void* stream);
/* The same pair with the query's own block left out of the rows: X[n] = [k_n | q_b * k_n] (2 D wide).  The first attention
 * layer W [q, k, q - k, q * k] = (Wb - Wc) k + Wd (q * k) + (Wa + Wc) q: its last term is one product per SAMPLE, added to the
 * per-position product as tzr_linear_rows' gathered row vector -- a third less contraction length than the 3 D form above.
 * Backward: dk_n (+)= dX_n[0:D] + q_b * dX_n[D:2D];  dq_b = sum_n k_n * dX_n[D:2D] + d_dq_add[b] (d_dq_add [B, D] or NULL: the
 * query's gradient through that per-sample term). */
int tzr_din_assemble2_fwd(const float* d_kv, int64_t kv_stride, const float* d_q, int64_t q_stride, const int32_t* d_seg, int64_t B,
                          int64_t N, int D, float* d_X, int64_t x_stride, void* stream);
int tzr_din_assemble2_bwd(const float* d_dX, int64_t x_stride, const float* d_kv, int64_t kv_stride, const float* d_q,
                          int64_t q_stride, const int32_t* d_seg, const int64_t* d_offsets, int64_t B, int64_t N, int D,
                          float* d_dkv, int64_t dkv_stride, int accumulate_dkv, const float* d_dq_add, int64_t dq_add_stride,
                          float* d_dq, int64_t dq_stride, void* stream);
int tzr_din_attn_fwd(const float* d_h, int64_t h_stride, int H, const float* d_w, const float* d_bias,
                     const float* d_kv, int64_t kv_stride, int D, const int64_t* d_offsets, int64_t B,
                     int64_t max_len, float* d_out, int64_t out_stride, float* d_p, void* stream);
int tzr_din_attn_bwd(const float* d_grad_out, int64_t grad_out_stride, const float* d_p, const float* d_kv,
                     int64_t kv_stride, int D, const int64_t* d_offsets, int64_t B, int64_t max_len, float* d_ds,
                     float* d_dkv, int64_t dkv_stride, void* stream);

/* ---- native step driver (csrc/step_driver.hip) ------------------------------------------------------------------
 * Replaces the host side of a steady-state train step of tzrec's pipeline (tzrec/utils/dist_util.py:221-303: Python +
 * torch.distributed calls per collective) for a sharded step that was cut into captured hipGraphs: ONE call queues the
 * graphs and the RCCL collectives between them.
 *
 * Communicator: RCCL reached through the librccl the process already holds (`librccl_path`: e.g. <torch>/lib/librccl.so;
 * NULL = default search); rank 0 makes the 128-byte unique id, the launcher carries it to every rank (torch.distributed's
 * store or one broadcast), every rank calls tzr_comm_create.  tzr_comm_all_to_all / _all_reduce are in stream order on
 * `stream` (bytes_per_peer: what each rank sends to and receives from every rank; all-reduce: fp32 sum or average).
 * Program: ops recorded once (tzr_step_add_*: each returns the op's index >= 0, or a negative TZR_ERR_*), replayed by
 * tzr_step_run: a graph op is hipGraphLaunch on `stream`; a collective runs on the program's own communication stream
 * behind everything queued on `stream` so far (sync != 0: `stream` waits for it on the spot; else a later wait op does);
 * nothing allocates, nothing synchronises the host.  Buffers and graph handles must outlive the program. */
int tzr_comm_available(const char* librccl_path); /* 1 / 0 */
int tzr_comm_version(const char* librccl_path);   /* ncclGetVersion, -1 when RCCL is not reachable */
int tzr_comm_unique_id(const char* librccl_path, void* out, size_t out_bytes); /* out_bytes >= 128 */
int tzr_comm_create(const char* librccl_path, const void* unique_id, size_t id_bytes, int world, int rank,
                    void** out_comm);
int tzr_comm_destroy(void* comm);
int tzr_comm_all_to_all(void* comm, const void* d_send, void* d_recv, int64_t bytes_per_peer, void* stream);
int tzr_comm_all_reduce(void* comm, float* d_buf, int64_t count, int average, void* stream);
int tzr_step_create(void** out_program);
int tzr_step_destroy(void* program);
int tzr_step_add_graph(void* program, void* graph_exec /* hipGraphExec_t */);
int tzr_step_add_all_to_all(void* program, void* comm, const void* d_send, void* d_recv, int64_t bytes_per_peer, int sync);
int tzr_step_add_all_reduce(void* program, void* comm, float* d_buf, int64_t count, int average, int sync);
int tzr_step_add_wait(void* program, int collective_op);
int tzr_step_num_ops(void* program);
int tzr_step_run(void* program, void* stream);

/* ---- export ------------------------------------------------------------------------------ */

/* Row-wise INT8 export of a table: replaces _quantize_quint8_rowwise_f16
 * (tzrec/utils/quant_util.py:25-131; reached from tzrec/utils/export_util.py:2353 and the
 * delta-embedding dump).  d_out[rows, dim + 4] bytes per row:
 * [dim uint8 values][float16 scale][float16 offset] (QUint8RowwiseF16), byte-exact with the
 * reference encoder.  d_w: float or (w_dtype = TZR_DT_F16) half rows, w_stride ELEMENTS apart.
 * d_first_bad: int64[3] device scratch; after the stream has drained it holds the smallest row
 * index with (0) a non-finite value, (1) an offset, (2) a scale outside the finite float16 range,
 * INT64_MAX where there is none -- the conditions the reference raises ValueError for; such rows'
 * output bytes are unspecified. */
int tzr_quantize_rows_q8f16(const void* d_w, int w_dtype, int64_t w_stride, int64_t rows, int dim,
                            uint8_t* d_out, int64_t* d_first_bad, void* stream);
/* dequantize_quint8_rowwise_f16 (tzrec/utils/quant_util.py:158-196): out[r, c] = q * scale + offset
 * (product rounded first), float32 [rows, dim]. */
int tzr_dequantize_rows_q8f16(const uint8_t* d_rows, int64_t rows, int dim, float* d_out,
                              int64_t out_stride, void* stream);

/* ---- delta-embedding tracker ---------------------------------------- */

/* Which rows were looked up since the last dump.  Replaces the id store of tzrec's ModelDeltaTracker
 * (tzrec/utils/delta_embedding_dump.py:352-641): torchrec DeltaStoreTrec.append per lookup (:478-513)
 * and compact / torch.cat(...).unique() per dump (:565-609).  Here the touched set of a table is a
 * bitmap in HBM, one bit per LOCAL row, uint32 words, (rows + 31) / 32 of them, bit r & 31 of word
 * r >> 5; the caller owns and zero-initialises it. */
typedef struct TzrDeltaSeg {
  uint32_t* bitmap; /* bitmap of the table this lookup segment reads; NULL = segment not tracked */
  int64_t rows;     /* local rows of that table: ids outside [0, rows) are counted, not marked   */
  int32_t key;      /* key segment of the id array the lookup reads (see tzr_delta_mark)         */
  int32_t reserved;
} TzrDeltaSeg; /* 24 bytes */

/* record_lookup (delta_embedding_dump.py:478-513): mark the ids of n_segs lookup segments.  Segment
 * s covers ids [o(k), o(k+1)) with k = d_segs[s].key and o(k) = d_key_offsets[k * key_stride], or
 * k * key_stride * uniform_len when d_key_offsets is NULL (a KJT: key_stride = B and the offsets
 * array, or uniform bags; the owner side of the sharded exchange: key_stride = 1 and the received
 * key starts).  *d_oob (nullable, caller-zeroed accumulator) += ids outside their table -- the
 * condition the reference raises ValueError for at dump time (:1026-1035). */
int tzr_delta_mark(const TzrDeltaSeg* d_segs, int n_segs, const int64_t* d_ids,
                   const int64_t* d_key_offsets, int64_t key_stride, int64_t uniform_len,
                   int64_t n_ids, int64_t* d_oob, void* stream);
/* get_unique (:565-609): *d_total += number of marked rows (caller-zeroed accumulator). */
size_t tzr_delta_collect_workspace(int64_t rows);
int tzr_delta_count(const uint32_t* d_bitmap, int64_t rows, int64_t* d_total, void* ws,
                    size_t ws_bytes, void* stream);
/* ... and the marked rows themselves, ascending, as id_base + local row (the reference's
 * ids.unique(sorted=True) + shard row_offset, :976,1036) into d_out_ids[0 : min(count, capacity)];
 * clear != 0 zeroes the bitmap in the same pass (delete_on_read). */
int tzr_delta_collect(uint32_t* d_bitmap, int64_t rows, int64_t id_base, int clear,
                      int64_t* d_out_ids, int64_t capacity, void* ws, size_t ws_bytes, void* stream);

/* Process-wide knobs; returns TZR_ERR_INVALID for unknown names.
 *   fwd_tile_b          samples per workgroup of the pooled forward (0 = by problem size)
 *   bwd_ch              lookups per chunk of the backward plan: 256 / 512 / 768 / 1024 (0 = by problem size)
 *   bwd_force_prep      1: geometry prologue as its own launch (the > 1024 features path)
 *   ia_bwd_plain        1: the D = 16 dot-interaction backward without its software pipeline (A/B switch)
 *   ia_bwd_wgs          workgroups of that backward (0 = by batch size)
 *   ia_gen_wgs          workgroups of the generalised MFMA backward (D != 16 or 33-64 rows; 0 = the resident set)
 *   ia_fwd_wgs          workgroups of the D = 16 dot-interaction forward (0 = by batch size)
 *   mlp_mfma            -1: tzr_mlp2_* / tzr_mlp_tail use their general LDS-tiled kernels for every shape (0: the MFMA
 *                       kernels of mlp_mfma.hip where the shape is DLRM-Criteo's)
 *   it_fwd_stagger      fused interaction forward: 1 = half of the waves build the next tile's row before the product, half behind it;
 *                       2 / 3 = all before / all behind; 0 = half and half when z is written, all behind when not
 *   wg_debug            phase-skipping bits of tzr_dot_interaction_top_wgrad for timing experiments (wrong results)
 *   it_wgs              persistent workgroups of the fused interaction + first-layer kernels (0 = 256, one per CU)
 *   bwd_one_wg_heavy    1: a heavy bucket of the backward plan is sorted by ONE workgroup instead of one per
 *                       1024-lookup tile.  Same plan, slower under heavy skew.  Set it when plans are built on a
 *                       stream other than the one the rest of the step runs on (NOTES.md, "Side-stream plan"). */
int tzr_tune(const char* name, int value);

/* ---- feature interaction ----------------------------------------------------------------- */

/* K9: DLRM dot interaction.  Replaces tzrec InteractionArch.forward
 * (tzrec/modules/interaction.py:80-91: bmm(X, X^T) then strict-upper-triangle gather, row-major
 * (i<j) order) fused with the concatenations of DLRM.predict (tzrec/models/dlrm.py:123-130).
 * X[b] = [dense[b] (optional, row 0); sparse[b, 0:F*D] as F rows of D].  n = F + (dense!=0).
 * out[b, 0 : n(n-1)/2] = X X^T upper triangle (exact fp32);
 * if cat_dense: out[b, P : P+D] = dense[b]; if cat_sparse: the F*D sparse floats follow.
 * D = 16 with n <= 32 (DLRM-Criteo) runs kernels specialised for that shape; any other D % 4 == 0
 * with n <= 64 the same MFMA 16x16x4 f32 scheme generalised (2 or 4 row blocks, a loop over
 * 16-column blocks of D); n > 64 a general LDS-staged VALU kernel, as long as one sample fits its
 * 60 KB of LDS: n (D + 1) floats forward, n (D + 1) + n (n + 1) backward; beyond that
 * TZR_ERR_UNSUPPORTED. */
int tzr_dot_interaction_fwd(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                            int64_t sparse_stride, int F, int D, int64_t B, float* d_out,
                            int64_t out_stride, int cat_dense, int cat_sparse, void* stream);
/* backward: dX = (G + G^T) X with G the upper-triangular gradient; concatenated pass-through
 * gradients are added.  d_grad_dense may be null when d_dense is null. */
int tzr_dot_interaction_bwd(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                            int64_t sparse_stride, int F, int D, int64_t B,
                            const float* d_grad_out, int64_t grad_out_stride, int cat_dense,
                            int cat_sparse, float* d_grad_dense, int64_t grad_dense_stride,
                            float* d_grad_sparse, int64_t grad_sparse_stride, void* stream);

/* K9b: the dot interaction FUSED with the first Linear of the MLP behind it (DLRM.predict,
 * tzrec/models/dlrm.py:123-135: `final_mlp(cat(interaction, dense, sparse))`, first layer [P + D n -> H]).
 * Layout of the interaction row as tzr_dot_interaction_fwd writes it with cat_dense = (dense != 0), cat_sparse = 1:
 * [P pairs | dense row | sparse rows], width P + D n.  W1 is the nn.Linear weight [H, ldw >= width], row-major.
 * Supported: D = 16, H = 64, ceil(P / 16) + n <= 64 column blocks, i.e. n <= 32 (tzr_dot_interaction_top_supported;
 * otherwise TZR_ERR_UNSUPPORTED and the caller runs the unfused ops).  Up to n = 29 the backward keeps two dz tiles in
 * LDS (one barrier per tile); n = 30 .. 32 run the same kernel with one tile (two barriers), same results.  fp32 MFMA, fixed summation order (deterministic).
 * B <= 2^30 samples per call (the kernels count tiles and samples in 32 bits; more: TZR_ERR_UNSUPPORTED).
 *   _top_fwd: y1[b] = act(z[b] W1^T + bias) (relu != 0: ReLU), [B, H]; the interaction row z[b] itself is written
 *             only when d_z is non-null (training keeps it for the weight gradient; inference does not need it).
 *   _top_bwd: from g1 = d(loss)/d(z W1^T + bias) [B, H]: grad_dense / grad_sparse = the interaction backward of
 *             dz = scale * g1 W1, with dz never leaving the chip.  d_scale: device scalar or null (= 1). */
int tzr_dot_interaction_top_supported(int F, int D, int has_dense, int H);
int tzr_dot_interaction_top_fwd(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                int64_t sparse_stride, int F, int D, int64_t B, const float* d_W1, int64_t ldw,
                                const float* d_bias, int H, int relu, float* d_z, int64_t z_stride, float* d_y1,
                                int64_t y1_stride, void* stream);
int tzr_dot_interaction_top_bwd(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                int64_t sparse_stride, int F, int D, int64_t B, const float* d_g1, int64_t g1_stride,
                                int H, const float* d_W1, int64_t ldw, const float* d_scale, float* d_grad_dense,
                                int64_t grad_dense_stride, float* d_grad_sparse, int64_t grad_sparse_stride,
                                void* stream);

/* K9c: the weight gradient of that first Linear, dW1 = scale * g1^T z [H, P + D n] (autograd of nn.Linear in
 * tzrec/modules/mlp.py:58-83 behind tzrec/models/dlrm.py:123-135), with the interaction rows z REBUILT from the
 * embeddings on the chip: training does not keep z (tzr_dot_interaction_top_fwd with d_z = null).  Same shapes as
 * tzr_dot_interaction_top_supported; exact-fp32 MFMA; the batch is summed in a fixed order (run-to-run identical).
 * d_ws: tzr_dot_interaction_top_wgrad_workspace(F, D, has_dense, H) bytes, 16-byte aligned (partial sums of the batch
 * slices).  d_scale: device scalar or null (= 1).  d_dW1 row pitch ldw >= P + D n; columns behind the width are left alone.
 * B = 0 writes zeros (the width still follows from F and whether d_dense is null). */
int64_t tzr_dot_interaction_top_wgrad_workspace(int F, int D, int has_dense, int H);
int tzr_dot_interaction_top_wgrad(const float* d_dense, int64_t dense_stride, const float* d_sparse,
                                  int64_t sparse_stride, int F, int D, int64_t B, const float* d_g1, int64_t g1_stride,
                                  int H, const float* d_scale, float* d_dW1, int64_t ldw, void* d_ws, int64_t ws_bytes,
                                  void* stream);

/* K10: FM second order.  Replaces tzrec FactorizationMachine.forward
 * (tzrec/modules/fm.py:27-42): out[b,:] = 0.5*((sum_f x_f)^2 - sum_f x_f^2), x:[B,F,D]. */
int tzr_fm_fwd(const float* d_x, int64_t x_stride, int F, int D, int64_t B, float* d_out,
               int64_t out_stride, void* stream);
int tzr_fm_bwd(const float* d_x, int64_t x_stride, int F, int D, int64_t B,
               const float* d_grad_out, int64_t grad_out_stride, float* d_grad_x,
               int64_t grad_x_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TZREC_HIP_H_ */
