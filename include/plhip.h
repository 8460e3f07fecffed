/*
 * plhip.h — C ABI of libplhip.so: the MI355X (gfx950) hash group-by aggregation path.
 *
 * This header is the drop-in boundary for ONE hot path of Polars 0.33.2 (reference repo
 * backend-devloper/polars): hash group-by aggregation over Arrow-backed columns.  Every entry point
 * cites the reference interface it replaces as `path:line` relative to the reference root.  A Rust
 * `polars-hip-sys` crate would bind these 1:1 (see INTEGRATION.md for the binding stubs).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns a plhip_status_t (0 = OK) and never unwinds; a human readable message for
 *     the calling thread's last failure is available from plhip_last_error();
 *   - columns follow the Arrow primitive layout the reference uses
 *     (crates/nano-arrow/src/array/primitive/mod.rs:48-52): a values buffer, an optional LSB-first
 *     validity bitmap (crates/nano-arrow/src/bitmap/utils/mod.rs:19,77-79) and an element offset that
 *     applies to both;
 *   - `location` says where the buffers live: host memory (the library stages them through pinned
 *     buffers) or device memory of the current HIP device (zero-copy; what bench.py uses);
 *   - all entry points are thread safe.  Every host thread owns a context: the device it is bound to, a HIP stream
 *     of its own, a grow-only workspace (about the largest query's working set) and its settings (strategy, last
 *     error, timings are per thread).  Calls from different threads do NOT serialise: they run on their own streams
 *     and the GPU interleaves them.  The only shared state is the pool of result buffers (locked, keyed by device)
 *     and the static array of plhip_group_by_agg_noalloc (see there).  A thread's context is released when the
 *     thread ends; plhip_release_workspace() gives the workspace back earlier;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     PLHIP_ERR_NO_DEVICE.
 */
#ifndef PLHIP_H
#define PLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLHIP_ABI_VERSION 1

/* Mirrors the PolarsError kinds this path can raise (crates/polars-error/src/lib.rs;
 * crates/polars-core/src/frame/group_by/mod.rs:59-73: ComputeError for zero keys, ShapeMismatch for
 * key/frame length mismatch). */
typedef enum plhip_status {
  PLHIP_OK = 0,
  PLHIP_ERR_COMPUTE = 1,           /* PolarsError::ComputeError */
  PLHIP_ERR_SHAPE_MISMATCH = 2,    /* PolarsError::ShapeMismatch */
  PLHIP_ERR_OOM = 3,               /* allocation failure (host or HBM) */
  PLHIP_ERR_UNSUPPORTED_DTYPE = 4, /* PolarsError::InvalidOperation for this path */
  PLHIP_ERR_HIP = 5,               /* a HIP runtime call failed; message holds hipGetErrorString */
  PLHIP_ERR_INVALID = 6,           /* NULL pointer / bad enum / bad index at the ABI */
  PLHIP_ERR_NO_DEVICE = 7          /* no gfx950 device visible; the library has no CPU path */
} plhip_status_t;

/* Physical dtypes of the hot path (crates/polars-core/src/datatypes/dtype.rs; Categorical keys arrive
 * as their u32 physical codes, crates/polars-core/src/frame/group_by/mod.rs:33-34). */
typedef enum plhip_dtype {
  PLHIP_I8 = 0,
  PLHIP_I16 = 1,
  PLHIP_I32 = 2,
  PLHIP_I64 = 3,
  PLHIP_U8 = 4,
  PLHIP_U16 = 5,
  PLHIP_U32 = 6,
  PLHIP_U64 = 7,
  PLHIP_F32 = 8,
  PLHIP_F64 = 9,
  /* Arrow boolean array: `values` is an LSB-first bitmap (one bit per row, `offset` in bits).  Accepted by plhip_group_by_agg /
   * plhip_group_by_agg_opts as a key and as a value column, with the reference's casts: a Boolean key groups as UInt8
   * (into_groups.rs:215-230) and comes back as a Boolean key column; sum(bool) is the u32 sum of the column cast to IdxSize
   * (aggregations/boolean.rs:107-109), mean(bool) the f64 mean of its cast (aggregations/dispatch.rs:164), min / max stay
   * Boolean (boolean.rs:20-106), count is a count.  Also accepted, with the same casts, by the CSR seams: plhip_group_tuples
   * (Boolean keys) and plhip_agg_groups (sum, mean, min, max, first, last of a Boolean column; first / last stay Boolean) --
   * by plhip_group_by_agg_sharded, and by plhip_vec_hash_seeded.  Other entry points reject the dtype. */
  PLHIP_BOOL = 10
} plhip_dtype_t;

typedef enum plhip_location { PLHIP_HOST = 0, PLHIP_DEVICE = 1 } plhip_location_t;

/* One single-chunk Arrow primitive array (the reference rechunks before this path:
 * crates/polars-lazy/src/physical_plan/executors/group_by.rs:66,
 * crates/polars-core/src/frame/group_by/aggregations/mod.rs:625). */
typedef struct plhip_column {
  const void *values;      /* buffer start (element 0 of the BUFFER, before `offset`) */
  const uint8_t *validity; /* LSB-first bitmap over the buffer, or NULL = no nulls */
  int64_t offset;          /* Arrow array offset in elements, applied to values and validity */
  int64_t length;          /* number of elements */
  int32_t dtype;           /* plhip_dtype_t */
  int32_t location;        /* plhip_location_t */
} plhip_column_t;

/* GroupByMethod subset on the path (crates/polars-core/src/frame/group_by/mod.rs:844-861). */
typedef enum plhip_agg_kind {
  PLHIP_AGG_SUM = 0,   /* aggregations/mod.rs:622-688 */
  PLHIP_AGG_MEAN = 1,  /* aggregations/mod.rs:701-779 (float), 941-1016 (int) */
  PLHIP_AGG_MIN = 2,   /* aggregations/mod.rs:459-537 */
  PLHIP_AGG_MAX = 3,   /* aggregations/mod.rs:539-620 */
  PLHIP_AGG_COUNT = 4, /* proxy.rs:452-466: group length INCLUDING nulls, as IdxSize (u32) */
  /* over CSR groups only (plhip_agg_groups): */
  PLHIP_AGG_FIRST = 5, /* value (null or not) at the group's first index; empty group -> null (dispatch.rs:48-100) */
  PLHIP_AGG_LAST = 6,  /* ... at its last index */
  PLHIP_AGG_VAR = 7,   /* online_variance over the valid values as f64 (crates/polars-arrow/src/kernels/take_agg/var.rs:11-40;
                          aggregations/mod.rs:781-803,1018-1035): f32 -> f32, everything else -> f64; count <= ddof -> null */
  PLHIP_AGG_STD = 8,   /* sqrt of it (aggregations/mod.rs:841-862,1059-1077) */
  PLHIP_AGG_MEDIAN = 9, /* agg_median = quantile 0.5, linear interpolation over the sorted valid values, NaN last
                          (dispatch.rs:98-113, chunked_array/ops/aggregate/quantile.rs:19-54,133-175): f32 -> f32,
                          everything else -> f64; no valid value -> null */
  PLHIP_AGG_N_UNIQUE = 10 /* agg_n_unique (dispatch.rs:82-107) = ChunkUnique::n_unique of the group's values
                          (chunked_array/ops/unique/mod.rs:136-171): sorted, then one per change of value; nulls are one
                          value, -0.0 == 0.0, every NaN counts (NaN != NaN); IdxSize (u32); empty group -> null */
} plhip_agg_kind_t;
#define PLHIP_AGG_DDOF(d) ((uint32_t)(d) << 8) /* delta degrees of freedom for VAR / STD, in plhip_agg_groups' flags */

typedef struct plhip_agg {
  int32_t column; /* index into the `values` array; ignored for COUNT */
  int32_t kind;   /* plhip_agg_kind_t */
} plhip_agg_t;

/* flags for plhip_group_by_agg */
#define PLHIP_FLAG_MAINTAIN_ORDER 1u /* group_by_stable: groups ordered by first row (hashing.rs:32-79) */
#define PLHIP_FLAG_OUTPUT_DEVICE 2u  /* result buffers stay in HBM (device pointers) */
#define PLHIP_FLAG_PARTIAL 4u        /* emit mergeable partial state instead of final values: MEAN becomes
                                        two columns (sum, valid count) as in AggregationExpr::
                                        evaluate_partitioned (crates/polars-lazy/src/physical_plan/
                                        expressions/aggregation.rs:283-365) */
#define PLHIP_FLAG_KEYS_SORTED 8u    /* the (single) key column carries the reference's sorted flag (ascending or
                                        descending, nulls contiguous at the front or the back): groups are the runs of
                                        equal consecutive values, in row order, no hashing
                                        (create_groups_from_sorted, crates/polars-core/src/frame/group_by/
                                        into_groups.rs:58-127,142-150; partition_to_groups, crates/polars-arrow/src/
                                        kernels/sort_partition.rs:71-116).  Values compare as values there: every
                                        NaN is a run of its own and -0.0 == 0.0.  Like the reference, the library
                                        trusts the flag: unsorted input yields one output row per run.  Ignored
                                        with more than one key. */

#define PLHIP_FLAG_KEYS_DENSE 16u    /* the (single, u32) key column holds the physical codes of a Categorical with a local
                                        rev-map: every valid code is < plhip_options_t::n_groups_hint (the rev-map's
                                        length).  The codes then index the tables directly -- no hashing, no probing, the
                                        null key in the last slot -- as group_tuples_perfect does (crates/polars-core/src/
                                        frame/group_by/perfect.rs:24-185, 190-227; taken for Categorical keys,
                                        into_groups.rs via CategoricalChunked::group_tuples_perfect).  Codes that never
                                        occur produce no group.  A code outside the promise costs time, never
                                        correctness: the hashed strategies redo the query. */

/* strategy override, PLHIP_STRATEGY env var or plhip_set_strategy(): mirrors the reference's
 * POLARS_NO_PARTITION / POLARS_FORCE_PARTITION switches
 * (crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:167-186). */
#define PLHIP_FLAG_EXACT_FOLD 64u     /* float SUM / MEAN (and the f64 sum behind i64 / u64 means) fold every group strictly in
                                       * ascending row order from 0 -- the association of the reference's GroupsProxy::Idx path
                                       * (take_agg_no_null_primitive_iter_unchecked, crates/polars-arrow/src/kernels/take_agg/
                                       * mod.rs:17-37; agg_sum / agg_mean, aggregations/mod.rs:622-651, 701-742) -- with f32 columns
                                       * folded IN f32 (aggregations/mod.rs:719-724) and the reference's 1-row shortcut
                                       * (`arr.get(first)`).  Results are then BIT-EXACT against that path and identical from run
                                       * to run; without the flag float sums are re-associated (LDS atomics; bound below).  The
                                       * query runs on the sort strategy (stable radix sort of (key, row), one sequential fold per
                                       * group; a group of >= 2048 rows is folded by ONE wave, 64 rows per trip): about 10x the
                                       * fused paths' time at q3's shape.  Not with PLHIP_FLAG_PARTIAL. */
typedef enum plhip_strategy {
  PLHIP_STRATEGY_AUTO = 0,
  PLHIP_STRATEGY_GLOBAL = 1,     /* every row updates the HBM-resident table */
  PLHIP_STRATEGY_LDS = 2,        /* per-workgroup LDS tables, then one merge into the HBM table */
  PLHIP_STRATEGY_PARTITIONED = 3,/* radix partition on hash bits, then one LDS table per partition */
  PLHIP_STRATEGY_DIRECT = 4,     /* LDS-only: per-workgroup LDS tables over row slices, merged in LDS (few groups) */
  PLHIP_STRATEGY_TILESORT = 5,   /* LDS-only: tiles sorted by hash partition in LDS, per-partition LDS tables, merged
                                    in LDS (many groups); AUTO picks DIRECT / TILESORT when the query qualifies and
                                    falls back to the strategies above otherwise */
  PLHIP_STRATEGY_SORT = 6,       /* radix sort of (packed key, row), run boundaries, one fold per run: any number of
                                    groups up to all-unique keys, immune to skew */
  PLHIP_STRATEGY_TILESORT2 = 7,  /* two partition levels (radix partition, then the tile sort inside every partition),
                                    one LDS table per leaf: more groups than TILESORT's 512 tables hold */
  PLHIP_STRATEGY_DENSE = 8       /* dense Categorical codes (PLHIP_FLAG_KEYS_DENSE) index LDS tables directly: code ranges x
                                    row slices, one pass over the rows, no exchange; AUTO takes it whenever the flag is set
                                    and the tables fit (<= 8 code ranges of one CU's LDS) */
} plhip_strategy_t;

/* Optional knowledge the caller has about the query.  struct_size = sizeof(plhip_options_t). */
typedef struct plhip_options {
  uint64_t struct_size;
  int64_t n_groups_hint; /* upper bound on the number of groups, 0 = unknown.  For a Categorical key the
                            reference knows it without looking at the data (the rev-map length it uses to size
                            group_tuples_perfect, crates/polars-core/src/frame/group_by/perfect.rs:190-227);
                            without a hint the library samples the keys, as estimate_unique_count does
                            (crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:115-157).
                            A wrong hint costs time, never correctness. */
  /* slice pushdown: GroupByExec carries `slice: Option<(i64, usize)>` and cuts the groups before it aggregates
     (crates/polars-lazy/src/physical_plan/executors/group_by.rs:79-82, proxy.rs:504-547).  Here the result
     frame is cut to rows [offset, offset + len) of the group order, negative offsets counting from the end
     exactly as slice_offsets does (crates/polars-core/src/utils/mod.rs:199-218); the columns then carry an
     Arrow offset.  Meaningful with PLHIP_FLAG_MAINTAIN_ORDER (otherwise the group order is unspecified, as in
     the reference). */
  int64_t slice_offset;
  uint64_t slice_len;
  uint32_t has_slice; /* 0: no slice */
  uint32_t pad;
} plhip_options_t;

typedef struct plhip_result plhip_result_t; /* opaque: a frame of output columns */
typedef struct plhip_groups plhip_groups_t; /* opaque: GroupsProxy::Idx as CSR */

/* ---- library ------------------------------------------------------------------------------------ */

/* ABI version of the loaded library (== PLHIP_ABI_VERSION of the header it was built from). */
int32_t plhip_abi_version(void);

/* Selects the HIP device for the calling thread and creates its stream/workspace pool.
 * Reference analogue: the global rayon POOL (crates/polars-core/src/lib.rs:49-62). */
int32_t plhip_init(int32_t device_id);

/* Frees the calling thread's grow-only workspace (hash tables, partitioned copies, partial dumps: sized by the largest
 * query the thread has run) and the pooled result buffers of its device.  The stream stays; the next call re-grows
 * what it needs.  The reference analogue is dropping the per-thread hash tables at the end of a query
 * (crates/polars-core/src/frame/group_by/hashing.rs:380-468 builds them per call); here they are kept between calls
 * because hipMalloc costs more than a small query.  Host-resident results (no PLHIP_FLAG_OUTPUT_DEVICE; plhip_group_tuples with
 * PLHIP_HOST) of 32 KiB and more live in page-locked buffers that are recycled the same way (up to 2 GiB per process; the copy out
 * of HBM is then one DMA at the link's rate): this call hands those back as well.  PLHIP_NO_PINNED_RESULTS=1 in the environment
 * keeps host results in plain malloc'ed memory. */
int32_t plhip_release_workspace(void);

/* Result buffers currently pooled for `device_id` (diagnostic: the pool is keyed by device, a buffer of one GPU is
 * never handed to a thread bound to another). */
int64_t plhip_pooled_buffers(int32_t device_id);

/* Message for the calling thread's last non-OK status ("" if none). Never NULL. */
const char *plhip_last_error(void);

/* Forces a strategy for the CALLING THREAD's later calls (the setting is per thread, like the context);
 * PLHIP_STRATEGY_AUTO restores the library's choice and overrides the PLHIP_STRATEGY environment variable. */
int32_t plhip_set_strategy(int32_t strategy /* plhip_strategy_t */);

/* Name of the strategy the last plhip_group_by_agg call on this thread ran ("global", "lds", "partitioned",
 * "direct", "tilesort", "tilesort2", "leafsort", "sort", "sorted_keys", "dense"); the POLARS_VERBOSE analogue
 * (crates/polars-lazy/src/physical_plan/executors/group_by.rs:124-126). */
const char *plhip_last_strategy(void);

/* The single-launch pipelines (direct, tilesort, dense) bracket their kernels with HIP events for plhip_last_kernel_ms /
 * _breakdown.  Three event records cost a few microseconds of a 50-100 us query: a host that runs many small queries has
 * them recorded on every `every`-th query of the calling thread only (1 = every query, the default; 0 = never);
 * plhip_last_kernel_ms then reports the last query that was timed.  No counterpart in the reference. */
int32_t plhip_set_timing_interval(int32_t every);

/* Milliseconds the device spent in the kernels of the last plhip_group_by_agg call on this thread
 * (HIP events on the library's own stream around the launches; staging copies excluded), and the
 * part of it spent in the dominant kernel.  Used by bench.py for `roofline.achieved`. */
int32_t plhip_last_kernel_ms(float *total_ms, float *dominant_ms, const char **dominant_kernel);

/* Blocking copy between buffers of result frames / groups handles and caller memory, on the library's own
 * HIP runtime (what a host layer uses to move a device-resident result into its own buffers).  No
 * counterpart in the reference: there results never leave host memory. */
int32_t plhip_memcpy(void *dst, int32_t dst_location, const void *src, int32_t src_location, size_t bytes);

/* Per-kernel device time of the last call when it ran the LDS-only pipeline (strategy "direct" /
 * "tilesort"): k_tile_sort (0 for "direct"), k_fold_rows / k_fold_runs, k_merge_emit. */
int32_t plhip_last_kernel_breakdown(float *tile_sort_ms, float *fold_ms, float *merge_emit_ms);

/* ---- hashing ------------------------------------------------------------------------------------ */

/* VecHash::vec_hash on keys[0] then VecHash::vec_hash_combine for keys[1..]
 * (crates/polars-core/src/hashing/vector_hasher.rs:78-139, series_to_hashes :467-483):
 *   h = folded_multiply(as_u64(v), 6364136223846793005) ; combine: h = folded_multiply(as_u64(v) ^ h, M)
 * Null rows hash to `null_hash` (get_null_hash_value :42-51 is seed dependent, so the caller supplies
 * it).  Float columns hash their bit pattern (:304-326).  `out` has keys[0].length u64 elements and
 * lives where `out_location` says. */
int32_t plhip_vec_hash(const plhip_column_t *keys, int32_t n_keys, uint64_t null_hash, uint64_t *out,
                       int32_t out_location);

/* The same with Boolean key columns among the keys (dtype PLHIP_BOOL: values = the Arrow bitmap).  VecHash for BooleanChunked
 * (vector_hasher.rs:241-302): a Boolean value hashes to random_state.hash_one(true) / hash_one(false) -- seed dependent like the
 * null hash, so the caller supplies both -- and a Boolean column behind the first one combines with
 *   _boost_hash_combine(l, h) = l ^ (h + 0x9e3779b9 + (l << 6) + (h >> 2))     (hashing/mod.rs:20-22)
 * instead of folded_multiply.  Integer and float columns hash as in plhip_vec_hash. */
typedef struct plhip_hash_seeds {
  uint64_t null_hash;  /* get_null_hash_value(random_state) */
  uint64_t true_hash;  /* random_state.hash_one(true) */
  uint64_t false_hash; /* random_state.hash_one(false) */
} plhip_hash_seeds_t;
int32_t plhip_vec_hash_seeded(const plhip_column_t *keys, int32_t n_keys, const plhip_hash_seeds_t *seeds, uint64_t *out,
                              int32_t out_location);

/* ---- fused group-by + aggregation ------------------------------------------------------------- */

/* Replaces group_by_helper / GroupByExec::execute end to end
 * (crates/polars-lazy/src/physical_plan/executors/group_by.rs:57-148) and the low-cardinality
 * PartitionGroupByExec (group_by_partitioned.rs:273-361): groups `keys` (null key = its own group,
 * float keys by bit pattern) and evaluates `aggs` over `values`.
 * Output frame = key columns (input dtypes) followed by one column per agg, in order
 * (group_by.rs:91-94).  Output dtypes follow the reference:
 *   SUM   i8/i16/u8/u16 -> i64 (series/implementations/mod.rs:167-174); i32,u32,i64,u64 wrap in their
 *         own type; f32 -> f32; f64 -> f64; empty/all-null group -> 0, never null
 *   MEAN  f32 -> f32, everything else -> f64; all-null group -> null
 *   MIN/MAX own dtype; floats ignore NaN; all-null group -> null
 *   COUNT u32
 * Float sums: the reference folds a group's values left to right in row order per thread and then adds the
 * threads' partials; here f64 (and f32, folded in f64 and rounded once) sums are re-associated -- LDS / global atomics
 * and fixed-order merges of per-workgroup partials -- so SUM / MEAN / VAR of floats agree with the reference to
 * |err| <= 1e-12 * sum|x| of the group (f64; 1e-5 relative for f32 columns), not bit for bit, and the atomic part can
 * differ between runs in the last bits.  Integer results, counts, keys and validity are bit-exact.
 * Errors: n_keys == 0 -> PLHIP_ERR_COMPUTE; length mismatch -> PLHIP_ERR_SHAPE_MISMATCH
 * (group_by/mod.rs:59-73). */
int32_t plhip_group_by_agg(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                           int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                           plhip_result_t **out);

/* plhip_group_by_agg with caller-side knowledge (`opts` may be NULL). */
int32_t plhip_group_by_agg_opts(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                                int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                const plhip_options_t *opts, plhip_result_t **out);

/* plhip_group_by_agg_opts over CHUNKED columns.  The reference rechunks the frame before this path
 * (`df.as_single_chunk_par()`, crates/polars-lazy/src/physical_plan/executors/group_by.rs:66; ChunkedArray::rechunk,
 * crates/polars-core/src/chunked_array/ops/chunkops.rs); here the chunks of every column -- host or device memory, each with its
 * own Arrow offset and optional validity bitmap -- are laid end to end in HBM by the library (values: one copy per chunk on the
 * library's stream; validity: a bit-granular copy kernel, chunks without a bitmap contribute ones) and the query runs on the
 * single-chunk columns.  A column with ONE chunk is passed through untouched (zero-copy for device memory).  All columns must
 * have the same total length; chunk boundaries need not line up between columns.  PLHIP_BOOL columns are accepted: their value
 * bitmaps are laid end to end with the same bit-granular copy. */
typedef struct plhip_chunked_column {
  const plhip_column_t *chunks; /* n_chunks single-chunk arrays of one dtype */
  int32_t n_chunks;
  int32_t pad;
} plhip_chunked_column_t;
int32_t plhip_group_by_agg_chunked(const plhip_chunked_column_t *keys, int32_t n_keys, const plhip_chunked_column_t *values,
                                   int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                   const plhip_options_t *opts, plhip_result_t **out);

/* plhip_group_by_agg_opts for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc,
 * processes with a stream of their own under HIP graph capture (which forbids allocation process-wide in the default
 * capture mode).  NO counterpart in the reference (there every aggregation allocates its output
 * Series); it exists for those callers only.  The call is BLOCKING and not itself capturable: it launches on the library's
 * private stream, waits for the group count through a pinned signal word and hipStreamSynchronize, and returns the count.
 * It calls no hipMalloc / hipFree / allocating routine: inputs and
 * outputs are the caller's device buffers, and everything the kernels need beyond them (flag block, dump headers,
 * the compact table dumps k_fold_rows hands to k_merge_emit) lives in one static device array of
 * PLHIP_NOALLOC_SCRATCH_BYTES per GPU, resolved with hipGetSymbolAddress when plhip_init binds the thread's device.
 *   - plhip_init must have been called on the calling thread (it creates the stream and the pinned signal word);
 *   - exactly one key column; keys / values device-resident; flags: 0 or PLHIP_FLAG_PARTIAL;
 *   - n_groups_hint >= 1 is required (an upper bound on the groups, e.g. a Categorical key's category count), and the
 *     groups must fit the DIRECT strategy (one LDS table, roughly <= 5 000 groups) within the scratch limit: anything
 *     larger is rejected with PLHIP_ERR_INVALID and a message that names the limit;
 *   - out_cols[n_out_cols] = the frame plhip_group_by_agg_opts would return (key column, then one column per
 *     aggregation; two for MEAN under PLHIP_FLAG_PARTIAL), as caller-allocated DEVICE buffers: dtype = the output
 *     dtype, offset 0, length = CAPACITY in rows, validity = a bitmap of (length / 64 + 1) * 8 bytes where the column
 *     can be null (a nullable key, MEAN / MIN / MAX of a nullable column), ignored elsewhere.  The rows written = *n_groups
 *     (group order unspecified, as in the reference); more groups than the smallest capacity -> PLHIP_ERR_COMPUTE.
 * All calls on one GPU share that GPU's array, which keeps the last call's contents (the entry point clears what must
 * start at zero): calls that overlap on one GPU race -- serialise them. */
#define PLHIP_NOALLOC_SCRATCH_BYTES (40u << 20)
int32_t plhip_group_by_agg_noalloc(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                                   int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                   int64_t n_groups_hint, const plhip_column_t *out_cols, int32_t n_out_cols,
                                   int64_t *n_groups);

/* ---- result frame accessors ------------------------------------------------------------------- */

int64_t plhip_result_num_rows(const plhip_result_t *r);    /* number of groups */
int32_t plhip_result_num_columns(const plhip_result_t *r); /* n_keys + n output columns */
/* Fills `col` with the i-th output column (location = host unless PLHIP_FLAG_OUTPUT_DEVICE). The
 * buffers belong to the result and die with plhip_result_free.
 * Deferred completion: plhip_group_by_agg[_opts] with PLHIP_FLAG_OUTPUT_DEVICE returns as soon as the frame's row
 * count is on the host (the last kernel publishes it through pinned memory); that kernel may still be retiring.
 * This call -- like plhip_result_pack_* and plhip_finalize_partial -- waits for it once, so a device pointer obtained
 * here always names complete data, readable from any stream; an error of the kernels that surfaces only then is
 * returned here.  plhip_result_num_rows / _num_columns / _record_bytes / _free never wait: a caller that chains queries
 * or wants only the count does not pay the ~7 us stream wait per call (a freed frame's pooled buffers carry the
 * producing stream with them, and whoever reuses them from another stream waits first).
 * PLHIP_NO_DEFERRED_SYNC=1 in the environment restores the blocking behaviour. */
int32_t plhip_result_column(const plhip_result_t *r, int32_t i, plhip_column_t *col);
void plhip_result_free(plhip_result_t *r);

/* ---- two-step API: the seams a Rust host layer would rebind -------------------------------------- */

/* IntoGroupsProxy::group_tuples (crates/polars-core/src/frame/group_by/into_groups.rs:12-19) for one
 * key and group_by_threaded_multiple_keys_flat (hashing.rs:380-468) for several: builds
 * GroupsProxy::Idx (proxy.rs:16-20) as CSR: first[G], offsets[G+1], indices[N]; indices ascending
 * inside a group, first[g] == indices[offsets[g]].  `sorted` != 0 orders groups by first
 * (hashing.rs:32-79); otherwise the order is unspecified, as in the reference. */
int32_t plhip_group_tuples(const plhip_column_t *keys, int32_t n_keys, int32_t sorted, int32_t out_location,
                           plhip_groups_t **out);
int64_t plhip_groups_num_groups(const plhip_groups_t *g);
int64_t plhip_groups_num_rows(const plhip_groups_t *g);
/* Pointers (u32 = IdxSize, crates/polars-utils/src/lib.rs:19-22) owned by the handle. */
int32_t plhip_groups_buffers(const plhip_groups_t *g, const uint32_t **first, const uint32_t **offsets,
                             const uint32_t **indices, int32_t *location);
void plhip_groups_free(plhip_groups_t *g);

/* Series::agg_sum / agg_mean / agg_min / agg_max / agg_first / agg_last / agg_var / agg_std on GroupsProxy::Idx
 * (crates/polars-core/src/series/series_trait.rs:129-154, aggregations/mod.rs:459-1077, dispatch.rs:48-100) and
 * GroupsProxy::group_count (proxy.rs:452-466) for kind == COUNT (values may be NULL then).  VAR / STD take their
 * ddof from flags (PLHIP_AGG_DDOF).
 * Returns a one-column result frame with G rows. */
int32_t plhip_agg_groups(const plhip_column_t *values, int32_t kind, const plhip_groups_t *groups,
                         uint32_t flags, plhip_result_t **out);

/* QuantileInterpolOptions (crates/polars-arrow/src/kernels/rolling/no_nulls/mod.rs:70-77) */
typedef enum plhip_quantile_interpol {
  PLHIP_QUANTILE_NEAREST = 0, PLHIP_QUANTILE_LOWER = 1, PLHIP_QUANTILE_HIGHER = 2, PLHIP_QUANTILE_MIDPOINT = 3, PLHIP_QUANTILE_LINEAR = 4
} plhip_quantile_interpol_t;
/* Series::agg_quantile on GroupsProxy::Idx (dispatch.rs:115-140 -> agg_quantile_generic, aggregations/mod.rs:333-414;
 * quantile_idx / linear_interpol / midpoint_interpol / generic_quantile, chunked_array/ops/aggregate/quantile.rs:19-61,
 * 133-179): the group's valid values sorted ascending (NaN last), index arithmetic in f64 with the group's null count
 * as the reference has it; f32 column -> f32 (the f64 answer cast), everything else -> f64; no valid value -> null; a
 * quantile outside [0, 1] -> an all-null column of the input dtype (mod.rs:346-349). */
int32_t plhip_agg_groups_quantile(const plhip_column_t *values, const plhip_groups_t *groups, double quantile,
                                  int32_t interpolation, uint32_t flags, plhip_result_t **out);

/* ---- exchange step of the row-sharded group-by (one process per GPU) ---------------------------- */

/* The reference runs the partitioned group-by between THREADS: local group-by with mergeable partials ->
 * concat of the partial frames -> regroup -> finalize (PartitionGroupByExec::run_partitions / execute,
 * crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:66-113, 320-360).  Between GPUs the
 * concat is ONE all-to-all of row records bucketed by the rank that owns each key; these two entry points
 * are its device side (polars_amd/distributed.py drives them).
 *
 * Record of a frame row = for every column its value bytes followed by one validity byte (0/1) if the column
 * has a bitmap; plhip_result_record_bytes = the record size of `r`. */
int64_t plhip_result_record_bytes(const plhip_result_t *r);

/* Packs the rows of the device-resident frame `r` (its first n_keys columns are the keys) into `records`
 * (device memory, num_rows * record_bytes), bucket after bucket by
 *     owner(row) = ((vec_hash(keys)[row] >> 33) & 0x7FFFFFFF) % n_parts        (plhip_vec_hash, same null_hash)
 * and returns the rows per owner in counts[n_parts] (host).  Order inside a bucket is unspecified. */
int32_t plhip_result_pack_by_owner(const plhip_result_t *r, int32_t n_keys, uint64_t null_hash, int32_t n_parts,
                                   void *records, int64_t *counts);

/* The receiving side: n_rows records (device memory) of the schema dtypes[n_cols] / nullable[n_cols] become a
 * device-resident frame (free it with plhip_result_free). */
int32_t plhip_records_to_frame(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable,
                               int32_t n_cols, plhip_result_t **out);

/* plhip_records_to_frame followed by the regroup of the partials (PartitionGroupByExec::execute,
 * group_by_partitioned.rs:340-360; AggregationExpr::finalize's per-kind merge, aggregation.rs:367-483: sum of
 * sums and counts, min of mins, max of maxs): the first n_keys columns are the keys, merge_kinds[n_cols -
 * n_keys] (PLHIP_AGG_SUM / _MIN / _MAX) says how each partial column merges.  flags / opts as in
 * plhip_group_by_agg_opts.  One call, no host round trip between the unpacking and the regroup. */
int32_t plhip_merge_records(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable,
                            int32_t n_cols, int32_t n_keys, const int32_t *merge_kinds, uint32_t flags,
                            const plhip_options_t *opts, plhip_result_t **out);

/* The same exchange without the counts round trip: fixed-capacity slots.  Every rank packs bucket b into slot b =
 * a 64-byte header {u64 rows, u64 overflowed} + slot_rows records (plhip_slot_bytes per slot), the slots travel with
 * ONE equal-split all-to-all, and the receiver reads the rows per slot from the headers.  slot_rows must be the
 * same on every rank (derive it from the query: the n_groups_hint).  A rank whose bucket exceeds slot_rows sets
 * `overflowed` in ALL its headers and keeps only the first slot_rows records of that bucket: every receiver then
 * sees the flag of every sender, and all ranks redo the exchange with plhip_result_pack_by_owner / counts.
 * counts[n_parts] (host) = the rows per bucket before clamping.  n_parts <= 64. */
int64_t plhip_slot_bytes(int64_t record_bytes, int64_t slot_rows);
int32_t plhip_result_pack_slots(const plhip_result_t *r, int32_t n_keys, uint64_t null_hash, int32_t n_parts,
                                int64_t slot_rows, void *slots, int64_t *counts);
/* plhip_merge_records over received slots: slot_counts[n_slots] (host) = the header's rows of every slot. */
int32_t plhip_merge_slots(const void *slots, int32_t n_slots, int64_t slot_rows, const int64_t *slot_counts,
                          const int32_t *dtypes, const uint8_t *nullable, int32_t n_cols, int32_t n_keys,
                          const int32_t *merge_kinds, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out);

/* AggregationExpr::finalize (crates/polars-lazy/src/physical_plan/expressions/aggregation.rs:367-483) on a
 * regrouped PARTIAL frame: sum partials wrap into the column's own sum type, counts become IdxSize (u32),
 * mean = sum / count (f32 sources round through f32 first; count 0 -> null), keys / min / max pass through.
 * `merged` = n_keys key columns, then per final one partial column (two for MEAN: sum, count).  out_cols[n_keys +
 * n_finals] are caller-allocated DEVICE buffers of num_rows elements of the final dtype (as plhip_group_by_agg
 * would produce); `validity` there = a bitmap of (num_rows / 64 + 1) * 8 bytes, required where the merged column
 * has one, optional for MEAN (NULL: not wanted), ignored elsewhere. */
typedef struct plhip_final {
  int32_t kind;      /* plhip_agg_kind_t */
  int32_t src_dtype; /* dtype of the aggregated source column (ignored for COUNT) */
} plhip_final_t;
int32_t plhip_finalize_partial(const plhip_result_t *merged, int32_t n_keys, const plhip_final_t *finals, int32_t n_finals,
                               const plhip_column_t *out_cols);


/* ---- pinned host buffers ("pinned-host staging") -----------------------------------------------------
 * Host columns are copied to HBM by every entry point that accepts PLHIP_HOST.  From pageable memory the runtime
 * stages the copy through its own bounce buffers; from page-locked memory it is one DMA at PCIe rate.  A host layer
 * whose Arrow buffers come from an allocator it controls (Polars' is pluggable) can take them from here.  The memory is
 * ordinary host memory to the CPU; free it with plhip_free_pinned.  No counterpart in the reference. */
int32_t plhip_alloc_pinned(size_t bytes, void **ptr);
int32_t plhip_free_pinned(void *ptr);

/* ---- string keys ----------------------------------------------------------------------------------
 * IntoGroupsProxy for Utf8Chunked / BinaryChunked (crates/polars-core/src/frame/group_by/into_groups.rs:232-290): the
 * reference hashes every string once and groups the (hash, bytes) pairs -- BytesHash, hashing/vector_hasher.rs:422-446,
 * equal when the hashes and then the bytes are equal; all nulls form one group.  plhip_utf8_encode does the same on the
 * device and hands back what a Categorical holds: dense u32 codes (code = index of the string's group; null rows keep the
 * column's validity) and the dictionary (string g = the string at group g's first row; it has one null entry when the
 * column has nulls -- no valid row carries that code).  The codes are then an ordinary key column for every entry point
 * of this header; as a single key with PLHIP_FLAG_KEYS_DENSE and n_groups_hint = plhip_dictionary_len they take the
 * dense-code path (perfect.rs:190-227), and the host decodes the result's key codes through the dictionary the way a
 * Categorical's rev-map does.  Every row's bytes are compared with its group's first row's (a 64-bit collision between
 * different strings makes the encoder start over with another seed), so groups are exact, never probabilistic. */
typedef struct plhip_utf8_column { /* one single-chunk Arrow LargeUtf8 / LargeBinary array (i64 offsets: Utf8Chunked in 0.33) */
  const int64_t *offsets;  /* offsets buffer start: entries offset .. offset + length are read */
  const uint8_t *data;     /* values buffer the offsets index */
  const uint8_t *validity; /* LSB-first bitmap over the buffer, or NULL */
  int64_t offset;          /* Arrow array offset in elements */
  int64_t length;
  int32_t location;        /* plhip_location_t, of all three buffers */
  int32_t pad;
} plhip_utf8_column_t;
typedef struct plhip_dictionary plhip_dictionary_t;
/* `codes`: a device-resident u32 column of col->length rows (validity: a copy of the column's, or NULL), owned by *dict
 * and valid until plhip_dictionary_free. */
int32_t plhip_utf8_encode(const plhip_utf8_column_t *col, plhip_dictionary_t **dict, plhip_column_t *codes);
int64_t plhip_dictionary_len(const plhip_dictionary_t *dict); /* entries (= groups of the column, the null group included) */
/* The dictionary as a LargeUtf8 array in device memory or (copied on first request) host memory; owned by the dictionary. */
int32_t plhip_dictionary_strings(plhip_dictionary_t *dict, int32_t location, plhip_utf8_column_t *out);
void plhip_dictionary_free(plhip_dictionary_t *dict);

/* ---- hash join on one integer key ---------------------------------------------------------------------
 * The row-index pairs of crates/polars-ops/src/frame/join/hash_join/: hash_join_tuples_inner (single_keys_inner.rs:38-140)
 * and hash_join_tuples_left (single_keys_left.rs:109-).  Inner: the table is built over the SHORTER relation
 * (det_hash_prone_order!, hash_join/mod.rs:42-58: the left one when the lengths are equal), the longer one probes in row
 * order, a probe row's matches come in ascending build-row order, and the pairs are swapped back so that left stays left --
 * the order of the pairs is the reference's.  Left: the left relation probes; a row without a match yields
 * (row, PLHIP_JOIN_NONE).  Keys are Option<T>: nulls join nulls (crates/polars/tests/it/core/joins.rs:574-586).  Both keys
 * must have the same integer dtype.  The host layer gathers the payload columns with the two index arrays
 * (DataFrame::_finish_join), which is outside this path. */
typedef enum plhip_join_how {
  PLHIP_JOIN_INNER = 0,
  PLHIP_JOIN_LEFT = 1,
  /* semi_anti_impl (single_keys_semi_anti.rs:33-): the left rows whose key occurs (SEMI) / does not occur (ANTI) among the right
   * keys, in left order, each once; right_idx comes back NULL */
  PLHIP_JOIN_SEMI = 2,
  PLHIP_JOIN_ANTI = 3,
  /* hash_join_tuples_outer (single_keys_outer.rs:62-): build / probe sides as INNER; a probe row without a match yields (row, NONE);
   * afterwards the build rows whose key no probe row had follow as (NONE, row) -- in the reference in its hash map's iteration
   * order, here in the build side's group order: that tail's order is unspecified on both sides.  Either index may be NONE. */
  PLHIP_JOIN_OUTER = 4
} plhip_join_how_t;
#define PLHIP_JOIN_NONE 0xFFFFFFFFu
typedef struct plhip_join plhip_join_t;
int32_t plhip_hash_join(const plhip_column_t *left, const plhip_column_t *right, int32_t how, int32_t out_location, plhip_join_t **out);
/* The same on several key columns (_inner_join_multiple_keys / _left_join_multiple_keys, hash_join/multiple_keys.rs:173-,251-: rows
 * hashed, equal when every column is equal -- compare_df_rows2 -- floats by bit pattern, nulls equal to nulls; build side and pair
 * order as above).  left_keys[i] and right_keys[i] have the same dtype (any dtype of plhip_dtype_t but PLHIP_BOOL).  The key
 * tuples of both relations are reduced to dense ids over their concatenation (plhip_group_tuples), then joined on the ids. */
int32_t plhip_hash_join_multi(const plhip_column_t *left_keys, const plhip_column_t *right_keys, int32_t n_keys, int32_t how,
                              int32_t out_location, plhip_join_t **out);
int64_t plhip_join_num_rows(const plhip_join_t *j);

/* The other half of a join: the payload gather.  DataFrame::_finish_join's inputs are `df.take_unchecked(left_idx)` and
 * `other._take_opt_chunked_unchecked(right_idx)` (crates/polars-ops/src/frame/join/hash_join/mod.rs:105-201; general.rs:9-): row i of
 * every output column is row idx[i] of the input column, null where idx[i] == PLHIP_JOIN_NONE (the missing side of a left / outer
 * join) or where the source row is null.  One call gathers all payload columns of a side on the device: `cols` fixed-width columns
 * of equal length (host or device), `idx` the join's index array (plhip_join_buffers: device-resident after a PLHIP_DEVICE join, so
 * the pairs never leave HBM between the two calls).  The result frame has n_cols columns of n_idx rows; a column carries a
 * validity bitmap when its source does or when PLHIP_TAKE_NULLABLE_IDX says the indices may hold PLHIP_JOIN_NONE.  Indices are
 * trusted (< the columns' length), as in the reference's unchecked takes.  flags: PLHIP_FLAG_OUTPUT_DEVICE, PLHIP_TAKE_NULLABLE_IDX. */
#define PLHIP_TAKE_NULLABLE_IDX 256u
int32_t plhip_take(const plhip_column_t *cols, int32_t n_cols, const uint32_t *idx, int64_t n_idx, int32_t idx_location, uint32_t flags,
                   plhip_result_t **out);
/* u32 = IdxSize pairs, owned by the handle, in `*location` memory */
int32_t plhip_join_buffers(const plhip_join_t *j, const uint32_t **left_idx, const uint32_t **right_idx, int32_t *location);
void plhip_join_free(plhip_join_t *j);

/* ---- the sharded path: one rank per GPU, the exchange inside the library ------------------------------ */

/* The reference's partitioned group-by (crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:273-361) runs
 * the aggregation per thread on a slice of the rows, hands every partial to the owner of its hash partition (:320-360)
 * and finalizes there (AggregationExpr::finalize, expressions/aggregation.rs:367-483).  Across GPUs the "threads" are
 * ranks, one per GPU, and the hand-over is an all-to-all of packed partial records over xGMI.  A communicator carries
 * the transport:
 *   - RCCL: plhip_comm_unique_id on one rank, the 128 bytes sent to the others by the host's own means (MPI, a file,
 *     torch.distributed), plhip_comm_init_rccl on every rank with its device bound (plhip_init).  The exchange is
 *     ncclGroupStart; ncclSend / ncclRecv to and from every peer; ncclGroupEnd on the library's stream: xGMI is point to
 *     point, so these n - 1 transfers ARE the all-to-all.  librccl.so is loaded at run time (PLHIP_RCCL_LIBRARY names
 *     another copy); an RCCL the process already holds is shared.
 *   - host transport: the host supplies blocking all-to-all functions over device buffers (a host with its own
 *     communicator; the one-GPU rehearsals of the tests).  The library drains its stream before it calls them. */
#define PLHIP_COMM_ID_BYTES 128
typedef struct plhip_comm plhip_comm_t;
/* peer p: send[p * bytes_per_peer, ...) -> the peer's recv[rank * bytes_per_peer, ...); 0 = success */
typedef int32_t (*plhip_alltoall_fn)(void *user, const void *send, void *recv, size_t bytes_per_peer);
/* variable split: byte counts per peer, buffers packed peer after peer */
typedef int32_t (*plhip_alltoallv_fn)(void *user, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes);
int32_t plhip_comm_unique_id(uint8_t id[PLHIP_COMM_ID_BYTES]);
int32_t plhip_comm_init_rccl(const uint8_t id[PLHIP_COMM_ID_BYTES], int32_t world, int32_t rank, plhip_comm_t **comm);
int32_t plhip_comm_init_host(int32_t world, int32_t rank, plhip_alltoall_fn all_to_all, plhip_alltoallv_fn all_to_all_v, void *user,
                             plhip_comm_t **comm);
void plhip_comm_free(plhip_comm_t *comm);
/* The communicator's size and this rank as the TRANSPORT reports them (RCCL: ncclCommCount / ncclCommUserRank; host transport:
 * what plhip_comm_init_host was given); *transport = 1 for RCCL, 0 for host callbacks.  bench.py's "n_gpus" is this world.
 * (The reference's counterpart is the thread count of its POOL, crates/polars-core/src/lib.rs:49-62.) */
int32_t plhip_comm_info(plhip_comm_t *comm, int32_t *world, int32_t *rank, int32_t *transport);

/* plhip_group_by_agg_opts over the rows of ALL ranks of `comm`: every rank passes its shard (same schema on every
 * rank: dtypes and which columns carry a validity bitmap) and gets its share of the groups, disjoint in the keys between
 * ranks; the shares together are the result.  One call = the whole of PartitionGroupByExec across GPUs:
 *   local pass with mergeable partials (PLHIP_FLAG_PARTIAL) -> records packed by owner = vec_hash(keys) % world
 *   -> exchange -> unpack + regroup (sum of sums and counts, min of mins, max of maxs) -> finalize.
 * With opts->n_groups_hint (a bound on the groups of the whole job, the same on every rank) the buckets travel in
 * fixed-capacity slots: ONE equal-split exchange, no counts round trip; a bucket that does not fit says so in its slot
 * header, every rank sees it and all fall back to the exact protocol (counts, then a variable-size exchange).
 * PLHIP_FLAG_EXACT_EXCHANGE forces the exact protocol.  flags: PLHIP_FLAG_OUTPUT_DEVICE, PLHIP_FLAG_EXACT_EXCHANGE,
 * PLHIP_FLAG_KEYS_DENSE.
 * With PLHIP_FLAG_KEYS_DENSE (one u32 / i32 key whose values are codes < opts->n_groups_hint on every rank: a Categorical
 * with a global rev-map, perfect.rs:190-227) the owners are code RANGES -- rank r owns [r * per, (r + 1) * per), the null key
 * sits behind the last code -- and nothing is packed, hashed or regrouped: every rank adds up its dense partial arrays, sends
 * each owner its range of them raw (ONE equal-split exchange of hint * state bytes, no keys on the wire; a reduce-scatter made
 * of point-to-point sends), the owner adds the `world` copies up and finalizes.  The host waits for the device ONCE per call
 * (RCCL transport).  A rank that cannot keep the promise (a code >= hint, misaligned buffers) flags its slot headers; every
 * rank sees the flag and all fall back to the hashed protocols above together.  The sums of a group are added in a fixed
 * order (slices of a rank, then ranks 0 .. world - 1): results do not change from run to run.
 * Collective: every rank of the communicator must call it with the same query.  A non-zero return on any rank is FATAL for
 * the communicator: the library has no collective error path, the peers of a rank that bailed out between the local pass and
 * the exchange stay blocked in their receive; tear the job down (plhip_comm_free, a new communicator) as after a failed
 * RCCL call. */
#define PLHIP_FLAG_EXACT_EXCHANGE 32u
int32_t plhip_group_by_agg_sharded(plhip_comm_t *comm, const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                                   int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                   const plhip_options_t *opts, plhip_result_t **out);

/* What the last plhip_group_by_agg_sharded call on this thread did (BASELINE config 4 asks for the shuffle's bytes and
 * time beside the compute): host wall-clock per phase, the kernels of the local pass (the pass over the rank's rows,
 * which is what a roofline is quoted on), bytes this rank sent. */
typedef struct plhip_sharded_stats {
  float partial_ms, pack_ms, exchange_ms, merge_ms, finalize_ms; /* host wall clock per phase */
  float partial_kernel_ms, partial_dominant_ms;                  /* device time of the local pass's kernels */
  float partial_breakdown_ms[3];                                  /* as plhip_last_kernel_breakdown */
  int64_t shuffle_bytes;                                          /* bytes this rank handed to the transport */
  int64_t partial_groups;                                         /* groups of the local pass (-1 under protocol 3: never counted) */
  int32_t protocol;                                               /* 1: slots (one exchange); 2: exact (counts + variable); 3: dense code ranges (one exchange) */
  int32_t host_syncs;                                             /* times the host waited for the device in the call */
  char partial_strategy[16];
  char partial_dominant_kernel[32];
} plhip_sharded_stats_t;
int32_t plhip_sharded_last_stats(plhip_sharded_stats_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* PLHIP_H */
