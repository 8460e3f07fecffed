/*
 * groupby_oracle.h — CPU restatement (plain C) of the reference's hash group-by aggregation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (polars_amd/, libplhip.so) may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
 * checker / the timed CPU baseline.
 *
 * Reference: backend-devloper/polars @ Polars 0.33.2 (Rust; cannot be compiled or imported in this
 * image: no rustc/cargo, no polars wheel).  Each function cites the reference file:line it follows.
 * Pinned by the reference's own known-answer tests transcribed in tests/golden/reference_known_answers.json
 * (see tests/test_oracle_golden.py).
 */
#ifndef GROUPBY_ORACLE_H
#define GROUPBY_ORACLE_H

#include <stdint.h>

#include "../include/plhip.h" /* plhip_column_t, plhip_dtype_t, plhip_agg_kind_t (descriptors only) */

#ifdef __cplusplus
extern "C" {
#endif

/* GroupsProxy::Idx (crates/polars-core/src/frame/group_by/proxy.rs:16-20) flattened to CSR. */
typedef struct orc_groups {
  uint32_t n_groups;
  uint32_t n_rows;
  uint32_t *first;   /* [n_groups]            first row of each group */
  uint32_t *offsets; /* [n_groups + 1]        CSR offsets into indices */
  uint32_t *indices; /* [n_rows]              row indices, ascending inside a group */
} orc_groups_t;

/* folded_multiply (crates/polars-core/src/hashing/vector_hasher.rs:37-40). */
uint64_t orc_folded_multiply(uint64_t s, uint64_t by);

/* series_to_hashes (vector_hasher.rs:467-483): vec_hash on keys[0] (:78-101), vec_hash_combine on the
 * rest (:103-139).  `null_h` = get_null_hash_value (:42-51), supplied by the caller (seed dependent). */
int orc_vec_hash(const plhip_column_t *keys, int n_keys, uint64_t null_h, uint64_t *out);

/* DataFrame::group_by_with_series (crates/polars-core/src/frame/group_by/mod.rs:53-97).
 * n_threads <= 1: group_by (hashing.rs:143-177) / group_by_multiple_keys (:470-523).
 * n_threads  > 1: group_by_threaded_slice / _iter (:182-316) / group_by_threaded_multiple_keys_flat
 *                 (:380-468) with P = largest power of two <= n_threads partitions
 *                 (crates/polars-core/src/utils/mod.rs:32-46), every thread scanning all rows and
 *                 keeping this_partition(x, t, P) (hashing/partition.rs:134-138).
 * sorted != 0: groups ordered by first (hashing.rs:32-79); else partition order, insertion order inside. */
int orc_group_tuples(const plhip_column_t *keys, int n_keys, int sorted, int n_threads, orc_groups_t *out);
void orc_groups_free(orc_groups_t *g);

/* Output dtype of `kind` over a column of `dtype` (aggregations/mod.rs, series/implementations/mod.rs:167-174). */
int orc_agg_out_dtype(int dtype, int kind);

/* Series::agg_{sum,mean,min,max} on GroupsProxy::Idx + group_count
 * (aggregations/mod.rs:459-779, 941-1016; polars-arrow take_agg/mod.rs:17-109; proxy.rs:452-466).
 * out_values: n_groups elements of orc_agg_out_dtype(); out_valid: n_groups bytes (1 = valid).
 * Sequential left fold in ascending row order per group, exactly as the reference's Idx path. */
int orc_agg(const plhip_column_t *values, int kind, const orc_groups_t *g, void *out_values,
            uint8_t *out_valid, int n_threads);

/* Partitioned order (PartitionGroupByExec::run_partitions, crates/polars-lazy/src/physical_plan/
 * executors/group_by_partitioned.rs:66-113,320-360; AggregationExpr::evaluate_partitioned/finalize,
 * expressions/aggregation.rs:283-398): rows split into n_parts contiguous parts, per-part partial fold,
 * then partials combined in part order.  Same signature as orc_agg. */
int orc_agg_partitioned(const plhip_column_t *values, int kind, const orc_groups_t *g, int n_parts,
                        void *out_values, uint8_t *out_valid);

/* Gather key output (GroupBy::keys_sliced, group_by/mod.rs:262-296): out[g] = key[first[g]]. */
/* Path B (streaming sink, polars-pipe PrimitiveGroupbySink): see groupby_oracle.c */
int orc_sink_group_by(const plhip_column_t *key, const plhip_column_t *values, int n_values, const int *agg_col, const int *agg_kind,
                      int n_aggs, int n_threads, int64_t morsel, int64_t max_groups, uint64_t *out_keys, uint8_t *out_key_valid,
                      double *out_vals, uint8_t *out_valid, int64_t *n_groups);
/* IntoGroupsProxy for Utf8Chunked / BinaryChunked (into_groups.rs:232-290): every string is hashed once (nulls get one
 * constant, null_h), the (hash, bytes) pairs are grouped with BytesHash equality (vector_hasher.rs:422-446: hash first, then
 * the bytes; two nulls are equal); groups are numbered by first occurrence as `group_by(.., sorted = true)` leaves them
 * (hashing.rs:32-79).  codes[i] = group of row i, first[g] = first row of group g (capacity n).  Returns the group count. */
int64_t orc_utf8_group_codes(const int64_t *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset, int64_t n,
                             uint32_t *codes, uint32_t *first);

/* hash_join_tuples_inner (polars-ops/src/frame/join/hash_join/single_keys_inner.rs:38-140) and hash_join_tuples_left
 * (single_keys_left.rs:109-) for one integer key column pair of the same dtype: build_tables over the shorter relation
 * (inner; det_hash_prone_order!, hash_join/mod.rs:42-58) or over the right one (left join), key -> rows in ascending
 * order; the other relation probes in row order.  Keys are Option<T>: null == null.  how: 0 inner, 1 left (no match:
 * right index 0xFFFFFFFF), 2 semi / 3 anti (single_keys_semi_anti.rs: left rows whose key is / is not among the right keys; out_right unused),
 * 4 outer (single_keys_outer.rs: as inner plus (row, None) for unmatched probe rows, then (None, row) for the build rows nobody
 * matched -- in build-row order here, hash-map order in the reference).  Pass out_left = NULL to get the pair count only.  Returns the number of pairs, -1 on error. */
int64_t orc_hash_join(const plhip_column_t *left, const plhip_column_t *right, int how, uint32_t *out_left, uint32_t *out_right);

/* _inner_join_multiple_keys / _left_join_multiple_keys (polars-ops/src/frame/join/hash_join/multiple_keys.rs:173-,251-): rows
 * hashed over all key columns, equal when every column is equal (compare_df_rows2: floats by bit pattern after
 * _to_physical_and_bit_repr, null == null); build side, probe order and `how` as orc_hash_join. */
int64_t orc_hash_join_multi(const plhip_column_t *left, const plhip_column_t *right, int n_keys, int how, uint32_t *out_left,
                            uint32_t *out_right);

int orc_take_keys(const plhip_column_t *key, const orc_groups_t *g, void *out_values, uint8_t *out_valid);

#ifdef __cplusplus
}
#endif
#endif
