// fast.hpp — the LDS-only aggregation pipeline (fast_path.hip): no global atomics on the row path.
//
//   few groups   : k_fold_rows   (row slices -> per-workgroup LDS tables -> compact dumps)
//                  k_merge_emit  (dumps -> merged LDS tables -> output columns)
//   many groups  : k_tile_sort   (each 8192-row tile is sorted by hash partition in LDS and written back
//                                 tile-contiguous; no histogram pre-pass, no global scatter, no atomics)
//                  k_fold_runs   (one partition group x tile segment per workgroup: gathers its runs from
//                                 every tile, LDS table, compact dump)
//                  k_merge_emit
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"
#include "plan.hpp"

#define TS_BLOCK 1024
#ifndef TS_R
#define TS_R 8
#endif
#define TS_TILE (TS_BLOCK * TS_R) // rows per sorted tile
#define TS_MAX_PBITS 9 // np + 1 histogram entries must not exceed TS_BLOCK (one per thread in the scan)

#define FP_MAX_MBITS 6
struct DumpHdr {
  uint32_t n_hash;   // entries with ordinary keys, grouped by merge bucket
  uint32_t has_sent; // then one entry for the key equal to the empty sentinel
  uint32_t has_null; // then one entry for the null key
  uint32_t pad;
  uint32_t boff[(1 << FP_MAX_MBITS) + 1]; // bucket b = entries [boff[b], boff[b + 1])
  uint32_t pad2[3];
};

// Compact copies of LDS tables: item i owns [base + i * item_stride, ...): keys[cap], then every
// accumulator's array[cap] (16-byte aligned each).
struct DumpRef {
  uint8_t *base;
  DumpHdr *hdr;
  uint64_t item_stride;
  uint32_t cap;
  uint32_t pad;
};

struct FoldRowsArgs {
  DumpRef D;
  uint32_t *flags;
  int64_t rows_per_item; // multiple of 4
  uint64_t kxor;
  uint32_t lbits;
  int32_t keymode; // 0: row_packed_key, 1: plain 4-byte column, 2: plain 8-byte column
  uint32_t mbits;  // the dump is grouped into 1 << mbits merge buckets
  uint32_t pad;
};

// Heavy hitters in the two-level pipeline: a leaf that holds a hot key has as many rows as the key has, and one
// workgroup would fold them all.  The level-2 tile sort adds up the long runs per leaf (hot_rows), k_hot_plan turns
// the leaves above a threshold into entries, and the leaf fold gives every such leaf `chunks` workgroups (chunk 0 = the
// leaf's own, the others from a block of extra workgroups behind the grid), each over a share of the leaf's tiles; they
// dump their tables, and the one that finishes last merges the dumps and writes the leaf's output rows.
#define HOT_MAX_LEAVES 256
#define HOT_MAX_CHUNKS 256
struct HotEntry { uint32_t leaf, chunks, first_extra, done; };
struct HotState {
  uint32_t n_hot, n_extra, pad[2];
  HotEntry e[HOT_MAX_LEAVES]; // ascending in leaf and in first_extra; chunk c of entry h dumps into item first_extra + h + c
};

struct FoldRunsArgs {
  DumpRef D;
  uint32_t *flags;
  const uint8_t *pkeys; // packed keys of the sorted tiles
  uint64_t out_cap;     // S == 1: the workgroup writes the output columns itself
  const uint32_t *offs; // [(np + 2)][n_tiles]: start of partition p inside tile t; row np + 1 = rows in the tile
  uint32_t lbits;
  uint32_t np;       // hash partitions of the tile sort (partition np = null keys)
  uint32_t w;        // partitions per group
  uint32_t n_pg;     // partition groups (the last one is the null partition when has_null_pg)
  uint32_t has_null_pg;
  uint32_t S;        // tile segments per group
  uint32_t n_tiles;
  uint32_t pbits; // log2(np): the hash bits the tile sort consumed
  uint32_t *host_sig; // pinned host memory (device address): the last workgroup stores seq << 32 | overflow << 31 | groups
  uint32_t seq;       // there, the word the host spins on (nullptr: no signal)
  uint32_t disjoint;  // two-level pipeline: the S segments are level-1 partitions (disjoint in the keys): every workgroup emits
  const uint32_t *seg_start; // ... and segment s covers the tiles [seg_start[s], seg_start[s + 1]) / TS_TILE (nullptr: equal split)
  HotState *hot;              // disjoint only, may be nullptr: the split leaves (the grid then has max_extra workgroups behind the leaves')
  const uint32_t *hot_of_leaf; // [n_pg * S]: entry index + 1, 0 = an ordinary leaf
  uint32_t max_extra;          // extra workgroups behind the leaves' (both hot modes)
  // single level (S == 1, one workgroup per partition, at most blockDim partitions): no plan kernel -- the workgroups plan
  // for themselves from self_hot[0 .. n_pg) = rows in long runs per partition (k_tile_sort), self_hot[n_pg .. 2 n_pg) = done
  // counters, self_hot[-2] = "some partition has long runs", self_hot[-1] = the queue of chunks, all zeroed by the host.  The
  // chunks >= 1 are taken from the queue by workgroups that have finished their own partition (max_extra bounds their number).  A partition with >= hot_min_rows such rows is split into ceil(rows / hot_chunk_rows)
  // chunks (<= HOT_MAX_CHUNKS, <= n_tiles); chunk 0 dumps into item max_extra + partition, chunk c >= 1 into item
  // first_extra + c - 1 (first_extra = exclusive prefix of the extra chunks over the partitions).
  uint32_t hot_min_rows;
  uint32_t *self_hot;
  uint32_t hot_chunk_rows, pad3;
  const uint8_t *skip; // may be nullptr: [n_pg * S] leaves with skip[leaf] != 0 were folded by k_leaf_sort
};

struct MergeArgs {
  DumpRef D;
  uint32_t *flags;
  uint64_t out_cap; // rows the output buffers can hold
  uint32_t lbits;
  uint32_t S;     // dumps per partition group
  uint32_t mbits; // merge workgroups per partition group = 1 << mbits (split by key hash)
  uint32_t pshift; // hash bits consumed by the tile sort (0 for row-slice dumps)
  uint32_t *host_sig; // as in FoldRunsArgs
  uint32_t seq;
  uint32_t pad;
};

// leaf_path.hip: the leaf kernel for the one-row-per-group regime (a wave sorts a leaf of <= 512 rows and folds its runs)
struct LeafSortArgs {
  const uint8_t *pkeys;      // packed keys of the level-2 sorted tiles (key_bytes each, < 2^55)
  const uint32_t *offs;      // as FoldRunsArgs::offs
  const uint32_t *seg_start; // first row of every level-1 partition (multiples of TS_TILE), [S + 1]
  uint32_t n_pg, S, n_tiles, n_leaves; // leaf = seg * n_pg + pg
  uint32_t *flags;           // [FLAG_NGROUPS] output-row cursor, [FLAG_OVERFLOW], [10] workgroups done, [11] leaves left over
  uint8_t *handled;          // [n_leaves]: 1 = folded here, 0 = left to k_fold_runs
  uint64_t out_cap;
  uint32_t *host_sig;        // the last workgroup: word 2 = leaves left over, words 0-1 = seq << 32 | overflow << 31 | groups
  uint32_t seq, pad;
};
void launch_leaf_sort(const Plan &p2, const LeafSortArgs &a, const OutPlan &o, int key_bytes, hipStream_t s);

int fast_shape_id(const Plan &p); // index of the pre-instantiated shape the plan matches, or -1 (run-time shape)
size_t fast_table_bytes(const Plan &p, uint32_t lbits, int key_bytes);  // LDS bytes of a fold / merge workgroup
uint64_t fast_dump_stride(const Plan &p, uint32_t cap, int key_bytes);
void launch_fold_rows(const Plan &p, const FoldRowsArgs &a, int key_bytes, bool quad, uint32_t n_items, uint32_t block, hipStream_t s);
// Rows that belong to tile t = [t * TS_TILE, ...).  Level 2 of the two-level pipeline: `lvl1` describes the level-1
// partitions the input consists of (tile-aligned starts, rows per partition) and the hash bits they consumed.
struct TileSortLevel1 {
  const uint32_t *part_start; // [np1 + 1], multiples of TS_TILE; nullptr: one flat input of P.n_rows rows
  const uint32_t *part_rows;  // [np1]
  uint32_t np1, pshift;       // pshift: hash bits consumed by level 1
  uint32_t *hot_rows;         // level 2, may be nullptr: [np1 << pbits] zeroed; runs of at least hot_run rows are added to their leaf's word
  uint32_t hot_run;
  uint32_t out_rows;          // scatter variant, hash partitions: rows the output columns hold (0: unknown -> the binary-search placement)
  uint32_t *hot_any;          // single level, may be nullptr: set to 1 when any long run was seen
  // scatter variant only, range partitions of dense codes (dense_path.hip, k_dense_part_hist has the same function): range_gpr != 0
  // -> partition = min((key - range_bias) / range_gpr, range_np - 1) instead of hash bits (range_recip = ceil(2^32 / range_gpr))
  uint32_t range_gpr, range_bias, range_recip, range_np;
};
// code / gpr for code < 2^31 without a division: the reciprocal estimate is off by at most one
__device__ __forceinline__ uint32_t range_part(uint32_t code, uint32_t gpr, uint32_t recip, uint32_t np) {
  uint32_t q = __umulhi(code, recip);
  if ((uint64_t)q * gpr > code) q--;
  else if ((uint64_t)(q + 1) * gpr <= code) q++;
  return q < np ? q : np - 1u; // (a code beyond the promised range: the histogram pass has raised the overflow flag)
}
// hot_rows[] (rows in long runs per leaf) -> H + hot_rows[] rewritten as entry index + 1 / 0.  A leaf is split when it has
// at least min_rows such rows, into ceil(rows / chunk_rows) chunks (at most HOT_MAX_CHUNKS, at most its tiles).
void launch_hot_plan(uint32_t *hot_rows, uint32_t n_leaves, uint32_t n_pg, const uint32_t *seg_start, uint32_t min_rows, uint32_t chunk_rows,
                     uint32_t max_extra, HotState *H, hipStream_t s);
void launch_tile_sort(const Plan &p, const PartOut &o, uint32_t pbits, uint32_t n_tiles, uint32_t *offs, int keymode, uint64_t kxor,
                      int key_bytes, hipStream_t s, const TileSortLevel1 *lvl1 = nullptr, bool scatter = false);
void launch_fold_runs(const Plan &p2, const FoldRunsArgs &a, const OutPlan &o, int key_bytes, uint32_t block, hipStream_t s);
void launch_merge_emit(const Plan &p, const MergeArgs &a, const OutPlan &o, int key_bytes, uint32_t n_pg, uint32_t block, hipStream_t s);
// distinct keys among `n_sample` rows taken every `stride` rows -> flags[FLAG_NGROUPS]
// `set`: ZEROED scratch of (1 << set_bits) + (1 << set_bits) / 16 u32: the fingerprints (at least 4x the sample), then
// two state bits per slot; flags[FLAG_F1] receives the number of sampled keys seen exactly once
void launch_estimate(const Plan &p, int64_t n_sample, int64_t stride, uint32_t *set, uint32_t set_bits, uint32_t *flags, hipStream_t s);
