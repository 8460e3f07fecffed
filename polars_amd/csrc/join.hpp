// join.hpp — hash join on one integer key (join_path.hip): the row-index pairs of an inner or left join.
//
// Reference: crates/polars-ops/src/frame/join/hash_join/ — build_tables (single_keys.rs:3-57: key -> Vec<IdxSize> of the
// SHORTER relation, rows in ascending order), probe_inner (single_keys_inner.rs:8-36: the longer relation's rows in order,
// every match appended in the table's order, pairs swapped back so that left stays left), hash_join_tuples_left
// (single_keys_left.rs:109-: the left relation always probes, a row without a match yields (row, None)).  Keys are
// Option<T>: two nulls are equal and join (crates/polars/tests/it/core/joins.rs:574-586, 3 x 5 rows -> 9 pairs).
//
// On the device the build side's key -> row list map is the CSR the group-by seam already builds (plhip_group_tuples:
// first / offsets / indices, lists ascending), plus an open-addressing table in HBM from key to group:
//   k_join_build  the G distinct keys (taken at the groups' first rows) claim slots -- they are distinct, so a claim flag per slot is
//                 all the synchronisation there is; the null group's index goes into a scalar
//   k_join_count  every probe row finds its group (or none) and writes how many pairs it yields
//   (scan)        exclusive prefix sums: where each probe row's pairs start -- the reference's order, reproduced exactly
//   k_join_fill   the pairs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct JoinKey {
  const uint8_t *values;   // element 0 of the logical slice
  const uint8_t *validity; // LSB-first bitmap or nullptr
  int64_t boff;
  int64_t n;
  uint32_t width; // bytes per key: 1, 2, 4, 8 (raw bits, zero-extended: both sides have the same dtype)
  uint32_t pad;
};

struct JoinSlot { // 16 bytes: one load per probe step
  uint64_t key;
  uint32_t val;  // group index
  uint32_t used; // claim flag (0: empty)
};

struct JoinTable {
  JoinSlot *slots;  // cap entries, zeroed by the caller
  uint32_t *g_null; // one word preset to JOIN_NONE: the group of the null key
  uint64_t mask;    // cap - 1
};

#define JOIN_NONE 0xFFFFFFFFu

void launch_join_build(const JoinKey &build, const uint32_t *first, uint32_t G, const JoinTable &t, hipStream_t s);
// gid[i] = the probe row's group or JOIN_NONE; cnt[i] = rows it yields; cnt[n] = 0.  mode 0: one per match (inner); 1: at least one
// (left); 2: one if matched (semi); 3: one if not matched (anti; gid is then reported as JOIN_NONE so that the fill writes the row alone)
// mode 4 (outer): as 1, and matched[g] = 1 for every group a probe row found (`matched`: G words, zeroed)
void launch_join_count(const JoinKey &probe, const JoinTable &t, const uint32_t *offsets, int mode, uint32_t *gid, uint64_t *cnt, uint32_t *matched,
                       hipStream_t s);
// outer join's tail: the rows of the build groups no probe row matched.  cnt[g] = rows of group g if unmatched, cnt[G] = 0;
// after the scan the fill writes (JOIN_NONE, row) pairs behind the probe rows' pairs
void launch_join_tail_count(uint32_t G, const uint32_t *matched, const uint32_t *offsets, uint64_t *cnt, hipStream_t s);
void launch_join_tail_fill(uint32_t G, const uint32_t *matched, const uint64_t *pos, const uint32_t *offsets, const uint32_t *indices,
                           uint32_t *out_probe, uint32_t *out_build, hipStream_t s);
size_t join_scan_temp_bytes(int64_t n);
hipError_t join_scan(void *temp, size_t temp_bytes, const uint64_t *cnt, uint64_t *pos, int64_t n, hipStream_t s); // exclusive, n + 1 entries
// out_build == nullptr (semi / anti): the probe rows with a non-zero count, each once
void launch_join_fill(int64_t n, const uint32_t *gid, const uint64_t *pos, const uint32_t *offsets, const uint32_t *indices, uint32_t *out_probe,
                      uint32_t *out_build, hipStream_t s);
// CSR groups (first / offsets / indices of plhip_group_tuples) -> the group of every row: gid[indices[j]] = g for offsets[g] <= j < offsets[g + 1]
void launch_csr_to_gid(const uint32_t *offsets, const uint32_t *indices, uint32_t G, int64_t n, uint32_t *gid, hipStream_t s);
// out[i] = vals[idx[i]] (elements of `width` bytes), null (bit 0 in out_valid, value 0) where idx[i] == JOIN_NONE or the source row is
// null; out_valid: (n + 63) / 64 whole words, or nullptr when neither can happen
void launch_take(const uint8_t *vals, int width, const uint8_t *validity, int64_t boff, const uint32_t *idx, int64_t n, uint8_t *out, uint64_t *out_valid,
                 hipStream_t s);
