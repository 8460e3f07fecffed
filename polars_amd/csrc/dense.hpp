// dense.hpp — the dense-code path (dense_path.hip): a Categorical key's physical codes index the tables directly.
//
// Reference: group_tuples_perfect (crates/polars-core/src/frame/group_by/perfect.rs:24-185) — the codes of a
// Categorical with a local rev-map are used as group indices: no hashing, no probing, the null key in the last
// slot; threads own disjoint code ranges and every thread scans all rows.  Here the "threads" are workgroups:
//
//   k_dense_fold  : the query is cut into PARTS (one per value column, counters attached to the cheapest one).
//                   A part's dense accumulator arrays (G + 1 groups x its state bytes, 32-bit where the reference
//                   wraps in 32 bits) are cut into R code RANGES that fit one CU's LDS; a workgroup owns
//                   (part, row slice, range): it streams the slice's key + value column with 16-byte loads, keeps
//                   the rows whose code falls into its range and folds them with LDS atomics.  The R workgroups of
//                   a slice read the same bytes at the same time and are placed on one XCD, so HBM delivers a
//                   slice once and the XCD's L2 serves the other R - 1 readers.  No row ever leaves the CU: there
//                   is no exchange pass, no sorted copy, no global atomic.
//   k_dense_merge : sums the slices' partial arrays per group (fixed order), drops the codes that never occurred,
//                   finalises (mean = sum / count ...) and writes the output columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"

#define DENSE_MAX_PARTS 4
#define DENSE_MAX_PART_ACCS 8
#define DENSE_MAX_RANGES 8
#define DENSE_BLOCK 1024
#define DENSE_BATCH (DENSE_BLOCK * 8) // rows one workgroup takes per trip of its main loop

enum DenseKind : uint32_t {
  DENSE_KIND_RT = 0,        // any part: accumulator list read at run time
  DENSE_KIND_CNT_SUM32 = 1, // 4-byte integer column: [CNT_ALL, ADD_U64 kept in 32 bits]      (q1, q3's v1 part)
  DENSE_KIND_SUMF64 = 2,    // f64 column: [ADD_F64]                                          (q3's v3 part)
  DENSE_KIND_SUM32 = 3,     // 4-byte integer column: [ADD_U64 kept in 32 bits]
  DENSE_KIND_CNTV_SUMF64 = 4, // nullable f64 column: [CNT_VALID, ADD_F64]                     (mean of a nullable column)
  DENSE_KIND_CNTV = 5,      // a nullable column's valid count alone: reads the key and the column's validity bits, never its values.
                            // The host splits [CNT_VALID, ADD_F64] into this and SUMF64 when 12 bytes of state per group would cost
                            // more ranges than 8 (db-benchmark's NA variant: 8 ranges -> 5, and the counting part reads 4 bytes a row)
};

struct DensePart {
  uint8_t acc[DENSE_MAX_PART_ACCS]; // index into Plan::accs
  uint8_t lw[DENSE_MAX_PART_ACCS];  // bytes per group in LDS and in the partial arrays: 4 or 8
  uint32_t loff[DENSE_MAX_PART_ACCS]; // byte offset of the accumulator's array inside a range table (gpr entries each)
  uint32_t doff[DENSE_MAX_PART_ACCS]; // ... inside a slice's partial arrays (R * gpr entries each)
  int32_t src;       // value column, -1: counters only
  uint32_t n_accs;
  uint32_t n_slices; // row slices: one group slot (R workgroups on one XCD lane) each
  uint32_t slot0;    // the part's first group slot
  uint32_t pad0;     // 1: the value column has a validity bitmap (straight-line kinds)
  uint32_t kind;     // DenseKind
  uint64_t dump_off;     // the part's partial arrays inside DenseArgs::dump
  uint64_t slice_stride; // bytes per slice there
  int64_t rows_per_slice; // multiple of DENSE_BATCH
};

struct DenseArgs {
  uint32_t n_parts;
  uint32_t R;       // code ranges per part
  uint32_t gpr;     // groups per range (multiple of 4)
  uint32_t n_slots; // hint + 1: slot `hint` is the null key's
  uint32_t hint;    // codes are < hint
  uint32_t xcds;    // workgroup w -> XCD lane w % xcds, position w / xcds there; group slot = (position / R) * xcds + lane
  uint32_t cnt_acc; // Plan::accs index of ACC_CNT_ALL (a group exists iff its count is not 0)
  uint32_t key_bias; // code = the key's raw 32 bits - key_bias (0 for Categorical codes; the column's minimum for a plain integer key
                     // whose measured value range is compact)
  uint32_t slot_base; // k_dense_merge over received owner slots: entry s is global slot slot_base + s (0 otherwise)
  uint32_t n_hdr;     // ... and the n_hdr slots (part[0].slice_stride apart in `dump`) start with a header whose first word flags a sender
                      // that could not keep the dense promise (0: `dump` holds k_dense_fold's partial arrays, no headers)
  // Partition mode (more groups than DENSE_MAX_RANGES ranges hold): the rows have been scattered by code range into R = np
  // partition-contiguous copies first (k_dense_part_hist + the tile sort's scatter variant), and a workgroup owns
  // (part, partition p, slice of the partition's rows): range p of the codes in ONE table, every row read once per part.
  uint32_t pmode;     // 1: partition mode
  uint32_t S;         // slices per partition (= DensePart::n_slices); a slice's partial arrays hold gpr entries
  const uint32_t *part_start; // [R + 1] first row of every partition in the partitioned copies (multiples of the tile)
  const uint32_t *part_rows;  // [R] its rows
  uint8_t *dump;
  uint32_t *flags;
  DensePart part[DENSE_MAX_PARTS];
};

// The sharded dense path (plhip_group_by_agg_sharded with PLHIP_FLAG_KEYS_DENSE): rank r owns the code range
// [r * per, (r + 1) * per).  k_dense_scatter adds up the slices' partial arrays like k_dense_merge and writes every owner's
// entries, still dense and raw, into that owner's slot of the send buffer: [header 256 B][accumulator arrays, `per` entries each].
// One equal-split all-to-all later every rank holds `world` such slots for its own range and k_dense_merge runs over them as if
// they were `world` row slices: a reduce-scatter made of point-to-point sends, no packing, no keys on the wire, no regroup.
#define DENSE_SLOT_HEADER 256u
struct DenseXchg {
  uint8_t *send;
  uint64_t slot_bytes; // multiple of 256
  uint32_t per;        // slots per owner (multiple of 4)
  uint32_t world;
  uint32_t local_bad;  // this rank could not run its fold (alignment, size): flagged in its headers, every rank then falls back
  uint32_t pad;
  uint32_t xoff[DENSE_MAX_PARTS][DENSE_MAX_PART_ACCS]; // byte offset of an accumulator's array inside a slot
};
static_assert(sizeof(Plan) + sizeof(OutPlan) + sizeof(DenseArgs) + 32 <= 4096, "k_dense_merge's arguments must fit the 4 KB kernel-argument segment");

void launch_dense_fold(const Plan &p, const DenseArgs &a, uint32_t n_wgs, size_t lds_bytes, hipStream_t s);
void launch_dense_merge(const Plan &p, const DenseArgs &a, const OutPlan &o, uint64_t out_cap, uint32_t *host_sig, uint32_t seq, hipStream_t s);
void launch_dense_scatter(const Plan &p, const DenseArgs &a, const DenseXchg &x, hipStream_t s);
// counts[p * n_tiles + tile] = rows of tile `tile` (TS_TILE rows each) whose code (key - bias) / gpr is p; a code >= n_codes raises
// flags[FLAG_OVERFLOW] (and is counted in the last partition, as the scatter will place it)
void launch_dense_part_hist(const Plan &p, uint32_t bias, uint32_t gpr, uint32_t recip, uint32_t np, uint32_t np_pow2, uint32_t n_codes,
                            uint32_t n_tiles, uint32_t *counts, uint32_t *flags, hipStream_t s);
