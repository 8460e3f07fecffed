// kernels.hpp — host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"

// Table flags (u32[16] in HBM): [0] overflow, [1] sentinel-key slot used, [2] null-key slot used,
// [3] number of groups (dense-area cursor of the LDS pass, then continued by the compaction kernel),
// [4] rows that overflowed an LDS table.
#define FLAG_OVERFLOW 0
#define FLAG_SENTINEL 1
#define FLAG_NULL 2
#define FLAG_NGROUPS 3
#define FLAG_LDS_SPILL 4
#define FLAG_DONE 5 // workgroups of the last kernel that have finished (fast path: completion signal)
#define FLAG_F1 6   // k_estimate: sampled keys seen exactly once (signed count)

struct TableRef {
  uint8_t *base;   // keys[slots] u64, then the accumulator arrays (AccSpec::offset)
  uint64_t cap;    // hashed part: power of two; slots cap / cap+1 are the reserved sentinel / null slots;
                   // slots from cap+2 on are the dense result area of the partitioned path
  uint32_t *flags; // u32[16]
};

size_t table_bytes(const Plan &p, uint64_t n_slots); // n_slots = cap + 2 + dense capacity
void plan_layout(Plan &p, uint64_t n_slots);         // sets AccSpec::offset for a table of n_slots slots

struct PartOut {
  uint8_t *keys;       // packed keys, key_bytes each
  uint32_t *rowidx;    // original row ids or nullptr
  uint8_t *vals[PLHIP_MAX_VALS];
  uint8_t *valid[PLHIP_MAX_VALS]; // byte per row, or nullptr when the source column has no validity
  uint32_t key_bytes;
  uint32_t pad;
};


#define PART_SCATTER_R 8 // rows per thread of the partition scatter: its tile is 8192 rows
size_t lds2_table_bytes(const Plan &p, uint32_t lbits, int key_bytes);
int plhip_num_cus();
void launch_lds_agg2(const Plan &p, TableRef t, const WorkItem *items, const uint32_t *n_items_dev, uint32_t max_items,
                     uint32_t *next_item_dev, uint32_t lbits, int key_bytes, uint64_t dense_cap, uint64_t *slots,
                     hipStream_t s);
void launch_part_hist(const Plan &p, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles, uint32_t *counts, hipStream_t s);
void launch_part_scan(uint32_t *counts, uint32_t np, uint32_t n_tiles, uint32_t *totals, hipStream_t s);
void launch_part_plan(const uint32_t *totals, uint32_t np, uint32_t *part_start, WorkItem *items, uint32_t *n_items,
                      uint32_t *next_item, uint32_t split_rows, hipStream_t s, uint32_t align = 1);
void launch_part_scatter(const Plan &p, const PartOut &o, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles,
                         const uint32_t *offs, const uint32_t *part_start, hipStream_t s);

void launch_table_init(const Plan &p, TableRef t, hipStream_t s);
void launch_agg_global(const Plan &p, TableRef t, hipStream_t s);
// appends the occupied slots of the hashed part (reserved slots included) to `slots`, continuing at
// flags[FLAG_NGROUPS]; afterwards flags[FLAG_NGROUPS] = G.
void launch_compact(const Plan &p, TableRef t, uint64_t *slots, hipStream_t s);
// sort keys for the optional ordering step: FIRST_ROW accumulator (u32 -> u64) or the packed key
void launch_gather_sort_keys(const Plan &p, TableRef t, const uint64_t *slots, uint64_t n, int by_first_acc,
                             uint64_t *out_keys, hipStream_t s);
// `o` and `kg` are DEVICE pointers (the plans together exceed the 4 KB kernel-argument segment)
void launch_finalize(const Plan &p, TableRef t, const OutPlan *o, const KeyGather *kg, const uint64_t *slots,
                     uint64_t n_groups, hipStream_t s);
// dense ids: rank[slot] = position of the slot in the slot list; ids[row] = rank[slot of row's key]
void launch_rank_scatter(const uint64_t *slots, uint64_t n, uint32_t *rank, hipStream_t s);
void launch_encode_rows(const Plan &p, TableRef t, const uint32_t *rank, uint32_t *ids, hipStream_t s);
void launch_sample_keys(const Plan &p, TableRef t, int64_t n_sample, int64_t stride, hipStream_t s);
size_t key_minmax_scratch_bytes(int n_keys);
// minmax[2k] = ~min, [2k + 1] = max; host_word: device alias of a pinned host word that receives min << 32 | max of key 0 (k_key_minmax_final)
void launch_key_minmax(const Plan &p, uint64_t *minmax, uint64_t *partials, hipStream_t s, uint64_t *host_word = nullptr);
void launch_vec_hash(const Plan &p, uint64_t null_hash, uint64_t true_hash, uint64_t false_hash, uint64_t *out, hipStream_t s);
// rechunk: n validity bits from bit src_off of `src` (nullptr = all valid) to bit dst_off of the zeroed bitmap `dst`
// Boolean columns at the boundary: bitmap -> one u8 / u32 (0 or 1) per row, and back (row i of `dst` = src[i] != 0; dst holds whole 64-bit words)
void launch_unpack_bool(const uint8_t *bits, int64_t bit_off, int64_t n, void *out, uint32_t width, hipStream_t s);
void launch_pack_bool(const void *src, uint32_t width, int64_t n, uint64_t *dst, hipStream_t s);
void launch_copy_bits(uint32_t *dst, int64_t dst_off, const uint8_t *src, int64_t src_off, int64_t n, hipStream_t s);
hipError_t sort_pairs_u64(uint64_t *keys, uint64_t *vals, uint64_t n, int end_bit, hipStream_t s);

// two-step API
hipError_t sort_pairs_u32(uint32_t *keys, uint32_t *vals, uint64_t n, int end_bit, hipStream_t s); // stable
// on caller-supplied buffers (no allocation, no copy back, no host wait); tmp: sort_pairs_tmp_bytes(n) bytes
size_t sort_pairs_tmp_bytes(uint64_t n);
hipError_t sort_pairs_u32_ws(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int end_bit, void *tmp,
                             size_t tmp_bytes, hipStream_t s); // stable
hipError_t sort_pairs_u64_ws(const uint64_t *kin, uint64_t *kout, const uint64_t *vin, uint64_t *vout, uint64_t n, int end_bit, void *tmp,
                             size_t tmp_bytes, hipStream_t s);
void launch_iota_u32(uint32_t *out, uint64_t n, hipStream_t s);
void launch_csr(const uint32_t *ids_sorted, const uint32_t *indices, uint64_t n, uint64_t G, uint32_t *offsets, uint32_t *first, hipStream_t s);
// `o.outs[0]` is the one output column; at most 4 accumulators; o.outs[0].validity must be zeroed
// long_scratch (agg_long_scratch_bytes(n) bytes, or nullptr): groups of more than AGL_CHUNK rows are cut into chunks, one workgroup each
#define AGL_CHUNK 4096u
size_t agg_long_scratch_bytes(uint64_t n);
void launch_agg_groups(const Plan &p, const OutPlan &o, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, bool wave,
                       void *long_scratch, hipStream_t s);

// first / last / var / std over CSR groups
#define AGG_EXT_FIRST 5
#define AGG_EXT_LAST 6
#define AGG_EXT_VAR 7
#define AGG_EXT_STD 8
struct AggExtArgs {
  const uint8_t *values;
  const uint8_t *validity; // LSB-first bitmap or nullptr
  int64_t boff;
  uint32_t dtype, width, kind, ddof, out_dtype, pad;
  uint8_t *out;
  uint8_t *out_validity; // zeroed bitmap
};
#define AGG_EXT_MEDIAN 9
#define AGG_EXT_N_UNIQUE 10
#define AGG_EXT_QUANTILE 11 // plhip_agg_groups_quantile only
hipError_t median_temp_bytes(uint64_t n, uint64_t G, size_t *bytes);
hipError_t launch_n_unique(const AggExtArgs &a, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, uint64_t *codes_a,
                           uint64_t *codes_b, uint32_t *gid_a, uint32_t *gid_b, uint32_t *marks, void *temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_quantile(const AggExtArgs &a, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, uint64_t *codes_a,
                           uint64_t *codes_b, uint32_t *gid_a, uint32_t *gid_b, void *temp, size_t temp_bytes, double q, int interp, hipStream_t s);
void launch_agg_groups_ext(const AggExtArgs &a, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, bool wave, void *long_scratch,
                           hipStream_t s);

// sort strategy (sort_path.hip)
hipError_t sort_path_temp_bytes(uint64_t n, int end_bit, size_t *bytes);
hipError_t launch_sort_groups(const Plan &p, uint64_t *keys_a, uint64_t *keys_b, uint32_t *rows_a, uint32_t *rows_b, uint32_t *flags,
                              uint32_t *gid, uint32_t *offsets, void *temp, size_t temp_bytes, int end_bit, uint32_t *dflags, hipStream_t s);
hipError_t launch_sorted_key_groups(const KeySpec &key, uint64_t n, uint32_t *rows, uint32_t *flags, uint32_t *gid, uint32_t *offsets, void *temp,
                                    size_t temp_bytes, uint32_t *dflags, hipStream_t s);
// gmap (optional, G u32): output row of every run (stable order)
void launch_fold_sorted(const Plan &p, const OutPlan &o, const KeyGather *kg_dev, const uint64_t *skeys, const uint32_t *rows,
                        const uint32_t *offsets, uint64_t G, uint32_t *long_list, uint32_t *long_count, hipStream_t s,
                        const uint32_t *gmap = nullptr);
hipError_t stable_order_temp_bytes(uint64_t G, size_t *bytes);
hipError_t launch_stable_order(const uint32_t *rows, const uint32_t *offsets, uint64_t G, uint32_t *buf, uint32_t *gmap, void *temp, size_t temp_bytes,
                               hipStream_t s);

// exchange step of the row-sharded group-by (exchange.hip): frame <-> row records
#define EX_MAX_PARTS 1024 // owner ranks
#define EX_MAX_SLOTS 64   // ... of the slotted layout
struct RecPlan {
  int64_t n_rows;
  int32_t n_cols, n_keys; // the first n_keys columns are the keys the owner is hashed from
  uint32_t n_parts, rec_bytes;
  const uint8_t *values[PLHIP_MAX_OUTS]; // column buffers (written by the unpack kernel)
  const uint8_t *validity[PLHIP_MAX_OUTS]; // LSB-first bitmaps or nullptr
  int64_t boff[PLHIP_MAX_OUTS];
  uint32_t dtype[PLHIP_MAX_OUTS], width[PLHIP_MAX_OUTS];
  uint32_t off[PLHIP_MAX_OUTS];      // field offset inside a record: value bytes, then one validity byte if has_valid
  uint8_t has_valid[PLHIP_MAX_OUTS];
  // slotted layout (fixed-capacity buckets, no counts exchange): bucket b = EX_SLOT_HDR header bytes
  // {u64 rows, u64 "this sender overflowed some bucket"} + slot_rows records, at b * slot_stride.  0 = compact layout.
  uint32_t slot_rows, pad2;
  uint64_t slot_stride;
  uint32_t slot_first[EX_MAX_SLOTS + 1]; // unpack: first output row of every slot (prefix sums of the rows per slot)
};
#define EX_SLOT_HDR 64
void launch_owner_count(const RecPlan &r, uint64_t null_hash, uint16_t *owner, uint32_t *counts, hipStream_t s);
void launch_pack_records(const RecPlan &r, const uint16_t *owner, const uint32_t *counts, uint32_t *cursor, uint8_t *records, hipStream_t s);
void launch_unpack_records(const RecPlan &r, const uint8_t *records, hipStream_t s);
void launch_slot_headers(const uint8_t *slots, uint64_t stride, uint32_t n_slots, uint64_t *out, hipStream_t s); // n_slots <= 64

// AggregationExpr::finalize on a regrouped partial frame (exchange.hip)
#define FIN_COPY 0  // keys, min, max: value and validity bit as they are
#define FIN_VALUE 1 // sum / count / mean through finalize_value (device.hpp)
struct FinOut {
  const uint8_t *a;       // partial column (FIN_VALUE: 8-byte sums / counts)
  const uint8_t *a_valid; // its bitmap (FIN_COPY) or nullptr
  const uint8_t *cnt;     // mean: the u64 valid-count column
  uint32_t mode, width;   // FIN_COPY: element width
  OutSpec spec;           // FIN_VALUE: kind / dtype / values; both: values, validity (may be nullptr)
};
struct FinPlan {
  int64_t n_rows;
  int32_t n_outs, pad;
  FinOut outs[PLHIP_MAX_OUTS];
};
void launch_finalize_partial(const FinPlan &f, hipStream_t s);
