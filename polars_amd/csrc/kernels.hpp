// kernels.hpp — host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"

// Table flags (u32[8] in HBM): [0] overflow, [1] sentinel-key slot used, [2] null-key slot used,
// [3] number of groups (written by the compaction kernel), [4] rows that overflowed an LDS table.
#define FLAG_OVERFLOW 0
#define FLAG_SENTINEL 1
#define FLAG_NULL 2
#define FLAG_NGROUPS 3
#define FLAG_LDS_SPILL 4

struct TableRef {
  uint8_t *base;   // keys[cap+2] u64, then the accumulator arrays (AccSpec::offset)
  uint64_t cap;    // power of two
  uint32_t *flags; // u32[8]
};

size_t table_bytes(const Plan &p, uint64_t cap); // also fills nothing; offsets come from plan_layout
void plan_layout(Plan &p, uint64_t cap);         // sets AccSpec::offset for a table of `cap` slots

void launch_table_init(const Plan &p, TableRef t, hipStream_t s);
void launch_agg_global(const Plan &p, TableRef t, hipStream_t s);
void launch_agg_lds(const Plan &p, TableRef t, uint32_t lds_cap, hipStream_t s);
// slots[has_null .. G) = occupied hashed/sentinel slots in arbitrary order, slots[0] = null slot if used;
// flags[FLAG_NGROUPS] = G.
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
void launch_key_minmax(const Plan &p, uint64_t *minmax /* [n_keys][2], pre-set to {~0, 0} */, hipStream_t s);
void launch_vec_hash(const Plan &p, uint64_t null_hash, uint64_t *out, hipStream_t s);
hipError_t sort_pairs_u64(uint64_t *keys, uint64_t *vals, uint64_t n, int end_bit, hipStream_t s);
