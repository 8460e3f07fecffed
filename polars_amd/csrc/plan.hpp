// plan.hpp — POD descriptions shared by the host planner (api.hip) and the gfx950 kernels (kernels.hip).
//
// A query is lowered to
//   * KeySpec[]  : how a row's key tuple becomes ONE packed u64 (range compression, nulls as code 0)
//   * ValSpec[]  : the value columns the aggregations read
//   * AccSpec[]  : the per-group accumulators (SoA arrays in the hash table) and the atomic op that
//                  folds a row (or a partial from an LDS table) into them
//   * OutSpec[]  : how accumulators are finalised into the reference's output columns
#pragma once
#include <stdint.h>

#define PLHIP_MAX_KEYS 16
#define PLHIP_MAX_VALS 16
#define PLHIP_MAX_ACCS 24
#define PLHIP_MAX_OUTS 32

// empty-slot sentinel of the key array; the (rare) real key equal to it lives in reserved slot CAP,
// the null key (single-key queries) in reserved slot CAP + 1.
#define PLHIP_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// vector_hasher.rs:14
#define PLHIP_MULTIPLE 6364136223846793005ull

enum AccOp : uint32_t {
  ACC_ADD_U64 = 0,   // integer sums (sign/zero-extended to 64 bits; i32/u32 outputs truncate = wrapping)
  ACC_ADD_F64 = 1,   // float sums and the f64 sum behind i64/u64 means
  ACC_MIN_I64 = 2,   // signed integer min (sign-extended)
  ACC_MAX_I64 = 3,
  ACC_MIN_U64 = 4,   // unsigned integer min, and float min on the order-preserving u64 encoding
  ACC_MAX_U64 = 5,
  ACC_CNT_ALL = 6,   // rows in the group, nulls included (u32)
  ACC_CNT_VALID = 7, // non-null rows of the source column (u32)
  ACC_FIRST_ROW = 8, // smallest row index of the group (u32 min) for maintain_order / GroupsIdx.first
  ACC_ADD_F32 = 9,   // PLHIP_FLAG_EXACT_FOLD only (the sort strategy's row-order fold): an f32 column folded IN f32 as the reference
                     // does (aggregations/mod.rs:719-724); the state is the f64 image of that f32, so the finalizers are ADD_F64's
};

// how the source value is turned into the accumulator operand
enum AccConv : uint32_t {
  CONV_INT = 0,      // integer value, sign/zero extended per dtype
  CONV_F64 = 1,      // value converted to double (f32, f64, i64, u64 sources)
  CONV_FMIN = 2,     // float -> ordered u64, NaN = largest   (MIN ignores NaN)
  CONV_FMAX = 3,     // float -> ordered u64, NaN = smallest  (MAX ignores NaN)
  CONV_NONE = 4,     // counters / row index
};

struct KeySpec {
  const uint8_t *values;   // element 0 of the logical slice
  const uint8_t *validity; // bitmap or nullptr
  int64_t boff;            // bit offset of element 0 in the bitmap
  uint64_t minv;           // smallest normalised value (0 for single-key queries)
  uint32_t dtype;          // plhip_dtype_t
  uint32_t shift;          // bit position of this key's code in the packed key
  uint32_t bits;           // code width in bits (64 = whole word)
  uint32_t nullable;       // 1: code 0 means null, values are stored +1   (multi-key only)
};

struct ValSpec {
  const uint8_t *values;
  const uint8_t *validity;
  int64_t boff;
  uint32_t dtype;
  uint32_t valid_bytes; // 1: `validity` is one byte per row (partitioned intermediates), 0: Arrow bitmap
};

struct AccSpec {
  uint32_t op;     // AccOp
  uint32_t conv;   // AccConv
  int32_t src;     // index into vals[] or -1
  uint32_t width;  // bytes per slot: 4 or 8
  uint64_t init;   // bit pattern a fresh slot starts from
  uint64_t offset; // byte offset of this accumulator's SoA array inside a table of `cap + 2` slots
};

struct Plan {
  int64_t n_rows;
  int32_t n_keys;
  int32_t n_vals;
  int32_t n_accs;
  int32_t single_key; // 1: packed key = normalised key bits, null -> reserved slot; 0: range-packed tuple
  const uint32_t *rowidx; // original row index of each row (partitioned intermediates) or nullptr = identity
  int32_t prepacked;      // 1: keys[0].values already holds the packed key (u32 or u64), use it as is
  int32_t exact_fold;     // 1: PLHIP_FLAG_EXACT_FOLD -- float sums fold strictly in ascending row order (sort strategy only)
  KeySpec keys[PLHIP_MAX_KEYS];
  ValSpec vals[PLHIP_MAX_VALS];
  AccSpec accs[PLHIP_MAX_ACCS];
};

// One unit of work of the LDS aggregation kernel: a row range that ONE workgroup folds into its LDS table.
// direct = 1: no other work item can hold the same keys (a whole hash partition) -> the groups are
// written straight to the dense area of the table; direct = 0: merged into the hashed HBM table.
struct WorkItem {
  int64_t row0, row1;
  uint32_t direct;
  uint32_t pad;
};

enum OutKind : uint32_t {
  OUT_KEY = 0,
  OUT_SUM_INT = 1,    // from ADD_U64, truncated to the output width
  OUT_SUM_F64 = 2,    // from ADD_F64 (f64 or f32 output)
  OUT_MEAN_INT = 3,   // (double)(int64)sum / count
  OUT_MEAN_UINT = 4,  // (double)(uint64)sum / count  (u32 and narrower accumulate unsigned)
  OUT_MEAN_F64 = 5,   // sum / count (f64 out), or through f32 rounding (f32 out)
  OUT_MIN_INT = 6,    // MIN/MAX_{I,U}64 truncated to the output width
  OUT_FMIN = 7,       // decode ordered u64, apply the len==1 / all-NaN rules
  OUT_FMAX = 8,
  OUT_COUNT = 9,
  OUT_RAW_U64 = 10,   // partial mode: accumulator as is (u64 / i64 / f64 bits)
  OUT_RAW_U32_AS_U64 = 11,
  OUT_RAW_U64_WRAP32 = 12, // OUT_RAW_U64 of an i32 / u32 SUM whose consumer wraps it in 32 bits (the sharded path's finalize): the
                           // 64-bit partial only has to be right modulo 2^32, so its accumulator may live in 32 bits
};

struct OutSpec {
  uint32_t kind;      // OutKind
  uint32_t dtype;     // plhip_dtype_t of the output column
  int32_t acc;        // main accumulator (or key index for OUT_KEY)
  int32_t cnt_valid;  // accumulator index of the valid count, or -1 (then cnt_all / always valid)
  int32_t cnt_all;    // accumulator index of the group length, or -1
  uint32_t pad;
  uint8_t *values;    // output buffer (device)
  uint8_t *validity;  // output bitmap (device) or nullptr when the column can never be null
};

struct OutPlan {
  int32_t n_outs;
  int32_t pad;
  OutSpec outs[PLHIP_MAX_OUTS];
};
// the fast path passes Plan + OutPlan + its own arguments by value: the kernel-argument segment is 4 KB
static_assert(sizeof(Plan) + sizeof(OutPlan) <= 3800, "Plan + OutPlan must fit the kernel-argument segment");

// Wide key tuples (more than 64 bits after range compression) are reduced to dense ids first; their
// output key columns are then gathered from the ORIGINAL columns at each group's first row, exactly as
// GroupBy::keys_sliced does (crates/polars-core/src/frame/group_by/mod.rs:262-296).
struct KeyGather {
  int32_t enabled;   // 0: unpack keys from the packed word; 1: gather okeys[k] at first_row
  int32_t first_acc; // accumulator index of ACC_FIRST_ROW
  KeySpec okeys[PLHIP_MAX_KEYS];
};
