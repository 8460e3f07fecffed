// device.hpp — device-side building blocks of the gfx950 group-by kernels: hashing, row decoding,
// accumulator updates on HBM and LDS tables.  Everything here is __device__ __forceinline__ so that the
// address space of table pointers (global vs LDS) is visible to the compiler at every atomic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/plhip.h"
#include "plan.hpp"

#define DEVINL __device__ __forceinline__

// ---- hashing (crates/polars-core/src/hashing/vector_hasher.rs:37-40) ----------------------------------
DEVINL uint64_t folded_multiply(uint64_t s, uint64_t by) {
  return (s * by) ^ __umul64hi(s, by);
}
DEVINL uint64_t hash_key(uint64_t packed) { return folded_multiply(packed, PLHIP_MULTIPLE); }

// Cheap 32-bit mixes for the partition id and the LDS slot (independent of each other and of the HBM
// table's folded-multiply slot): Fibonacci hashing spreads dense integer keys evenly over the top bits.
DEVINL uint32_t mix_a(uint64_t key) { return (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u; }
DEVINL uint32_t part_of(uint32_t a, uint32_t pbits) { return pbits ? a >> (32 - pbits) : 0u; }
DEVINL uint32_t lds_home(uint32_t a, uint32_t lbits) { return ((a ^ (a >> 15)) * 0xC2B2AE3Du) >> (32 - lbits); }

// ---- column access -----------------------------------------------------------------------------------
DEVINL bool bit_valid(const uint8_t *validity, int64_t boff, int64_t row) {
  if (!validity) return true;
  const int64_t b = boff + row;
  return (validity[b >> 3] >> (b & 7)) & 1;
}
DEVINL bool val_valid(const ValSpec &v, int64_t row) {
  if (!v.validity) return true;
  if (v.valid_bytes) return v.validity[row] != 0;
  const int64_t b = v.boff + row;
  return (v.validity[b >> 3] >> (b & 7)) & 1;
}

constexpr uint64_t SIGN64 = 0x8000000000000000ull;

// Normalised key: an order-preserving (for integers) injective map of the key bits into u64.
// Floats keep their raw bit pattern: the reference groups float keys by bits
// (crates/polars-core/src/frame/group_by/into_groups.rs:167-174).
DEVINL uint64_t load_key_norm(const uint8_t *p, uint32_t dtype, int64_t row) {
  switch (dtype) {
    // signed keys of <= 4 bytes normalise inside 32 bits, so that they fit a 32-bit table slot
    case PLHIP_I8: return (uint64_t)((uint32_t)(int32_t)((const int8_t *)p)[row] ^ 0x80000000u);
    case PLHIP_I16: return (uint64_t)((uint32_t)(int32_t)((const int16_t *)p)[row] ^ 0x80000000u);
    case PLHIP_I32: return (uint64_t)(((const uint32_t *)p)[row] ^ 0x80000000u);
    case PLHIP_I64: return ((const uint64_t *)p)[row] ^ SIGN64;
    case PLHIP_U8: return ((const uint8_t *)p)[row];
    case PLHIP_U16: return ((const uint16_t *)p)[row];
    case PLHIP_U32: case PLHIP_F32: return ((const uint32_t *)p)[row];
    default: return ((const uint64_t *)p)[row];
  }
}
DEVINL uint64_t key_denorm(uint64_t norm, uint32_t dtype) {
  switch (dtype) {
    case PLHIP_I8: case PLHIP_I16: case PLHIP_I32: return norm ^ 0x80000000ull; // stored truncated to the width
    case PLHIP_I64: return norm ^ SIGN64;
    default: return norm;
  }
}

// as_u64 of the hashing path (crates/polars-core/src/hashing/partition.rs:5-65)
DEVINL uint64_t load_as_u64(const uint8_t *p, uint32_t dtype, int64_t row) {
  switch (dtype) {
    case PLHIP_I8: return (uint64_t)(int64_t)((const int8_t *)p)[row];
    case PLHIP_I16: return (uint64_t)(int64_t)((const int16_t *)p)[row];
    case PLHIP_U8: return ((const uint8_t *)p)[row];
    case PLHIP_U16: return ((const uint16_t *)p)[row];
    case PLHIP_I32: case PLHIP_U32: case PLHIP_F32: return ((const uint32_t *)p)[row];
    default: return ((const uint64_t *)p)[row];
  }
}

DEVINL uint64_t load_int64(const uint8_t *p, uint32_t dtype, int64_t row) {
  switch (dtype) {
    case PLHIP_I8: return (uint64_t)(int64_t)((const int8_t *)p)[row];
    case PLHIP_I16: return (uint64_t)(int64_t)((const int16_t *)p)[row];
    case PLHIP_I32: return (uint64_t)(int64_t)((const int32_t *)p)[row];
    case PLHIP_U8: return ((const uint8_t *)p)[row];
    case PLHIP_U16: return ((const uint16_t *)p)[row];
    case PLHIP_U32: return ((const uint32_t *)p)[row];
    default: return ((const uint64_t *)p)[row];
  }
}
DEVINL double load_double(const uint8_t *p, uint32_t dtype, int64_t row) {
  switch (dtype) {
    case PLHIP_F32: return (double)((const float *)p)[row];
    case PLHIP_F64: return ((const double *)p)[row];
    case PLHIP_U64: return (double)((const uint64_t *)p)[row];
    case PLHIP_U8: case PLHIP_U16: case PLHIP_U32: return (double)load_int64(p, dtype, row);
    default: return (double)(int64_t)load_int64(p, dtype, row);
  }
}

// Order-preserving u64 encoding of a double: -inf < ... < -0 < +0 < ... < +inf.  NaN is mapped by the
// caller (largest for MIN, smallest for MAX: compare_fn_nan_max / compare_fn_nan_min,
// crates/polars-arrow/src/kernels/rolling/mod.rs:29-69).
DEVINL uint64_t f64_ordered(double d) {
  const uint64_t u = (uint64_t)__double_as_longlong(d);
  return (u & SIGN64) ? ~u : (u | SIGN64);
}
DEVINL double f64_from_ordered(uint64_t e) {
  const uint64_t u = (e & SIGN64) ? (e & ~SIGN64) : ~e;
  return __longlong_as_double((long long)u);
}

// Packed key of a row.  single_key: normalised bits, `isnull` tells the caller to use the reserved null
// slot.  Multi-key: range-compressed codes (null = 0) shifted into one word; `isnull` stays false.
DEVINL uint64_t row_packed_key(const Plan &P, int64_t row, bool &isnull) {
  isnull = false;
  if (P.prepacked)
    return P.keys[0].dtype == PLHIP_U32 ? (uint64_t)((const uint32_t *)P.keys[0].values)[row]
                                        : ((const uint64_t *)P.keys[0].values)[row];
  if (P.single_key) {
    const KeySpec &k = P.keys[0];
    if (!bit_valid(k.validity, k.boff, row)) { isnull = true; return 0; }
    return load_key_norm(k.values, k.dtype, row);
  }
  uint64_t packed = 0;
  for (int i = 0; i < P.n_keys; i++) {
    const KeySpec &k = P.keys[i];
    uint64_t code = 0;
    if (bit_valid(k.validity, k.boff, row)) code = load_key_norm(k.values, k.dtype, row) - k.minv + k.nullable;
    packed |= code << k.shift;
  }
  return packed;
}

// ---- the accumulator operand of one row --------------------------------------------------------------
// Returns false when the row does not touch this accumulator (null source value).
DEVINL bool acc_operand(const Plan &P, const AccSpec &a, int64_t row, uint64_t &operand) {
  if (a.op == ACC_CNT_ALL) { operand = 1; return true; }
  if (a.op == ACC_FIRST_ROW) { operand = P.rowidx ? (uint64_t)P.rowidx[row] : (uint64_t)row; return true; }
  const ValSpec &v = P.vals[a.src];
  if (!val_valid(v, row)) return false;
  switch (a.conv) {
    case CONV_INT: operand = load_int64(v.values, v.dtype, row); break;
    case CONV_F64: operand = (uint64_t)__double_as_longlong(load_double(v.values, v.dtype, row)); break;
    case CONV_FMIN: { const double d = load_double(v.values, v.dtype, row); operand = (d != d) ? ~0ull : f64_ordered(d); break; }
    case CONV_FMAX: { const double d = load_double(v.values, v.dtype, row); operand = (d != d) ? 0ull : f64_ordered(d); break; }
    default: operand = 1; break; // CNT_VALID
  }
  return true;
}

// ---- folding an operand into a slot: the same code serves HBM (global_atomic_*) and LDS (ds_*) tables;
// `base` is the table base in the matching address space. -----------------------------------------------
template <typename BasePtr>
DEVINL void acc_apply(BasePtr base, const AccSpec &a, uint64_t slot, uint64_t operand) {
  auto *p = base + a.offset + slot * a.width;
  switch (a.op) {
    case ACC_ADD_U64: atomicAdd((unsigned long long *)p, (unsigned long long)operand); break;
    case ACC_ADD_F64: atomicAdd((double *)p, __longlong_as_double((long long)operand)); break;
    case ACC_MIN_I64: atomicMin((long long *)p, (long long)operand); break;
    case ACC_MAX_I64: atomicMax((long long *)p, (long long)operand); break;
    case ACC_MIN_U64: atomicMin((unsigned long long *)p, (unsigned long long)operand); break;
    case ACC_MAX_U64: atomicMax((unsigned long long *)p, (unsigned long long)operand); break;
    case ACC_CNT_ALL: case ACC_CNT_VALID: atomicAdd((unsigned int *)p, (unsigned int)operand); break;
    case ACC_FIRST_ROW: atomicMin((unsigned int *)p, (unsigned int)operand); break;
  }
}

// ---- HBM table: find-or-insert -------------------------------------------------------------------------
// Linear probing on the u64 key array with a 64-bit CAS claim.  A stale (L1) read can only show EMPTY for
// a slot that has since been claimed; the CAS then returns the real owner, so the result is exact without
// any fence.  Returns the slot, or ~0 if the probe limit was hit (caller raises the overflow flag).
#define PLHIP_GLOBAL_PROBE_LIMIT 8192
DEVINL uint64_t gtable_slot(uint64_t *keys, uint64_t cap, uint32_t *flags, uint64_t key, bool isnull) {
  if (isnull) { if (!flags[2]) flags[2] = 1; return cap + 1; }
  if (key == PLHIP_EMPTY_KEY) { if (!flags[1]) flags[1] = 1; return cap; }
  const uint64_t mask = cap - 1;
  uint64_t h = hash_key(key);
  uint64_t s = (h ^ (h >> 32)) & mask;
  const uint64_t limit = cap < PLHIP_GLOBAL_PROBE_LIMIT ? cap : PLHIP_GLOBAL_PROBE_LIMIT;
  for (uint64_t i = 0; i < limit; i++) {
    uint64_t k = keys[s];
    if (k == key) return s;
    if (k == PLHIP_EMPTY_KEY) {
      k = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)PLHIP_EMPTY_KEY, (unsigned long long)key);
      if (k == PLHIP_EMPTY_KEY || k == key) return s;
    }
    s = (s + 1) & mask;
  }
  flags[0] = 1;
  return ~0ull;
}

// ---- accumulators -> the reference's output columns (shared by k_finalize and the fast path's merge) ----
DEVINL void store_by_dtype_int(uint8_t *out, uint32_t dtype, uint64_t i, uint64_t v) {
  switch (dtype) {
    case PLHIP_I8: case PLHIP_U8: out[i] = (uint8_t)v; break;
    case PLHIP_I16: case PLHIP_U16: ((uint16_t *)out)[i] = (uint16_t)v; break;
    case PLHIP_I32: case PLHIP_U32: case PLHIP_F32: ((uint32_t *)out)[i] = (uint32_t)v; break;
    default: ((uint64_t *)out)[i] = v; break;
  }
}
DEVINL void store_float(uint8_t *out, uint32_t dtype, uint64_t i, double v) {
  if (dtype == PLHIP_F32) ((float *)out)[i] = (float)v;
  else ((double *)out)[i] = v;
}

// Output key column `S.acc` of group i from the packed key word.  Returns the key's validity.
DEVINL bool finalize_key(const Plan &P, const OutSpec &S, uint64_t packed, bool null_slot, bool sentinel_slot, uint64_t i) {
  const KeySpec &K = P.keys[S.acc];
  bool valid = true;
  uint64_t norm;
  if (P.single_key) {
    valid = !null_slot;
    norm = sentinel_slot ? PLHIP_EMPTY_KEY : packed;
  } else {
    uint64_t code = K.bits >= 64 ? packed : ((packed >> K.shift) & ((1ull << K.bits) - 1));
    if (K.nullable) { valid = code != 0; code -= 1; }
    norm = code + K.minv;
  }
  store_by_dtype_int(S.values, K.dtype, i, valid ? key_denorm(norm, K.dtype) : 0);
  return valid;
}

// One aggregate output (everything except OUT_KEY) of group i.  Returns its validity.
DEVINL bool finalize_value(const OutSpec &S, uint64_t acc, uint32_t cnt_all, uint32_t cnt_valid, uint64_t i) {
  bool valid = true;
  switch (S.kind) {
    case OUT_SUM_INT: // wrapping sum in the output width (aggregations/mod.rs:622-651); never null
      store_by_dtype_int(S.values, S.dtype, i, acc);
      break;
    case OUT_SUM_F64:
      store_float(S.values, S.dtype, i, __longlong_as_double((long long)acc));
      break;
    case OUT_MEAN_INT: case OUT_MEAN_UINT: { // aggregations/mod.rs:941-991
      valid = cnt_valid != 0;
      const double sum = S.kind == OUT_MEAN_INT ? (double)(long long)acc : (double)acc;
      store_float(S.values, S.dtype, i, valid ? sum / (double)cnt_valid : 0.0);
      break;
    }
    case OUT_MEAN_F64: { // aggregations/mod.rs:701-742: f32 sums are f32 before the division
      valid = cnt_valid != 0;
      double sum = __longlong_as_double((long long)acc);
      if (S.dtype == PLHIP_F32) sum = (double)(float)sum;
      store_float(S.values, S.dtype, i, valid ? sum / (double)cnt_valid : 0.0);
      break;
    }
    case OUT_MIN_INT:
      valid = S.cnt_valid >= 0 ? cnt_valid != 0 : true;
      store_by_dtype_int(S.values, S.dtype, i, valid ? acc : 0);
      break;
    case OUT_FMIN: case OUT_FMAX: {
      // aggregations/mod.rs:459-620 with TakeExtremum :240-259: fold from T::MAX (min) / T::MIN (max)
      // ignoring NaN, so +-inf and all-NaN groups clamp to the init; a 1-row group returns its value.
      valid = cnt_valid != 0;
      const bool is_min = S.kind == OUT_FMIN;
      const double big = S.dtype == PLHIP_F32 ? 3.4028234663852886e38 : 1.7976931348623157e308;
      double r = 0.0;
      if (valid) {
        const bool only_nan = is_min ? acc == ~0ull : acc == 0ull;
        if (cnt_all == 1) r = only_nan ? __longlong_as_double(0x7ff8000000000000ll) : f64_from_ordered(acc);
        else if (only_nan) r = is_min ? big : -big;
        else {
          r = f64_from_ordered(acc);
          if (is_min && r > big) r = big;
          if (!is_min && r < -big) r = -big;
        }
      }
      store_float(S.values, S.dtype, i, r);
      break;
    }
    case OUT_COUNT: ((uint32_t *)S.values)[i] = (uint32_t)acc; break;
    case OUT_RAW_U64: case OUT_RAW_U64_WRAP32: ((uint64_t *)S.values)[i] = acc; break;
    case OUT_RAW_U32_AS_U64: ((uint64_t *)S.values)[i] = acc; break;
    default: break;
  }
  return valid;
}

// ---- batched column access shared by the LDS pipeline (fast_path.hip) and the sort strategy (sort_path.hip) ----
DEVINL uint32_t wbytes(uint32_t dt) {
  switch (dt) {
    case PLHIP_I8: case PLHIP_U8: return 1;
    case PLHIP_I16: case PLHIP_U16: return 2;
    case PLHIP_I32: case PLHIP_U32: case PLHIP_F32: return 4;
    default: return 8;
  }
}


// raw column bits of B rows, zero-extended.  QUAD: the B == 4 rows are consecutive, row[0] % 4 == 0 and the
// column base is aligned to 4 elements, so one (two for 8-byte types) vector load fetches them.
template <int B, bool QUAD>
DEVINL void load_raw(const uint8_t *vals, uint32_t w, const uint32_t (&row)[B], uint64_t (&raw)[B]) {
  if (QUAD) { // every group of four rows is one aligned quad: vector loads
    static_assert(!QUAD || B % 4 == 0, "quads");
    switch (w) {
      case 1:
#pragma unroll
        for (int g = 0; g < B / 4; g++) {
          const uint32_t q = *(const uint32_t *)(vals + (size_t)row[4 * g]);
#pragma unroll
          for (int b = 0; b < 4; b++) raw[4 * g + b] = (q >> (8 * b)) & 0xFFu;
        }
        break;
      case 2:
#pragma unroll
        for (int g = 0; g < B / 4; g++) {
          const uint2 q = *(const uint2 *)(vals + (size_t)row[4 * g] * 2);
          raw[4 * g] = q.x & 0xFFFFu; raw[4 * g + 1] = q.x >> 16; raw[4 * g + 2] = q.y & 0xFFFFu; raw[4 * g + 3] = q.y >> 16;
        }
        break;
      case 4:
#pragma unroll
        for (int g = 0; g < B / 4; g++) {
          const uint4 q = *(const uint4 *)(vals + (size_t)row[4 * g] * 4);
          raw[4 * g] = q.x; raw[4 * g + 1] = q.y; raw[4 * g + 2] = q.z; raw[4 * g + 3] = q.w;
        }
        break;
      default:
#pragma unroll
        for (int g = 0; g < B / 4; g++) {
          const uint4 q0 = *(const uint4 *)(vals + (size_t)row[4 * g] * 8);
          const uint4 q1 = *(const uint4 *)(vals + (size_t)row[4 * g] * 8 + 16);
          raw[4 * g] = (uint64_t)q0.x | ((uint64_t)q0.y << 32); raw[4 * g + 1] = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
          raw[4 * g + 2] = (uint64_t)q1.x | ((uint64_t)q1.y << 32); raw[4 * g + 3] = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
        }
        break;
    }
    return;
  }
  switch (w) {
#define LD_RAW(T) _Pragma("unroll") for (int b = 0; b < B; b++) raw[b] = (uint64_t)((const T *)vals)[row[b]]; break;
    case 1: LD_RAW(uint8_t)
    case 2: LD_RAW(uint16_t)
    case 4: LD_RAW(uint32_t)
    default: LD_RAW(uint64_t)
#undef LD_RAW
  }
}

// raw bits -> accumulator operands.  Sign extension is a pair of uniform shifts (0 for unsigned and 64-bit
// types), so the integer path has no per-dtype branch at all.
template <int B>
DEVINL void raw_to_operands(uint32_t dtype, uint32_t conv, const uint64_t (&raw)[B], uint64_t (&opnd)[B]) {
  const bool sgn = dtype == PLHIP_I8 || dtype == PLHIP_I16 || dtype == PLHIP_I32;
  const uint32_t sh = sgn ? 64u - 8u * wbytes(dtype) : 0u;
  int64_t x[B];
#pragma unroll
  for (int b = 0; b < B; b++) x[b] = (int64_t)(raw[b] << sh) >> sh;
  if (conv == CONV_INT) {
#pragma unroll
    for (int b = 0; b < B; b++) opnd[b] = (uint64_t)x[b];
    return;
  }
  double d[B];
  if (dtype == PLHIP_F64) {
#pragma unroll
    for (int b = 0; b < B; b++) d[b] = __longlong_as_double((long long)raw[b]);
  } else if (dtype == PLHIP_F32) {
#pragma unroll
    for (int b = 0; b < B; b++) d[b] = (double)__uint_as_float((uint32_t)raw[b]);
  } else if (dtype == PLHIP_U64) {
#pragma unroll
    for (int b = 0; b < B; b++) d[b] = (double)raw[b];
  } else {
#pragma unroll
    for (int b = 0; b < B; b++) d[b] = (double)x[b];
  }
#pragma unroll
  for (int b = 0; b < B; b++) {
    if (conv == CONV_F64) opnd[b] = (uint64_t)__double_as_longlong(d[b]);
    else if (conv == CONV_FMIN) opnd[b] = (d[b] != d[b]) ? ~0ull : f64_ordered(d[b]);
    else opnd[b] = (d[b] != d[b]) ? 0ull : f64_ordered(d[b]);
  }
}


