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
