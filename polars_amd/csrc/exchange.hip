// exchange.hip — the device side of the ONE exchange step of the row-sharded group-by: a partial
// frame becomes row records bucketed by owner rank, and received records become a frame again.  The reference
// does the same between threads by concatenating the partial frames (PartitionGroupByExec::run_partitions,
// crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:66-113, 320-360); between GPUs the
// "concat" is an all-to-all of these records.
//
//   k_owner_count   owner(row) = ((vec_hash(keys) >> 33) & 0x7FFFFFFF) % n_parts, rows per owner
//   k_pack_records  every workgroup reserves a range per owner (one global atomic per owner) and writes its rows'
//                   records there: bucket after bucket, order inside a bucket unspecified
//   k_unpack        records -> columns + LSB-first validity bitmaps
//
// HBM-bound byte work on G (number of groups) rows: no hashing table, no per-row global atomics.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "kernels.hpp"

namespace {

#define EX_BLOCK 256

DEVINL uint32_t owner_of(const RecPlan &R, uint64_t null_h, int64_t row) {
  uint64_t h = bit_valid(R.validity[0], R.boff[0], row) ? folded_multiply(load_as_u64(R.values[0], R.dtype[0], row), PLHIP_MULTIPLE) : null_h;
  for (int k = 1; k < R.n_keys; k++) {
    const uint64_t v = bit_valid(R.validity[k], R.boff[k], row) ? load_as_u64(R.values[k], R.dtype[k], row) : null_h;
    h = folded_multiply(v ^ h, PLHIP_MULTIPLE);
  }
  return (uint32_t)(((h >> 33) & 0x7FFFFFFFull) % R.n_parts);
}

__global__ void __launch_bounds__(EX_BLOCK) k_owner_count(RecPlan R, uint64_t null_h, uint16_t *__restrict__ owner, uint32_t *__restrict__ counts) {
  __shared__ uint32_t hist[EX_MAX_PARTS];
  for (uint32_t p = threadIdx.x; p < R.n_parts; p += EX_BLOCK) hist[p] = 0;
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * EX_BLOCK + threadIdx.x;
  if (row < R.n_rows) {
    const uint32_t o = owner_of(R, null_h, row);
    owner[row] = (uint16_t)o;
    atomicAdd(&hist[o], 1u);
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < R.n_parts; p += EX_BLOCK)
    if (hist[p]) atomicAdd(&counts[p], hist[p]);
}

DEVINL void copy_field(uint8_t *dst, const uint8_t *src, uint32_t w) {
  switch (w) { // record fields are not aligned: bytes
    case 1: dst[0] = src[0]; break;
    case 2: dst[0] = src[0]; dst[1] = src[1]; break;
    case 4:
#pragma unroll
      for (int i = 0; i < 4; i++) dst[i] = src[i];
      break;
    default:
#pragma unroll
      for (int i = 0; i < 8; i++) dst[i] = src[i];
      break;
  }
}

// counts[] = rows per owner (complete: k_owner_count has finished), cursor[] zeroed
__global__ void __launch_bounds__(EX_BLOCK) k_pack_records(RecPlan R, const uint16_t *__restrict__ owner, const uint32_t *__restrict__ counts,
                                                           uint32_t *__restrict__ cursor, uint8_t *__restrict__ records) {
  __shared__ uint32_t hist[EX_MAX_PARTS]; // rows of this workgroup per owner, then the start of its range
  __shared__ uint32_t start[EX_MAX_PARTS]; // first record of the owner's bucket
  for (uint32_t p = threadIdx.x; p < R.n_parts; p += EX_BLOCK) hist[p] = 0;
  if (threadIdx.x == 0) {
    uint32_t run = 0, over = 0;
    for (uint32_t p = 0; p < R.n_parts; p++) {
      start[p] = R.slot_rows ? 0u : run;
      run += counts[p];
      if (R.slot_rows && counts[p] > R.slot_rows) over = 1;
    }
    if (R.slot_rows && blockIdx.x == 0) // the slot headers: rows in the bucket, "this sender overflowed somewhere"
      for (uint32_t p = 0; p < R.n_parts; p++) {
        unsigned long long *h = (unsigned long long *)(records + (uint64_t)p * R.slot_stride);
        h[0] = counts[p]; h[1] = over;
      }
  }
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * EX_BLOCK + threadIdx.x;
  const bool live = row < R.n_rows;
  uint32_t o = 0, rank = 0;
  if (live) {
    o = owner[row];
    rank = atomicAdd(&hist[o], 1u); // position inside the workgroup's range
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < R.n_parts; p += EX_BLOCK) {
    const uint32_t c = hist[p];
    hist[p] = start[p] + (c ? atomicAdd(&cursor[p], c) : 0u);
  }
  __syncthreads();
  if (!live) return;
  const uint32_t pos = hist[o] + rank;
  if (R.slot_rows && pos >= R.slot_rows) return; // overflow: the header says so, the exchange is redone exactly
  uint8_t *rec = R.slot_rows ? records + (uint64_t)o * R.slot_stride + EX_SLOT_HDR + (uint64_t)pos * R.rec_bytes
                             : records + (uint64_t)pos * R.rec_bytes;
  for (int c = 0; c < R.n_cols; c++) {
    const uint32_t w = R.width[c];
    copy_field(rec + R.off[c], R.values[c] + (uint64_t)row * w, w);
    if (R.has_valid[c]) rec[R.off[c] + w] = bit_valid(R.validity[c], R.boff[c], row) ? 1 : 0;
  }
}

// a wave covers 64 consecutive rows starting at a multiple of 64: it owns one 64-bit word of every bitmap
__global__ void __launch_bounds__(EX_BLOCK) k_unpack(RecPlan R, const uint8_t *__restrict__ records) {
  const int64_t row = (int64_t)blockIdx.x * EX_BLOCK + threadIdx.x;
  const bool live = row < R.n_rows;
  const uint8_t *rec = records + (uint64_t)(live ? row : 0) * R.rec_bytes;
  if (R.slot_rows) { // slotted layout: output row -> (slot, position)
    uint32_t sl = 0;
    const uint32_t r32 = live ? (uint32_t)row : 0u;
    while (sl + 1 < R.n_parts && R.slot_first[sl + 1] <= r32) sl++;
    rec = records + (uint64_t)sl * R.slot_stride + EX_SLOT_HDR + (uint64_t)(r32 - R.slot_first[sl]) * R.rec_bytes;
  }
  for (int c = 0; c < R.n_cols; c++) {
    const uint32_t w = R.width[c];
    if (live) copy_field((uint8_t *)R.values[c] + (uint64_t)row * w, rec + R.off[c], w);
    if (R.has_valid[c]) {
      const unsigned long long m = __ballot(live && rec[R.off[c] + w] != 0);
      if ((threadIdx.x & 63) == 0 && row < R.n_rows) ((unsigned long long *)R.validity[c])[row >> 6] = m;
    }
  }
}

// one thread per group; a wave owns one 64-bit word of every output bitmap
__global__ void __launch_bounds__(EX_BLOCK) k_finalize_partial(FinPlan F) {
  const int64_t row = (int64_t)blockIdx.x * EX_BLOCK + threadIdx.x;
  const bool live = row < F.n_rows;
  const int64_t r = live ? row : 0;
  for (int o = 0; o < F.n_outs; o++) {
    const FinOut &X = F.outs[o];
    bool valid = true;
    if (X.mode == FIN_COPY) {
      if (live) copy_field(X.spec.values + (uint64_t)r * X.width, X.a + (uint64_t)r * X.width, X.width);
      valid = bit_valid(X.a_valid, 0, r);
    } else if (live) {
      const uint64_t acc = ((const uint64_t *)X.a)[r];
      const uint32_t cnt = X.cnt ? (uint32_t)((const uint64_t *)X.cnt)[r] : 1u; // group sizes are IdxSize (u32) in the reference
      valid = finalize_value(X.spec, acc, cnt, cnt, (uint64_t)r);
    }
    if (X.spec.validity) {
      const unsigned long long m = __ballot(live && valid);
      if ((threadIdx.x & 63) == 0 && live) ((unsigned long long *)X.spec.validity)[row >> 6] = m;
    }
  }
}

} // namespace

// the received slots' 16-byte headers {rows, overflow flag} -> one small block of (pinned, device-mapped) host memory
__global__ void __launch_bounds__(64) k_slot_headers(const uint8_t *__restrict__ slots, uint64_t stride, uint32_t n_slots, uint64_t *__restrict__ out) {
  const uint32_t p = threadIdx.x;
  if (p < n_slots) {
    const uint64_t *h = (const uint64_t *)(slots + (uint64_t)p * stride);
    out[2 * p] = h[0];
    out[2 * p + 1] = h[1];
  }
}
void launch_slot_headers(const uint8_t *slots, uint64_t stride, uint32_t n_slots, uint64_t *out, hipStream_t s) {
  hipLaunchKernelGGL(k_slot_headers, dim3(1), dim3(64), 0, s, slots, stride, n_slots, out);
}

void launch_finalize_partial(const FinPlan &f, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize_partial, dim3((unsigned)((f.n_rows + EX_BLOCK - 1) / EX_BLOCK)), dim3(EX_BLOCK), 0, s, f);
}
void launch_owner_count(const RecPlan &r, uint64_t null_hash, uint16_t *owner, uint32_t *counts, hipStream_t s) {
  hipLaunchKernelGGL(k_owner_count, dim3((unsigned)((r.n_rows + EX_BLOCK - 1) / EX_BLOCK)), dim3(EX_BLOCK), 0, s, r, null_hash, owner, counts);
}
void launch_pack_records(const RecPlan &r, const uint16_t *owner, const uint32_t *counts, uint32_t *cursor, uint8_t *records, hipStream_t s) {
  const int64_t blocks = (r.n_rows + EX_BLOCK - 1) / EX_BLOCK; // (an empty frame still writes its slot headers)
  hipLaunchKernelGGL(k_pack_records, dim3((unsigned)(blocks ? blocks : 1)), dim3(EX_BLOCK), 0, s, r, owner, counts, cursor, records);
}
void launch_unpack_records(const RecPlan &r, const uint8_t *records, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((r.n_rows + EX_BLOCK - 1) / EX_BLOCK)), dim3(EX_BLOCK), 0, s, r, records);
}
