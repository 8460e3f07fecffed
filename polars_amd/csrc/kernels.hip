// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the hash group-by aggregation path.
//
// The path is HBM-bound integer/byte work: no MFMA.  What matters is coalesced column reads, keeping the
// per-group accumulators in LDS (ds_* atomics) whenever the groups a workgroup sees fit there, and
// touching HBM-resident tables with as few (scattered, memory-side) atomics as possible.
//
//   k_table_init   fill an HBM table with EMPTY keys / accumulator identities
//   k_agg_global   every row -> find-or-insert in the HBM table + global atomics  (any cardinality; the
//                  fallback and the correctness anchor)
//   k_agg_lds      persistent workgroups, one LDS open-addressing table each (ds_cmpst_b64 claim, ds_add /
//                  ds_min / ds_max folds), rows that do not fit spill to the HBM table, then ONE merge of
//                  every LDS slot into the HBM table
//   k_compact      occupied slots -> dense slot list + group count
//   k_finalize     accumulators -> the reference's output columns (dtype / null / NaN rules)
//   k_key_minmax   per key column min/max of the normalised bits (range compression of multi-key tuples)
//   k_vec_hash     VecHash::vec_hash / vec_hash_combine
#include "kernels.hpp"

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "device.hpp"

// ---------------------------------------------------------------------------------------------------------
// layout
// ---------------------------------------------------------------------------------------------------------
static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

void plan_layout(Plan &p, uint64_t n_slots) {
  uint64_t off = align_up(n_slots * 8, 256);
  for (int i = 0; i < p.n_accs; i++) {
    p.accs[i].offset = off;
    off += align_up(n_slots * p.accs[i].width, 256);
  }
}

size_t table_bytes(const Plan &p, uint64_t n_slots) {
  uint64_t off = align_up(n_slots * 8, 256);
  for (int i = 0; i < p.n_accs; i++) off += align_up(n_slots * p.accs[i].width, 256);
  return off;
}

// ---------------------------------------------------------------------------------------------------------
// HBM table init
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_table_init(Plan P, TableRef T) {
  const uint64_t n = T.cap + 2;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    ((uint64_t *)T.base)[i] = PLHIP_EMPTY_KEY;
    for (int a = 0; a < P.n_accs; a++) {
      const AccSpec &A = P.accs[a];
      if (A.width == 8) ((uint64_t *)(T.base + A.offset))[i] = A.init;
      else ((uint32_t *)(T.base + A.offset))[i] = (uint32_t)A.init;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 16) T.flags[threadIdx.x] = 0;
}

void launch_table_init(const Plan &p, TableRef t, hipStream_t s) {
  uint64_t n = t.cap + 2;
  int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_table_init, dim3(grid), dim3(256), 0, s, p, t);
}

// ---------------------------------------------------------------------------------------------------------
// strategy "global": every row updates the HBM table
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_agg_global(Plan P, TableRef T) {
  uint64_t *keys = (uint64_t *)T.base;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < P.n_rows; row += stride) {
    bool isnull;
    const uint64_t key = row_packed_key(P, row, isnull);
    const uint64_t slot = gtable_slot(keys, T.cap, T.flags, key, isnull);
    if (slot == ~0ull) continue; // overflow flag raised; the host retries with a larger table
    for (int a = 0; a < P.n_accs; a++) {
      const AccSpec &A = P.accs[a];
      uint64_t operand;
      if (!acc_operand(P, A, row, operand)) continue;
      // The first-row minimum only ever moves down: a row that cannot lower it skips the atomic (a stale read shows an
      // older, LARGER value: then the atomic runs as before).  Without this one key on 30 % of 1e7 rows serialised 3e6
      // returning atomics on one address (plhip_group_tuples: 47 ms).
      if (A.op == ACC_FIRST_ROW && ((const volatile uint32_t *)(T.base + A.offset))[slot] <= (uint32_t)operand) continue;
      acc_apply(T.base, A, slot, operand);
    }
  }
}

// keys-only pass over every `stride`-th row: the sampled cardinality estimate
__global__ void __launch_bounds__(256) k_sample_keys(Plan P, TableRef T, int64_t n_sample, int64_t stride) {
  uint64_t *keys = (uint64_t *)T.base;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_sample; i += gs) {
    bool isnull;
    const uint64_t key = row_packed_key(P, i * stride, isnull);
    gtable_slot(keys, T.cap, T.flags, key, isnull);
  }
}

void launch_sample_keys(const Plan &p, TableRef t, int64_t n_sample, int64_t stride, hipStream_t s) {
  if (n_sample <= 0) return;
  hipLaunchKernelGGL(k_sample_keys, dim3((unsigned)((n_sample + 255) / 256)), dim3(256), 0, s, p, t, n_sample, stride);
}

void launch_agg_global(const Plan &p, TableRef t, hipStream_t s) {
  int64_t blocks = (p.n_rows + 255) / 256;
  int grid = (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
  hipLaunchKernelGGL(k_agg_global, dim3(grid), dim3(256), 0, s, p, t);
}

// ---------------------------------------------------------------------------------------------------------
// compaction: occupied slots -> slot list (+ group count in flags[FLAG_NGROUPS])
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_compact(TableRef T, uint64_t *slots) {
  const uint64_t *keys = (const uint64_t *)T.base;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < T.cap + 2; base += stride) {
    const uint64_t i = base + threadIdx.x;
    bool occ = false;
    if (i < T.cap) occ = keys[i] != PLHIP_EMPTY_KEY;
    else if (i == T.cap) occ = T.flags[FLAG_SENTINEL] != 0;
    else if (i == T.cap + 1) occ = T.flags[FLAG_NULL] != 0;
    // one counter atomic per wave: ballot + popcount prefix
    const unsigned long long m = __ballot(occ);
    uint32_t wbase = 0;
    if (m) {
      if (lane == 0) wbase = atomicAdd(&T.flags[FLAG_NGROUPS], (uint32_t)__popcll(m));
      wbase = __shfl(wbase, 0);
      if (occ) slots[wbase + __popcll(m & ((1ull << lane) - 1))] = i;
    }
  }
}

void launch_compact(const Plan &p, TableRef t, uint64_t *slots, hipStream_t s) {
  (void)p;
  uint64_t n = t.cap + 2;
  int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, s, t, slots);
}

// ---------------------------------------------------------------------------------------------------------
// ordering support
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_sort_keys(Plan P, TableRef T, const uint64_t *slots, uint64_t n,
                                                          int first_acc, uint64_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t s = slots[i];
  out[i] = ((const uint32_t *)(T.base + P.accs[first_acc].offset))[s];
}

void launch_gather_sort_keys(const Plan &p, TableRef t, const uint64_t *slots, uint64_t n, int by_first_acc,
                             uint64_t *out_keys, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_gather_sort_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, t, slots, n,
                     by_first_acc, out_keys);
}

// Output ordering only (maintain_order / sort_keys): a library radix sort of (u64 key, u64 slot) pairs.
hipError_t sort_pairs_u64(uint64_t *keys, uint64_t *vals, uint64_t n, int end_bit, hipStream_t s) {
  if (n < 2) return hipSuccess;
  uint64_t *kout = nullptr, *vout = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipMalloc(&kout, n * 8);
  if (e != hipSuccess) return e;
  e = hipMalloc(&vout, n * 8);
  if (e != hipSuccess) { hipFree(kout); return e; }
  e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, kout, vals, vout, n, 0, end_bit, s);
  if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, kout, vals, vout, n, 0, end_bit, s);
  if (e == hipSuccess) e = hipMemcpyAsync(keys, kout, n * 8, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(vals, vout, n * 8, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(kout); hipFree(vout); hipFree(tmp);
  return e;
}

// ---------------------------------------------------------------------------------------------------------
// finalize: accumulators -> output columns
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_finalize(Plan P, TableRef T, const OutPlan *__restrict__ Op,
                                                  const KeyGather *__restrict__ KG, const uint64_t *slots, uint64_t G) {
  const OutPlan &O = *Op;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // a wave covers an aligned 64-group span
  const bool live = i < G;
  const uint64_t s = live ? slots[i] : 0;
  const uint64_t packed = live ? ((const uint64_t *)T.base)[s] : 0;
  for (int o = 0; o < O.n_outs; o++) {
    const OutSpec &S = O.outs[o];
    bool valid = live;
    if (live) {
      uint64_t acc = 0;
      if (S.kind != OUT_KEY && S.acc >= 0) {
        const AccSpec &A = P.accs[S.acc];
        acc = A.width == 8 ? ((const uint64_t *)(T.base + A.offset))[s] : ((const uint32_t *)(T.base + A.offset))[s];
      }
      const uint32_t cnt_all = S.cnt_all >= 0 ? ((const uint32_t *)(T.base + P.accs[S.cnt_all].offset))[s] : 0;
      const uint32_t cnt_valid = S.cnt_valid >= 0 ? ((const uint32_t *)(T.base + P.accs[S.cnt_valid].offset))[s] : cnt_all;
      if (S.kind == OUT_KEY) {
        if (KG->enabled) { // keys_sliced: take the original key column at the group's first row
          const KeySpec &K = KG->okeys[S.acc];
          const int64_t row = ((const uint32_t *)(T.base + P.accs[KG->first_acc].offset))[s];
          valid = bit_valid(K.validity, K.boff, row);
          store_by_dtype_int(S.values, K.dtype, i, valid ? key_denorm(load_key_norm(K.values, K.dtype, row), K.dtype) : 0);
        } else {
          // only the hashed part can hold the reserved sentinel / null slots
          valid = finalize_key(P, S, packed, P.single_key && s == T.cap + 1, P.single_key && s == T.cap, i);
        }
      } else {
        valid = finalize_value(S, acc, cnt_all, cnt_valid, i);
      }
    }
    if (S.validity) {
      const unsigned long long m = __ballot(valid);
      if ((threadIdx.x & 63) == 0 && (i & ~63ull) < G) ((uint64_t *)S.validity)[i >> 6] = m;
    }
  }
}

void launch_finalize(const Plan &p, TableRef t, const OutPlan *o, const KeyGather *kg, const uint64_t *slots,
                     uint64_t n_groups, hipStream_t s) {
  if (!n_groups) return;
  hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, p, t, o, kg, slots, n_groups);
}

// ---------------------------------------------------------------------------------------------------------
// dense ids for wide key tuples
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rank_scatter(const uint64_t *slots, uint64_t n, uint32_t *rank) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank[slots[i]] = (uint32_t)i;
}
void launch_rank_scatter(const uint64_t *slots, uint64_t n, uint32_t *rank, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_rank_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slots, n, rank);
}

__global__ void __launch_bounds__(256) k_encode_rows(Plan P, TableRef T, const uint32_t *rank, uint32_t *ids) {
  uint64_t *keys = (uint64_t *)T.base;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < P.n_rows; row += gs) {
    bool isnull;
    const uint64_t key = row_packed_key(P, row, isnull);
    const uint64_t slot = gtable_slot(keys, T.cap, T.flags, key, isnull); // every key is present: lookup only
    ids[row] = slot == ~0ull ? 0u : rank[slot];
  }
}
void launch_encode_rows(const Plan &p, TableRef t, const uint32_t *rank, uint32_t *ids, hipStream_t s) {
  int64_t blocks = (p.n_rows + 255) / 256;
  int grid = (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
  hipLaunchKernelGGL(k_encode_rows, dim3(grid), dim3(256), 0, s, p, t, rank, ids);
}

// ---------------------------------------------------------------------------------------------------------
// key range pre-pass (multi-key tuples)
// ---------------------------------------------------------------------------------------------------------
// 4- and 8-byte columns without nulls stream as 16-byte vectors, four requests in flight per lane (the scalar loop below ran
// at 2.8 TB/s: one 8-byte load per lane per trip); everything else (narrow types, validity bitmaps) keeps the scalar loop.
typedef uint32_t mm_u32x4 __attribute__((ext_vector_type(4)));
DEVINL void mm_fold(uint64_t v, uint64_t &lo, uint64_t &hi) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
template <int W>
DEVINL void mm_fold_vec(const mm_u32x4 &q, uint64_t x, uint64_t &lo, uint64_t &hi) {
  if (W == 8) {
    mm_fold(((uint64_t)q.x | ((uint64_t)q.y << 32)) ^ x, lo, hi);
    mm_fold(((uint64_t)q.z | ((uint64_t)q.w << 32)) ^ x, lo, hi);
  } else {
    mm_fold((uint64_t)(q.x ^ (uint32_t)x), lo, hi); mm_fold((uint64_t)(q.y ^ (uint32_t)x), lo, hi);
    mm_fold((uint64_t)(q.z ^ (uint32_t)x), lo, hi); mm_fold((uint64_t)(q.w ^ (uint32_t)x), lo, hi);
  }
}
template <int W>
DEVINL void mm_stream(const uint8_t *values, int64_t n_rows, uint64_t x, uint64_t &lo, uint64_t &hi) {
  const mm_u32x4 *p = (const mm_u32x4 *)values;
  const int64_t nv = n_rows * W / 16; // whole vectors; the last rows are block 0's
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // a workgroup takes 4 * blockDim CONSECUTIVE vectors per trip (16 KB at 256 threads): a lane's four requests sit 4 KB apart.  (With
  // the four a whole grid stride apart -- 4 MB at 1 024 workgroups -- config 3's three 800 MB key columns streamed at 4.9 TB/s, 492 us;
  // an odd workgroup count 394 us, this shape 347 us = 6.9 TB/s; plain instead of non-temporal loads 443 us.)  Vectors
  // [n_tiles * tile, nv) are left to the grid-stride loop below.
  const int64_t tile = (int64_t)blockDim.x * 4, n_tiles = nv / tile;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const mm_u32x4 *q = p + t * tile + threadIdx.x; // the last one read is q[3 * blockDim.x] = p[(t + 1) * tile - blockDim.x + threadIdx.x], < nv
    const mm_u32x4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + blockDim.x);
    const mm_u32x4 c = __builtin_nontemporal_load(q + 2 * blockDim.x), d = __builtin_nontemporal_load(q + 3 * blockDim.x);
    mm_fold_vec<W>(a, x, lo, hi); mm_fold_vec<W>(b, x, lo, hi); mm_fold_vec<W>(c, x, lo, hi); mm_fold_vec<W>(d, x, lo, hi);
  }
  for (int64_t i = n_tiles * tile + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) mm_fold_vec<W>(p[i], x, lo, hi);
  if (blockIdx.x == 0) {
    const int64_t r = nv * (16 / W) + threadIdx.x;
    if (r < n_rows) mm_fold(W == 8 ? ((const uint64_t *)values)[r] ^ x : (uint64_t)(((const uint32_t *)values)[r] ^ (uint32_t)x), lo, hi);
  }
}

// Every workgroup writes ONE (~min, max) pair per key column to `partials`; k_key_minmax_final folds them.  (One atomic pair per
// wave on the result words themselves -- 8 192 waves on two addresses -- took 190 us at 1e7 rows: same-address atomics retire
// at ~88 per us.)
#define MM_GRID 1024
__global__ void __launch_bounds__(256) k_key_minmax(Plan P, uint64_t *partials) {
  __shared__ uint64_t sh[4][2];
  const int k = blockIdx.y;
  const KeySpec &K = P.keys[k];
  uint64_t lo = ~0ull, hi = 0;
  const uint32_t dt = K.dtype;
  const bool vec = !K.validity && (((uintptr_t)K.values) & 15) == 0; // (uniform over the workgroup)
  if (vec && (dt == PLHIP_I64 || dt == PLHIP_U64 || dt == PLHIP_F64)) {
    mm_stream<8>(K.values, P.n_rows, dt == PLHIP_I64 ? SIGN64 : 0ull, lo, hi);
  } else if (vec && (dt == PLHIP_I32 || dt == PLHIP_U32 || dt == PLHIP_F32)) {
    mm_stream<4>(K.values, P.n_rows, dt == PLHIP_I32 ? 0x80000000ull : 0ull, lo, hi);
  } else {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < P.n_rows; row += stride) {
      if (!bit_valid(K.validity, K.boff, row)) continue;
      mm_fold(load_key_norm(K.values, K.dtype, row), lo, hi);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = lo; sh[threadIdx.x >> 6][1] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { lo = sh[w][0] < lo ? sh[w][0] : lo; hi = sh[w][1] > hi ? sh[w][1] : hi; }
    uint64_t *p = partials + ((uint64_t)k * gridDim.x + blockIdx.x) * 2;
    p[0] = lo; p[1] = hi; // (lo > hi: this workgroup saw no valid row)
  }
}
// minmax[2k] = ~min (0 when no row is valid), minmax[2k + 1] = max: the host's measure_key_columns reads them
// host_word (may be nullptr; one key column of at most 32 bits): the pair also goes to the host as ONE 64-bit word in pinned memory,
// min << 32 | max (1 << 32 | 0, i.e. min > max, when no row is valid) -- one relaxed system-scope store, nothing to order
__global__ void __launch_bounds__(256) k_key_minmax_final(const uint64_t *__restrict__ partials, uint32_t n_blocks, uint64_t *minmax,
                                                          uint64_t *host_word) {
  __shared__ uint64_t sh[4][2];
  const int k = blockIdx.x;
  uint64_t lo = ~0ull, hi = 0;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) {
    const uint64_t l = partials[((uint64_t)k * n_blocks + b) * 2], h = partials[((uint64_t)k * n_blocks + b) * 2 + 1];
    if (l <= h) { lo = l < lo ? l : lo; hi = h > hi ? h : hi; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = lo; sh[threadIdx.x >> 6][1] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { lo = sh[w][0] < lo ? sh[w][0] : lo; hi = sh[w][1] > hi ? sh[w][1] : hi; }
    minmax[2 * k] = lo <= hi ? ~lo : 0ull;
    minmax[2 * k + 1] = lo <= hi ? hi : 0ull;
    if (host_word && k == 0)
      __hip_atomic_store(host_word, lo <= hi ? ((lo & 0xFFFFFFFFull) << 32) | (hi & 0xFFFFFFFFull) : (1ull << 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

size_t key_minmax_scratch_bytes(int n_keys) { return (size_t)n_keys * MM_GRID * 16; }

// partials: key_minmax_scratch_bytes(p.n_keys) bytes of scratch
void launch_key_minmax(const Plan &p, uint64_t *minmax, uint64_t *partials, hipStream_t s, uint64_t *host_word) {
  int64_t blocks = (p.n_rows + 4095) / 4096;
  // (an odd count once the column is long: the workgroups' trips are then not a power of two apart)
  int grid = (int)(blocks < 1 ? 1 : (blocks > MM_GRID - 3 ? MM_GRID - 3 : blocks));
  hipLaunchKernelGGL(k_key_minmax, dim3(grid, p.n_keys), dim3(256), 0, s, p, partials);
  hipLaunchKernelGGL(k_key_minmax_final, dim3(p.n_keys), dim3(256), 0, s, partials, (uint32_t)grid, minmax, host_word);
}

// ---------------------------------------------------------------------------------------------------------
// rechunk: a chunk's validity bits copied to bit position dst_off of the single-chunk bitmap (zeroed before)
// ---------------------------------------------------------------------------------------------------------
// A thread builds one destination 32-bit word from up to 32 source bits (src == nullptr: the chunk has no nulls, all ones).
// Words that the chunk only partly covers (its first and last) are shared with the neighbouring chunks: atomicOr.
__global__ void __launch_bounds__(256) k_copy_bits(uint32_t *__restrict__ dst, int64_t dst_off, const uint8_t *__restrict__ src, int64_t src_off,
                                                   int64_t n) {
  const int64_t w0 = dst_off >> 5, w1 = (dst_off + n + 31) >> 5; // destination words touched
  const int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= w1) return;
  const int64_t b0 = w << 5;
  const int64_t lo = b0 > dst_off ? b0 : dst_off, hi = b0 + 32 < dst_off + n ? b0 + 32 : dst_off + n; // destination bits [lo, hi)
  uint32_t word = 0;
  for (int64_t b = lo; b < hi; b++) {
    const int64_t sb = src_off + (b - dst_off);
    const uint32_t bit = src ? (src[sb >> 3] >> (sb & 7)) & 1u : 1u;
    word |= bit << (uint32_t)(b - b0);
  }
  if (hi - lo == 32) dst[w] = word;
  else if (word) atomicOr(&dst[w], word);
}
namespace {
__global__ void __launch_bounds__(256) k_unpack_bool(const uint8_t *bits, int64_t bit_off, int64_t n, void *out, uint32_t width) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t b = bit_off + i;
  const uint32_t v = (bits[b >> 3] >> (b & 7)) & 1u;
  if (width == 1) ((uint8_t *)out)[i] = (uint8_t)v; else ((uint32_t *)out)[i] = v;
}
__global__ void __launch_bounds__(256) k_pack_bool(const void *src, uint32_t width, int64_t n, uint64_t *dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; // (a wave covers 64 consecutive rows: one output word)
  const bool v = i < n && (width == 1 ? ((const uint8_t *)src)[i] != 0 : ((const uint32_t *)src)[i] != 0u);
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (i & ~63ll) < n) dst[i >> 6] = m;
}
} // namespace

void launch_unpack_bool(const uint8_t *bits, int64_t bit_off, int64_t n, void *out, uint32_t width, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_unpack_bool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bits, bit_off, n, out, width);
}

void launch_pack_bool(const void *src, uint32_t width, int64_t n, uint64_t *dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pack_bool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, width, n, dst);
}

void launch_copy_bits(uint32_t *dst, int64_t dst_off, const uint8_t *src, int64_t src_off, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  const int64_t words = ((dst_off + n + 31) >> 5) - (dst_off >> 5);
  hipLaunchKernelGGL(k_copy_bits, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, dst, dst_off, src, src_off, n);
}

// ---------------------------------------------------------------------------------------------------------
// VecHash (crates/polars-core/src/hashing/vector_hasher.rs:78-139, 467-483)
// ---------------------------------------------------------------------------------------------------------
// A Boolean key column (PLHIP_BOOL: `values` is a bitmap; KeySpec::minv carries the bit offset of element 0) hashes to one of the
// caller's constants and, behind the first column, combines with _boost_hash_combine (hashing/mod.rs:20-22) instead of
// folded_multiply: VecHash for BooleanChunked, vector_hasher.rs:241-302.
DEVINL uint64_t boost_hash_combine(uint64_t l, uint64_t r) { return l ^ (r + (0x9e3779b9ull + (l << 6) + (r >> 2))); }
__global__ void __launch_bounds__(256) k_vec_hash(Plan P, uint64_t null_h, uint64_t true_h, uint64_t false_h, uint64_t *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < P.n_rows; row += stride) {
    uint64_t h = 0;
    for (int k = 0; k < P.n_keys; k++) {
      const KeySpec &K = P.keys[k];
      const bool valid = bit_valid(K.validity, K.boff, row);
      if (K.dtype == PLHIP_BOOL) {
        const int64_t b = (int64_t)K.minv + row;
        const uint64_t l = !valid ? null_h : (((K.values[b >> 3] >> (b & 7)) & 1) ? true_h : false_h);
        h = k == 0 ? l : boost_hash_combine(l, h);
      } else if (k == 0) h = valid ? folded_multiply(load_as_u64(K.values, K.dtype, row), PLHIP_MULTIPLE) : null_h;
      else h = folded_multiply((valid ? load_as_u64(K.values, K.dtype, row) : null_h) ^ h, PLHIP_MULTIPLE);
    }
    out[row] = h;
  }
}

void launch_vec_hash(const Plan &p, uint64_t null_hash, uint64_t true_hash, uint64_t false_hash, uint64_t *out, hipStream_t s) {
  int64_t blocks = (p.n_rows + 255) / 256;
  int grid = (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
  hipLaunchKernelGGL(k_vec_hash, dim3(grid), dim3(256), 0, s, p, null_hash, true_hash, false_hash, out);
}

// ---------------------------------------------------------------------------------------------------------
// two-step API: GroupsProxy::Idx as CSR, and Series::agg_* over it
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_iota_u32(uint32_t *out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint32_t)i;
}

// ids_sorted[] ascending: offsets[g] = first position of id g (every id in [0, G) occurs), offsets[G] = n
__global__ void __launch_bounds__(256) k_csr_offsets(const uint32_t *__restrict__ ids_sorted, uint64_t n, uint64_t G,
                                                     uint32_t *__restrict__ offsets) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (i == 0 || ids_sorted[i] != ids_sorted[i - 1])) offsets[ids_sorted[i]] = (uint32_t)i;
  if (i == 0) offsets[G] = (uint32_t)n;
}

__global__ void __launch_bounds__(256) k_csr_first(const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ indices, uint64_t G,
                                                   uint32_t *__restrict__ first) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) first[g] = indices[offsets[g]];
}

hipError_t sort_pairs_u32(uint32_t *keys, uint32_t *vals, uint64_t n, int end_bit, hipStream_t s) {
  if (n < 2) return hipSuccess;
  uint32_t *kout = nullptr, *vout = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipMalloc(&kout, n * 4);
  if (e != hipSuccess) return e;
  e = hipMalloc(&vout, n * 4);
  if (e != hipSuccess) { hipFree(kout); return e; }
  e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, kout, vals, vout, n, 0, end_bit, s);
  if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, kout, vals, vout, n, 0, end_bit, s); // stable
  if (e == hipSuccess) e = hipMemcpyAsync(keys, kout, n * 4, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(vals, vout, n * 4, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(kout); hipFree(vout); hipFree(tmp);
  return e;
}

// The same sorts on caller-supplied buffers: no allocation, no copy back, no host wait (the hipMalloc / hipFree / synchronise
// of the forms above cost 10 of plhip_group_tuples' 12.6 ms at 1e7 rows).  `tmp` holds sort_pairs_tmp_bytes(n) bytes.
size_t sort_pairs_tmp_bytes(uint64_t n) {
  size_t a = 0, b = 0;
  if (rocprim::radix_sort_pairs(nullptr, a, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 32) != hipSuccess) a = 0;
  if (rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, n, 0, 64) != hipSuccess) b = 0;
  return (a > b ? a : b) + 256;
}
hipError_t sort_pairs_u32_ws(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int end_bit, void *tmp,
                             size_t tmp_bytes, hipStream_t s) { // stable
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}
hipError_t sort_pairs_u64_ws(const uint64_t *kin, uint64_t *kout, const uint64_t *vin, uint64_t *vout, uint64_t n, int end_bit, void *tmp,
                             size_t tmp_bytes, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}

void launch_iota_u32(uint32_t *out, uint64_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n);
}
void launch_csr(const uint32_t *ids_sorted, const uint32_t *indices, uint64_t n, uint64_t G, uint32_t *offsets, uint32_t *first, hipStream_t s) {
  hipLaunchKernelGGL(k_csr_offsets, dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)), dim3(256), 0, s, ids_sorted, n, G, offsets);
  if (G) hipLaunchKernelGGL(k_csr_first, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, offsets, indices, G, first);
}

// One accumulator state folded sequentially (registers, no atomics): the same AccSpec semantics as the hash
// tables, applied in the order the rows are visited.
DEVINL void acc_fold_local(const AccSpec &a, uint64_t &state, uint64_t operand) {
  switch (a.op) {
    case ACC_ADD_U64: state += operand; break;
    case ACC_ADD_F64: state = (uint64_t)__double_as_longlong(__longlong_as_double((long long)state) + __longlong_as_double((long long)operand)); break;
    case ACC_MIN_I64: state = (long long)operand < (long long)state ? operand : state; break;
    case ACC_MAX_I64: state = (long long)operand > (long long)state ? operand : state; break;
    case ACC_MIN_U64: state = operand < state ? operand : state; break;
    case ACC_MAX_U64: state = operand > state ? operand : state; break;
    case ACC_CNT_ALL: case ACC_CNT_VALID: state += operand; break;
    case ACC_FIRST_ROW: state = operand < state ? operand : state; break;
  }
}

// Series::agg_{sum,mean,min,max} / group_count on GroupsProxy::Idx (aggregations/mod.rs:459-779,941-1016;
// take_agg/mod.rs:17-109): WAVE = false: one thread walks a group's index list front to back, the
// reference's left fold; WAVE = true (long groups): a wave walks it 64 rows at a time and combines the
// lane partials in a fixed butterfly order.
template <bool WAVE>
__global__ void __launch_bounds__(256) k_agg_groups(Plan P, OutPlan O, const uint32_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ indices, uint64_t G, uint32_t long_min) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t g = WAVE ? t >> 6 : t;
  const uint32_t lane = threadIdx.x & 63;
  // groups of more than long_min rows belong to k_agg_groups_long (several workgroups each)
  const bool live = g < G && offsets[g + 1] - offsets[g] <= long_min;
  uint64_t st[4];
  for (int a = 0; a < 4; a++) st[a] = a < P.n_accs ? P.accs[a].init : 0;
  if (live) {
    const uint32_t b = offsets[g], e = offsets[g + 1];
    if (WAVE) { // four gathers in flight per lane; a lane still folds its rows i, i + 64, i + 128, ... in that order
      for (uint32_t i0 = b + lane; i0 < e; i0 += 256) {
        int64_t row[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t i = i0 + 64u * (uint32_t)k; ok[k] = i < e; row[k] = indices[ok[k] ? i : b]; }
        for (int a = 0; a < P.n_accs && a < 4; a++) {
          uint64_t operand[4];
          bool v[4];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = acc_operand(P, P.accs[a], row[k], operand[k]) && ok[k];
#pragma unroll
          for (int k = 0; k < 4; k++) if (v[k]) acc_fold_local(P.accs[a], st[a], operand[k]);
        }
      }
    } else {
      for (uint32_t i = b; i < e; i++) {
        const int64_t row = indices[i];
        for (int a = 0; a < P.n_accs && a < 4; a++) {
          uint64_t operand;
          if (acc_operand(P, P.accs[a], row, operand)) acc_fold_local(P.accs[a], st[a], operand);
        }
      }
    }
  }
  if (WAVE) {
    for (int off = 32; off > 0; off >>= 1)
      for (int a = 0; a < P.n_accs && a < 4; a++) {
        const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(st[a] >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)st[a], off);
        acc_fold_local(P.accs[a], st[a], other);
      }
    if (lane != 0) return;
  }
  if (!live) return; // WAVE = false: a wave covers an aligned span of 64 groups below
  const OutSpec &S = O.outs[0];
  const uint32_t cnt_all = S.cnt_all >= 0 ? (uint32_t)st[S.cnt_all] : 0;
  const uint32_t cnt_valid = S.cnt_valid >= 0 ? (uint32_t)st[S.cnt_valid] : cnt_all;
  const bool valid = finalize_value(S, S.acc >= 0 ? st[S.acc] : 0, cnt_all, cnt_valid, g);
  if (S.validity) atomicOr((unsigned int *)S.validity + (g >> 5), valid ? 1u << (g & 31) : 0u);
}

// first / last / var / std over CSR groups.  first / last: the value (null or not) at the group's
// first / last index (crates/polars-core/src/frame/group_by/aggregations/agg_list.rs and dispatch.rs: take by
// `first` / the last index of `all`).  var / std: online_variance (crates/polars-arrow/src/kernels/take_agg/
// var.rs:11-40), nulls skipped, values as f64, None when count <= ddof; a thread walks a short group in index order
// (the reference's own update sequence), a wave walks a long one 64 rows at a time and merges the lanes' (count,
// mean, M2) triples pairwise (Chan et al.).
DEVINL double value_as_f64(const uint8_t *p, uint32_t dtype, int64_t row) {
  switch (dtype) {
    case PLHIP_I8: return (double)((const int8_t *)p)[row];
    case PLHIP_I16: return (double)((const int16_t *)p)[row];
    case PLHIP_I32: return (double)((const int32_t *)p)[row];
    case PLHIP_I64: return (double)((const int64_t *)p)[row];
    case PLHIP_U8: return (double)p[row];
    case PLHIP_U16: return (double)((const uint16_t *)p)[row];
    case PLHIP_U32: return (double)((const uint32_t *)p)[row];
    case PLHIP_U64: return (double)((const uint64_t *)p)[row];
    case PLHIP_F32: return (double)((const float *)p)[row];
    default: return ((const double *)p)[row];
  }
}

template <bool WAVE>
__global__ void __launch_bounds__(256) k_agg_groups_ext(AggExtArgs A, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ indices,
                                                        uint64_t G, uint32_t long_min) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t g = WAVE ? t >> 6 : t;
  const uint32_t lane = threadIdx.x & 63;
  const bool moments = A.kind == AGG_EXT_VAR || A.kind == AGG_EXT_STD;
  const bool live = g < G && (!moments || offsets[g + 1] - offsets[g] <= long_min); // (first / last read one row whatever the length)
  const uint32_t b = live ? offsets[g] : 0, e = live ? offsets[g + 1] : 0;
  bool valid = false;
  if (A.kind == AGG_EXT_FIRST || A.kind == AGG_EXT_LAST) {
    if (WAVE && lane != 0) return;
    if (!live) return;
    if (e > b) {
      const int64_t row = indices[A.kind == AGG_EXT_FIRST ? b : e - 1];
      valid = bit_valid(A.validity, A.boff, row);
      const uint32_t w = A.width;
      for (uint32_t i = 0; i < w; i++) A.out[g * w + i] = valid ? A.values[(uint64_t)row * w + i] : 0;
    } else {
      for (uint32_t i = 0; i < A.width; i++) A.out[g * A.width + i] = 0;
    }
  } else {
    double mean = 0.0, m2 = 0.0, cnt = 0.0;
    for (uint32_t i = b + (WAVE ? lane : 0); i < e; i += (WAVE ? 64 : 1)) {
      const int64_t row = indices[i];
      if (!bit_valid(A.validity, A.boff, row)) continue;
      const double v = value_as_f64(A.values, A.dtype, row);
      cnt += 1.0;
      const double d1 = v - mean;
      mean = d1 / cnt + mean;
      m2 += d1 * (v - mean);
    }
    if (WAVE) {
      for (int off = 32; off > 0; off >>= 1) {
        const double c2 = __shfl_xor(cnt, off), mean2 = __shfl_xor(mean, off), m22 = __shfl_xor(m2, off);
        const double ct = cnt + c2;
        if (ct > 0.0) {
          const double d = mean2 - mean;
          m2 = m2 + m22 + d * d * (cnt * c2 / ct);
          mean = mean + d * (c2 / ct);
        }
        cnt = ct;
      }
      if (lane != 0) return;
    }
    if (!live) return;
    valid = e > b && cnt > (double)A.ddof;
    double r = valid ? m2 / (cnt - (double)A.ddof) : 0.0;
    if (A.kind == AGG_EXT_STD) r = sqrt(r);
    if (A.out_dtype == PLHIP_F32) ((float *)A.out)[g] = (float)r;
    else ((double *)A.out)[g] = r;
  }
  if (valid) atomicOr((unsigned int *)A.out_validity + (g >> 5), 1u << (g & 31)); // zeroed bitmap
}

// ---- long groups (skew): a group of more than AGL_CHUNK rows is cut into chunks, one workgroup each; the workgroup that
// finishes last folds the chunks' partial states in chunk order (a fixed order: the result does not depend on timing) and
// writes the group's output row.  One wave -- or, for short average groups, one THREAD -- walking a 3e6-row group alone
// took 35 ms (sum) at 1e7 rows; the fused paths split their hot leaves the same way (fast_path.hip).
struct LongItem { uint32_t g, chunk, nchunks, base; };

__global__ void __launch_bounds__(256) k_long_items(const uint32_t *__restrict__ offsets, uint64_t G, uint32_t long_min, uint32_t ch, LongItem *items,
                                                    uint32_t *n_items, uint32_t max_items) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t len = offsets[g + 1] - offsets[g];
  if (len <= long_min) return;
  const uint32_t nch = (len + ch - 1) / ch; // <= 2 * len / ch: the items of all long groups fit 2 * n / ch
  const uint32_t base = atomicAdd(n_items, nch);
  if (base + nch > max_items) return; // (cannot happen; a group left out here would lose its row, so the bound is checked on the host side too)
  for (uint32_t c = 0; c < nch; c++) items[base + c] = LongItem{(uint32_t)g, c, nch, base};
}

DEVINL uint64_t ld_agent(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(256) k_agg_groups_long(Plan P, OutPlan O, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ indices,
                                                         const LongItem *__restrict__ items, const uint32_t *__restrict__ n_items, uint64_t *partial,
                                                         uint32_t *done, uint32_t ch) {
  __shared__ uint64_t sh[4][4];
  __shared__ uint32_t s_last;
  if (blockIdx.x >= *n_items) return;
  const LongItem it = items[blockIdx.x];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t gb = offsets[it.g], ge = offsets[it.g + 1];
  const uint32_t b = gb + it.chunk * ch, e = b + ch < ge ? b + ch : ge;
  uint64_t st[4];
  for (int a = 0; a < 4; a++) st[a] = a < P.n_accs ? P.accs[a].init : 0;
  for (uint32_t i0 = b + threadIdx.x; i0 < e; i0 += 1024) { // four gathers in flight per lane
    int64_t row[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = i0 + 256u * (uint32_t)k; ok[k] = i < e; row[k] = indices[ok[k] ? i : b]; }
    for (int a = 0; a < P.n_accs && a < 4; a++) {
      uint64_t operand[4];
      bool v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = acc_operand(P, P.accs[a], row[k], operand[k]) && ok[k];
#pragma unroll
      for (int k = 0; k < 4; k++) if (v[k]) acc_fold_local(P.accs[a], st[a], operand[k]);
    }
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int a = 0; a < P.n_accs && a < 4; a++) {
      const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(st[a] >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)st[a], off);
      acc_fold_local(P.accs[a], st[a], other);
    }
  if (lane == 0) for (int a = 0; a < 4; a++) sh[wave][a] = st[a];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) for (int a = 0; a < P.n_accs && a < 4; a++) acc_fold_local(P.accs[a], st[a], sh[w][a]);
    for (int a = 0; a < 4; a++) partial[(uint64_t)blockIdx.x * 4 + a] = st[a];
    __threadfence(); // the chunk's state is visible device-wide before the chunk counts as done
    s_last = atomicAdd(&done[it.base], 1u) == it.nchunks - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  __threadfence(); // the other chunks' states, not stale lines of this XCD's L2
  for (int a = 0; a < 4; a++) st[a] = a < P.n_accs ? P.accs[a].init : 0;
  for (uint32_t c = 0; c < it.nchunks; c++)
    for (int a = 0; a < P.n_accs && a < 4; a++) acc_fold_local(P.accs[a], st[a], ld_agent(&partial[(uint64_t)(it.base + c) * 4 + a]));
  const OutSpec &S = O.outs[0];
  const uint32_t cnt_all = S.cnt_all >= 0 ? (uint32_t)st[S.cnt_all] : 0;
  const uint32_t cnt_valid = S.cnt_valid >= 0 ? (uint32_t)st[S.cnt_valid] : cnt_all;
  const bool valid = finalize_value(S, S.acc >= 0 ? st[S.acc] : 0, cnt_all, cnt_valid, it.g);
  if (S.validity) atomicOr((unsigned int *)S.validity + (it.g >> 5), valid ? 1u << (it.g & 31) : 0u);
}

DEVINL void chan_merge(double &cnt, double &mean, double &m2, double c2, double mean2, double m22) {
  const double ct = cnt + c2;
  if (ct > 0.0) {
    const double d = mean2 - mean;
    m2 = m2 + m22 + d * d * (cnt * c2 / ct);
    mean = mean + d * (c2 / ct);
  }
  cnt = ct;
}

__global__ void __launch_bounds__(256) k_agg_groups_ext_long(AggExtArgs A, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ indices,
                                                             const LongItem *__restrict__ items, const uint32_t *__restrict__ n_items, uint64_t *partial,
                                                             uint32_t *done, uint32_t ch) {
  __shared__ double sh[4][3];
  __shared__ uint32_t s_last;
  if (blockIdx.x >= *n_items) return;
  const LongItem it = items[blockIdx.x];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t gb = offsets[it.g], ge = offsets[it.g + 1];
  const uint32_t b = gb + it.chunk * ch, e = b + ch < ge ? b + ch : ge;
  double mean = 0.0, m2 = 0.0, cnt = 0.0;
  for (uint32_t i = b + threadIdx.x; i < e; i += 256) {
    const int64_t row = indices[i];
    if (!bit_valid(A.validity, A.boff, row)) continue;
    const double v = value_as_f64(A.values, A.dtype, row);
    cnt += 1.0;
    const double d1 = v - mean;
    mean = d1 / cnt + mean;
    m2 += d1 * (v - mean);
  }
  for (int off = 32; off > 0; off >>= 1) chan_merge(cnt, mean, m2, __shfl_xor(cnt, off), __shfl_xor(mean, off), __shfl_xor(m2, off));
  if (lane == 0) { sh[wave][0] = cnt; sh[wave][1] = mean; sh[wave][2] = m2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) chan_merge(cnt, mean, m2, sh[w][0], sh[w][1], sh[w][2]);
    uint64_t *p = partial + (uint64_t)blockIdx.x * 4;
    p[0] = (uint64_t)__double_as_longlong(cnt); p[1] = (uint64_t)__double_as_longlong(mean); p[2] = (uint64_t)__double_as_longlong(m2);
    __threadfence();
    s_last = atomicAdd(&done[it.base], 1u) == it.nchunks - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  __threadfence();
  cnt = 0.0; mean = 0.0; m2 = 0.0;
  for (uint32_t c = 0; c < it.nchunks; c++) {
    const uint64_t *p = partial + (uint64_t)(it.base + c) * 4;
    chan_merge(cnt, mean, m2, __longlong_as_double((long long)ld_agent(p)), __longlong_as_double((long long)ld_agent(p + 1)),
               __longlong_as_double((long long)ld_agent(p + 2)));
  }
  const bool valid = cnt > (double)A.ddof;
  double r = valid ? m2 / (cnt - (double)A.ddof) : 0.0;
  if (A.kind == AGG_EXT_STD) r = sqrt(r);
  if (A.out_dtype == PLHIP_F32) ((float *)A.out)[it.g] = (float)r;
  else ((double *)A.out)[it.g] = r;
  if (valid) atomicOr((unsigned int *)A.out_validity + (it.g >> 5), 1u << (it.g & 31));
}

size_t agg_long_scratch_bytes(uint64_t n) {
  const uint64_t max_items = 2 * (n / AGL_CHUNK) + 2;
  return 256 + max_items * (sizeof(LongItem) + 32 + 4) + 256;
}

// scratch: n_items (u32, 256 B slot) | items | partial (4 u64 each) | done (u32 each)
template <typename F>
static void launch_long(const uint32_t *offsets, uint64_t n, uint64_t G, void *scratch, hipStream_t s, F launch) {
  const uint32_t max_items = (uint32_t)(2 * (n / AGL_CHUNK) + 2);
  uint8_t *sb = (uint8_t *)scratch;
  uint32_t *n_items = (uint32_t *)sb;
  LongItem *items = (LongItem *)(sb + 256);
  uint64_t *partial = (uint64_t *)(sb + 256 + (size_t)max_items * sizeof(LongItem));
  uint32_t *done = (uint32_t *)((uint8_t *)partial + (size_t)max_items * 32);
  (void)hipMemsetAsync(n_items, 0, 256, s);
  (void)hipMemsetAsync(done, 0, (size_t)max_items * 4, s);
  hipLaunchKernelGGL(k_long_items, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, offsets, G, (uint32_t)AGL_CHUNK, (uint32_t)AGL_CHUNK, items, n_items, max_items);
  launch(max_items, items, n_items, partial, done);
}

void launch_agg_groups_ext(const AggExtArgs &a, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, bool wave, void *long_scratch,
                           hipStream_t s) {
  if (!G) return;
  const bool split = long_scratch && n > AGL_CHUNK && (a.kind == AGG_EXT_VAR || a.kind == AGG_EXT_STD);
  const uint32_t long_min = split ? (uint32_t)AGL_CHUNK : 0xFFFFFFFFu;
  if (wave) hipLaunchKernelGGL(k_agg_groups_ext<true>, dim3((unsigned)((G * 64 + 255) / 256)), dim3(256), 0, s, a, offsets, indices, G, long_min);
  else hipLaunchKernelGGL(k_agg_groups_ext<false>, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, a, offsets, indices, G, long_min);
  if (split)
    launch_long(offsets, n, G, long_scratch, s, [&](uint32_t max_items, LongItem *items, uint32_t *n_items, uint64_t *partial, uint32_t *done) {
      hipLaunchKernelGGL(k_agg_groups_ext_long, dim3(max_items), dim3(256), 0, s, a, offsets, indices, items, n_items, partial, done, (uint32_t)AGL_CHUNK);
    });
}

void launch_agg_groups(const Plan &p, const OutPlan &o, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, bool wave,
                       void *long_scratch, hipStream_t s) {
  if (!G) return;
  const bool split = long_scratch && n > AGL_CHUNK && p.n_accs > 0; // (group_count reads no rows)
  const uint32_t long_min = split ? (uint32_t)AGL_CHUNK : 0xFFFFFFFFu;
  if (wave) hipLaunchKernelGGL(k_agg_groups<true>, dim3((unsigned)((G * 64 + 255) / 256)), dim3(256), 0, s, p, o, offsets, indices, G, long_min);
  else hipLaunchKernelGGL(k_agg_groups<false>, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, p, o, offsets, indices, G, long_min);
  if (split)
    launch_long(offsets, n, G, long_scratch, s, [&](uint32_t max_items, LongItem *items, uint32_t *n_items, uint64_t *partial, uint32_t *done) {
      hipLaunchKernelGGL(k_agg_groups_long, dim3(max_items), dim3(256), 0, s, p, o, offsets, indices, items, n_items, partial, done, (uint32_t)AGL_CHUNK);
    });
}
