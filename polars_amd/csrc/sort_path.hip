// sort_path.hip — the strategy for more groups than one partition level brings into LDS (up to all-unique
// keys, e.g. BASELINE configs 3 and 5), and for any skew: radix sort of (packed key, row) pairs, run
// boundaries, then one fold per run.
//
//   k_pack_rows      packed key of every row (the Plan's range-compressed tuple) + row ids
//   rocprim          stable LSD radix sort of the pairs on the used key bits (library sort: the pass count, not
//                    the kernel, is what this tier is about; HBM-bound)
//   k_run_flags      1 where a new key starts; inclusive scan -> group id; k_run_offsets -> CSR offsets
//   k_fold_sorted    one thread per run: gathers the value rows (ascending row order = the reference's fold
//                    order), folds every accumulator in registers, writes the output columns; long runs are
//                    queued and folded by one workgroup each (k_fold_sorted_long), so a heavy key cannot
//                    serialise on one lane
//
// No hash table, no atomics on the row path; cost independent of the number of groups.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "device.hpp"
#include "kernels.hpp"

namespace {

#define SP_BLOCK 256
#define SP_LONG 2048 // runs at least this long go to the per-workgroup kernel
#define SP_HUGE 131072u // ... and runs at least this long are shared by several workgroups

__global__ void __launch_bounds__(256) k_pack_rows(Plan P, uint64_t *__restrict__ keys, uint32_t *__restrict__ rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < P.n_rows; r += stride) {
    bool isn;
    keys[r] = row_packed_key(P, r, isn);
    rows[r] = (uint32_t)r;
  }
}

__global__ void __launch_bounds__(256) k_run_flags(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// gid = inclusive scan of the flags: run g (0-based) starts where flags[i] == 1 and gid[i] == g + 1
__global__ void __launch_bounds__(256) k_run_offsets(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ gid, uint64_t n,
                                                     uint32_t *__restrict__ offsets, uint32_t *__restrict__ dflags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) offsets[gid[i] - 1] = (uint32_t)i;
  if (i == n - 1) { offsets[gid[i]] = (uint32_t)n; dflags[FLAG_NGROUPS] = gid[i]; }
}

// Sorted-key fast path (create_groups_from_sorted, crates/polars-core/src/frame/group_by/into_groups.rs:58-127;
// partition_to_groups, crates/polars-arrow/src/kernels/sort_partition.rs:71-116): a group is a maximal run of
// equal consecutive values, compared as VALUES (`val != first`: every NaN is a run of its own, -0.0 == 0.0), the
// nulls (contiguous at the front or the back of a sorted column) one run.
__global__ void __launch_bounds__(256) k_sorted_run_flags(KeySpec K, uint64_t n, uint32_t *__restrict__ flags, uint32_t *__restrict__ rows) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rows[i] = (uint32_t)i;
  bool fresh = i == 0;
  if (!fresh) {
    const bool va = bit_valid(K.validity, K.boff, (int64_t)i), vb = bit_valid(K.validity, K.boff, (int64_t)i - 1);
    if (va != vb) fresh = true;
    else if (va) {
      const uint64_t a = load_as_u64(K.values, K.dtype, (int64_t)i), b = load_as_u64(K.values, K.dtype, (int64_t)i - 1);
      if (K.dtype == PLHIP_F32) fresh = __uint_as_float((uint32_t)a) != __uint_as_float((uint32_t)b);
      else if (K.dtype == PLHIP_F64) fresh = __longlong_as_double((long long)a) != __longlong_as_double((long long)b);
      else fresh = a != b;
    }
  }
  flags[i] = fresh ? 1u : 0u;
}

DEVINL void fold_local(const AccSpec &a, uint64_t &state, uint64_t operand) {
  switch (a.op) {
    case ACC_ADD_U64: state += operand; break;
    case ACC_ADD_F64: state = (uint64_t)__double_as_longlong(__longlong_as_double((long long)state) + __longlong_as_double((long long)operand)); break;
    case ACC_ADD_F32: // the f32 fold of the reference (aggregations/mod.rs:719-724); state and operand are f64 images of f32 values
      state = (uint64_t)__double_as_longlong((double)((float)__longlong_as_double((long long)state) + (float)__longlong_as_double((long long)operand)));
      break;
    case ACC_MIN_I64: state = (long long)operand < (long long)state ? operand : state; break;
    case ACC_MAX_I64: state = (long long)operand > (long long)state ? operand : state; break;
    case ACC_MIN_U64: state = operand < state ? operand : state; break;
    case ACC_MAX_U64: state = operand > state ? operand : state; break;
    case ACC_CNT_ALL: case ACC_CNT_VALID: state += operand; break;
    case ACC_FIRST_ROW: state = operand < state ? operand : state; break;
  }
}

// accumulators of run g -> output row g (sst[a * SP_BLOCK + lane_slot] holds accumulator a)
// wave_rows: the caller's lanes hold CONSECUTIVE output rows g = (64-aligned base) + lane (k_fold_sorted without a group map): a
// column's validity bits of the wave then go out as two 32-bit ORs instead of one atomic per lane -- with about one row per group
// the per-lane form put 32 atomics on every word (the harness's q10 with NAs, 1e7 groups, six nullable key columns: 3.8 ms of a
// 5.3 ms query).  Still an OR into the zeroed bitmap: the long runs' rows are emitted by other kernels into the same words.
template <bool WAVE_ROWS = false>
DEVINL void emit_run(const Plan &P, const OutPlan &O, const KeyGather *KG, const uint64_t *sst, uint32_t slot, uint64_t packed,
                     uint32_t first_row, uint64_t g) {
  for (int o = 0; o < O.n_outs; o++) {
    const OutSpec &S = O.outs[o];
    bool valid;
    if (S.kind == OUT_KEY) {
      if (KG && KG->enabled) { // keys_sliced: the original key column at the group's first row
        const KeySpec &K = KG->okeys[S.acc];
        valid = bit_valid(K.validity, K.boff, first_row);
        store_by_dtype_int(S.values, K.dtype, g, valid ? key_denorm(load_key_norm(K.values, K.dtype, first_row), K.dtype) : 0);
      } else {
        valid = finalize_key(P, S, packed, false, false, g);
      }
    } else {
      const uint64_t acc = S.acc >= 0 ? sst[S.acc * SP_BLOCK + slot] : 0;
      const uint32_t cnt_all = S.cnt_all >= 0 ? (uint32_t)sst[S.cnt_all * SP_BLOCK + slot] : 0;
      const uint32_t cnt_valid = S.cnt_valid >= 0 ? (uint32_t)sst[S.cnt_valid * SP_BLOCK + slot] : cnt_all;
      valid = finalize_value(S, acc, cnt_all, cnt_valid, g);
    }
    if (WAVE_ROWS) {
      const unsigned long long m = __ballot(S.validity != nullptr && valid); // (lanes that left early are in neither mask)
      const unsigned long long act = __ballot(1);
      const uint32_t lane = threadIdx.x & 63u;
      const uint64_t w0 = (g - lane) >> 5; // the wave's rows are words w0 and w0 + 1 of the bitmap
      // each half's word is written by that half's first lane still here (lane 0 / 32 may have left: a long run, g >= G)
      const uint32_t lo = (uint32_t)act, hi = (uint32_t)(act >> 32);
      if (lo && lane == (uint32_t)__ffs((int)lo) - 1u && (uint32_t)m) atomicOr((unsigned int *)S.validity + w0, (uint32_t)m);
      if (hi && lane == 32u + (uint32_t)__ffs((int)hi) - 1u && (uint32_t)(m >> 32)) atomicOr((unsigned int *)S.validity + w0 + 1, (uint32_t)(m >> 32));
    } else if (S.validity && valid) atomicOr((unsigned int *)S.validity + (g >> 5), 1u << (g & 31)); // zeroed bitmap
  }
}

__global__ void __launch_bounds__(SP_BLOCK) k_fold_sorted(Plan P, OutPlan O, const KeyGather *__restrict__ KG,
                                                          const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ rows,
                                                          const uint32_t *__restrict__ offsets, uint64_t G,
                                                          uint32_t *__restrict__ long_list, uint32_t *__restrict__ long_count,
                                                          const uint32_t *__restrict__ gmap, uint32_t *__restrict__ long_done,
                                                          uint64_t *__restrict__ long_part, uint32_t max_runs) {
  extern __shared__ __align__(16) uint64_t sst[]; // [n_accs][SP_BLOCK]
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t b = offsets[g], e = offsets[g + 1];
  if (P.exact_fold && e - b >= SP_LONG) { // one wave folds the run in row order (k_fold_sorted_exact_long)
    long_list[atomicAdd(long_count, 1u)] = (uint32_t)g;
    return;
  }
  if (e - b >= SP_HUGE) { // queued from the list's far end, with the combined state of its chunks at the identities
    const uint32_t hi = atomicAdd(long_count + 1, 1u);
    long_list[max_runs - 1 - hi] = (uint32_t)g;
    long_done[hi] = 0u;
    for (int a = 0; a < P.n_accs; a++) long_part[(uint64_t)hi * P.n_accs + a] = P.accs[a].init;
    return;
  }
  if (e - b >= SP_LONG) {
    long_list[atomicAdd(long_count, 1u)] = (uint32_t)g;
    return;
  }
  // accumulators are ordered by source column: every value row is gathered ONCE and serves all of them
  for (int a = 0; a < P.n_accs; a++) sst[a * SP_BLOCK + threadIdx.x] = P.accs[a].init;
  for (int a0 = 0; a0 < P.n_accs;) {
    const int src = P.accs[a0].src;
    int a1 = a0 + 1;
    while (a1 < P.n_accs && P.accs[a1].src == src) a1++;
    for (uint32_t i = b; i < e; i += 4) {
      uint32_t r[4];
      uint32_t okm = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (i + k < e) okm |= 1u << k;
        r[k] = rows[i + k < e ? i + k : b];
      }
      uint64_t raw[4] = {0, 0, 0, 0};
      uint32_t dtype = 0;
      if (src >= 0) {
        const ValSpec V = P.vals[src];
        dtype = V.dtype;
        load_raw<4, false>(V.values, wbytes(V.dtype), r, raw); // four gathers in flight
        if (V.validity) {
#pragma unroll
          for (int k = 0; k < 4; k++) if (!val_valid(V, (int64_t)r[k])) okm &= ~(1u << k);
        }
      }
      for (int a = a0; a < a1; a++) {
        const AccSpec A = P.accs[a];
        uint64_t st = sst[a * SP_BLOCK + threadIdx.x];
        if (A.op == ACC_CNT_ALL || A.op == ACC_CNT_VALID) st += (uint64_t)__popc(okm);
        else if (A.op == ACC_FIRST_ROW) {
#pragma unroll
          for (int k = 0; k < 4; k++) if ((okm >> k) & 1) st = r[k] < st ? r[k] : st;
        } else {
          uint64_t opnd[4];
          raw_to_operands<4>(dtype, A.conv, raw, opnd);
#pragma unroll
          for (int k = 0; k < 4; k++) if ((okm >> k) & 1) fold_local(A, st, opnd[k]); // index order: the reference's fold order
          // the reference's 1-row shortcut (`arr.get(first)`, aggregations/mod.rs:631-633): the value itself, not 0.0 + value
          // (which would turn -0.0 into +0.0)
          if (P.exact_fold && e - b == 1 && (okm & 1) && (A.op == ACC_ADD_F64 || A.op == ACC_ADD_F32)) st = opnd[0];
        }
        sst[a * SP_BLOCK + threadIdx.x] = st;
      }
    }
    a0 = a1;
  }
  if (gmap) emit_run(P, O, KG, sst, threadIdx.x, skeys ? skeys[b] : 0ull, rows[b], gmap[g]); // stable (first-row) order: rows anywhere
  else emit_run<true>(P, O, KG, sst, threadIdx.x, skeys ? skeys[b] : 0ull, rows[b], g);
}

// Long runs (>= SP_LONG rows): one workgroup per run (lanes stride the index list, partial states are combined in a
// fixed tree).  Huge runs (>= SP_HUGE: a heavy hitter, the null keys) are cut into chunks of SP_LCHUNK rows that the
// workgroups share round-robin; a chunk's states go into the run's combined state with one global atomic per
// accumulator, and the chunk that finishes last writes the output row (f64 sums then add up in completion order).
#define SP_LCHUNK 16384u
DEVINL void combine_global(const AccSpec &a, uint64_t *p, uint64_t state) {
  switch (a.op) {
    case ACC_ADD_U64: case ACC_CNT_ALL: case ACC_CNT_VALID: atomicAdd((unsigned long long *)p, (unsigned long long)state); break;
    case ACC_ADD_F64: atomicAdd((double *)p, __longlong_as_double((long long)state)); break;
    case ACC_MIN_I64: atomicMin((long long *)p, (long long)state); break;
    case ACC_MAX_I64: atomicMax((long long *)p, (long long)state); break;
    case ACC_MIN_U64: case ACC_FIRST_ROW: atomicMin((unsigned long long *)p, (unsigned long long)state); break;
    case ACC_MAX_U64: atomicMax((unsigned long long *)p, (unsigned long long)state); break;
  }
}
// rows [cb, ce) of the sorted index list -> sst[a * SP_BLOCK] for every accumulator a (all threads of the workgroup)
DEVINL void reduce_rows(const Plan &P, const uint32_t *__restrict__ rows, uint64_t *sst, uint32_t cb, uint32_t ce) {
  for (int a = 0; a < P.n_accs; a++) {
    const AccSpec A = P.accs[a];
    uint64_t st = A.init;
    for (uint32_t i0 = cb + threadIdx.x; i0 < ce; i0 += 4 * SP_BLOCK) { // four gathers in flight
      uint32_t r[4];
      bool ok[4];
      uint64_t operand[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint32_t i = i0 + (uint32_t)k * SP_BLOCK; ok[k] = i < ce; r[k] = rows[ok[k] ? i : cb]; }
#pragma unroll
      for (int k = 0; k < 4; k++) ok[k] = acc_operand(P, A, (int64_t)r[k], operand[k]) && ok[k];
#pragma unroll
      for (int k = 0; k < 4; k++) if (ok[k]) fold_local(A, st, operand[k]);
    }
    uint64_t *col = sst + a * SP_BLOCK;
    col[threadIdx.x] = st;
    __syncthreads();
    for (uint32_t off = SP_BLOCK / 2; off > 0; off >>= 1) {
      if (threadIdx.x < off) {
        uint64_t mine = col[threadIdx.x];
        fold_local(A, mine, col[threadIdx.x + off]);
        col[threadIdx.x] = mine;
      }
      __syncthreads();
    }
  }
}
__global__ void __launch_bounds__(SP_BLOCK) k_fold_sorted_long(Plan P, OutPlan O, const KeyGather *__restrict__ KG,
                                                               const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ rows,
                                                               const uint32_t *__restrict__ offsets,
                                                               const uint32_t *__restrict__ long_list, const uint32_t *__restrict__ long_count,
                                                               const uint32_t *__restrict__ gmap, uint32_t *__restrict__ long_done,
                                                               uint64_t *__restrict__ long_part, uint32_t max_runs) {
  extern __shared__ __align__(16) uint64_t sst[]; // [n_accs][SP_BLOCK]
  const uint32_t count = long_count[0], n_huge = long_count[1];
  for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
    const uint64_t g = long_list[li];
    const uint32_t b = offsets[g], e = offsets[g + 1];
    reduce_rows(P, rows, sst, b, e);
    if (threadIdx.x == 0) emit_run(P, O, KG, sst, 0, skeys ? skeys[b] : 0ull, rows[b], gmap ? gmap[g] : g);
    __syncthreads();
  }
  uint32_t base = 0; // chunks of the huge runs before this one
  for (uint32_t hi = 0; hi < n_huge; hi++) {
    const uint64_t g = long_list[max_runs - 1 - hi];
    const uint32_t b = offsets[g], e = offsets[g + 1];
    const uint32_t C = (e - b + SP_LCHUNK - 1) / SP_LCHUNK;
    for (uint32_t c = (blockIdx.x + gridDim.x - base % gridDim.x) % gridDim.x; c < C; c += gridDim.x) {
      const uint32_t cb = b + c * SP_LCHUNK, ce = e - cb > SP_LCHUNK ? cb + SP_LCHUNK : e;
      reduce_rows(P, rows, sst, cb, ce);
      if (threadIdx.x == 0) {
        for (int a = 0; a < P.n_accs; a++) combine_global(P.accs[a], &long_part[(uint64_t)hi * P.n_accs + a], sst[a * SP_BLOCK]);
        __threadfence();
        if (atomicAdd(&long_done[hi], 1u) == C - 1) { // the last chunk of the run
          __threadfence();
          for (int a = 0; a < P.n_accs; a++)
            sst[a * SP_BLOCK] = __hip_atomic_load(&long_part[(uint64_t)hi * P.n_accs + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          emit_run(P, O, KG, sst, 0, skeys ? skeys[b] : 0ull, rows[b], gmap ? gmap[g] : g);
        }
      }
      __syncthreads();
    }
    base += C;
  }
}

// PLHIP_FLAG_EXACT_FOLD, runs of >= SP_LONG rows: ONE wave per run.  Per trip the lanes gather 64 consecutive rows of the run
// (coalesced index reads), then the float sums are folded lane after lane in row order -- every lane computes the same
// chain from the broadcast operands, so the state stays wave-uniform -- while the order-independent accumulators (integer
// sums, extrema, counts) fold per lane and meet in a xor-shuffle tree at the end.
__global__ void __launch_bounds__(SP_BLOCK) k_fold_sorted_exact_long(Plan P, OutPlan O, const KeyGather *__restrict__ KG,
                                                                     const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ rows,
                                                                     const uint32_t *__restrict__ offsets,
                                                                     const uint32_t *__restrict__ long_list, const uint32_t *__restrict__ long_count,
                                                                     const uint32_t *__restrict__ gmap) {
  extern __shared__ __align__(16) uint64_t sst[]; // [n_accs][SP_BLOCK]
  const uint32_t count = long_count[0];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = SP_BLOCK / 64;
  for (uint32_t li = blockIdx.x * waves + wave; li < count; li += gridDim.x * waves) {
    const uint64_t g = long_list[li];
    const uint32_t b = offsets[g], e = offsets[g + 1];
    for (int a = 0; a < P.n_accs; a++) {
      const AccSpec A = P.accs[a];
      const bool ordered = A.op == ACC_ADD_F64 || A.op == ACC_ADD_F32;
      uint64_t st = A.init;
      for (uint32_t i0 = b; i0 < e; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool ok = i < e;
        const uint32_t r = rows[ok ? i : b];
        uint64_t operand = 0;
        ok = acc_operand(P, A, (int64_t)r, operand) && ok;
        if (ordered) {
          unsigned long long m = __ballot(ok);
          while (m) { // ascending lanes = ascending rows of the run
            const int k = __ffsll((long long)m) - 1;
            const uint64_t v = ((uint64_t)(uint32_t)__shfl((int)(operand >> 32), k) << 32) | (uint32_t)__shfl((int)operand, k);
            fold_local(A, st, v);
            m &= m - 1;
          }
        } else if (ok) fold_local(A, st, operand);
      }
      if (!ordered) {
        for (int off = 32; off > 0; off >>= 1) {
          const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(st >> 32), off) << 32) | (uint32_t)__shfl_xor((int)st, off);
          fold_local(A, st, o);
        }
      }
      if (lane == 0) sst[a * SP_BLOCK + wave * 64] = st;
    }
    if (lane == 0) emit_run(P, O, KG, sst, wave * 64, skeys ? skeys[b] : 0ull, rows[b], gmap ? gmap[g] : g);
  }
}

} // namespace

// scratch bytes rocprim needs for the sort and the scan of n elements
hipError_t sort_path_temp_bytes(uint64_t n, int end_bit, size_t *bytes) {
  size_t a = 0, b = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, a, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, end_bit);
  if (e != hipSuccess) return e;
  e = rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, n, rocprim::plus<uint32_t>());
  *bytes = (a > b ? a : b) + 256;
  return e;
}

// keys_a/rows_a receive the packed rows; the sorted pairs end in keys_b/rows_b.  offsets: n + 1 u32.
// flags / gid: n u32 each.  On return dflags[FLAG_NGROUPS] = number of runs (read it after the stream).
hipError_t launch_sort_groups(const Plan &p, uint64_t *keys_a, uint64_t *keys_b, uint32_t *rows_a, uint32_t *rows_b, uint32_t *flags,
                              uint32_t *gid, uint32_t *offsets, void *temp, size_t temp_bytes, int end_bit, uint32_t *dflags, hipStream_t s) {
  const uint64_t n = (uint64_t)p.n_rows;
  int64_t blocks = (p.n_rows + 255) / 256;
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, s, p, keys_a, rows_a);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_a, keys_b, rows_a, rows_b, n, 0, end_bit, s); // stable
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_run_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys_b, n, flags);
  e = rocprim::inclusive_scan(temp, temp_bytes, flags, gid, n, rocprim::plus<uint32_t>(), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_run_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flags, gid, n, offsets, dflags);
  return hipGetLastError();
}

// ---- median / quantile over CSR groups (agg_median -> agg_quantile_generic(0.5, Linear), crates/polars-core/src/frame/group_by/
// aggregations/mod.rs:420-457 + dispatch.rs:98-113; quantile_idx / linear_interpol, chunked_array/ops/aggregate/
// quantile.rs:19-54): the group's valid values sorted ascending (NaN last), float_idx = (m - 1) / 2, result =
// lower + (float_idx - floor) * (upper - lower) in f64 (an f32 column's answer is that, cast); no valid value -> null.
// Values travel as order-preserving u64 codes of their f64 value in CSR order, nulls as the largest code, one
// segmented radix sort (rocPRIM) orders every group, one thread per group picks the middle.
namespace {
DEVINL double median_value(const uint8_t *p, uint32_t dtype, int64_t row) {
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
__global__ void __launch_bounds__(256) k_gather_ordered(AggExtArgs A, const uint32_t *__restrict__ indices, uint64_t n, uint64_t *__restrict__ codes) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = indices[i];
  uint64_t c = ~0ull; // null: behind everything
  if (bit_valid(A.validity, A.boff, row)) {
    double d = median_value(A.values, A.dtype, row);
    if (d != d) d = __longlong_as_double(0x7ff8000000000000ll); // one NaN, behind +inf
    c = f64_ordered(d);
  }
  codes[i] = c;
}
// ---- n_unique over CSR groups (agg_n_unique, dispatch.rs:82-107 -> ChunkUnique::n_unique, chunked_array/ops/unique/
// mod.rs:136-171: the group's values sorted, one per change of value with Option<T> / IEEE comparison): exact
// order-preserving u64 codes (integers by value, floats by f64 value with -0.0 folded onto 0.0 and one NaN code), nulls
// as the largest code, the same segmented sort, then every element that differs from its predecessor in the segment
// (or is a NaN: NaN != NaN) counts one for its group.  A real value whose code is the nulls' (u64::MAX / i64::MAX) would
// merge with them: both sightings are recorded per group and the count corrected.
#define NU_NAN_CODE 0xFFF8000000000000ull // f64_ordered(canonical NaN)
DEVINL uint64_t exact_code(const uint8_t *p, uint32_t dtype, int64_t row) {
  switch (dtype) {
    case PLHIP_I8: return (uint64_t)(int64_t)((const int8_t *)p)[row] ^ 0x8000000000000000ull;
    case PLHIP_I16: return (uint64_t)(int64_t)((const int16_t *)p)[row] ^ 0x8000000000000000ull;
    case PLHIP_I32: return (uint64_t)(int64_t)((const int32_t *)p)[row] ^ 0x8000000000000000ull;
    case PLHIP_I64: return (uint64_t)((const int64_t *)p)[row] ^ 0x8000000000000000ull;
    case PLHIP_U8: return p[row];
    case PLHIP_U16: return ((const uint16_t *)p)[row];
    case PLHIP_U32: return ((const uint32_t *)p)[row];
    case PLHIP_U64: return ((const uint64_t *)p)[row];
    default: {
      double d = dtype == PLHIP_F32 ? (double)((const float *)p)[row] : ((const double *)p)[row];
      if (d != d) return NU_NAN_CODE;
      if (d == 0.0) d = 0.0; // -0.0 == 0.0
      return f64_ordered(d);
    }
  }
}
DEVINL uint32_t segment_of(const uint32_t *__restrict__ offsets, uint64_t G, uint32_t i) { // last g with offsets[g] <= i
  uint64_t lo = 0, hi = G;
  while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] <= i) lo = mid; else hi = mid; }
  return (uint32_t)lo;
}
// marks[g] bit 0: the group holds a null, bit 1: it holds a value whose code is the nulls'
__global__ void __launch_bounds__(256) k_gather_exact(AggExtArgs A, const uint32_t *__restrict__ indices, const uint32_t *__restrict__ offsets,
                                                      uint64_t n, uint64_t G, uint64_t *__restrict__ codes, uint32_t *__restrict__ marks) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = indices[i];
  uint64_t c = ~0ull;
  uint32_t mark = 1u;
  if (bit_valid(A.validity, A.boff, row)) {
    c = exact_code(A.values, A.dtype, row);
    mark = c == ~0ull ? 2u : 0u;
  }
  codes[i] = c;
  if (mark) {
    const uint32_t g = segment_of(offsets, G, (uint32_t)i);
    const uint32_t bits = mark << ((g & 15u) * 2u);
    if ((marks[g >> 4] & bits) != bits) atomicOr(&marks[g >> 4], bits); // (bits only ever get set: a stale read costs an atomic, never a mark)
  }
}
__global__ void __launch_bounds__(256) k_count_changes(AggExtArgs A, const uint64_t *__restrict__ sorted, const uint32_t *__restrict__ offsets,
                                                       uint64_t n, uint64_t G, uint32_t is_float) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < n; // (no early return: the wave votes below)
  const uint64_t c = in ? sorted[i] : 0;
  bool counts = in && (i == 0 || sorted[i - 1] != c || (is_float && c == NU_NAN_CODE));
  uint32_t g = 0;
  bool have_g = false;
  if (in && !counts) { // equal to its predecessor: still the first of its segment?
    g = segment_of(offsets, G, (uint32_t)i);
    have_g = true;
    counts = offsets[g] == (uint32_t)i;
  }
  if (counts && !have_g) g = segment_of(offsets, G, (uint32_t)i);
  // neighbours in the sorted column mostly belong to one group: one atomic per (wave, group), not per element (a long
  // group of distinct values would put its whole length on one address otherwise)
  unsigned long long active = __ballot(counts);
  while (active) {
    const int leader = __ffsll((long long)active) - 1;
    const uint32_t gl = (uint32_t)__shfl((int)g, leader);
    const unsigned long long m = __ballot(counts && g == gl);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd((unsigned int *)A.out + gl, (unsigned int)__popcll(m));
    active &= ~m;
  }
}
__global__ void __launch_bounds__(256) k_n_unique_finish(AggExtArgs A, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ marks, uint64_t G) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  if (offsets[g + 1] == offsets[g]) return; // empty group: null
  const uint32_t m = (marks[g >> 4] >> ((g & 15u) * 2u)) & 3u;
  if (m == 3u) ((uint32_t *)A.out)[g] += 1u; // nulls and the largest value shared one code
  atomicOr((unsigned int *)A.out_validity + (g >> 5), 1u << (g & 31));
}
// interp: QuantileInterpolOptions (crates/polars-arrow/src/kernels/rolling/no_nulls/mod.rs:70-77): 0 Nearest, 1 Lower, 2 Higher,
// 3 Midpoint, 4 Linear.  quantile_idx (quantile.rs:19-44) counts in the sorted column with the nulls in front, so the
// group's null count enters the index arithmetic exactly as there; everything is f64 and an f32 column's answer is
// the f64 one cast (quantile.rs:231-246, 296-310).
__global__ void __launch_bounds__(256) k_quantile(AggExtArgs A, const uint64_t *__restrict__ sorted, const uint32_t *__restrict__ offsets, uint64_t G,
                                                  double q, int interp) {
#pragma clang fp contract(off) // no fused multiply-adds: the reference's arithmetic rounds after every operation
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t b = offsets[g], e = offsets[g + 1];
  uint32_t lo = b, hi = e; // first null code of the segment
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted[mid] == ~0ull) hi = mid; else lo = mid + 1; }
  const uint32_t m = lo - b, len = e - b;
  const double nc = (double)(len - m);
  bool valid = m > 0;
  double r = 0.0;
  if (valid) {
    const double float_idx = ((double)m - 1.0) * q + nc;
    double base_f = interp == 0 ? (double)m * q + nc : (interp == 2 ? ceil(float_idx) : float_idx);
    uint64_t base = (uint64_t)base_f; // `as usize`: truncation
    if (base > (uint64_t)len - 1) base = (uint64_t)len - 1;
    const uint64_t top = (uint64_t)ceil(float_idx);
    const uint32_t idx = (uint32_t)(base - (uint64_t)(len - m)); // among the valid values
    const double lower = f64_from_ordered(sorted[b + idx]);
    r = lower;
    if ((interp == 3 || interp == 4) && top != base) {
      const double upper = f64_from_ordered(sorted[b + (idx + 1 < m ? idx + 1 : idx)]);
      if (lower != upper) r = interp == 3 ? (lower + upper) / 2.0 : (float_idx - (double)base) * (upper - lower) + lower;
    }
  }
  if (A.out_dtype == PLHIP_F32) ((float *)A.out)[g] = (float)r;
  else ((double *)A.out)[g] = r;
  if (valid) atomicOr((unsigned int *)A.out_validity + (g >> 5), 1u << (g & 31));
}
} // namespace

// ---- sorting the values inside every group, whatever the group sizes -------------------------------------------------------
// rocPRIM's segmented sort gives a segment to one workgroup: one key on 30 % of 1e7 rows made it 91 ms (median) against
// 1.9 ms on uniform groups.  When some group is long, two device-wide STABLE radix sorts replace it -- by value, then by
// group id -- whose cost does not depend on the group sizes at all; the result layout is the same (groups in CSR order,
// values ascending inside).
#define SEG_LONG (1u << 15)
namespace {
__global__ void __launch_bounds__(256) k_max_seg(const uint32_t *__restrict__ offsets, uint64_t G, uint32_t *out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t len = g < G ? offsets[g + 1] - offsets[g] : 0u;
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)len, off); len = o > len ? o : len; }
  if ((threadIdx.x & 63) == 0 && len) atomicMax(out, len);
}
__global__ void __launch_bounds__(256) k_fill_gid(const uint32_t *__restrict__ offsets, uint64_t G, uint64_t n, uint32_t *__restrict__ gid) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gid[i] = segment_of(offsets, G, (uint32_t)i);
}
int bits_of(uint64_t x) { int b = 0; while (x) { b++; x >>= 1; } return b < 1 ? 1 : b; }
} // namespace

hipError_t median_temp_bytes(uint64_t n, uint64_t G, size_t *bytes) {
  size_t a = 0, b = 0, c = 0;
  hipError_t e = rocprim::segmented_radix_sort_keys(nullptr, a, (uint64_t *)nullptr, (uint64_t *)nullptr, (unsigned int)n, (unsigned int)G,
                                                    (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0, 64);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, c, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, n, 0, 32);
  *bytes = std::max(a, std::max(b, c)) + 512; // + the (aligned) word k_max_seg writes
  return e;
}

// codes_a (n u64, filled) -> *sorted (codes_a or codes_b): ascending inside every CSR segment.  gid_a / gid_b: n u32 each.
static hipError_t sort_in_segments(const uint32_t *offsets, uint64_t n, uint64_t G, uint64_t *codes_a, uint64_t *codes_b, uint32_t *gid_a,
                                   uint32_t *gid_b, void *temp, size_t temp_bytes, const uint64_t **sorted, hipStream_t s) {
  uint32_t *d_max = (uint32_t *)((uint8_t *)temp + ((temp_bytes - 256) & ~(size_t)255));
  uint32_t max_len = 0;
  hipError_t e = hipMemsetAsync(d_max, 0, 4, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_max_seg, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, offsets, G, d_max);
  e = hipMemcpyAsync(&max_len, d_max, 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  const size_t tb = (temp_bytes - 256) & ~(size_t)255;
  if (max_len <= SEG_LONG && !getenv("PLHIP_FORCE_TWO_SORTS")) {
    *sorted = codes_b;
    size_t t2 = tb;
    return rocprim::segmented_radix_sort_keys(temp, t2, codes_a, codes_b, (unsigned int)n, (unsigned int)G, offsets, offsets + 1, 0, 64, s);
  }
  hipLaunchKernelGGL(k_fill_gid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, offsets, G, n, gid_a);
  size_t t2 = tb;
  e = rocprim::radix_sort_pairs(temp, t2, codes_a, codes_b, gid_a, gid_b, n, 0, 64, s);
  if (e != hipSuccess) return e;
  t2 = tb;
  e = rocprim::radix_sort_pairs(temp, t2, gid_b, gid_a, codes_b, codes_a, n, 0, bits_of(G ? G - 1 : 0), s); // stable: value order survives
  *sorted = codes_a;
  return e;
}

// codes_a / codes_b: n u64 each; gid_a / gid_b: n u32 each
hipError_t launch_quantile(const AggExtArgs &a, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, uint64_t *codes_a,
                           uint64_t *codes_b, uint32_t *gid_a, uint32_t *gid_b, void *temp, size_t temp_bytes, double q, int interp, hipStream_t s) {
  if (!G) return hipSuccess;
  const uint64_t *sorted = codes_b;
  if (n) {
    hipLaunchKernelGGL(k_gather_ordered, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, indices, n, codes_a);
    hipError_t e = sort_in_segments(offsets, n, G, codes_a, codes_b, gid_a, gid_b, temp, temp_bytes, &sorted, s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_quantile, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, a, sorted, offsets, G, q, interp);
  return hipGetLastError();
}

// marks: (G / 16 + 1) u32; A.out: G u32 -- both zeroed here
hipError_t launch_n_unique(const AggExtArgs &a, const uint32_t *offsets, const uint32_t *indices, uint64_t n, uint64_t G, uint64_t *codes_a,
                           uint64_t *codes_b, uint32_t *gid_a, uint32_t *gid_b, uint32_t *marks, void *temp, size_t temp_bytes, hipStream_t s) {
  if (!G) return hipSuccess;
  hipError_t e = hipMemsetAsync(marks, 0, (G / 16 + 1) * 4, s);
  if (e == hipSuccess) e = hipMemsetAsync(a.out, 0, G * 4, s);
  if (e != hipSuccess) return e;
  if (n) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_gather_exact, dim3(grid), dim3(256), 0, s, a, indices, offsets, n, G, codes_a, marks);
    const uint64_t *sorted = codes_b;
    e = sort_in_segments(offsets, n, G, codes_a, codes_b, gid_a, gid_b, temp, temp_bytes, &sorted, s);
    if (e != hipSuccess) return e;
    const uint32_t is_float = a.dtype == PLHIP_F32 || a.dtype == PLHIP_F64;
    hipLaunchKernelGGL(k_count_changes, dim3(grid), dim3(256), 0, s, a, sorted, offsets, n, G, is_float);
  }
  hipLaunchKernelGGL(k_n_unique_finish, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, a, offsets, marks, G);
  return hipGetLastError();
}

// the same without the sort: `key` is a column the caller declares sorted; rows = 0..n-1
hipError_t launch_sorted_key_groups(const KeySpec &key, uint64_t n, uint32_t *rows, uint32_t *flags, uint32_t *gid, uint32_t *offsets, void *temp,
                                    size_t temp_bytes, uint32_t *dflags, hipStream_t s) {
  hipLaunchKernelGGL(k_sorted_run_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, key, n, flags, rows);
  hipError_t e = rocprim::inclusive_scan(temp, temp_bytes, flags, gid, n, rocprim::plus<uint32_t>(), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_run_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flags, gid, n, offsets, dflags);
  return hipGetLastError();
}

// group_by_stable (hashing.rs:32-79: groups ordered by their first row): first[g] = the run's first (= smallest) row,
// a radix sort of (first, g), and gmap[g] = the position of g in that order, which the fold writes its output row to
namespace {
__global__ void __launch_bounds__(256) k_run_first(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ offsets, uint64_t G,
                                                   uint32_t *__restrict__ first, uint32_t *__restrict__ gid) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) { first[g] = rows[offsets[g]]; gid[g] = (uint32_t)g; }
}
__global__ void __launch_bounds__(256) k_invert(const uint32_t *__restrict__ sorted_g, uint64_t G, uint32_t *__restrict__ gmap) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < G) gmap[sorted_g[i]] = (uint32_t)i;
}
} // namespace
hipError_t stable_order_temp_bytes(uint64_t G, size_t *bytes) {
  return rocprim::radix_sort_pairs(nullptr, *bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, G, 0, 32);
}
// buf: 4 * G u32 (first, gid, first sorted, gid sorted); gmap: G u32
hipError_t launch_stable_order(const uint32_t *rows, const uint32_t *offsets, uint64_t G, uint32_t *buf, uint32_t *gmap, void *temp, size_t temp_bytes,
                               hipStream_t s) {
  if (!G) return hipSuccess;
  uint32_t *first = buf, *gid = buf + G, *first_s = buf + 2 * G, *gid_s = buf + 3 * G;
  hipLaunchKernelGGL(k_run_first, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, rows, offsets, G, first, gid);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, first, first_s, gid, gid_s, G, 0, 32, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_invert, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, gid_s, G, gmap);
  return hipGetLastError();
}

void launch_fold_sorted(const Plan &p, const OutPlan &o, const KeyGather *kg_dev, const uint64_t *skeys, const uint32_t *rows,
                        const uint32_t *offsets, uint64_t G, uint32_t *long_list, uint32_t *long_count, hipStream_t s, const uint32_t *gmap) {
  if (!G) return;
  // behind the list: a done counter per long run, then its combined accumulator states (the caller sized the buffer)
  const uint64_t max_runs = (uint64_t)p.n_rows / SP_LONG + 2;
  uint32_t *long_done = long_list + max_runs;
  uint64_t *long_part = (uint64_t *)(((uintptr_t)(long_done + max_runs) + 7) & ~(uintptr_t)7);
  const size_t lds = (size_t)(p.n_accs ? p.n_accs : 1) * SP_BLOCK * 8;
  hipLaunchKernelGGL(k_fold_sorted, dim3((unsigned)((G + SP_BLOCK - 1) / SP_BLOCK)), dim3(SP_BLOCK), lds, s, p, o, kg_dev, skeys, rows, offsets, G,
                     long_list, long_count, gmap, long_done, long_part, (uint32_t)max_runs);
  const uint64_t max_long = (uint64_t)p.n_rows / SP_LONG + 1;
  if (p.exact_fold) {
    const uint64_t blocks = (max_long + SP_BLOCK / 64 - 1) / (SP_BLOCK / 64);
    hipLaunchKernelGGL(k_fold_sorted_exact_long, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(SP_BLOCK), lds, s, p, o, kg_dev, skeys, rows,
                       offsets, long_list, long_count, gmap);
    return;
  }
  hipLaunchKernelGGL(k_fold_sorted_long, dim3((unsigned)(max_long < 4096 ? max_long : 4096)), dim3(SP_BLOCK), lds, s, p, o, kg_dev, skeys, rows,
                     offsets, long_list, long_count, gmap, long_done, long_part, (uint32_t)max_runs);
}
