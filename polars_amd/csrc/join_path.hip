// join_path.hip — kernels of the single-key hash join (join.hpp has the design and the reference lines).
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "join.hpp"

namespace {

#define JOIN_BLOCK 256

__device__ __forceinline__ uint64_t jk_load(const JoinKey &k, int64_t i) {
  switch (k.width) {
    case 1: return k.values[i];
    case 2: return ((const uint16_t *)k.values)[i];
    case 4: return ((const uint32_t *)k.values)[i];
    default: return ((const uint64_t *)k.values)[i];
  }
}

__device__ __forceinline__ bool jk_valid(const JoinKey &k, int64_t i) {
  if (!k.validity) return true;
  const int64_t b = k.boff + i;
  return (k.validity[b >> 3] >> (b & 7)) & 1u;
}

__device__ __forceinline__ uint64_t jk_slot(uint64_t key, uint64_t mask) {
  const uint64_t m = key * 0x9E3779B97F4A7C15ull;
  return (m ^ (m >> 32)) & mask;
}

__global__ void __launch_bounds__(JOIN_BLOCK) k_join_build(JoinKey b, const uint32_t *first, uint32_t G, JoinTable t) {
  const uint32_t g = blockIdx.x * JOIN_BLOCK + threadIdx.x;
  if (g >= G) return;
  const int64_t f = first[g];
  if (!jk_valid(b, f)) { *t.g_null = g; return; } // (one group at most)
  const uint64_t key = jk_load(b, f);
  for (uint64_t s = jk_slot(key, t.mask), tries = 0; tries <= t.mask; s = (s + 1) & t.mask, tries++)
    if (atomicCAS(&t.slots[s].used, 0u, 1u) == 0u) { t.slots[s].key = key; t.slots[s].val = g; return; } // the groups' keys are distinct: nothing to compare
}

__global__ void __launch_bounds__(JOIN_BLOCK) k_join_count(JoinKey p, JoinTable t, const uint32_t *offsets, int mode, uint32_t *gid, uint64_t *cnt,
                                                           uint32_t *matched) {
  const int64_t i = (int64_t)blockIdx.x * JOIN_BLOCK + threadIdx.x;
  if (i > p.n) return;
  if (i == p.n) { cnt[i] = 0; return; }
  uint32_t g = JOIN_NONE;
  if (!jk_valid(p, i)) g = *t.g_null; // Option<T>: None == None (joins.rs:574-586)
  else {
    const uint64_t key = jk_load(p, i);
    for (uint64_t s = jk_slot(key, t.mask), tries = 0; tries <= t.mask; s = (s + 1) & t.mask, tries++) {
      const uint4 e = *(const uint4 *)&t.slots[s]; // {key lo, key hi, val, used}
      if (!e.w) break;
      if ((((uint64_t)e.y << 32) | e.x) == key) { g = e.z; break; }
    }
  }
  if (mode == 2 || mode == 3) { // semi / anti: membership only
    const bool keep = (g != JOIN_NONE) == (mode == 2);
    gid[i] = JOIN_NONE;
    cnt[i] = keep ? 1ull : 0ull;
    return;
  }
  gid[i] = g;
  if (mode == 4 && g != JOIN_NONE) matched[g] = 1u; // (every writer stores the same value)
  cnt[i] = g != JOIN_NONE ? (uint64_t)(offsets[g + 1] - offsets[g]) : (mode == 1 || mode == 4 ? 1ull : 0ull);
}

__global__ void __launch_bounds__(JOIN_BLOCK) k_join_fill(int64_t n, const uint32_t *gid, const uint64_t *pos, const uint32_t *offsets,
                                                          const uint32_t *indices, uint32_t *out_probe, uint32_t *out_build) {
  const int64_t i = (int64_t)blockIdx.x * JOIN_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = gid[i];
  const uint64_t p0 = pos[i];
  if (g == JOIN_NONE) {
    if (pos[i + 1] > p0) { out_probe[p0] = (uint32_t)i; if (out_build) out_build[p0] = JOIN_NONE; } // left join: (row, None); semi / anti: the row
    return;
  }
  const uint32_t o0 = offsets[g], o1 = offsets[g + 1];
  for (uint32_t j = o0; j < o1; j++) { out_probe[p0 + (j - o0)] = (uint32_t)i; out_build[p0 + (j - o0)] = indices[j]; }
}

__global__ void __launch_bounds__(JOIN_BLOCK) k_join_tail_count(uint32_t G, const uint32_t *matched, const uint32_t *offsets, uint64_t *cnt) {
  const uint32_t g = blockIdx.x * JOIN_BLOCK + threadIdx.x;
  if (g > G) return;
  cnt[g] = (g < G && !matched[g]) ? (uint64_t)(offsets[g + 1] - offsets[g]) : 0ull;
}

__global__ void __launch_bounds__(JOIN_BLOCK) k_join_tail_fill(uint32_t G, const uint32_t *matched, const uint64_t *pos, const uint32_t *offsets,
                                                               const uint32_t *indices, uint32_t *out_probe, uint32_t *out_build) {
  const uint32_t g = blockIdx.x * JOIN_BLOCK + threadIdx.x;
  if (g >= G || matched[g]) return;
  const uint64_t p0 = pos[g];
  const uint32_t o0 = offsets[g], o1 = offsets[g + 1];
  for (uint32_t j = o0; j < o1; j++) { out_probe[p0 + (j - o0)] = JOIN_NONE; out_build[p0 + (j - o0)] = indices[j]; }
}

__global__ void __launch_bounds__(JOIN_BLOCK) k_csr_to_gid(const uint32_t *offsets, const uint32_t *indices, uint32_t G, int64_t n, uint32_t *gid) {
  const int64_t j = (int64_t)blockIdx.x * JOIN_BLOCK + threadIdx.x;
  if (j >= n) return;
  uint32_t lo = 0, hi = G; // the last g with offsets[g] <= j
  while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((int64_t)offsets[mid] <= j) lo = mid; else hi = mid; }
  gid[indices[j]] = lo;
}

} // namespace

void launch_csr_to_gid(const uint32_t *offsets, const uint32_t *indices, uint32_t G, int64_t n, uint32_t *gid, hipStream_t s) {
  if (n <= 0 || !G) return;
  hipLaunchKernelGGL(k_csr_to_gid, dim3((unsigned)((n + JOIN_BLOCK - 1) / JOIN_BLOCK)), dim3(JOIN_BLOCK), 0, s, offsets, indices, G, n, gid);
}

void launch_join_build(const JoinKey &build, const uint32_t *first, uint32_t G, const JoinTable &t, hipStream_t s) {
  if (!G) return;
  hipLaunchKernelGGL(k_join_build, dim3((G + JOIN_BLOCK - 1) / JOIN_BLOCK), dim3(JOIN_BLOCK), 0, s, build, first, G, t);
}

void launch_join_count(const JoinKey &probe, const JoinTable &t, const uint32_t *offsets, int mode, uint32_t *gid, uint64_t *cnt, uint32_t *matched,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_join_count, dim3((unsigned)((probe.n + 1 + JOIN_BLOCK - 1) / JOIN_BLOCK)), dim3(JOIN_BLOCK), 0, s, probe, t, offsets, mode, gid, cnt,
                     matched);
}

void launch_join_tail_count(uint32_t G, const uint32_t *matched, const uint32_t *offsets, uint64_t *cnt, hipStream_t s) {
  hipLaunchKernelGGL(k_join_tail_count, dim3((G + 1 + JOIN_BLOCK - 1) / JOIN_BLOCK), dim3(JOIN_BLOCK), 0, s, G, matched, offsets, cnt);
}

void launch_join_tail_fill(uint32_t G, const uint32_t *matched, const uint64_t *pos, const uint32_t *offsets, const uint32_t *indices,
                           uint32_t *out_probe, uint32_t *out_build, hipStream_t s) {
  if (!G) return;
  hipLaunchKernelGGL(k_join_tail_fill, dim3((G + JOIN_BLOCK - 1) / JOIN_BLOCK), dim3(JOIN_BLOCK), 0, s, G, matched, pos, offsets, indices, out_probe, out_build);
}

size_t join_scan_temp_bytes(int64_t n) {
  size_t b = 0;
  if (rocprim::exclusive_scan(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>()) != hipSuccess) return 0;
  return b + 256;
}

hipError_t join_scan(void *temp, size_t temp_bytes, const uint64_t *cnt, uint64_t *pos, int64_t n, hipStream_t s) {
  return rocprim::exclusive_scan(temp, temp_bytes, cnt, pos, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), s);
}

void launch_join_fill(int64_t n, const uint32_t *gid, const uint64_t *pos, const uint32_t *offsets, const uint32_t *indices, uint32_t *out_probe,
                      uint32_t *out_build, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_join_fill, dim3((unsigned)((n + JOIN_BLOCK - 1) / JOIN_BLOCK)), dim3(JOIN_BLOCK), 0, s, n, gid, pos, offsets, indices, out_probe, out_build);
}

// ---- payload gather (DataFrame::take_unchecked / _take_opt_chunked_unchecked behind _finish_join, crates/polars-ops/src/frame/join/
// hash_join/mod.rs:105-201, general.rs:9-): out[i] = col[idx[i]], null where idx[i] == JOIN_NONE or the source row is null.  A thread per
// output row, the width as a template argument, validity as one 64-bit ballot per wave (whole words: the bitmap is written, never ORed).
template <typename T>
__global__ void __launch_bounds__(256) k_take(const T *__restrict__ vals, const uint8_t *__restrict__ validity, int64_t boff, const uint32_t *__restrict__ idx,
                                              int64_t n, T *__restrict__ out, uint64_t *__restrict__ out_valid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // (no early return: the wave votes)
  bool ok = false;
  T v = 0;
  if (i < n) {
    const uint32_t r = idx[i];
    ok = r != JOIN_NONE;
    if (ok && validity) { const int64_t b = boff + (int64_t)r; ok = (validity[b >> 3] >> (b & 7)) & 1; }
    if (ok) v = vals[r];
    out[i] = v;
  }
  const unsigned long long m = __ballot(ok);
  if (out_valid && (threadIdx.x & 63) == 0 && (i & ~(int64_t)63) < n) out_valid[i >> 6] = m;
}

void launch_take(const uint8_t *vals, int width, const uint8_t *validity, int64_t boff, const uint32_t *idx, int64_t n, uint8_t *out, uint64_t *out_valid,
                 hipStream_t s) {
  if (n <= 0) return;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (width) {
    case 1: hipLaunchKernelGGL(k_take<uint8_t>, grid, block, 0, s, (const uint8_t *)vals, validity, boff, idx, n, (uint8_t *)out, out_valid); break;
    case 2: hipLaunchKernelGGL(k_take<uint16_t>, grid, block, 0, s, (const uint16_t *)vals, validity, boff, idx, n, (uint16_t *)out, out_valid); break;
    case 4: hipLaunchKernelGGL(k_take<uint32_t>, grid, block, 0, s, (const uint32_t *)vals, validity, boff, idx, n, (uint32_t *)out, out_valid); break;
    default: hipLaunchKernelGGL(k_take<uint64_t>, grid, block, 0, s, (const uint64_t *)vals, validity, boff, idx, n, (uint64_t *)out, out_valid); break;
  }
}
