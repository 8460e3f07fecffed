// utf8_path.hip — kernels of the string-key encoder (utf8.hpp has the design and the reference lines).
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "utf8.hpp"

namespace {

#define UTF8_BLOCK 256

__device__ __forceinline__ uint64_t fmul(uint64_t a, uint64_t b) { // folded multiply (the shape of ahash's folded_multiply, vector_hasher.rs)
  return (a * b) ^ __umul64hi(a, b);
}

__device__ __forceinline__ uint64_t load8(const uint8_t *p) { // strings start at any byte
  uint64_t w;
  __builtin_memcpy(&w, p, 8);
  return w;
}

__device__ __forceinline__ uint64_t load_tail(const uint8_t *p, uint32_t len) { // 1 .. 7 bytes, never reads past the string
  uint64_t w = 0;
  for (uint32_t j = 0; j < len; j++) w |= (uint64_t)p[j] << (8u * j);
  return w;
}

__device__ __forceinline__ bool row_valid(const Utf8View &v, int64_t i) {
  if (!v.validity) return true;
  const int64_t b = v.boff + i;
  return (v.validity[b >> 3] >> (b & 7)) & 1u;
}

#define UTF8_NULL_HASH 0x9E3779B97F4A7C15ull // get_null_hash_value's role: all null rows agree on it (vector_hasher.rs:18-29)

__global__ void __launch_bounds__(UTF8_BLOCK) k_utf8_hash(Utf8View v, uint64_t seed, uint64_t *h, uint32_t *iota) {
  const int64_t i = (int64_t)blockIdx.x * UTF8_BLOCK + threadIdx.x;
  if (i >= v.n) return;
  iota[i] = (uint32_t)i;
  if (!row_valid(v, i)) { h[i] = UTF8_NULL_HASH; return; }
  const int64_t a = v.offsets[i], b = v.offsets[i + 1];
  uint64_t len = (uint64_t)(b - a);
  const uint8_t *p = v.data + a;
  const bool wide = len >= 8; // the last 1 .. 7 bytes of such a string are read as the 8 bytes that END the string (one load, no byte loop)
  uint64_t x = seed ^ fmul(len + 1, 0xA0761D6478BD642Full);
  for (; len >= 8; len -= 8, p += 8) x = fmul(x ^ load8(p), 0xE7037ED1A0B428DBull);
  if (len) x = fmul(x ^ (wide ? load8(p + len - 8) : load_tail(p, (uint32_t)len)), 0x8EBC6AF09C88C6E3ull);
  x = fmul(x ^ seed, 0x589965CC75374CC3ull);
  if (x == UTF8_HT_EMPTY) x = 0x1D8E4E27C47D124Full; // the table's empty mark is not a hash (a clash here is caught like any collision)
  h[i] = x;
}

__device__ __forceinline__ uint64_t slot_of(uint64_t h, uint64_t mask) { return (h ^ (h >> 29)) & mask; }

__global__ void __launch_bounds__(UTF8_BLOCK) k_ht_build(const uint64_t *keys, const uint32_t *first, uint32_t G, Utf8Slot *table, uint64_t mask) {
  const uint32_t g = blockIdx.x * UTF8_BLOCK + threadIdx.x;
  if (g >= G) return;
  const uint64_t h = keys[g];
  for (uint64_t s = slot_of(h, mask), tries = 0; tries <= mask; s = (s + 1) & mask, tries++) { // (load <= 0.5: an empty slot exists)
    const unsigned long long old = atomicCAS((unsigned long long *)&table[s].hash, (unsigned long long)UTF8_HT_EMPTY, (unsigned long long)h);
    if (old == UTF8_HT_EMPTY) { table[s].group = g; table[s].first = first[g]; return; }
  }
}

// Every row looks its hash up (its code), then BytesHash equality (vector_hasher.rs:437-446) of the row against its group's first row
__global__ void __launch_bounds__(UTF8_BLOCK) k_ht_probe_verify(Utf8View v, const uint64_t *h, const Utf8Slot *table, uint64_t mask, uint32_t *codes,
                                                                uint32_t *bad) {
  const int64_t i = (int64_t)blockIdx.x * UTF8_BLOCK + threadIdx.x;
  if (i >= v.n) return;
  const uint64_t x = h[i];
  uint32_t code = 0;
  int64_t f = 0;
  bool found = false;
  for (uint64_t s = slot_of(x, mask), tries = 0; tries <= mask; s = (s + 1) & mask, tries++) {
    const uint4 e = *(const uint4 *)&table[s]; // {hash lo, hash hi, group, first}
    const uint64_t k = ((uint64_t)e.y << 32) | e.x;
    if (k == x) { code = e.z; f = e.w; found = true; break; }
    if (k == UTF8_HT_EMPTY) break;
  }
  codes[i] = code;
  if (!found) { atomicOr(bad, 2u); return; } // every row's hash is one of the groups': cannot happen
  if (f == i) return;
  const bool vi = row_valid(v, i), vf = row_valid(v, f);
  bool same = vi == vf;
  if (same && vi) {
    const int64_t a = v.offsets[i], b = v.offsets[i + 1], fa = v.offsets[f], fb = v.offsets[f + 1];
    uint64_t len = (uint64_t)(b - a);
    same = len == (uint64_t)(fb - fa);
    const bool wide = len >= 8;
    const uint8_t *p = v.data + a, *q = v.data + fa;
    for (; same && len >= 8; len -= 8, p += 8, q += 8) same = load8(p) == load8(q);
    if (same && len) same = wide ? load8(p + len - 8) == load8(q + len - 8) : load_tail(p, (uint32_t)len) == load_tail(q, (uint32_t)len);
  }
  if (!same) atomicOr(bad, 1u);
}

__global__ void __launch_bounds__(UTF8_BLOCK) k_utf8_lengths(Utf8View v, const uint32_t *first, uint32_t G, int64_t *lens) {
  const uint32_t g = blockIdx.x * UTF8_BLOCK + threadIdx.x;
  if (g > G) return;
  int64_t len = 0;
  if (g < G) {
    const int64_t f = first[g];
    if (row_valid(v, f)) len = v.offsets[f + 1] - v.offsets[f];
  }
  lens[g] = len; // (entry G = 0: the exclusive scan's last entry is the total)
}

// 16 lanes per dictionary entry
__global__ void __launch_bounds__(UTF8_BLOCK) k_utf8_gather(Utf8View v, const uint32_t *first, uint32_t G, const int64_t *offs, uint8_t *out,
                                                            uint32_t *dvalid) {
  const uint32_t g = (uint32_t)(((uint64_t)blockIdx.x * UTF8_BLOCK + threadIdx.x) >> 4), l = threadIdx.x & 15u;
  if (g >= G) return;
  const int64_t f = first[g];
  if (!row_valid(v, f)) return;
  if (l == 0) atomicOr(&dvalid[g >> 5], 1u << (g & 31u));
  const uint8_t *src = v.data + v.offsets[f];
  uint8_t *dst = out + offs[g];
  const int64_t len = offs[g + 1] - offs[g];
  for (int64_t j = l; j < len; j += 16) dst[j] = src[j];
}

} // namespace

void launch_utf8_hash(const Utf8View &v, uint64_t seed, uint64_t *h, uint32_t *iota, hipStream_t s) {
  if (v.n <= 0) return;
  hipLaunchKernelGGL(k_utf8_hash, dim3((unsigned)((v.n + UTF8_BLOCK - 1) / UTF8_BLOCK)), dim3(UTF8_BLOCK), 0, s, v, seed, h, iota);
}

void launch_ht_build(const uint64_t *keys, const uint32_t *first, uint32_t G, Utf8Slot *table, uint64_t mask, hipStream_t s) {
  if (!G) return;
  hipLaunchKernelGGL(k_ht_build, dim3((G + UTF8_BLOCK - 1) / UTF8_BLOCK), dim3(UTF8_BLOCK), 0, s, keys, first, G, table, mask);
}

void launch_ht_probe_verify(const Utf8View &v, const uint64_t *h, const Utf8Slot *table, uint64_t mask, uint32_t *codes, uint32_t *bad, hipStream_t s) {
  if (v.n <= 0) return;
  hipLaunchKernelGGL(k_ht_probe_verify, dim3((unsigned)((v.n + UTF8_BLOCK - 1) / UTF8_BLOCK)), dim3(UTF8_BLOCK), 0, s, v, h, table, mask, codes, bad);
}

void launch_utf8_lengths(const Utf8View &v, const uint32_t *first, uint32_t G, int64_t *lens, hipStream_t s) {
  hipLaunchKernelGGL(k_utf8_lengths, dim3((G + 1 + UTF8_BLOCK - 1) / UTF8_BLOCK), dim3(UTF8_BLOCK), 0, s, v, first, G, lens);
}

size_t utf8_scan_temp_bytes(uint32_t G) {
  size_t b = 0;
  if (rocprim::exclusive_scan(nullptr, b, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, (size_t)G + 1, rocprim::plus<int64_t>()) != hipSuccess) return 0;
  return b + 256;
}

hipError_t utf8_scan(void *temp, size_t temp_bytes, const int64_t *lens, int64_t *offs, uint32_t G, hipStream_t s) {
  return rocprim::exclusive_scan(temp, temp_bytes, lens, offs, (int64_t)0, (size_t)G + 1, rocprim::plus<int64_t>(), s);
}

void launch_utf8_gather(const Utf8View &v, const uint32_t *first, uint32_t G, const int64_t *offs, uint8_t *out, uint32_t *dvalid, hipStream_t s) {
  if (!G) return;
  const uint64_t threads = (uint64_t)G * 16u;
  hipLaunchKernelGGL(k_utf8_gather, dim3((unsigned)((threads + UTF8_BLOCK - 1) / UTF8_BLOCK)), dim3(UTF8_BLOCK), 0, s, v, first, G, offs, out, dvalid);
}
