// ubench.hip — hardware-rate microbenchmarks behind the design choices in DESIGN.md (development aid,
// not part of libplhip.so).  Build: make -C scripts/ubench ; run on the GPU box: scripts/ubench/ubench <test>
//   bw       streaming reads: 1..3 columns, dwordx4 loads, grid sweep
//   mm       the key min / max pass over an 800 MB column: loop shapes x non-temporal loads x grid
//   ldsatom  LDS atomic throughput by type and table size (random slots)
//   gatom    global atomic throughput, random slots, by table size
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

static float time_ms(hipStream_t s, int reps, const std::function<void()> &f) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(a, s));
  for (int i = 0; i < reps; i++) f();
  CHECK(hipEventRecord(b, s));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  return ms / reps;
}

__global__ void k_fill(uint32_t *p, size_t n, uint32_t seed, uint32_t mod) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
    p[i] = mod ? 1 + x % mod : x;
  }
}

// ---- bw ------------------------------------------------------------------------------------------------
template <int NCOL>
__global__ void __launch_bounds__(256) k_bw(const uint4 *a, const uint4 *b, const uint4 *c, size_t nvec, uint32_t *out) {
  uint32_t acc = 0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < nvec; i += st) {
    uint4 x = a[i];
    acc += x.x ^ x.y ^ x.z ^ x.w;
    if (NCOL > 1) { uint4 y = b[i]; acc += y.x ^ y.y ^ y.z ^ y.w; }
    if (NCOL > 2) { uint4 z = c[2 * i]; uint4 w = c[2 * i + 1]; acc += z.x ^ z.y ^ z.z ^ z.w ^ w.x ^ w.y ^ w.z ^ w.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

static void test_bw() {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  for (size_t n : {(size_t)10000000, (size_t)100000000}) {
    uint32_t *a, *b, *c, *out;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&c, n * 8)); CHECK(hipMalloc(&out, 64));
    k_fill<<<4096, 256, 0, s>>>(a, n, 1, 0); k_fill<<<4096, 256, 0, s>>>(b, n, 2, 0); k_fill<<<4096, 256, 0, s>>>(c, 2 * n, 3, 0);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
      float t1 = time_ms(s, 20, [&] { k_bw<1><<<grid, 256, 0, s>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, n / 4, out); });
      float t2 = time_ms(s, 20, [&] { k_bw<2><<<grid, 256, 0, s>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, n / 4, out); });
      float t3 = time_ms(s, 20, [&] { k_bw<3><<<grid, 256, 0, s>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, n / 4, out); });
      printf("bw n=%zu grid=%d  1col(4B/row) %.1f us %.0f GB/s | 2col(8B) %.1f us %.0f GB/s | 3col(16B) %.1f us %.0f GB/s\n", n, grid,
             t1 * 1e3, n * 4 / t1 / 1e6, t2 * 1e3, n * 8 / t2 / 1e6, t3 * 1e3, n * 16 / t3 / 1e6);
    }
    CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(c)); CHECK(hipFree(out));
  }
}

// ---- mm: the key min / max pass (kernels.hip k_key_minmax) as loop shapes, 64-bit keys ------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct MmArgs { const uint8_t *values; uint64_t pad[40]; int64_t n_rows; }; // (the pointer arrives inside a by-value struct, as in Plan)
__device__ __forceinline__ void mmf(uint64_t v, uint64_t &lo, uint64_t &hi) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
__device__ __forceinline__ void mmv(const u32x4 &q, uint64_t &lo, uint64_t &hi) {
  mmf((uint64_t)q.x | ((uint64_t)q.y << 32), lo, hi); mmf((uint64_t)q.z | ((uint64_t)q.w << 32), lo, hi);
}
// V: 0 = four requests a grid stride apart (round 2), 1 = four consecutive 4 KB pieces per workgroup trip, 2 = one request per trip,
// 3 = eight consecutive pieces; NT: non-temporal loads
template <int V, bool NT>
__global__ void __launch_bounds__(256) k_mm(MmArgs A, uint64_t *out) {
  const u32x4 *p = (const u32x4 *)A.values;
  const int64_t nv = A.n_rows / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  uint64_t lo = ~0ull, hi = 0;
  auto ld = [&](const u32x4 *q) { return NT ? __builtin_nontemporal_load(q) : *q; };
  if (V == 0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nv; i += 4 * stride) {
      const u32x4 a = ld(p + i), b = ld(p + i + stride), c = ld(p + i + 2 * stride), d = ld(p + i + 3 * stride);
      mmv(a, lo, hi); mmv(b, lo, hi); mmv(c, lo, hi); mmv(d, lo, hi);
    }
    for (; i < nv; i += stride) mmv(p[i], lo, hi);
  } else if (V == 1 || V == 3) {
    constexpr int U = V == 1 ? 4 : 8;
    const int64_t tile = (int64_t)blockDim.x * U, n_tiles = nv / tile;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
      const u32x4 *q = p + t * tile + threadIdx.x;
      u32x4 x[U];
#pragma unroll
      for (int u = 0; u < U; u++) x[u] = ld(q + u * blockDim.x);
#pragma unroll
      for (int u = 0; u < U; u++) mmv(x[u], lo, hi);
    }
    for (int64_t i = n_tiles * tile + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) mmv(p[i], lo, hi);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) mmv(ld(p + i), lo, hi);
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
    lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = lo; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = hi; }
}

static void test_mm() {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const size_t n = 100000000;
  uint32_t *a; uint64_t *out;
  CHECK(hipMalloc(&a, n * 8)); CHECK(hipMalloc(&out, 16384 * 4 * 16));
  k_fill<<<4096, 256, 0, s>>>(a, 2 * n, 1, 0);
  MmArgs A{}; A.values = (const uint8_t *)a; A.n_rows = (int64_t)n;
  for (int grid : {1021, 1024, 2048, 4093, 8192, 16381}) {
    float t[8];
    t[0] = time_ms(s, 10, [&] { k_mm<0, true><<<grid, 256, 0, s>>>(A, out); });
    t[1] = time_ms(s, 10, [&] { k_mm<0, false><<<grid, 256, 0, s>>>(A, out); });
    t[2] = time_ms(s, 10, [&] { k_mm<1, true><<<grid, 256, 0, s>>>(A, out); });
    t[3] = time_ms(s, 10, [&] { k_mm<1, false><<<grid, 256, 0, s>>>(A, out); });
    t[4] = time_ms(s, 10, [&] { k_mm<2, true><<<grid, 256, 0, s>>>(A, out); });
    t[5] = time_ms(s, 10, [&] { k_mm<2, false><<<grid, 256, 0, s>>>(A, out); });
    t[6] = time_ms(s, 10, [&] { k_mm<3, true><<<grid, 256, 0, s>>>(A, out); });
    t[7] = time_ms(s, 10, [&] { k_mm<3, false><<<grid, 256, 0, s>>>(A, out); });
    printf("mm 800 MB grid=%5d us: strided4 nt %.0f plain %.0f | tile4 nt %.0f plain %.0f | single nt %.0f plain %.0f | tile8 nt %.0f plain %.0f\n", grid,
           t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3, t[6] * 1e3, t[7] * 1e3);
  }
  CHECK(hipFree(a)); CHECK(hipFree(out));
}

// ---- ldsatom -------------------------------------------------------------------------------------------
enum { OP_ADD32, OP_ADD64, OP_ADDF64, OP_MINU64, OP_READ32, OP_CAS32 };
template <int OP>
__global__ void __launch_bounds__(256) k_ldsatom(uint32_t smask, int iters, uint64_t *out) {
  extern __shared__ __align__(16) uint8_t sm[];
  uint64_t *t64 = (uint64_t *)sm;
  uint32_t *t32 = (uint32_t *)sm;
  for (uint32_t i = threadIdx.x; i <= smask; i += blockDim.x) t64[i] = 0;
  __syncthreads();
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x = x * 1664525u + 1013904223u;
      const uint32_t slot = (x >> 10) & smask;
      if (OP == OP_ADD32) atomicAdd(&t32[slot], 1u);
      if (OP == OP_ADD64) atomicAdd((unsigned long long *)&t64[slot], (unsigned long long)x);
      if (OP == OP_ADDF64) atomicAdd((double *)&t64[slot], (double)x);
      if (OP == OP_MINU64) atomicMin((unsigned long long *)&t64[slot], (unsigned long long)x);
      if (OP == OP_READ32) acc += t32[slot];
      if (OP == OP_CAS32) acc += atomicCAS(&t32[slot], 0xFFFFFFFFu, x);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t64[acc & smask] + acc;
}

static void test_ldsatom() {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  uint64_t *out;
  CHECK(hipMalloc(&out, 8 * 65536));
  const int iters = 64;
  const char *names[] = {"add_u32", "add_u64", "add_f64", "min_u64", "read_b32", "cas_b32"};
  for (uint32_t slots : {64u, 128u, 1024u, 4096u}) {
    for (int wg_per_cu : {1, 4, 8}) {
      const int grid = 256 * wg_per_cu * 4; // 4 rounds of resident blocks
      float t[6];
      t[0] = time_ms(s, 5, [&] { k_ldsatom<OP_ADD32><<<grid, 256, slots * 8, s>>>(slots - 1, iters, out); });
      t[1] = time_ms(s, 5, [&] { k_ldsatom<OP_ADD64><<<grid, 256, slots * 8, s>>>(slots - 1, iters, out); });
      t[2] = time_ms(s, 5, [&] { k_ldsatom<OP_ADDF64><<<grid, 256, slots * 8, s>>>(slots - 1, iters, out); });
      t[3] = time_ms(s, 5, [&] { k_ldsatom<OP_MINU64><<<grid, 256, slots * 8, s>>>(slots - 1, iters, out); });
      t[4] = time_ms(s, 5, [&] { k_ldsatom<OP_READ32><<<grid, 256, slots * 8, s>>>(slots - 1, iters, out); });
      t[5] = time_ms(s, 5, [&] { k_ldsatom<OP_CAS32><<<grid, 256, slots * 8, s>>>(slots - 1, iters, out); });
      const double ops = (double)grid * 256 * iters * 8;
      printf("ldsatom slots=%u grid=%d(lds %u B/wg):", slots, grid, slots * 8);
      for (int i = 0; i < 6; i++) printf(" %s %.1f Gop/s (%.2f lanes/clk/CU)", names[i], ops / t[i] / 1e6, ops / t[i] / 1e6 / 256 / 2.4);
      printf("\n");
    }
  }
}

// ---- gatom ---------------------------------------------------------------------------------------------
template <int OP>
__global__ void __launch_bounds__(256) k_gatom(uint64_t *tab, uint64_t mask, int per_thread) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 999u;
  for (int i = 0; i < per_thread; i++) {
    x = x * 1664525u + 1013904223u;
    uint64_t slot = ((uint64_t)(x >> 4) * 2654435761ull >> 7) & mask;
    if (OP == 0) atomicAdd((unsigned long long *)&tab[slot], 1ull);
    if (OP == 1) atomicAdd((double *)&tab[slot], 1.0);
    if (OP == 2) atomicAdd((unsigned int *)&tab[slot], 1u);
    if (OP == 3) atomicMax((unsigned long long *)&tab[slot], (unsigned long long)x);
  }
}
static void test_gatom() {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const char *names[] = {"add_u64", "add_f64", "add_u32", "max_u64"};
  for (uint64_t slots : {(uint64_t)128, (uint64_t)1 << 14, (uint64_t)1 << 18, (uint64_t)1 << 24, (uint64_t)1 << 28}) {
    uint64_t *tab;
    CHECK(hipMalloc(&tab, slots * 8));
    CHECK(hipMemset(tab, 0, slots * 8));
    const int grid = 8192, per = 16;
    const double ops = (double)grid * 256 * per;
    printf("gatom slots=%llu (%.1f MB):", (unsigned long long)slots, slots * 8 / 1e6);
    float t;
    t = time_ms(s, 5, [&] { k_gatom<0><<<grid, 256, 0, s>>>(tab, slots - 1, per); }); printf(" %s %.2f Gop/s", names[0], ops / t / 1e6);
    t = time_ms(s, 5, [&] { k_gatom<1><<<grid, 256, 0, s>>>(tab, slots - 1, per); }); printf(" %s %.2f Gop/s", names[1], ops / t / 1e6);
    t = time_ms(s, 5, [&] { k_gatom<2><<<grid, 256, 0, s>>>(tab, slots - 1, per); }); printf(" %s %.2f Gop/s", names[2], ops / t / 1e6);
    t = time_ms(s, 5, [&] { k_gatom<3><<<grid, 256, 0, s>>>(tab, slots - 1, per); }); printf(" %s %.2f Gop/s", names[3], ops / t / 1e6);
    printf("\n");
    CHECK(hipFree(tab));
  }
}

int main(int argc, char **argv) {
  const char *t = argc > 1 ? argv[1] : "all";
  CHECK(hipSetDevice(0));
  if (!strcmp(t, "bw") || !strcmp(t, "all")) test_bw();
  if (!strcmp(t, "mm") || !strcmp(t, "all")) test_mm();
  if (!strcmp(t, "ldsatom") || !strcmp(t, "all")) test_ldsatom();
  if (!strcmp(t, "gatom") || !strcmp(t, "all")) test_gatom();
  return 0;
}
