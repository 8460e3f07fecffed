// dense_path.hip — the dense-code path for gfx950 (see dense.hpp): Categorical codes index LDS tables directly.
//
// HBM / L2 / LDS-atomic bound integer and f64 work: no MFMA.  Every row is read from HBM once per part (the R
// workgroups of a row slice share an XCD's L2), folded with LDS atomics by the one workgroup whose code range holds
// its key, and never written back: the only intermediates are the per-slice partial arrays (a few tens of MB).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dense.hpp"
#include "device.hpp"
#include "fast.hpp"
#include "kernels.hpp"

extern __shared__ __align__(16) uint8_t dsm[];

namespace {

DEVINL uint32_t al16u(uint32_t x) { return (x + 15u) & ~15u; }

DEVINL void dense_atomic(uint8_t *p, uint32_t op, uint32_t lw, uint64_t operand) {
  switch (op) {
    case ACC_ADD_U64:
      if (lw == 4) atomicAdd((unsigned int *)p, (unsigned int)operand); // the reference wraps in the column's 32 bits
      else atomicAdd((unsigned long long *)p, (unsigned long long)operand);
      break;
    case ACC_ADD_F64: atomicAdd((double *)p, __longlong_as_double((long long)operand)); break;
    case ACC_MIN_I64: atomicMin((long long *)p, (long long)operand); break;
    case ACC_MAX_I64: atomicMax((long long *)p, (long long)operand); break;
    case ACC_MIN_U64: atomicMin((unsigned long long *)p, (unsigned long long)operand); break;
    case ACC_MAX_U64: atomicMax((unsigned long long *)p, (unsigned long long)operand); break;
    default: atomicAdd((unsigned int *)p, 1u); break; // ACC_CNT_ALL / ACC_CNT_VALID
  }
}

DEVINL uint64_t dense_combine(uint32_t op, uint64_t a, uint64_t b) {
  switch (op) {
    case ACC_ADD_F64: return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    case ACC_MIN_I64: return (long long)b < (long long)a ? b : a;
    case ACC_MAX_I64: return (long long)b > (long long)a ? b : a;
    case ACC_MIN_U64: return b < a ? b : a;
    case ACC_MAX_U64: return b > a ? b : a;
    default: return a + b; // integer sums and counters
  }
}

// B rows of one lane on their way from the columns to the tables.  Everything that comes from global memory -- the key
// codes, the value column's raw bits, the validity bits of both -- is requested by dense_load; dense_process touches
// registers and LDS only, so that the next batch's loads stay in flight while this one is folded (a load inside the
// fold would make the wave wait for every older load first: vmcnt counts in order).
template <int B>
struct DenseRows {
  uint32_t key[B];
  uint64_t raw[B];
  uint32_t knull, vnull; // bit b: row b's key / value is null
};

// validity bits of the aligned quad of rows starting at `q` (q % 4 == 0): bit b set = row q + b is NULL
DEVINL uint32_t quad_nulls(const uint8_t *bm, int64_t boff, uint32_t q) {
  const int64_t bit = boff + (int64_t)q;
  const uint32_t sh = (uint32_t)bit & 7u;
  uint32_t w = bm[bit >> 3];
  if (sh > 4u) w |= (uint32_t)bm[(bit >> 3) + 1] << 8; // (only when the bitmap's offset is not a multiple of 4)
  return ~(w >> sh) & 0xFu;
}

template <int KIND, int B, bool QUAD, bool KV>
DEVINL void dense_load(const Plan &P, const DensePart &D, const uint32_t (&row)[B], DenseRows<B> &X) {
  uint64_t kraw[B];
  load_raw<B, QUAD>(P.keys[0].values, 4u, row, kraw);
#pragma unroll
  for (int b = 0; b < B; b++) X.key[b] = (uint32_t)kraw[b];
  if (KIND == DENSE_KIND_CNT_SUM32 || KIND == DENSE_KIND_SUM32) load_raw<B, QUAD>(P.vals[D.src].values, 4u, row, X.raw);
  else if (KIND == DENSE_KIND_SUMF64) load_raw<B, QUAD>(P.vals[D.src].values, 8u, row, X.raw);
  else if (D.src >= 0) load_raw<B, QUAD>(P.vals[D.src].values, wbytes(P.vals[D.src].dtype), row, X.raw);
  else {
#pragma unroll
    for (int b = 0; b < B; b++) X.raw[b] = 0;
  }
  X.knull = 0; X.vnull = 0;
  if (KV) {
    const uint8_t *bm = P.keys[0].validity;
    const int64_t boff = P.keys[0].boff;
    if (QUAD) {
#pragma unroll
      for (int g = 0; g < B / 4; g++) X.knull |= quad_nulls(bm, boff, row[4 * g]) << (4 * g);
    } else {
#pragma unroll
      for (int b = 0; b < B; b++) if (!bit_valid(bm, boff, (int64_t)row[b])) X.knull |= 1u << b;
    }
  }
  if (KIND == DENSE_KIND_RT && D.src >= 0 && P.vals[D.src].validity) {
    const ValSpec V = P.vals[D.src];
    if (QUAD && !V.valid_bytes) {
#pragma unroll
      for (int g = 0; g < B / 4; g++) X.vnull |= quad_nulls(V.validity, V.boff, row[4 * g]) << (4 * g);
    } else {
#pragma unroll
      for (int b = 0; b < B; b++) if (!val_valid(V, (int64_t)row[b])) X.vnull |= 1u << b;
    }
  }
}

// codes -> range-relative slots, then LDS atomics under the in-range mask (registers and LDS only)
template <int KIND, int B, bool KV>
DEVINL void dense_process(const Plan &P, const DenseArgs &A, const DensePart &D, uint8_t *tab, uint32_t c0, uint32_t cnt, bool is_last,
                          const DenseRows<B> &X, uint32_t &bad) {
  uint32_t code[B], rel[B];
#pragma unroll
  for (int b = 0; b < B; b++) code[b] = X.key[b] - A.key_bias;
  if (is_last) { // (uniform) the last range holds the null key's slot `hint` (perfect.rs:51-56) and sees every code >= hint
#pragma unroll
    for (int b = 0; b < B; b++) {
      const bool isnull = KV && ((X.knull >> b) & 1u);
      if (!isnull && code[b] >= A.hint) { bad = 1; code[b] = 0xFFFFFFFFu; } // not a code of this rev-map: the caller's promise is broken
      // All null keys of the slice meet in ONE slot, and same-address LDS atomics serialise (~5 cycles a row: db-benchmark's
      // NA variant took 0.40 ms instead of 0.06).  So every lane folds its null keys into a replica of that slot of its own
      // (64 of them behind the range's `cnt` entries, conflict-free), and the replicas are added up before the dump.
      if (isnull) code[b] = c0 + cnt + (threadIdx.x & 63u);
    }
  } else if (KV) { // a null key's code bits mean nothing: keep the row out of this range
#pragma unroll
    for (int b = 0; b < B; b++) if ((X.knull >> b) & 1u) code[b] = 0xFFFFFFFFu;
  }
  const uint32_t cnt_r = cnt + ((KV && is_last) ? 64u : 0u); // entries + the null slot's replicas
  uint32_t inm = 0;
#pragma unroll
  for (int b = 0; b < B; b++) {
    rel[b] = code[b] - c0;
    if (rel[b] < cnt_r) inm |= 1u << b;
  }
  if (KIND == DENSE_KIND_CNT_SUM32) {
    uint8_t *t0 = tab + D.loff[0], *t1 = tab + D.loff[1];
#pragma unroll
    for (int b = 0; b < B; b++)
      if ((inm >> b) & 1) {
        atomicAdd((unsigned int *)(t0 + rel[b] * 4u), 1u);
        atomicAdd((unsigned int *)(t1 + rel[b] * 4u), (unsigned int)X.raw[b]);
      }
    return;
  }
  if (KIND == DENSE_KIND_SUM32) {
    uint8_t *t0 = tab + D.loff[0];
#pragma unroll
    for (int b = 0; b < B; b++)
      if ((inm >> b) & 1) atomicAdd((unsigned int *)(t0 + rel[b] * 4u), (unsigned int)X.raw[b]);
    return;
  }
  if (KIND == DENSE_KIND_SUMF64) {
    uint8_t *t0 = tab + D.loff[0];
#pragma unroll
    for (int b = 0; b < B; b++)
      if ((inm >> b) & 1) atomicAdd((double *)(t0 + rel[b] * 8u), __longlong_as_double((long long)X.raw[b]));
    return;
  }
  // run-time part
  const uint32_t vdt = D.src >= 0 ? P.vals[D.src].dtype : (uint32_t)PLHIP_I32;
  for (uint32_t i = 0; i < D.n_accs; i++) {
    const uint32_t op = P.accs[D.acc[i]].op, conv = P.accs[D.acc[i]].conv, lw = D.lw[i];
    uint8_t *arr = tab + D.loff[i];
    if (op == ACC_CNT_ALL) {
#pragma unroll
      for (int b = 0; b < B; b++) if ((inm >> b) & 1) atomicAdd((unsigned int *)(arr + rel[b] * 4u), 1u);
      continue;
    }
    const uint32_t m = inm & ~X.vnull;
    if (op == ACC_CNT_VALID) {
#pragma unroll
      for (int b = 0; b < B; b++) if ((m >> b) & 1) atomicAdd((unsigned int *)(arr + rel[b] * 4u), 1u);
      continue;
    }
    uint64_t opnd[B];
    raw_to_operands<B>(vdt, conv, X.raw, opnd);
#pragma unroll
    for (int b = 0; b < B; b++) if ((m >> b) & 1) dense_atomic(arr + rel[b] * lw, op, lw, opnd[b]);
  }
}

// ---- the streamlined row loop of the straight-line kinds (no validity bitmaps anywhere) ----------------------------------
// A wave keeps TWO batches in flight behind the one it folds (three register buffers, 12 KB of loads per wave, ~200 KB per
// CU): what a workgroup can pull out of its XCD's L2 is bounded by the bytes it has in flight, and with one batch per wave a
// CU reached 65 GB/s.  The compiler's s_waitcnt placement cannot express that (it waited for the loads it had just issued),
// so the loads and their waits are written by hand: global_load_dwordx4 with an SGPR base + 32-bit lane offset, counted
// s_waitcnt vmcnt(N) before a buffer's first use (vmcnt retires in order), and an empty asm that ties the buffer's registers
// to that wait.  Nothing else in the loop touches vector memory (LDS atomics count in lgkmcnt).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// VQ: 16-byte loads per quad of rows of the value column: 1 (4-byte values) or 2 (8-byte values); kb / vb: the byte of the
// key's / the value's validity bitmap that holds the quad's four bits (bitmap offsets are multiples of 4 here: host check)
template <int VQ>
struct DenseBuf {
  u32x4 k[2];
  u32x4 v[2 * VQ];
  uint32_t kb[2], vb[2];
};
DEVINL u32x4 gload16(const uint8_t *sbase, uint32_t voff) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
DEVINL uint32_t gload1(const uint8_t *sbase, uint32_t voff) {
  uint32_t v;
  asm volatile("global_load_ubyte %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
struct DenseBases { // wave-uniform: the slice's first element of every stream, and the bit offsets of its first row
  const uint8_t *k, *v, *kbm, *vbm;
  uint32_t ksh0, vsh0; // 0 or 4
};
template <int VQ, bool KV, bool VV, int VL = VQ> // VL: value loads per quad (0: the part never reads the values, DENSE_KIND_CNTV)
DEVINL void dense_issue(DenseBuf<VQ> &X, const DenseBases &S, uint32_t row_in_slice) {
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const uint32_t q = row_in_slice + ((uint32_t)g * DENSE_BLOCK + threadIdx.x) * 4u; // first row of the lane's quad g
    X.k[g] = gload16(S.k, q * 4u);
#pragma unroll
    for (int h = 0; h < VL; h++) X.v[g * VQ + h] = gload16(S.v, q * (4u * VQ) + 16u * h);
    if (KV) X.kb[g] = gload1(S.kbm, (S.ksh0 + q) >> 3);
    if (VV) X.vb[g] = gload1(S.vbm, (S.vsh0 + q) >> 3);
  }
}
template <int N> DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int VQ, bool KV, bool VV, int VL = VQ>
DEVINL void dense_pin(DenseBuf<VQ> &X) {
#pragma unroll
  for (int g = 0; g < 2; g++) { u32x4 t = X.k[g]; asm volatile("" : "+v"(t)::"memory"); X.k[g] = t; }
#pragma unroll
  for (int g = 0; g < 2 * VL; g++) { u32x4 t = X.v[g]; asm volatile("" : "+v"(t)::"memory"); X.v[g] = t; }
#pragma unroll
  for (int g = 0; g < 2; g++) {
    if (KV) { uint32_t t = X.kb[g]; asm volatile("" : "+v"(t)::"memory"); X.kb[g] = t; }
    if (VV) { uint32_t t = X.vb[g]; asm volatile("" : "+v"(t)::"memory"); X.vb[g] = t; }
  }
}
template <int KIND, int VQ, bool KV, bool VV>
DEVINL void dense_fold_buf(const DenseArgs &A, const DensePart &D, uint8_t *tab, uint32_t c0, uint32_t cnt, bool is_last, const DenseBuf<VQ> &X,
                           const DenseBases &S, uint32_t row_in_slice, uint32_t &bad) {
  constexpr bool TWO = KIND == DENSE_KIND_CNT_SUM32 || KIND == DENSE_KIND_CNTV_SUMF64;
  uint8_t *t0 = tab + D.loff[0], *t1 = tab + D.loff[TWO ? 1 : 0];
  const uint32_t cnt_r = cnt + ((KV && is_last) ? 64u : 0u); // entries + the null slot's 64 replicas (dense_process)
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const uint32_t q = row_in_slice + ((uint32_t)g * DENSE_BLOCK + threadIdx.x) * 4u;
    const uint32_t key[4] = {X.k[g].x, X.k[g].y, X.k[g].z, X.k[g].w};
    const uint32_t knull = KV ? ~(X.kb[g] >> ((S.ksh0 + q) & 4u)) & 0xFu : 0u;
    const uint32_t vnull = VV ? ~(X.vb[g] >> ((S.vsh0 + q) & 4u)) & 0xFu : 0u;
    uint32_t rel[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      uint32_t code = key[b] - A.key_bias;
      const bool isnull = KV && ((knull >> b) & 1u);
      if (is_last) { // (uniform test)
        if (!isnull && code >= A.hint) { bad = 1; code = 0xFFFFFFFFu; } // not a code of this rev-map
        if (isnull) code = c0 + cnt + (threadIdx.x & 63u);              // the lane's replica of the null slot
      } else if (isnull) code = 0xFFFFFFFFu;
      rel[b] = code - c0;
    }
    if (KIND == DENSE_KIND_CNTV) {
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (rel[b] < cnt_r && !((vnull >> b) & 1u)) atomicAdd((unsigned int *)(t0 + rel[b] * 4u), 1u);
    } else if (KIND == DENSE_KIND_SUMF64 || KIND == DENSE_KIND_CNTV_SUMF64) {
      const u32x4 a = X.v[2 * g], b2 = X.v[2 * g + 1];
      const double d[4] = {__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z), __hiloint2double((int)b2.y, (int)b2.x),
                           __hiloint2double((int)b2.w, (int)b2.z)};
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (rel[b] < cnt_r && !((vnull >> b) & 1u)) {
          if (KIND == DENSE_KIND_CNTV_SUMF64) atomicAdd((unsigned int *)(t0 + rel[b] * 4u), 1u);
          atomicAdd((double *)(t1 + rel[b] * 8u), d[b]);
        }
    } else {
      const uint32_t v[4] = {X.v[g].x, X.v[g].y, X.v[g].z, X.v[g].w};
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (rel[b] < cnt_r) {
          if (KIND == DENSE_KIND_CNT_SUM32) atomicAdd((unsigned int *)(t0 + rel[b] * 4u), 1u);
          atomicAdd((unsigned int *)(t1 + rel[b] * 4u), ((vnull >> b) & 1u) ? 0u : v[b]); // a null adds nothing to a sum
        }
    }
  }
}
template <int KIND, bool KV, bool VV>
DEVINL uint32_t dense_fold_rows_fast(const Plan &P, const DenseArgs &A, const DensePart &D, uint8_t *tab, uint32_t c0, uint32_t cnt, bool is_last,
                                     int64_t row0, uint32_t nb, uint32_t &bad) {
  constexpr int VQ = (KIND == DENSE_KIND_SUMF64 || KIND == DENSE_KIND_CNTV_SUMF64) ? 2 : 1;
  constexpr int VL = KIND == DENSE_KIND_CNTV ? 0 : VQ;          // value loads per quad
  constexpr int NL = 2 + 2 * VL + (KV ? 2 : 0) + (VV ? 2 : 0); // loads per batch
  DenseBases S;
  S.k = P.keys[0].values + row0 * 4; S.v = P.vals[D.src].values + row0 * (4 * VQ);
  S.kbm = nullptr; S.vbm = nullptr; S.ksh0 = 0; S.vsh0 = 0;
  if (KV) { const int64_t bit = P.keys[0].boff + row0; S.kbm = P.keys[0].validity + (bit >> 3); S.ksh0 = (uint32_t)bit & 7u; }
  if (VV) { const int64_t bit = P.vals[D.src].boff + row0; S.vbm = P.vals[D.src].validity + (bit >> 3); S.vsh0 = (uint32_t)bit & 7u; }
  DenseBuf<VQ> b0, b1, b2;
  if (nb > 0) dense_issue<VQ, KV, VV, VL>(b0, S, 0u);
  if (nb > 1) dense_issue<VQ, KV, VV, VL>(b1, S, (uint32_t)DENSE_BATCH);
#define DENSE_STEP(CUR, NEXT)                                                                     \
  {                                                                                               \
    if (t + 2 < nb) dense_issue<VQ, KV, VV, VL>(NEXT, S, (t + 2) * (uint32_t)DENSE_BATCH);        \
    const uint32_t ahead = nb - 1 - t; /* batches issued behind this one */                       \
    if (ahead >= 2) wait_vm<2 * NL>(); else if (ahead == 1) wait_vm<NL>(); else wait_vm<0>();     \
    dense_pin<VQ, KV, VV, VL>(CUR);                                                               \
    dense_fold_buf<KIND, VQ, KV, VV>(A, D, tab, c0, cnt, is_last, CUR, S, t * (uint32_t)DENSE_BATCH, bad); \
    ++t;                                                                                          \
  }
  uint32_t t = 0;
  while (t < nb) {
    DENSE_STEP(b0, b2)
    if (t >= nb) break;
    DENSE_STEP(b1, b0)
    if (t >= nb) break;
    DENSE_STEP(b2, b1)
  }
#undef DENSE_STEP
  return nb;
}

template <int KIND, bool KV>
DEVINL void dense_fold_rows(const Plan &P, const DenseArgs &A, const DensePart &D, uint8_t *tab, uint32_t c0, uint32_t cnt, bool is_last,
                            int64_t row0, int64_t row1, uint32_t &bad) {
  // whole batches: two aligned quads per lane, 16-byte loads; the next batch is requested before this one is folded
  auto mkrows = [&](int64_t base, uint32_t (&row)[8]) {
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const uint32_t q = (uint32_t)base + ((uint32_t)g * DENSE_BLOCK + threadIdx.x) * 4u;
#pragma unroll
      for (int b = 0; b < 4; b++) row[4 * g + b] = q + (uint32_t)b;
    }
  };
  const uint32_t nb = (uint32_t)((row1 - row0) / DENSE_BATCH);
  uint32_t row[8];
  uint32_t t = 0;
  if constexpr (KIND == DENSE_KIND_RT) { // (the run-time part has no registers to spare for a batch in flight)
    DenseRows<8> X;
    for (; t < nb; t++) {
      mkrows(row0 + (int64_t)t * DENSE_BATCH, row);
      dense_load<KIND, 8, true, KV>(P, D, row, X);
      dense_process<KIND, 8, KV>(P, A, D, tab, c0, cnt, is_last, X, bad);
    }
  } else { // the straight-line kinds: hand-pipelined loads (bitmap offsets are multiples of 4: host check)
    if (D.pad0) t = dense_fold_rows_fast<KIND, KV, true>(P, A, D, tab, c0, cnt, is_last, row0, nb, bad);
    else t = dense_fold_rows_fast<KIND, KV, false>(P, A, D, tab, c0, cnt, is_last, row0, nb, bad);
  }
  for (int64_t r = row0 + (int64_t)nb * DENSE_BATCH + threadIdx.x; r < row1; r += DENSE_BLOCK) { // the slice's last rows: one per lane
    const uint32_t row1_[1] = {(uint32_t)r};
    DenseRows<1> X;
    dense_load<DENSE_KIND_RT, 1, false, KV>(P, D, row1_, X); // (one row per lane: the run-time form of every kind)
    dense_process<DENSE_KIND_RT, 1, KV>(P, A, D, tab, c0, cnt, is_last, X, bad);
  }
}

// RT = false: every part is one of the straight-line kinds (fewer registers).  KV: the key column has a validity bitmap.
template <bool RT, bool KV>
__global__ void __launch_bounds__(DENSE_BLOCK) k_dense_fold(Plan P, DenseArgs A) {
  __shared__ uint32_t sh[4];
  // workgroup -> (group slot, range): the R workgroups of a slot share blockIdx.x % xcds, i.e. (as observed, for speed only)
  // an XCD and its L2, which serves R - 1 of the slot's R reads of every row; part p owns the slots [slot0, slot0 + n_slices)
  uint32_t range, slice;
  int p = -1;
  if (A.pmode) { // partition mode: workgroup -> (part, partition, slice of the partition); the partition's index is the range's
    // (32-bit divisions run on the vector unit; readfirstlane hands the wave-uniform results back to the scalar side, where the
    // row loop's inline-asm loads want their base pointers)
    slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x % A.S));
    range = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x / A.S) % A.R));
    p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (A.S * A.R)));
    if (p >= (int)A.n_parts) return;
  } else {
    const uint32_t x = blockIdx.x % A.xcds, q = blockIdx.x / A.xcds;
    const uint32_t gs = (q / A.R) * A.xcds + x;
    range = q % A.R;
    for (uint32_t i = 0; i < A.n_parts; i++)
      if (gs >= A.part[i].slot0 && gs < A.part[i].slot0 + A.part[i].n_slices) p = (int)i;
    if (p < 0) return;
    slice = gs - A.part[p].slot0;
  }
  const DensePart &D = A.part[p];
  const uint32_t c0 = range * A.gpr;
  const uint32_t cnt = c0 < A.n_slots ? (A.n_slots - c0 < A.gpr ? A.n_slots - c0 : A.gpr) : 0u;
  const bool is_last = cnt != 0 && c0 + cnt == A.n_slots;
  if (threadIdx.x == 0) sh[0] = 0;
  const uint32_t n_init = A.gpr + (KV ? 64u : 0u); // (a nullable key: 64 replicas of the null slot behind the entries, host-side layout)
  for (uint32_t i = 0; i < D.n_accs; i++) {
    const uint64_t init = P.accs[D.acc[i]].init;
    if (D.lw[i] == 8) for (uint32_t j = threadIdx.x; j < n_init; j += DENSE_BLOCK) ((uint64_t *)(dsm + D.loff[i]))[j] = init;
    else for (uint32_t j = threadIdx.x; j < n_init; j += DENSE_BLOCK) ((uint32_t *)(dsm + D.loff[i]))[j] = (uint32_t)init;
  }
  __syncthreads();
  int64_t row0 = (int64_t)slice * D.rows_per_slice, row1 = row0 + D.rows_per_slice;
  if (A.pmode) { // the partition's rows, cut into S slices of whole batches
    const int64_t pr = (int64_t)A.part_rows[range], p0 = (int64_t)A.part_start[range];
    const int64_t per = ((pr + A.S - 1) / A.S + DENSE_BATCH - 1) / DENSE_BATCH * DENSE_BATCH;
    row0 = p0 + (int64_t)slice * per; row1 = row0 + per;
    if (row1 > p0 + pr) row1 = p0 + pr;
    if ((uint64_t)per * 8u >= 0xFFFFFFFFull) { // (the row loop addresses a slice with 32-bit byte offsets: a partition this large -- extreme
      if (threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1; // skew at >= 5e8 rows -- goes back to the hashed path)
      row1 = row0;
    }
  } else if (row1 > P.n_rows) row1 = P.n_rows;
  if (row0 > row1) row0 = row1;
  { // wave-uniform by construction (block index, uniform loads); said so to the compiler: the row loop's loads take their base in SGPRs
    row0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)row0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)row0));
    row1 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)row1 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)row1));
  }
  uint32_t bad = 0;
  if (cnt) {
    switch (D.kind) {
      case DENSE_KIND_CNT_SUM32: dense_fold_rows<DENSE_KIND_CNT_SUM32, KV>(P, A, D, dsm, c0, cnt, is_last, row0, row1, bad); break;
      case DENSE_KIND_SUMF64: dense_fold_rows<DENSE_KIND_SUMF64, KV>(P, A, D, dsm, c0, cnt, is_last, row0, row1, bad); break;
      case DENSE_KIND_SUM32: dense_fold_rows<DENSE_KIND_SUM32, KV>(P, A, D, dsm, c0, cnt, is_last, row0, row1, bad); break;
      case DENSE_KIND_CNTV_SUMF64: dense_fold_rows<DENSE_KIND_CNTV_SUMF64, KV>(P, A, D, dsm, c0, cnt, is_last, row0, row1, bad); break;
      case DENSE_KIND_CNTV: dense_fold_rows<DENSE_KIND_CNTV, KV>(P, A, D, dsm, c0, cnt, is_last, row0, row1, bad); break;
      default:
        if constexpr (RT) dense_fold_rows<DENSE_KIND_RT, KV>(P, A, D, dsm, c0, cnt, is_last, row0, row1, bad);
        break;
    }
  }
  if (bad) sh[0] = 1;
  __syncthreads();
  if (sh[0] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1; // a code outside [0, hint): the caller falls back to the hashed path
  if (KV && is_last && threadIdx.x < D.n_accs) { // the null slot's replicas -> the slot itself (entry cnt - 1 of the last range)
    const uint32_t i = threadIdx.x, op = P.accs[D.acc[i]].op, lw = D.lw[i];
    uint8_t *arr = dsm + D.loff[i];
    uint64_t acc = lw == 8 ? ((uint64_t *)arr)[cnt - 1] : (uint64_t)((uint32_t *)arr)[cnt - 1];
    for (uint32_t r = 0; r < 64; r++) acc = dense_combine(op, acc, lw == 8 ? ((uint64_t *)arr)[cnt + r] : (uint64_t)((uint32_t *)arr)[cnt + r]);
    if (lw == 8) ((uint64_t *)arr)[cnt - 1] = acc; else ((uint32_t *)arr)[cnt - 1] = (uint32_t)acc;
  }
  __syncthreads();
  // the range's arrays -> the slice's partial arrays (plain 16-byte stores)
  uint8_t *dslice = A.dump + D.dump_off + (uint64_t)(A.pmode ? range * A.S + slice : slice) * D.slice_stride;
  for (uint32_t i = 0; i < D.n_accs; i++) {
    const uint32_t lw = D.lw[i], chunks = al16u(A.gpr * lw) >> 4;
    const uint4 *src = (const uint4 *)(dsm + D.loff[i]);
    uint4 *dst = (uint4 *)(dslice + D.doff[i] + (uint64_t)(A.pmode ? 0u : c0) * lw); // (partition mode: a slice's arrays hold its range only)
    for (uint32_t c = threadIdx.x; c < chunks; c += DENSE_BLOCK) dst[c] = src[c];
  }
}

// gb groups per workgroup, W "ways": thread (w, j) owns the four consecutive groups 4j .. 4j + 3 and adds up slices w, w + W, ...
// of their entries in every accumulator's partial arrays (16-byte loads, MG_U slices in flight, fixed order: the result does
// not depend on the launch); the W partial results meet pairwise in LDS, then the workgroup compacts the groups that
// occurred and writes their output rows (thread k writes row base + k of every column: coalesced; validity bits as three
// 32-bit ORs per wave).  ONE global atomic per workgroup carries both the output-row reservation and the done count.
// (Measured, q3: ~6 us of the kernel's ~21 are the boundary behind k_dense_fold's 38 MB of dirty partial arrays, ~9 the
// dependent rounds of loads -- every slice is a far-away, freshly written page -- ~4 the reservation atomics, which
// serialise on one address; spreading accumulators over threads to get one round of loads was measured and was slower.)
#define MG_U 4
// the slices' partial arrays of this workgroup's gb groups, added up: part[a][0][0 .. gb) holds the finals afterwards
__device__ __forceinline__ void dense_sum_slices(const Plan &P, const DenseArgs &A, uint64_t *part, uint32_t gb, uint32_t W) {
  const uint32_t tid = threadIdx.x;
  const uint32_t q4 = gb >> 2;                // quads of groups per workgroup
  const uint32_t j = tid % q4, w4 = tid / q4; // (w4 < W: blockDim = q4 * W)
  const uint32_t g0 = blockIdx.x * gb + 4u * j;
  const bool have = g0 < A.R * A.gpr; // the partial arrays hold R * gpr entries (a multiple of 4)
  for (uint32_t p = 0; p < A.n_parts; p++) {
    const DensePart &D = A.part[p];
    for (uint32_t i = 0; i < D.n_accs; i++) {
      const uint32_t a = D.acc[i], op = P.accs[a].op, lw = D.lw[i];
      const uint64_t init = lw == 8 ? P.accs[a].init : (uint64_t)(uint32_t)P.accs[a].init;
      uint64_t acc[4] = {init, init, init, init};
      if (have) {
        const uint8_t *col = A.dump + D.dump_off + D.doff[i] + (uint64_t)g0 * lw;
        if (A.pmode) { // the S slices of the quad's partition, gpr entries each (gpr is a multiple of 4: a quad has one partition)
          const uint32_t pp = g0 / A.gpr;
          col = A.dump + D.dump_off + (uint64_t)pp * A.S * D.slice_stride + D.doff[i] + (uint64_t)(g0 - pp * A.gpr) * lw;
        }
        for (uint32_t s0 = w4; s0 < D.n_slices; s0 += MG_U * W) {
          uint64_t v[MG_U][4];
#pragma unroll
          for (int u = 0; u < MG_U; u++) {
            const uint32_t sl = s0 + (uint32_t)u * W;
#pragma unroll
            for (int e = 0; e < 4; e++) v[u][e] = init;
            if (sl < D.n_slices) {
              const uint8_t *ptr = col + (uint64_t)sl * D.slice_stride;
              if (lw == 8) {
                const uint4 x0 = *(const uint4 *)ptr, x1 = *(const uint4 *)(ptr + 16);
                v[u][0] = (uint64_t)x0.x | ((uint64_t)x0.y << 32); v[u][1] = (uint64_t)x0.z | ((uint64_t)x0.w << 32);
                v[u][2] = (uint64_t)x1.x | ((uint64_t)x1.y << 32); v[u][3] = (uint64_t)x1.z | ((uint64_t)x1.w << 32);
              } else {
                const uint4 x0 = *(const uint4 *)ptr;
                v[u][0] = x0.x; v[u][1] = x0.y; v[u][2] = x0.z; v[u][3] = x0.w;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < MG_U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] = dense_combine(op, acc[e], v[u][e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; e++) part[((size_t)a * W + w4) * gb + 4u * j + e] = acc[e];
    }
  }
  // the W ways meet pairwise (W is a power of two): every thread still holds its quad of groups
  for (uint32_t half = W >> 1; half >= 1; half >>= 1) {
    __syncthreads();
    if (w4 < half) {
      for (int a = 0; a < P.n_accs; a++) {
        const uint32_t op = P.accs[a].op;
        uint64_t *lo = part + ((size_t)a * W + w4) * gb + 4u * j, *hi = part + ((size_t)a * W + w4 + half) * gb + 4u * j;
#pragma unroll
        for (int e = 0; e < 4; e++) lo[e] = dense_combine(op, lo[e], hi[e]);
      }
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(DENSE_BLOCK) k_dense_merge(Plan P, DenseArgs A, OutPlan O, uint64_t out_cap, uint32_t *host_sig, uint32_t seq, uint32_t gb,
                                                             uint32_t W) {
  __shared__ __align__(16) uint32_t tmp[DENSE_BLOCK / 64];
  __shared__ uint32_t s_base, s_n;
  uint64_t *part = (uint64_t *)dsm;                                       // [n_accs][W][gb]; way 0 ends up holding the finals
  uint32_t *list = (uint32_t *)(dsm + (size_t)P.n_accs * W * gb * 8u);    // [gb]
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  dense_sum_slices(P, A, part, gb, W);
  const uint32_t gl = tid, w = tid < gb ? 0u : 1u; // from here on: one thread per group (the first gb threads)
  const uint32_t g = blockIdx.x * gb + (tid < gb ? tid : 0u);
  const bool live = tid < gb && g < A.n_slots;
#define FIN(a, s) part[(size_t)(a) * W * gb + (s)]
  const bool present = w == 0 && live && (uint32_t)FIN(A.cnt_acc, gl) != 0u;
  // block-wide exclusive scan of `present`
  const unsigned long long m = __ballot(present);
  const uint32_t wave = tid >> 6, nw = blockDim.x >> 6;
  if (lane == 0) tmp[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t x = 0; x < nw; x++) { const uint32_t c = tmp[x]; if (x < wave) before += c; total += c; }
  if (present) list[before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = gl;
  if (tid == 0) {
    // one 64-bit atomic: low word = output rows reserved so far, high word = workgroups done
    const unsigned long long old = atomicAdd((unsigned long long *)(A.flags + 8), (1ull << 32) | (unsigned long long)total);
    s_base = (uint32_t)old;
    s_n = (uint32_t)(old >> 32);
  }
  __syncthreads();
  const uint32_t n = total;
  const uint64_t gbase = s_base;
  const bool over = n && gbase + n > out_cap;
  if (!over) {
    for (uint32_t k0 = 0; k0 < n; k0 += blockDim.x) {
      const uint32_t k = k0 + tid;
      const bool livek = k < n;
      const uint32_t s = livek ? list[k] : 0u;
      const uint32_t gg = A.slot_base + blockIdx.x * gb + s;
      const uint64_t i = gbase + k;
      for (int o = 0; o < O.n_outs; o++) {
        const OutSpec &S = O.outs[o];
        bool valid = false;
        if (livek) {
          if (S.kind == OUT_KEY) { // the key's normalised bits (device.hpp load_key_norm) from its slot
            const uint32_t rawk = gg + A.key_bias;
            valid = finalize_key(P, S, (uint64_t)(P.keys[0].dtype == PLHIP_I32 ? rawk ^ 0x80000000u : rawk), gg == A.hint, false, i);
          }
          else {
            const uint64_t acc = S.acc >= 0 ? FIN(S.acc, s) : 0ull;
            const uint32_t cnt_all = S.cnt_all >= 0 ? (uint32_t)FIN(S.cnt_all, s) : 0u;
            const uint32_t cnt_valid = S.cnt_valid >= 0 ? (uint32_t)FIN(S.cnt_valid, s) : cnt_all;
            valid = finalize_value(S, acc, cnt_all, cnt_valid, i);
          }
        }
        if (S.validity) { // a wave covers 64 consecutive output rows
          const unsigned long long mk = __ballot(valid);
          if (mk) {
            const uint64_t i0 = gbase + (k0 + (tid & ~63u));
            const uint32_t sft = (uint32_t)(i0 & 31);
            const unsigned long long lo = mk << sft, hi = sft ? mk >> (64 - sft) : 0ull;
            const uint32_t word = lane == 0 ? (uint32_t)lo : lane == 1 ? (uint32_t)(lo >> 32) : (uint32_t)hi;
            if (lane < 3 && word) atomicOr((unsigned int *)S.validity + (i0 >> 5) + lane, word);
          }
        }
      }
    }
  }
#undef FIN
  if (tid == 0) {
    if (over) atomicOr(&A.flags[FLAG_OVERFLOW], 1u);
    if (s_n == gridDim.x - 1 && host_sig) { // every other workgroup has reserved its rows: the total is known
      uint32_t o = atomicAdd(&A.flags[FLAG_OVERFLOW], 0u) | (over ? 1u : 0u);
      for (uint32_t h = 0; h < A.n_hdr; h++) o |= *(const uint32_t *)(A.dump + (uint64_t)h * A.part[0].slice_stride); // a sender off the dense path
      const uint64_t groups = gbase + n;
      A.flags[FLAG_NGROUPS] = (uint32_t)groups; // (for a host that reads the flag block instead of the signal)
      // this workgroup reserved last: nobody touches the counters any more, and `o` is final (had any workgroup run past out_cap, the
      // last reservation would have too).  A clean run leaves the flag block ready for the next dense query (api.hip flags_clean = 2).
      if (!o) { A.flags[8] = 0u; A.flags[9] = 0u; }
      __hip_atomic_store((uint64_t *)host_sig, ((uint64_t)seq << 32) | (o ? 0x80000000u : 0u) | ((uint32_t)groups & 0x7FFFFFFFu), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The sharded dense path's send side: the same sums, written raw and dense into the owners' slots (dense.hpp).
__global__ void __launch_bounds__(DENSE_BLOCK) k_dense_scatter(Plan P, DenseArgs A, DenseXchg X, uint32_t gb, uint32_t W) {
  uint64_t *part = (uint64_t *)dsm; // [n_accs][W][gb]
  const uint32_t tid = threadIdx.x;
  dense_sum_slices(P, A, part, gb, W);
  const uint32_t g = blockIdx.x * gb + tid;
  if (tid < gb && g < X.per * X.world) {
    const uint32_t owner = g / X.per, s = g - owner * X.per;
    uint8_t *slot = X.send + (uint64_t)owner * X.slot_bytes;
    for (uint32_t p = 0; p < A.n_parts; p++) {
      const DensePart &D = A.part[p];
      for (uint32_t i = 0; i < D.n_accs; i++) {
        const uint64_t v = part[(size_t)D.acc[i] * W * gb + tid];
        if (D.lw[i] == 8) *(uint64_t *)(slot + X.xoff[p][i] + (uint64_t)s * 8u) = v;
        else *(uint32_t *)(slot + X.xoff[p][i] + (uint64_t)s * 4u) = (uint32_t)v;
      }
    }
  }
  if (blockIdx.x == 0 && tid < X.world) { // the header every peer receives: did this rank keep the dense promise?
    uint32_t *h = (uint32_t *)(X.send + (uint64_t)tid * X.slot_bytes);
    h[0] = (A.flags[FLAG_OVERFLOW] | X.local_bad) ? 1u : 0u;
    h[1] = 0u; h[2] = X.per; h[3] = X.world;
  }
}

// Partition mode, pass 1: rows per (code range, tile) -- what the tile sort's scatter variant places its runs by.  One
// workgroup per TS_TILE rows, the key column in 16-byte loads, LDS atomics on the range counters.
__global__ void __launch_bounds__(1024) k_dense_part_hist(Plan P, uint32_t bias, uint32_t gpr, uint32_t recip, uint32_t np, uint32_t np_pow2,
                                                         uint32_t n_codes, uint32_t n_tiles, uint32_t *__restrict__ counts, uint32_t *__restrict__ flags) {
  __shared__ uint32_t h[1u << TS_MAX_PBITS];
  __shared__ uint32_t s_bad;
  for (uint32_t i = threadIdx.x; i < np_pow2; i += 1024) h[i] = 0;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * TS_TILE;
  const int64_t r1 = r0 + TS_TILE < P.n_rows ? r0 + TS_TILE : P.n_rows;
  const uint32_t *keys = (const uint32_t *)P.keys[0].values;
  uint32_t bad = 0;
  for (int64_t q = r0 + (int64_t)threadIdx.x * 4; q < r1; q += 4096) {
    uint32_t k[4];
    uint32_t nk = 4;
    if (q + 4 <= r1 && (((uintptr_t)(keys + q)) & 15u) == 0) { const uint4 v = *(const uint4 *)(keys + q); k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w; }
    else { nk = (uint32_t)(r1 - q < 4 ? r1 - q : 4); for (uint32_t b = 0; b < nk; b++) k[b] = keys[q + b]; }
    for (uint32_t b = 0; b < nk; b++) {
      const uint32_t code = k[b] - bias;
      if (code >= n_codes) bad = 1;
      atomicAdd(&h[range_part(code, gpr, recip, np)], 1u);
    }
  }
  if (bad) s_bad = 1;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < np_pow2; i += 1024) counts[(uint64_t)i * n_tiles + blockIdx.x] = h[i];
  if (threadIdx.x == 0 && s_bad) flags[FLAG_OVERFLOW] = 1; // a code outside the promised range: the caller falls back to the hashed path
}

} // namespace

void launch_dense_part_hist(const Plan &p, uint32_t bias, uint32_t gpr, uint32_t recip, uint32_t np, uint32_t np_pow2, uint32_t n_codes,
                            uint32_t n_tiles, uint32_t *counts, uint32_t *flags, hipStream_t s) {
  hipLaunchKernelGGL(k_dense_part_hist, dim3(n_tiles), dim3(1024), 0, s, p, bias, gpr, recip, np, np_pow2, n_codes, n_tiles, counts, flags);
}

template <bool RT, bool KV>
static void launch_dense_fold_t(const Plan &p, const DenseArgs &a, uint32_t n_wgs, size_t lds_bytes, hipStream_t s) {
  static bool once = false;
  if (!once) { // the whole LDS of a CU but the kernel's few static words
    if (hipFuncSetAttribute((const void *)k_dense_fold<RT, KV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) (void)hipGetLastError();
    once = true;
  }
  hipLaunchKernelGGL((k_dense_fold<RT, KV>), dim3(n_wgs), dim3(DENSE_BLOCK), lds_bytes, s, p, a);
}

void launch_dense_fold(const Plan &p, const DenseArgs &a, uint32_t n_wgs, size_t lds_bytes, hipStream_t s) {
  bool rt = false;
  for (uint32_t i = 0; i < a.n_parts; i++) rt |= a.part[i].kind == DENSE_KIND_RT;
  const bool kv = p.keys[0].validity != nullptr;
  if (rt) { if (kv) launch_dense_fold_t<true, true>(p, a, n_wgs, lds_bytes, s); else launch_dense_fold_t<true, false>(p, a, n_wgs, lds_bytes, s); }
  else { if (kv) launch_dense_fold_t<false, true>(p, a, n_wgs, lds_bytes, s); else launch_dense_fold_t<false, false>(p, a, n_wgs, lds_bytes, s); }
}

static void dense_merge_shape(const Plan &p, uint32_t groups, uint32_t &gb, uint32_t &W) {
  // gb groups per workgroup (about 512 workgroups when there are that many groups), W >= 4 ways of gb / 4 threads each
  gb = 16;
  while (gb < 1024 && (groups + gb - 1) / gb > 512) gb <<= 1;
  static const int gb_env = getenv("PLHIP_MERGE_GB") ? atoi(getenv("PLHIP_MERGE_GB")) : 0; // (experiments)
  if (gb_env >= 16 && gb_env <= 1024 && (gb_env & (gb_env - 1)) == 0) gb = (uint32_t)gb_env;
  W = DENSE_BLOCK / (gb / 4);
  if (W > 64) W = 64;
  while (W > 4 && (size_t)p.n_accs * W * gb * 8 + gb * 4 > 64 * 1024) W >>= 1; // the ways' partial results of every accumulator live in LDS
  while (gb > 16 && (size_t)p.n_accs * W * gb * 8 + gb * 4 > 64 * 1024) gb >>= 1;
}

void launch_dense_merge(const Plan &p, const DenseArgs &a, const OutPlan &o, uint64_t out_cap, uint32_t *host_sig, uint32_t seq, hipStream_t s) {
  uint32_t gb, W;
  dense_merge_shape(p, a.n_slots, gb, W);
  const uint32_t block = (gb / 4) * W; // >= gb
  const size_t lds = (size_t)p.n_accs * W * gb * 8 + (size_t)gb * 4;
  const uint32_t grid = std::max<uint32_t>(1u, (a.n_slots + gb - 1) / gb); // (a rank that owns no slot still reports its zero groups)
  hipLaunchKernelGGL(k_dense_merge, dim3(grid), dim3(block), lds, s, p, a, o, out_cap, host_sig, seq, gb, W);
}

void launch_dense_scatter(const Plan &p, const DenseArgs &a, const DenseXchg &x, hipStream_t s) {
  uint32_t gb, W;
  const uint32_t groups = x.per * x.world;
  dense_merge_shape(p, groups, gb, W);
  const uint32_t block = (gb / 4) * W;
  const size_t lds = (size_t)p.n_accs * W * gb * 8 + (size_t)gb * 4;
  hipLaunchKernelGGL(k_dense_scatter, dim3((groups + gb - 1) / gb), dim3(block), lds, s, p, a, x, gb, W);
}
