// lds_path.hip — the fast path: LDS-resident aggregation tables fed either by the input columns directly
// (few groups) or by a radix partition on hash bits (many groups).
//
//   k_part_hist     per tile, per partition row counts            (reads the key columns once)
//   k_part_scan     exclusive scan over tiles inside each partition
//   k_part_plan     partition starts + the work-item list of the aggregation pass
//   k_part_scatter  rows -> partition-contiguous copies of (packed key, value columns[, validity, row id])
//   k_lds_agg2      one work item (row range) per workgroup pass: LDS open-addressing table with
//                   ds_cmpst claim + ds_add/ds_min/ds_max folds; a whole partition writes its groups straight
//                   to the dense result area, a slice of a split partition (or of the unpartitioned input)
//                   merges into the hashed HBM table
//
// Everything here is HBM/LDS-bound integer and f64-atomic work; no MFMA.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "kernels.hpp"

#define A2_BLOCK 1024
#define A2_R 4
#define A2_PROBE_LIMIT 48
#define PART_BLOCK 1024
#define PART_MAX 4096

extern __shared__ __align__(16) uint8_t smem2[];

DEVINL uint32_t dtype_bytes(uint32_t dt) {
  switch (dt) {
    case PLHIP_I8: case PLHIP_U8: return 1;
    case PLHIP_I16: case PLHIP_U16: return 2;
    case PLHIP_I32: case PLHIP_U32: case PLHIP_F32: return 4;
    default: return 8;
  }
}

template <typename KS> struct KSEmpty;
template <> struct KSEmpty<uint32_t> { static constexpr uint32_t v = 0xFFFFFFFFu; };
template <> struct KSEmpty<uint64_t> { static constexpr uint64_t v = 0xFFFFFFFFFFFFFFFFull; };

DEVINL uint32_t cas_slot(uint32_t *p, uint32_t expect, uint32_t val) { return atomicCAS(p, expect, val); }
DEVINL uint64_t cas_slot(uint64_t *p, uint64_t expect, uint64_t val) {
  return atomicCAS((unsigned long long *)p, (unsigned long long)expect, (unsigned long long)val);
}

// LDS find-or-insert.  ~0u = no room along this probe sequence (the row is folded into the HBM table).
template <typename KS>
DEVINL uint32_t lds2_slot(KS *lkeys, uint32_t lbits, uint32_t *lused, uint64_t key, bool isnull) {
  const uint32_t lcap = 1u << lbits;
  if (isnull) { lused[1] = 1; return lcap + 1; }
  const KS k0 = (KS)key;
  if (k0 == KSEmpty<KS>::v) { lused[0] = 1; return lcap; }
  const uint32_t mask = lcap - 1;
  uint32_t s = lds_home(mix_a(key), lbits);
  for (int i = 0; i < A2_PROBE_LIMIT; i++) {
    KS k = lkeys[s];
    if (k == k0) return s;
    if (k == KSEmpty<KS>::v) {
      k = cas_slot(&lkeys[s], KSEmpty<KS>::v, k0);
      if (k == KSEmpty<KS>::v || k == k0) return s;
    }
    s = (s + 1) & mask;
  }
  return ~0u;
}

// Operands of A2_R rows for one accumulator: the dtype/conversion switches are wave-uniform and sit
// OUTSIDE the per-row loops, so the per-row cost is one load, one convert, one LDS atomic.
DEVINL void load_operands(const ValSpec &V, uint32_t conv, const int64_t (&rows)[A2_R], const bool (&ok)[A2_R],
                          uint64_t (&opnd)[A2_R]) {
  if (conv == CONV_INT) {
    switch (V.dtype) {
#define LD_INT(T, EXT)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < A2_R; r++) opnd[r] = ok[r] ? (uint64_t)(EXT)((const T *)V.values)[rows[r]] : 0; \
  break;
      case PLHIP_I8: LD_INT(int8_t, int64_t)
      case PLHIP_I16: LD_INT(int16_t, int64_t)
      case PLHIP_I32: LD_INT(int32_t, int64_t)
      case PLHIP_U8: LD_INT(uint8_t, uint64_t)
      case PLHIP_U16: LD_INT(uint16_t, uint64_t)
      case PLHIP_U32: LD_INT(uint32_t, uint64_t)
      default: LD_INT(uint64_t, uint64_t)
#undef LD_INT
    }
    return;
  }
  double d[A2_R];
  switch (V.dtype) {
#define LD_F(T)                                                                                        \
  _Pragma("unroll") for (int r = 0; r < A2_R; r++) d[r] = ok[r] ? (double)((const T *)V.values)[rows[r]] : 0.0; \
  break;
    case PLHIP_F32: LD_F(float)
    case PLHIP_F64: LD_F(double)
    case PLHIP_I64: LD_F(int64_t)
    case PLHIP_U64: LD_F(uint64_t)
    case PLHIP_I32: LD_F(int32_t)
    case PLHIP_U32: LD_F(uint32_t)
    case PLHIP_I16: LD_F(int16_t)
    case PLHIP_U16: LD_F(uint16_t)
    case PLHIP_I8: LD_F(int8_t)
    default: LD_F(uint8_t)
#undef LD_F
  }
#pragma unroll
  for (int r = 0; r < A2_R; r++) {
    if (conv == CONV_F64) opnd[r] = (uint64_t)__double_as_longlong(d[r]);
    else if (conv == CONV_FMIN) opnd[r] = (d[r] != d[r]) ? ~0ull : f64_ordered(d[r]);
    else opnd[r] = (d[r] != d[r]) ? 0ull : f64_ordered(d[r]);
  }
}

#define A2_NPRE 3 // value columns whose tile is prefetched into registers

template <typename KS>
struct TileRegs {
  KS key[A2_R];
  uint64_t raw[A2_NPRE][A2_R]; // raw bits, zero-extended from the column width
  uint32_t vok[A2_NPRE];       // validity, bit r
  uint32_t live, isnull;       // bit r
};

// Issues every load of one tile (keys + the first `npre` value columns); nothing here waits on them
// except the generic multi-column key path, whose packing needs the loaded bits.
template <typename KS>
DEVINL void load_tile(const Plan &P, const KeySpec &K0, bool fast4, bool fast8, uint64_t kxor, uint32_t npre,
                      int64_t tbase, int64_t row1, uint32_t tid, TileRegs<KS> &t) {
  t.live = 0; t.isnull = 0;
  int64_t rows[A2_R];
#pragma unroll
  for (int r = 0; r < A2_R; r++) {
    const int64_t row = tbase + (int64_t)r * A2_BLOCK + tid;
    if (row < row1) t.live |= 1u << r;
    rows[r] = row < row1 ? row : row1 - 1; // clamped: every load below is unconditional, so they issue back to back
  }
  if (fast4) {
#pragma unroll
    for (int r = 0; r < A2_R; r++) t.key[r] = (KS)(((const uint32_t *)K0.values)[rows[r]] ^ (uint32_t)kxor);
  } else if (fast8) {
#pragma unroll
    for (int r = 0; r < A2_R; r++) t.key[r] = (KS)(((const uint64_t *)K0.values)[rows[r]] ^ kxor);
  } else {
#pragma unroll
    for (int r = 0; r < A2_R; r++) {
      bool isn = false;
      t.key[r] = (KS)row_packed_key(P, rows[r], isn);
      if (isn) t.isnull |= 1u << r;
    }
  }
#pragma unroll
  for (int v = 0; v < A2_NPRE; v++) {
    t.vok[v] = t.live;
    if ((uint32_t)v >= npre) continue;
    const ValSpec V = P.vals[v];
    switch (dtype_bytes(V.dtype)) {
#define LD_RAW(T) _Pragma("unroll") for (int r = 0; r < A2_R; r++) t.raw[v][r] = (uint64_t)((const T *)V.values)[rows[r]]; break;
      case 1: LD_RAW(uint8_t)
      case 2: LD_RAW(uint16_t)
      case 4: LD_RAW(uint32_t)
      default: LD_RAW(uint64_t)
#undef LD_RAW
    }
    if (V.validity) {
      uint32_t m = 0;
#pragma unroll
      for (int r = 0; r < A2_R; r++) if (val_valid(V, rows[r])) m |= 1u << r;
      t.vok[v] = m & t.live;
    }
  }
}

// raw column bits -> accumulator operands (same conversions as load_operands)
DEVINL void raw_operands(uint32_t dtype, uint32_t conv, const uint64_t (&raw)[A2_R], uint64_t (&opnd)[A2_R]) {
  if (conv == CONV_INT) {
    switch (dtype) {
#define CV_INT(T) _Pragma("unroll") for (int r = 0; r < A2_R; r++) opnd[r] = (uint64_t)(int64_t)(T)raw[r]; break;
      case PLHIP_I8: CV_INT(int8_t)
      case PLHIP_I16: CV_INT(int16_t)
      case PLHIP_I32: CV_INT(int32_t)
      default:
#pragma unroll
        for (int r = 0; r < A2_R; r++) opnd[r] = raw[r]; // unsigned widths are already zero-extended
        break;
#undef CV_INT
    }
    return;
  }
  double d[A2_R];
  switch (dtype) {
#define CV_F(EXPR) _Pragma("unroll") for (int r = 0; r < A2_R; r++) d[r] = (EXPR); break;
    case PLHIP_F32: CV_F((double)__uint_as_float((uint32_t)raw[r]))
    case PLHIP_F64: CV_F(__longlong_as_double((long long)raw[r]))
    case PLHIP_I64: CV_F((double)(int64_t)raw[r])
    case PLHIP_I32: CV_F((double)(int32_t)raw[r])
    case PLHIP_I16: CV_F((double)(int16_t)raw[r])
    case PLHIP_I8: CV_F((double)(int8_t)raw[r])
    default: CV_F((double)raw[r])
#undef CV_F
  }
#pragma unroll
  for (int r = 0; r < A2_R; r++) {
    if (conv == CONV_F64) opnd[r] = (uint64_t)__double_as_longlong(d[r]);
    else if (conv == CONV_FMIN) opnd[r] = (d[r] != d[r]) ? ~0ull : f64_ordered(d[r]);
    else opnd[r] = (d[r] != d[r]) ? 0ull : f64_ordered(d[r]);
  }
}

template <typename KS>
__global__ void __launch_bounds__(A2_BLOCK) // 8 waves per SIMD = two 1024-thread workgroups per CU
k_lds_agg2(Plan P, TableRef T, const WorkItem *__restrict__ items, const uint32_t *__restrict__ n_items_p,
           uint32_t *next_item, uint32_t lbits, uint64_t dense_cap, uint64_t *slots) {
  __shared__ uint32_t lused[2];
  __shared__ uint32_t loff[PLHIP_MAX_ACCS];
  __shared__ uint32_t sh_item, sh_count, sh_cursor, sh_base;
  const uint32_t lcap = 1u << lbits, lslots = lcap + 2;
  const uint32_t tid = threadIdx.x;
  KS *lkeys = (KS *)smem2;
  if (tid == 0) {
    uint32_t off = (lslots * (uint32_t)sizeof(KS) + 7u) & ~7u;
    for (int a = 0; a < P.n_accs; a++) { loff[a] = off; off += (lslots * P.accs[a].width + 7u) & ~7u; }
  }
  uint64_t *gkeys = (uint64_t *)T.base;
  const uint32_t n_items = *n_items_p;
  // fast key read: one plain column of 4 or 8 bytes without validity (always true for partitioned input)
  const KeySpec K0 = P.keys[0];
  const bool fast4 = (P.prepacked && K0.dtype == PLHIP_U32) ||
                     (!P.prepacked && P.single_key && !K0.validity && (K0.dtype == PLHIP_I32 || K0.dtype == PLHIP_U32 || K0.dtype == PLHIP_F32));
  const bool fast8 = (P.prepacked && K0.dtype != PLHIP_U32) ||
                     (!P.prepacked && P.single_key && !K0.validity && (K0.dtype == PLHIP_I64 || K0.dtype == PLHIP_U64 || K0.dtype == PLHIP_F64));
  const uint64_t kxor = P.prepacked ? 0ull : (K0.dtype == PLHIP_I32 ? 0x80000000ull : (K0.dtype == PLHIP_I64 ? SIGN64 : 0ull));
  const uint32_t npre = P.n_vals < A2_NPRE ? (uint32_t)P.n_vals : (uint32_t)A2_NPRE;

  for (;;) {
    __syncthreads(); // previous item fully emitted; loff visible on the first trip
    if (tid == 0) { sh_item = atomicAdd(next_item, 1u); sh_count = 0; sh_cursor = 0; lused[0] = lused[1] = 0; }
    __syncthreads();
    const uint32_t it = sh_item;
    if (it >= n_items) break; // every wave of every workgroup reaches this exit
    const WorkItem W = items[it];

    for (uint32_t i = tid; i < lslots; i += A2_BLOCK) {
      lkeys[i] = KSEmpty<KS>::v;
      for (int a = 0; a < P.n_accs; a++) {
        if (P.accs[a].width == 8) ((uint64_t *)(smem2 + loff[a]))[i] = P.accs[a].init;
        else ((uint32_t *)(smem2 + loff[a]))[i] = (uint32_t)P.accs[a].init;
      }
    }
    __syncthreads();

    uint32_t spilled = 0;
    // Software pipeline: the columns of tile t+1 are in flight (registers) while tile t is probed and
    // folded, so a wave exposes one HBM round trip per item instead of one per column per tile.
    TileRegs<KS> cur, nxt;
    if (W.row0 < W.row1) load_tile<KS>(P, K0, fast4, fast8, kxor, npre, W.row0, W.row1, tid, cur);
    for (int64_t tbase = W.row0; tbase < W.row1; tbase += (int64_t)A2_BLOCK * A2_R) {
      const int64_t tnext = tbase + (int64_t)A2_BLOCK * A2_R;
      if (tnext < W.row1) load_tile<KS>(P, K0, fast4, fast8, kxor, npre, tnext, W.row1, tid, nxt);
      int64_t rows[A2_R];
      bool live[A2_R];
      uint32_t slot[A2_R];
#pragma unroll
      for (int r = 0; r < A2_R; r++) { rows[r] = tbase + (int64_t)r * A2_BLOCK + tid; live[r] = (cur.live >> r) & 1; }
#pragma unroll
      for (int r = 0; r < A2_R; r++) {
        slot[r] = 0;
        if (!live[r]) continue;
        const bool isnull = (cur.isnull >> r) & 1;
        slot[r] = lds2_slot<KS>(lkeys, lbits, lused, (uint64_t)cur.key[r], isnull);
        if (slot[r] == ~0u) { // no room in LDS: fold this row into the hashed HBM table
          live[r] = false;
          spilled++;
          const uint64_t gs = gtable_slot(gkeys, T.cap, T.flags, (uint64_t)cur.key[r], isnull);
          if (gs != ~0ull)
            for (int a = 0; a < P.n_accs; a++) {
              const AccSpec &A = P.accs[a];
              uint64_t operand;
              if (acc_operand(P, A, rows[r], operand)) acc_apply(T.base, A, gs, operand);
            }
        }
      }
      for (int a = 0; a < P.n_accs; a++) {
        const AccSpec A = P.accs[a];
        uint8_t *lb = smem2 + loff[a];
        if (A.op == ACC_CNT_ALL) {
#pragma unroll
          for (int r = 0; r < A2_R; r++) if (live[r]) atomicAdd((unsigned int *)(lb + slot[r] * 4), 1u);
          continue;
        }
        if (A.op == ACC_FIRST_ROW) {
#pragma unroll
          for (int r = 0; r < A2_R; r++)
            if (live[r]) atomicMin((unsigned int *)(lb + slot[r] * 4), P.rowidx ? P.rowidx[rows[r]] : (uint32_t)rows[r]);
          continue;
        }
        const ValSpec V = P.vals[A.src];
        bool ok[A2_R];
        uint64_t opnd[A2_R];
        if (A.src < (int)npre) { // operands come from the prefetched registers
          uint32_t vok = 0;
          uint64_t raw[A2_R];
          // wave-uniform select of the prefetched column (register arrays cannot be indexed dynamically)
#pragma unroll
          for (int v = 0; v < A2_NPRE; v++)
            if (A.src == v) {
              vok = cur.vok[v];
#pragma unroll
              for (int r = 0; r < A2_R; r++) raw[r] = cur.raw[v][r];
            }
#pragma unroll
          for (int r = 0; r < A2_R; r++) ok[r] = live[r] && ((vok >> r) & 1);
          if (A.op == ACC_CNT_VALID) {
#pragma unroll
            for (int r = 0; r < A2_R; r++) if (ok[r]) atomicAdd((unsigned int *)(lb + slot[r] * 4), 1u);
            continue;
          }
          raw_operands(V.dtype, A.conv, raw, opnd);
        } else {
          if (V.validity) {
#pragma unroll
            for (int r = 0; r < A2_R; r++) ok[r] = live[r] && val_valid(V, rows[r]);
          } else {
#pragma unroll
            for (int r = 0; r < A2_R; r++) ok[r] = live[r];
          }
          if (A.op == ACC_CNT_VALID) {
#pragma unroll
            for (int r = 0; r < A2_R; r++) if (ok[r]) atomicAdd((unsigned int *)(lb + slot[r] * 4), 1u);
            continue;
          }
          load_operands(V, A.conv, rows, ok, opnd);
        }
        switch (A.op) {
#define APPLY(EXPR) _Pragma("unroll") for (int r = 0; r < A2_R; r++) if (ok[r]) { auto *p = lb + slot[r] * 8; EXPR; } break;
          case ACC_ADD_U64: APPLY(atomicAdd((unsigned long long *)p, (unsigned long long)opnd[r]))
          case ACC_ADD_F64: APPLY(atomicAdd((double *)p, __longlong_as_double((long long)opnd[r])))
          case ACC_MIN_I64: APPLY(atomicMin((long long *)p, (long long)opnd[r]))
          case ACC_MAX_I64: APPLY(atomicMax((long long *)p, (long long)opnd[r]))
          case ACC_MIN_U64: APPLY(atomicMin((unsigned long long *)p, (unsigned long long)opnd[r]))
          case ACC_MAX_U64: APPLY(atomicMax((unsigned long long *)p, (unsigned long long)opnd[r]))
#undef APPLY
          default: break;
        }
      }
      cur = nxt;
    }
    if (spilled) atomicAdd(&T.flags[FLAG_LDS_SPILL], spilled);
    __syncthreads();

    if (W.direct) {
      // a whole partition: nobody else holds these keys -> dense result area, plain stores
      uint32_t mine = 0;
      for (uint32_t i = tid; i < lcap + 1; i += A2_BLOCK)
        mine += (i < lcap) ? (lkeys[i] != KSEmpty<KS>::v) : (lused[0] != 0);
      if (mine) atomicAdd(&sh_count, mine);
      __syncthreads();
      if (tid == 0) sh_base = sh_count ? atomicAdd(&T.flags[FLAG_NGROUPS], sh_count) : 0;
      __syncthreads();
      const uint64_t gbase = sh_base;
      const int lane = tid & 63;
      for (uint32_t i0 = 0; i0 < lcap + 1; i0 += A2_BLOCK) {
        const uint32_t i = i0 + tid;
        bool occ = false;
        if (i < lcap) occ = lkeys[i] != KSEmpty<KS>::v;
        else if (i == lcap) occ = lused[0] != 0;
        const unsigned long long m = __ballot(occ);
        uint32_t wbase = 0;
        if (m) {
          if (lane == 0) wbase = atomicAdd(&sh_cursor, (uint32_t)__popcll(m));
          wbase = __shfl(wbase, 0);
        }
        if (occ) {
          const uint64_t d = gbase + wbase + __popcll(m & ((1ull << lane) - 1));
          if (d >= dense_cap) T.flags[FLAG_OVERFLOW] = 1;
          else {
            const uint64_t gs = T.cap + 2 + d;
            gkeys[gs] = (i == lcap) ? (uint64_t)KSEmpty<KS>::v : (uint64_t)lkeys[i];
            for (int a = 0; a < P.n_accs; a++) {
              const AccSpec &A = P.accs[a];
              if (A.width == 8) ((uint64_t *)(T.base + A.offset))[gs] = ((const uint64_t *)(smem2 + loff[a]))[i];
              else ((uint32_t *)(T.base + A.offset))[gs] = ((const uint32_t *)(smem2 + loff[a]))[i];
            }
            slots[d] = gs;
          }
        }
      }
    } else {
      // merge: one HBM find-or-insert per occupied LDS slot, one global atomic per touched accumulator
      for (uint32_t i = tid; i < lslots; i += A2_BLOCK) {
        bool occupied, isnull = false;
        uint64_t key = 0;
        if (i < lcap) { const KS k = lkeys[i]; occupied = k != KSEmpty<KS>::v; key = (uint64_t)k; }
        else if (i == lcap) { occupied = lused[0] != 0; key = (uint64_t)KSEmpty<KS>::v; }
        else { occupied = lused[1] != 0; isnull = true; }
        if (!occupied) continue;
        const uint64_t gs = gtable_slot(gkeys, T.cap, T.flags, key, isnull);
        if (gs == ~0ull) continue;
        for (int a = 0; a < P.n_accs; a++) {
          const AccSpec &A = P.accs[a];
          const uint64_t part = (A.width == 8) ? ((const uint64_t *)(smem2 + loff[a]))[i] : (uint64_t)((const uint32_t *)(smem2 + loff[a]))[i];
          if (part == A.init) continue; // identity: nothing to fold
          acc_apply(T.base, A, gs, part);
        }
      }
    }
  }
}

size_t lds2_table_bytes(const Plan &p, uint32_t lbits, int key_bytes) {
  const size_t lslots = (1u << lbits) + 2;
  size_t off = (lslots * key_bytes + 7) & ~(size_t)7;
  for (int a = 0; a < p.n_accs; a++) off += (lslots * p.accs[a].width + 7) & ~(size_t)7;
  return off;
}

static int g_cus2 = 0;
static int num_cus2() {
  if (!g_cus2) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cus2 = prop.multiProcessorCount;
    if (g_cus2 <= 0) g_cus2 = 256;
  }
  return g_cus2;
}
int plhip_num_cus() { return num_cus2(); }

void launch_lds_agg2(const Plan &p, TableRef t, const WorkItem *items, const uint32_t *n_items_dev, uint32_t max_items,
                     uint32_t *next_item_dev, uint32_t lbits, int key_bytes, uint64_t dense_cap, uint64_t *slots,
                     hipStream_t s) {
  const size_t bytes = lds2_table_bytes(p, lbits, key_bytes);
  const int wg_per_cu = bytes <= 78 * 1024 ? 2 : 1;
  uint32_t grid = (uint32_t)(num_cus2() * wg_per_cu);
  if (grid > max_items) grid = max_items ? max_items : 1;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_lds_agg2<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    (void)hipFuncSetAttribute((const void *)k_lds_agg2<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    attr_set = true;
  }
  if (key_bytes == 4)
    hipLaunchKernelGGL(k_lds_agg2<uint32_t>, dim3(grid), dim3(A2_BLOCK), bytes, s, p, t, items, n_items_dev, next_item_dev, lbits, dense_cap, slots);
  else
    hipLaunchKernelGGL(k_lds_agg2<uint64_t>, dim3(grid), dim3(A2_BLOCK), bytes, s, p, t, items, n_items_dev, next_item_dev, lbits, dense_cap, slots);
}

// ---------------------------------------------------------------------------------------------------------
// radix partition on hash bits
// ---------------------------------------------------------------------------------------------------------
DEVINL uint64_t part_row_key(const Plan &P, int64_t row) {
  bool isnull;
  return row_packed_key(P, row, isnull); // partitioned plans carry nulls inside the packed code
}

__global__ void __launch_bounds__(PART_BLOCK) k_part_hist(Plan P, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles,
                                                          uint32_t *__restrict__ counts) {
  __shared__ uint32_t h[PART_MAX];
  const uint32_t np = 1u << pbits;
  for (uint32_t p = threadIdx.x; p < np; p += PART_BLOCK) h[p] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t r1 = r0 + tile_rows < P.n_rows ? r0 + tile_rows : P.n_rows;
  for (int64_t row = r0 + threadIdx.x; row < r1; row += PART_BLOCK)
    atomicAdd(&h[part_of(mix_a(part_row_key(P, row)), pbits)], 1u);
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < np; p += PART_BLOCK) counts[(uint64_t)p * n_tiles + blockIdx.x] = h[p];
}

// The same histogram for tuples of 2 .. 4 eight-byte key columns without nulls (config 3: three i64 ids; launch_part_hist checks the
// plan): row_packed_key's loop over a run-time number of columns of run-time type makes every load wait for the one before (3.8 TB/s
// on config 3's 2.4 GB of keys).  Here the same arithmetic -- (norm - minv + nullable) << shift, OR-ed -- runs over loads that are all
// requested first, PH_ROWS rows at a time.  Only workgroups whose row count is a whole multiple of PH_ROWS * PART_BLOCK take that
// loop (every row then lies below r1 by construction); the last, ragged tile keeps the row-at-a-time loop.  (A kernel of its own:
// with both loops in k_part_hist the single-key histogram of config 5 went from 2.1 to 3.2 ms.)
#define PH_ROWS 4
__global__ void __launch_bounds__(PART_BLOCK) k_part_hist_wide(Plan P, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles,
                                                               uint32_t *__restrict__ counts) {
  __shared__ uint32_t h[PART_MAX];
  const uint32_t np = 1u << pbits;
  for (uint32_t p = threadIdx.x; p < np; p += PART_BLOCK) h[p] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t r1 = r0 + tile_rows < P.n_rows ? r0 + tile_rows : P.n_rows;
  if (((r1 - r0) % (PH_ROWS * PART_BLOCK)) == 0) {
    const int nk = P.n_keys;
    uint64_t flip[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < nk) flip[i] = P.keys[i].dtype == PLHIP_I64 ? SIGN64 : 0ull; // load_key_norm of an 8-byte column
    for (int64_t row = r0 + threadIdx.x; row < r1; row += PH_ROWS * PART_BLOCK) {
      uint64_t raw[PH_ROWS][4];
#pragma unroll
      for (int u = 0; u < PH_ROWS; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) raw[u][i] = i < nk ? ((const uint64_t *)P.keys[i].values)[row + (int64_t)u * PART_BLOCK] : 0ull;
#pragma unroll
      for (int u = 0; u < PH_ROWS; u++) {
        uint64_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (i < nk) packed |= ((raw[u][i] ^ flip[i]) - P.keys[i].minv + P.keys[i].nullable) << P.keys[i].shift;
        atomicAdd(&h[part_of(mix_a(packed), pbits)], 1u);
      }
    }
  } else {
    for (int64_t row = r0 + threadIdx.x; row < r1; row += PART_BLOCK)
      atomicAdd(&h[part_of(mix_a(part_row_key(P, row)), pbits)], 1u);
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < np; p += PART_BLOCK) counts[(uint64_t)p * n_tiles + blockIdx.x] = h[p];
}

// one workgroup per partition: counts[p][*] -> exclusive prefix over tiles (in place), totals[p].  The row is walked in
// blocks of 1024 entries -- thread t takes the four consecutive entries 4t .. 4t + 3, so a wave reads and writes 1 KB
// contiguous -- with the running total carried from block to block (the first version gave every thread one long contiguous
// chunk: lanes strided by the chunk, 0.84 ms for the 512 x 122 071 counts of a 1e9-row input, 0.6 TB/s).
__global__ void __launch_bounds__(256) k_part_scan(uint32_t *__restrict__ counts, uint32_t n_tiles, uint32_t *__restrict__ totals) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry;
  uint32_t *c = counts + (uint64_t)blockIdx.x * n_tiles;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_tiles; base += 1024) {
    const uint32_t i0 = base + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = i0 + k < n_tiles ? c[i0 + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine; // inclusive scan over the wave
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= (uint32_t)off) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < n_tiles) c[i0 + k] = run; run += v[k]; }
    __syncthreads();
    if (threadIdx.x == 255) carry = run; // (the last thread's running total is the block's)
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// single workgroup: partition starts (exclusive scan of totals) + the aggregation pass's work items
// `align` (a power of two, 1 = packed): every partition starts at a multiple of it (the two-level pipeline runs
// whole 8192-row tiles inside each partition); items may be nullptr then.
__global__ void __launch_bounds__(1024) k_part_plan(const uint32_t *__restrict__ totals, uint32_t np, uint32_t *__restrict__ part_start,
                                                    WorkItem *__restrict__ items, uint32_t *__restrict__ n_items, uint32_t *next_item,
                                                    uint32_t split_rows, uint32_t align) {
  __shared__ uint32_t sums[1024];
  const uint32_t chunk = (np + 1023) / 1024;
  const uint32_t b = threadIdx.x * chunk, e = b + chunk < np ? b + chunk : np;
  uint32_t s = 0;
  for (uint32_t i = b; i < e; i++) s += (totals[i] + align - 1) & ~(align - 1);
  sums[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = threadIdx.x >= (uint32_t)off ? sums[threadIdx.x - off] : 0;
    __syncthreads();
    sums[threadIdx.x] += v;
    __syncthreads();
  }
  if (threadIdx.x == 0) { *n_items = 0; *next_item = 0; }
  __syncthreads();
  uint32_t run = threadIdx.x ? sums[threadIdx.x - 1] : 0;
  for (uint32_t p = b; p < e; p++) {
    const uint32_t n = totals[p];
    part_start[p] = run;
    if (n && items) {
      if (n <= split_rows) {
        const uint32_t k = atomicAdd(n_items, 1u);
        items[k] = WorkItem{(int64_t)run, (int64_t)run + n, 1u, 0u};
      } else { // a skewed partition: slices that merge into the hashed table
        const uint32_t pieces = (n + split_rows - 1) / split_rows;
        const uint32_t k = atomicAdd(n_items, pieces);
        for (uint32_t j = 0; j < pieces; j++) {
          const int64_t a = (int64_t)run + (int64_t)j * split_rows;
          const int64_t z = (int64_t)run + (j + 1 == pieces ? (int64_t)n : (int64_t)(j + 1) * split_rows);
          items[k + j] = WorkItem{a, z, 0u, 0u};
        }
      }
    }
    run += (n + align - 1) & ~(align - 1);
  }
  if (threadIdx.x == 1023) part_start[np] = sums[1023];
}


// Exclusive scan of v[0..n) in LDS by one workgroup of PART_BLOCK threads (n <= PART_MAX); `tmp` holds one
// word per wave.  Returns with a barrier passed; v[] holds the exclusive prefix.
DEVINL void block_exclusive_scan(uint32_t *v, uint32_t n, uint32_t *tmp) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t per = (n + PART_BLOCK - 1) / PART_BLOCK; // consecutive entries per thread (<= 4)
  const uint32_t b = tid * per;
  uint32_t loc[PART_MAX / PART_BLOCK];
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t i = 0; i < PART_MAX / PART_BLOCK; i++) {
    loc[i] = (i < per && b + i < n) ? v[b + i] : 0;
    sum += loc[i];
  }
  uint32_t inc = sum; // inclusive scan across the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= (uint32_t)off) inc += t;
  }
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < PART_BLOCK / 64 ? tmp[lane] : 0;
    uint32_t winc = w;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t t = __shfl_up(winc, off);
      if (lane >= (uint32_t)off) winc += t;
    }
    if (lane < PART_BLOCK / 64) tmp[lane] = winc - w; // exclusive wave offsets
  }
  __syncthreads();
  uint32_t run = tmp[wave] + inc - sum;
#pragma unroll
  for (uint32_t i = 0; i < PART_MAX / PART_BLOCK; i++)
    if (i < per && b + i < n) { v[b + i] = run; run += loc[i]; }
  __syncthreads();
}

// Scatter with an in-LDS counting sort of the tile: rows are ranked inside their partition (ds_add_rtn),
// staged at their sorted position, then written out run by run, so every (tile, partition) run is one
// contiguous, coalesced store stream instead of 4-byte stores to 2^pbits different lines.
template <int R>
__global__ void __launch_bounds__(PART_BLOCK) k_part_scatter(Plan P, PartOut O, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles,
                                                             const uint32_t *__restrict__ offs, const uint32_t *__restrict__ part_start) {
  __shared__ uint32_t tmp[PART_BLOCK / 64];
  const uint32_t np = 1u << pbits, tid = threadIdx.x;
  uint32_t *hist = (uint32_t *)smem2;              // [np] counts, then exclusive local starts
  uint32_t *delta = hist + np;                      // [np] global run start - local run start
  uint64_t *stage = (uint64_t *)(delta + np);       // [R * PART_BLOCK] one column of the sorted tile
  uint16_t *spart = (uint16_t *)(stage + R * PART_BLOCK); // [R * PART_BLOCK] partition of each sorted position
  for (uint32_t p = tid; p < np; p += PART_BLOCK) hist[p] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t r1 = r0 + tile_rows < P.n_rows ? r0 + tile_rows : P.n_rows;
  const uint32_t nt = (uint32_t)(r1 - r0);
  uint64_t key[R];
  uint32_t pr[R]; // partition | rank << 12 ... later: sorted position
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int64_t row = r0 + (int64_t)r * PART_BLOCK + tid;
    live[r] = row < r1;
    key[r] = live[r] ? part_row_key(P, row) : 0;
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    pr[r] = 0;
    if (live[r]) {
      const uint32_t p = part_of(mix_a(key[r]), pbits);
      pr[r] = p | (atomicAdd(&hist[p], 1u) << 12);
    }
  }
  __syncthreads();
  for (uint32_t p = tid; p < np; p += PART_BLOCK) delta[p] = part_start[p] + offs[(uint64_t)p * n_tiles + blockIdx.x];
  block_exclusive_scan(hist, np, tmp);
  for (uint32_t p = tid; p < np; p += PART_BLOCK) delta[p] -= hist[p];
#pragma unroll
  for (int r = 0; r < R; r++)
    if (live[r]) {
      const uint32_t p = pr[r] & 0xFFFu;
      const uint32_t pos = hist[p] + (pr[r] >> 12);
      pr[r] = pos;
      spart[pos] = (uint16_t)p;
      if (O.key_bytes == 4) ((uint32_t *)stage)[pos] = (uint32_t)key[r];
      else stage[pos] = key[r];
    }
  __syncthreads();
  uint32_t dest[R];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const uint32_t j = i * PART_BLOCK + tid;
    dest[i] = j < nt ? j + delta[spart[j]] : 0;
    if (j < nt) {
      if (O.key_bytes == 4) ((uint32_t *)O.keys)[dest[i]] = ((const uint32_t *)stage)[j];
      else ((uint64_t *)O.keys)[dest[i]] = stage[j];
    }
  }
  if (O.rowidx) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) if (live[r]) ((uint32_t *)stage)[pr[r]] = (uint32_t)(r0 + (int64_t)r * PART_BLOCK + tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; i++) { const uint32_t j = i * PART_BLOCK + tid; if (j < nt) O.rowidx[dest[i]] = ((const uint32_t *)stage)[j]; }
  }
  for (int v = 0; v < P.n_vals; v++) {
    const ValSpec V = P.vals[v];
    const uint32_t w = dtype_bytes(V.dtype);
    __syncthreads(); // the previous column has left the staging buffer
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (!live[r]) continue;
      const int64_t row = r0 + (int64_t)r * PART_BLOCK + tid;
      switch (w) {
        case 1: ((uint8_t *)stage)[pr[r]] = V.values[row]; break;
        case 2: ((uint16_t *)stage)[pr[r]] = ((const uint16_t *)V.values)[row]; break;
        case 4: ((uint32_t *)stage)[pr[r]] = ((const uint32_t *)V.values)[row]; break;
        default: stage[pr[r]] = ((const uint64_t *)V.values)[row]; break;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; i++) {
      const uint32_t j = i * PART_BLOCK + tid;
      if (j >= nt) continue;
      switch (w) {
        case 1: O.vals[v][dest[i]] = ((const uint8_t *)stage)[j]; break;
        case 2: ((uint16_t *)O.vals[v])[dest[i]] = ((const uint16_t *)stage)[j]; break;
        case 4: ((uint32_t *)O.vals[v])[dest[i]] = ((const uint32_t *)stage)[j]; break;
        default: ((uint64_t *)O.vals[v])[dest[i]] = stage[j]; break;
      }
    }
    if (O.valid[v]) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; r++)
        if (live[r]) ((uint8_t *)stage)[pr[r]] = val_valid(V, r0 + (int64_t)r * PART_BLOCK + tid) ? 1 : 0;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < R; i++) { const uint32_t j = i * PART_BLOCK + tid; if (j < nt) O.valid[v][dest[i]] = ((const uint8_t *)stage)[j]; }
    }
  }
}

void launch_part_hist(const Plan &p, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles, uint32_t *counts, hipStream_t s) {
  bool wide = !p.prepacked && !p.single_key && p.n_keys >= 2 && p.n_keys <= 4; // tuples of 8-byte columns without nulls
  for (int i = 0; wide && i < p.n_keys; i++)
    wide = !p.keys[i].validity && (p.keys[i].dtype == PLHIP_I64 || p.keys[i].dtype == PLHIP_U64 || p.keys[i].dtype == PLHIP_F64);
  if (wide) hipLaunchKernelGGL(k_part_hist_wide, dim3(n_tiles), dim3(PART_BLOCK), 0, s, p, pbits, tile_rows, n_tiles, counts);
  else hipLaunchKernelGGL(k_part_hist, dim3(n_tiles), dim3(PART_BLOCK), 0, s, p, pbits, tile_rows, n_tiles, counts);
}
void launch_part_scan(uint32_t *counts, uint32_t np, uint32_t n_tiles, uint32_t *totals, hipStream_t s) {
  hipLaunchKernelGGL(k_part_scan, dim3(np), dim3(256), 0, s, counts, n_tiles, totals);
}
void launch_part_plan(const uint32_t *totals, uint32_t np, uint32_t *part_start, WorkItem *items, uint32_t *n_items,
                      uint32_t *next_item, uint32_t split_rows, hipStream_t s, uint32_t align) {
  hipLaunchKernelGGL(k_part_plan, dim3(1), dim3(1024), 0, s, totals, np, part_start, items, n_items, next_item, split_rows, align);
}
// tile_rows must be PART_SCATTER_R * PART_BLOCK or less (the tile is sorted in LDS)
void launch_part_scatter(const Plan &p, const PartOut &o, uint32_t pbits, int64_t tile_rows, uint32_t n_tiles,
                         const uint32_t *offs, const uint32_t *part_start, hipStream_t s) {
  const uint32_t np = 1u << pbits;
  const size_t bytes = (size_t)np * 8 + (size_t)PART_SCATTER_R * PART_BLOCK * 10;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_part_scatter<PART_SCATTER_R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_part_scatter<PART_SCATTER_R>, dim3(n_tiles), dim3(PART_BLOCK), bytes, s, p, o, pbits, tile_rows, n_tiles, offs, part_start);
}
