// fast_path.hip — the LDS-only aggregation pipeline for gfx950 (see fast.hpp for the kernel list).
//
// Measured on MI355X (scripts/ubench): streaming reads 6-7 TB/s; global atomics ~25 Gop/s on random
// addresses and <2 Gop/s on ~100 hot ones; LDS atomics 3 (f64 add) to 10 (u32 add) lanes/clk/CU on random
// slots.  So rows are never folded with global atomics here: every accumulator update is an LDS atomic,
// workgroups hand their tables on as compact dumps (plain coalesced stores), and a merge kernel folds the
// dumps in LDS again and writes the output columns.  HBM/LDS-bound integer work: no MFMA.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "fast.hpp"

extern __shared__ __align__(16) uint8_t fsm[];

namespace {

template <typename KS> struct KE;
template <> struct KE<uint32_t> { static constexpr uint32_t v = 0xFFFFFFFFu; };
template <> struct KE<uint64_t> { static constexpr uint64_t v = 0xFFFFFFFFFFFFFFFFull; };

DEVINL uint32_t al8(uint32_t x) { return (x + 7u) & ~7u; }
DEVINL uint64_t al16(uint64_t x) { return (x + 15ull) & ~15ull; }

DEVINL uint32_t wbytes(uint32_t dt) {
  switch (dt) {
    case PLHIP_I8: case PLHIP_U8: return 1;
    case PLHIP_I16: case PLHIP_U16: return 2;
    case PLHIP_I32: case PLHIP_U32: case PLHIP_F32: return 4;
    default: return 8;
  }
}

DEVINL uint32_t cas_key(uint32_t *p, uint32_t expect, uint32_t val) { return atomicCAS(p, expect, val); }
DEVINL uint64_t cas_key(uint64_t *p, uint64_t expect, uint64_t val) {
  return atomicCAS((unsigned long long *)p, (unsigned long long)expect, (unsigned long long)val);
}

// used[0]: sentinel slot taken, used[1]: null slot taken, used[2]: table full
#define FP_PROBE_LIMIT 96
template <typename KS>
DEVINL uint32_t ltab_slot(KS *lkeys, uint32_t lbits, uint32_t *used, KS k0) {
  const uint32_t lcap = 1u << lbits;
  if (k0 == KE<KS>::v) { used[0] = 1; return lcap; }
  const uint32_t mask = lcap - 1;
  uint32_t s = lds_home(mix_a((uint64_t)k0), lbits);
  for (int i = 0; i < FP_PROBE_LIMIT; i++) {
    KS k = lkeys[s];
    if (k == k0) return s;
    if (k == KE<KS>::v) {
      k = cas_key(&lkeys[s], KE<KS>::v, k0);
      if (k == KE<KS>::v || k == k0) return s;
    }
    s = (s + 1) & mask;
  }
  used[2] = 1;
  return ~0u;
}

template <typename KS>
DEVINL void ltab_init(const Plan &P, uint8_t *sm, uint32_t lslots) {
  KS *lkeys = (KS *)sm;
  for (uint32_t i = threadIdx.x; i < lslots; i += blockDim.x) lkeys[i] = KE<KS>::v;
  uint32_t off = al8(lslots * (uint32_t)sizeof(KS));
  for (int a = 0; a < P.n_accs; a++) {
    const uint64_t init = P.accs[a].init;
    if (P.accs[a].width == 8) {
      for (uint32_t i = threadIdx.x; i < lslots; i += blockDim.x) ((uint64_t *)(sm + off))[i] = init;
    } else {
      for (uint32_t i = threadIdx.x; i < lslots; i += blockDim.x) ((uint32_t *)(sm + off))[i] = (uint32_t)init;
    }
    off += al8(lslots * P.accs[a].width);
  }
}

// raw column bits of B rows, zero-extended.  QUAD: the B == 4 rows are consecutive, row[0] % 4 == 0 and the
// column base is aligned to 4 elements, so one (two for 8-byte types) vector load fetches them.
template <int B, bool QUAD>
DEVINL void load_raw(const uint8_t *vals, uint32_t w, const uint32_t (&row)[B], uint64_t (&raw)[B]) {
  if (QUAD) {
    const size_t r0 = row[0];
    switch (w) {
      case 1: {
        const uint32_t q = *(const uint32_t *)(vals + r0);
#pragma unroll
        for (int b = 0; b < B; b++) raw[b] = (q >> (8 * b)) & 0xFFu;
        break;
      }
      case 2: {
        const uint2 q = *(const uint2 *)(vals + r0 * 2);
        raw[0] = q.x & 0xFFFFu; raw[1] = q.x >> 16; raw[2 % B] = q.y & 0xFFFFu; raw[3 % B] = q.y >> 16;
        break;
      }
      case 4: {
        const uint4 q = *(const uint4 *)(vals + r0 * 4);
        raw[0] = q.x; raw[1 % B] = q.y; raw[2 % B] = q.z; raw[3 % B] = q.w;
        break;
      }
      default: {
        const uint4 q0 = *(const uint4 *)(vals + r0 * 8);
        const uint4 q1 = *(const uint4 *)(vals + r0 * 8 + 16);
        raw[0] = (uint64_t)q0.x | ((uint64_t)q0.y << 32); raw[1 % B] = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
        raw[2 % B] = (uint64_t)q1.x | ((uint64_t)q1.y << 32); raw[3 % B] = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
        break;
      }
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

// Folds B rows per lane into the workgroup's LDS table.  row[] are in-bounds for every lane (dead lanes are
// clamped by the caller); bit b of `live` says whether row[b] counts.  The dtype / op switches are
// wave-uniform and sit outside the per-row loops.
template <typename KS, int B, bool QUAD>
DEVINL void fold_batch(const Plan &P, uint8_t *sm, uint32_t lbits, uint32_t *used, const uint32_t (&row)[B], uint32_t live,
                       int keymode, uint64_t kxor, bool force_null) {
  const uint32_t lcap = 1u << lbits, lslots = lcap + 2;
  KS *lkeys = (KS *)sm;
  const KeySpec &K0 = P.keys[0];
  uint32_t slot[B];
  if (force_null) {
    if (live) used[1] = 1;
#pragma unroll
    for (int b = 0; b < B; b++) slot[b] = lcap + 1;
  } else {
    KS key[B];
    uint32_t nullm = 0;
    if (keymode == 0) {
#pragma unroll
      for (int b = 0; b < B; b++) {
        bool isn = false;
        key[b] = (KS)row_packed_key(P, (int64_t)row[b], isn);
        if (isn) nullm |= 1u << b;
      }
    } else {
      uint64_t raw[B];
      load_raw<B, QUAD>(K0.values, keymode == 1 ? 4u : 8u, row, raw);
#pragma unroll
      for (int b = 0; b < B; b++) key[b] = (KS)(raw[b] ^ kxor);
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
      slot[b] = 0;
      if (!((live >> b) & 1)) continue;
      if ((nullm >> b) & 1) { used[1] = 1; slot[b] = lcap + 1; continue; }
      slot[b] = ltab_slot<KS>(lkeys, lbits, used, key[b]);
      if (slot[b] == ~0u) { live &= ~(1u << b); slot[b] = 0; }
    }
  }
  uint32_t off = al8(lslots * (uint32_t)sizeof(KS));
  int cur_src = -1;
  uint64_t raw[B];
  uint32_t ok = 0;
  for (int a = 0; a < P.n_accs; a++) {
    const AccSpec A = P.accs[a];
    uint8_t *lb = sm + off;
    off += al8(lslots * A.width);
    if (A.op == ACC_CNT_ALL) {
#pragma unroll
      for (int b = 0; b < B; b++) if ((live >> b) & 1) atomicAdd((unsigned int *)(lb + slot[b] * 4), 1u);
      continue;
    }
    if (A.op == ACC_FIRST_ROW) {
#pragma unroll
      for (int b = 0; b < B; b++)
        if ((live >> b) & 1) atomicMin((unsigned int *)(lb + slot[b] * 4), P.rowidx ? P.rowidx[row[b]] : row[b]);
      continue;
    }
    if (A.src != cur_src) { // accumulators are ordered by source column: one load serves all of them
      cur_src = A.src;
      const ValSpec V = P.vals[A.src];
      load_raw<B, QUAD>(V.values, wbytes(V.dtype), row, raw);
      ok = live;
      if (V.validity) {
#pragma unroll
        for (int b = 0; b < B; b++) if (!val_valid(V, (int64_t)row[b])) ok &= ~(1u << b);
      }
    }
    if (A.op == ACC_CNT_VALID) {
#pragma unroll
      for (int b = 0; b < B; b++) if ((ok >> b) & 1) atomicAdd((unsigned int *)(lb + slot[b] * 4), 1u);
      continue;
    }
    uint64_t opnd[B];
    raw_to_operands<B>(P.vals[A.src].dtype, A.conv, raw, opnd);
    switch (A.op) {
#define APPLY(EXPR) _Pragma("unroll") for (int b = 0; b < B; b++) if ((ok >> b) & 1) { auto *p = lb + slot[b] * 8; EXPR; } break;
      case ACC_ADD_U64: APPLY(atomicAdd((unsigned long long *)p, (unsigned long long)opnd[b]))
      case ACC_ADD_F64: APPLY(atomicAdd((double *)p, __longlong_as_double((long long)opnd[b])))
      case ACC_MIN_I64: APPLY(atomicMin((long long *)p, (long long)opnd[b]))
      case ACC_MAX_I64: APPLY(atomicMax((long long *)p, (long long)opnd[b]))
      case ACC_MIN_U64: APPLY(atomicMin((unsigned long long *)p, (unsigned long long)opnd[b]))
      case ACC_MAX_U64: APPLY(atomicMax((unsigned long long *)p, (unsigned long long)opnd[b]))
#undef APPLY
      default: break;
    }
  }
}

// Occupied slots -> list[] (u16 slot numbers, ordinary keys first, then the sentinel slot, then the null
// slot).  Returns the number of ordinary entries; every thread of the workgroup must call it.
template <typename KS>
DEVINL uint32_t build_list(const KS *lkeys, uint32_t lcap, uint16_t *list, uint32_t *sh_cnt) {
  const uint32_t lane = threadIdx.x & 63;
  if (threadIdx.x == 0) *sh_cnt = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < lcap; i0 += blockDim.x) {
    const uint32_t i = i0 + threadIdx.x;
    const bool occ = i < lcap && lkeys[i] != KE<KS>::v;
    const unsigned long long m = __ballot(occ);
    if (m) {
      uint32_t wbase = 0;
      if (lane == 0) wbase = atomicAdd(sh_cnt, (uint32_t)__popcll(m));
      wbase = __shfl(wbase, 0);
      if (occ) list[wbase + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)i;
    }
  }
  __syncthreads();
  return *sh_cnt;
}

// LDS table -> compact dump of item `item`
template <typename KS>
DEVINL void dump_table(const Plan &P, uint8_t *sm, uint32_t lbits, uint32_t *used, uint32_t *sh_cnt, uint16_t *list,
                       const DumpRef &D, uint32_t item) {
  const uint32_t lcap = 1u << lbits, lslots = lcap + 2;
  const KS *lkeys = (const KS *)sm;
  const uint32_t n_hash = build_list<KS>(lkeys, lcap, list, sh_cnt);
  const uint32_t hs = used[0] ? 1u : 0u, hn = used[1] ? 1u : 0u;
  if (threadIdx.x == 0) {
    if (hs) list[n_hash] = (uint16_t)lcap;
    if (hn) list[n_hash + hs] = (uint16_t)(lcap + 1);
    DumpHdr h;
    h.n_hash = n_hash; h.has_sent = hs; h.has_null = hn; h.pad = 0;
    D.hdr[item] = h;
  }
  __syncthreads();
  const uint32_t n = n_hash + hs + hn;
  uint8_t *base = D.base + (uint64_t)item * D.item_stride;
  KS *okeys = (KS *)base;
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t s = list[k];
    okeys[k] = s < lcap ? lkeys[s] : KE<KS>::v;
  }
  uint32_t loff = al8(lslots * (uint32_t)sizeof(KS));
  uint64_t goff = al16((uint64_t)D.cap * sizeof(KS));
  for (int a = 0; a < P.n_accs; a++) {
    const uint32_t w = P.accs[a].width;
    if (w == 8) {
      for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) ((uint64_t *)(base + goff))[k] = ((const uint64_t *)(sm + loff))[list[k]];
    } else {
      for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) ((uint32_t *)(base + goff))[k] = ((const uint32_t *)(sm + loff))[list[k]];
    }
    loff += al8(lslots * w);
    goff += al16((uint64_t)D.cap * w);
  }
}

DEVINL uint32_t table_lds_bytes(const Plan &P, uint32_t lslots, uint32_t key_bytes) {
  uint32_t off = al8(lslots * key_bytes);
  for (int a = 0; a < P.n_accs; a++) off += al8(lslots * P.accs[a].width);
  return off;
}

// ---------------------------------------------------------------------------------------------------------
// few groups: row slices of the input columns
// ---------------------------------------------------------------------------------------------------------
template <typename KS, bool QUAD>
__global__ void __launch_bounds__(1024) k_fold_rows(Plan P, FoldRowsArgs A) {
  __shared__ __align__(16) uint32_t sh[8]; // [0..2] used flags, [4] list counter
  const uint32_t lslots = (1u << A.lbits) + 2;
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  ltab_init<KS>(P, fsm, lslots);
  __syncthreads();
  const int64_t row0 = (int64_t)blockIdx.x * A.rows_per_item;
  const int64_t row1 = row0 + A.rows_per_item < P.n_rows ? row0 + A.rows_per_item : P.n_rows;
  for (int64_t base = row0 + (int64_t)threadIdx.x * 4; base < row1; base += (int64_t)blockDim.x * 4) {
    uint32_t row[4];
    uint32_t live = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int64_t r = base + b;
      if (r < row1) live |= 1u << b;
      row[b] = (uint32_t)(r < row1 ? r : base);
    }
    // a partial quad at the very end of the column may not be read with a vector load
    if (QUAD && live != 0xFu) fold_batch<KS, 4, false>(P, fsm, A.lbits, sh, row, live, A.keymode, A.kxor, false);
    else fold_batch<KS, 4, QUAD>(P, fsm, A.lbits, sh, row, live, A.keymode, A.kxor, false);
  }
  __syncthreads();
  if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
  uint16_t *list = (uint16_t *)(fsm + table_lds_bytes(P, lslots, sizeof(KS)));
  dump_table<KS>(P, fsm, A.lbits, sh, &sh[4], list, A.D, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// many groups, pass 1: sort every tile by hash partition (in LDS) and write it back tile-contiguous
// ---------------------------------------------------------------------------------------------------------
// exclusive scan of v[0..n), n <= blockDim.x = TS_BLOCK, one entry per thread
DEVINL uint32_t block_scan_excl(uint32_t x, uint32_t *tmp) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= (uint32_t)off) inc += t;
  }
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const uint32_t wv = lane < TS_BLOCK / 64 ? tmp[lane] : 0;
    uint32_t winc = wv;
#pragma unroll
    for (int off = 1; off < TS_BLOCK / 64; off <<= 1) {
      const uint32_t t = __shfl_up(winc, off);
      if (lane >= (uint32_t)off) winc += t;
    }
    if (lane < TS_BLOCK / 64) tmp[lane] = winc - wv;
  }
  __syncthreads();
  return tmp[wave] + inc - x;
}

// one column of the tile: registers -> sorted position in LDS -> coalesced 16-byte stores
template <typename T>
DEVINL void stage_and_store(uint8_t *stage, const T (&x)[TS_R], const uint32_t (&pos)[TS_R], uint32_t live, uint8_t *dst, uint32_t nt) {
#pragma unroll
  for (int r = 0; r < TS_R; r++) if ((live >> r) & 1) ((T *)stage)[pos[r]] = x[r];
  __syncthreads();
  const uint32_t nchunk = (nt * (uint32_t)sizeof(T) + 15u) >> 4; // the destination is padded to 16 bytes
  for (uint32_t c = threadIdx.x; c < nchunk; c += TS_BLOCK) ((uint4 *)dst)[c] = ((const uint4 *)stage)[c];
  __syncthreads();
}

template <typename KS>
__global__ void __launch_bounds__(TS_BLOCK) k_tile_sort(Plan P, PartOut O, uint32_t pbits, uint32_t n_tiles, uint32_t *__restrict__ offs,
                                                        int keymode, uint64_t kxor) {
  __shared__ __align__(16) uint32_t hist[(1 << TS_MAX_PBITS) + 4];
  __shared__ __align__(16) uint32_t tmp[TS_BLOCK / 64];
  const uint32_t np = 1u << pbits, nh = np + 1; // partition np holds the rows with a null key
  const uint32_t tid = threadIdx.x, tile = blockIdx.x;
  for (uint32_t p = tid; p < nh; p += TS_BLOCK) hist[p] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)tile * TS_TILE;
  const uint32_t nt = (uint32_t)((P.n_rows - r0) < TS_TILE ? (P.n_rows - r0) : TS_TILE);
  uint32_t row[TS_R], pos[TS_R];
  KS key[TS_R];
  uint32_t live = 0;
#pragma unroll
  for (int r = 0; r < TS_R; r++) {
    const uint32_t j = (uint32_t)r * TS_BLOCK + tid;
    if (j < nt) live |= 1u << r;
    row[r] = (uint32_t)(r0 + (j < nt ? j : 0));
  }
  uint32_t nullm = 0;
  if (keymode == 0) {
#pragma unroll
    for (int r = 0; r < TS_R; r++) {
      bool isn = false;
      key[r] = (KS)row_packed_key(P, (int64_t)row[r], isn);
      if (isn) { nullm |= 1u << r; key[r] = 0; }
    }
  } else {
    uint64_t raw[TS_R];
    load_raw<TS_R, false>(P.keys[0].values, keymode == 1 ? 4u : 8u, row, raw);
#pragma unroll
    for (int r = 0; r < TS_R; r++) key[r] = (KS)(raw[r] ^ kxor);
  }
#pragma unroll
  for (int r = 0; r < TS_R; r++) {
    pos[r] = 0;
    if (!((live >> r) & 1)) continue;
    const uint32_t p = ((nullm >> r) & 1) ? np : part_of(mix_a((uint64_t)key[r]), pbits);
    pos[r] = p | (atomicAdd(&hist[p], 1u) << 12);
  }
  __syncthreads();
  const uint32_t mine = tid < nh ? hist[tid] : 0;
  const uint32_t excl = block_scan_excl(mine, tmp);
  if (tid < nh) {
    hist[tid] = excl;
    offs[(uint64_t)tid * n_tiles + tile] = excl;
  }
  if (tid == 0) offs[(uint64_t)(np + 1) * n_tiles + tile] = nt;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < TS_R; r++) if ((live >> r) & 1) pos[r] = hist[pos[r] & 0xFFFu] + (pos[r] >> 12);

  stage_and_store<KS>(fsm, key, pos, live, O.keys + (size_t)r0 * sizeof(KS), nt);
  for (int v = 0; v < P.n_vals; v++) {
    const ValSpec V = P.vals[v];
    const uint32_t w = wbytes(V.dtype);
    uint64_t raw[TS_R];
    load_raw<TS_R, false>(V.values, w, row, raw);
    switch (w) {
#define STAGE(T) { T x[TS_R]; _Pragma("unroll") for (int r = 0; r < TS_R; r++) x[r] = (T)raw[r]; \
                   stage_and_store<T>(fsm, x, pos, live, O.vals[v] + (size_t)r0 * sizeof(T), nt); } break;
      case 1: STAGE(uint8_t)
      case 2: STAGE(uint16_t)
      case 4: STAGE(uint32_t)
      default: STAGE(uint64_t)
#undef STAGE
    }
    if (O.valid[v]) {
      uint8_t x[TS_R];
#pragma unroll
      for (int r = 0; r < TS_R; r++) x[r] = val_valid(V, (int64_t)row[r]) ? 1 : 0;
      stage_and_store<uint8_t>(fsm, x, pos, live, O.valid[v] + (size_t)r0, nt);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// many groups, pass 2: one (partition group, tile segment) per workgroup
// ---------------------------------------------------------------------------------------------------------
#define FR_U 4 // tiles whose runs one wave concatenates
#define FR_B 4 // rows per lane per pass
template <typename KS>
__global__ void __launch_bounds__(1024) k_fold_runs(Plan P, FoldRunsArgs A) {
  __shared__ __align__(16) uint32_t sh[8];
  const uint32_t lslots = (1u << A.lbits) + 2;
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  ltab_init<KS>(P, fsm, lslots);
  __syncthreads();
  const uint32_t item = blockIdx.x, pg = item / A.S, seg = item % A.S;
  const bool null_pg = A.has_null_pg && pg == A.n_pg - 1;
  const uint32_t p0 = null_pg ? A.np : pg * A.w, p1 = null_pg ? A.np + 1 : p0 + A.w;
  const uint32_t per = (A.n_tiles + A.S - 1) / A.S;
  const uint32_t t0 = seg * per, t1 = t0 + per < A.n_tiles ? t0 + per : A.n_tiles;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t *o0 = A.offs + (uint64_t)p0 * A.n_tiles, *o1 = A.offs + (uint64_t)p1 * A.n_tiles;
  for (uint32_t t = t0 + wave * FR_U; t < t1; t += nw * FR_U) {
    // lanes 0..U-1 fetch the run bounds of tiles t..t+U-1; every lane then gets the prefix sums
    uint32_t st = 0, len = 0;
    if (lane < FR_U && t + lane < t1) { st = o0[t + lane]; len = o1[t + lane] - st; }
    uint32_t s_[FR_U], c_[FR_U + 1];
    c_[0] = 0;
#pragma unroll
    for (int u = 0; u < FR_U; u++) {
      s_[u] = (uint32_t)__shfl((int)st, u) + (t + u) * (uint32_t)TS_TILE;
      c_[u + 1] = c_[u] + (uint32_t)__shfl((int)len, u);
    }
    const uint32_t tot = c_[FR_U];
    for (uint32_t j0 = 0; j0 < tot; j0 += 64 * FR_B) {
      uint32_t row[FR_B];
      uint32_t live = 0;
#pragma unroll
      for (int b = 0; b < FR_B; b++) {
        uint32_t j = j0 + (uint32_t)b * 64 + lane;
        if (j < tot) live |= 1u << b;
        else j = 0; // clamp dead lanes to the first row of the group (tot > 0 here)
        uint32_t base = s_[0], c = 0;
#pragma unroll
        for (int u = 1; u < FR_U; u++) if (j >= c_[u]) { base = s_[u]; c = c_[u]; }
        row[b] = base + (j - c);
      }
      // the first tile of the group may hold no rows of this partition: clamp to any in-bounds row
      fold_batch<KS, FR_B, false>(P, fsm, A.lbits, sh, row, live, P.keys[0].dtype == PLHIP_U32 ? 1 : 2, 0ull, null_pg);
    }
  }
  __syncthreads();
  if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
  uint16_t *list = (uint16_t *)(fsm + table_lds_bytes(P, lslots, sizeof(KS)));
  dump_table<KS>(P, fsm, A.lbits, sh, &sh[4], list, A.D, item);
}

// ---------------------------------------------------------------------------------------------------------
// merge the dumps of one partition group (split over 2^mbits workgroups by key hash) and write the outputs
// ---------------------------------------------------------------------------------------------------------
DEVINL uint32_t merge_bucket(uint64_t key, uint32_t mbits) {
  return mbits ? (mix_a(key) * 0x85EBCA77u) >> (32 - mbits) : 0u;
}

template <typename KS>
__global__ void __launch_bounds__(1024) k_merge_emit(Plan P, MergeArgs A, const OutPlan *__restrict__ Op) {
  __shared__ __align__(16) uint32_t sh[8];
  __shared__ __align__(16) uint32_t loff[PLHIP_MAX_ACCS];
  const uint32_t lcap = 1u << A.lbits, lslots = lcap + 2;
  KS *lkeys = (KS *)fsm;
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    uint32_t off = al8(lslots * (uint32_t)sizeof(KS));
    for (int a = 0; a < P.n_accs; a++) { loff[a] = off; off += al8(lslots * P.accs[a].width); }
  }
  ltab_init<KS>(P, fsm, lslots);
  __syncthreads();
  const uint32_t pg = blockIdx.x >> A.mbits, m = blockIdx.x & ((1u << A.mbits) - 1);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (uint32_t s = wave; s < A.S; s += nw) {
    const uint32_t item = pg * A.S + s;
    const DumpHdr h = A.D.hdr[item];
    const uint8_t *base = A.D.base + (uint64_t)item * A.D.item_stride;
    const KS *dkeys = (const KS *)base;
    const uint32_t n = h.n_hash + (m == 0 ? h.has_sent + h.has_null : 0);
    for (uint32_t k = lane; k < n; k += 64) {
      uint32_t slot;
      if (k < h.n_hash) {
        const KS key = dkeys[k];
        if (merge_bucket((uint64_t)key, A.mbits) != m) continue;
        slot = ltab_slot<KS>(lkeys, A.lbits, sh, key);
        if (slot == ~0u) continue;
      } else if (k == h.n_hash && h.has_sent) { sh[0] = 1; slot = lcap; }
      else { sh[1] = 1; slot = lcap + 1; }
      uint64_t goff = al16((uint64_t)A.D.cap * sizeof(KS));
      for (int a = 0; a < P.n_accs; a++) {
        const AccSpec Ac = P.accs[a];
        const uint64_t part = Ac.width == 8 ? ((const uint64_t *)(base + goff))[k] : (uint64_t)((const uint32_t *)(base + goff))[k];
        goff += al16((uint64_t)A.D.cap * Ac.width);
        if (part == Ac.init) continue; // identity: nothing to fold
        uint8_t *p = fsm + loff[a] + slot * Ac.width;
        switch (Ac.op) {
          case ACC_ADD_U64: atomicAdd((unsigned long long *)p, (unsigned long long)part); break;
          case ACC_ADD_F64: atomicAdd((double *)p, __longlong_as_double((long long)part)); break;
          case ACC_MIN_I64: atomicMin((long long *)p, (long long)part); break;
          case ACC_MAX_I64: atomicMax((long long *)p, (long long)part); break;
          case ACC_MIN_U64: atomicMin((unsigned long long *)p, (unsigned long long)part); break;
          case ACC_MAX_U64: atomicMax((unsigned long long *)p, (unsigned long long)part); break;
          case ACC_CNT_ALL: case ACC_CNT_VALID: atomicAdd((unsigned int *)p, (unsigned int)part); break;
          case ACC_FIRST_ROW: atomicMin((unsigned int *)p, (unsigned int)part); break;
        }
      }
    }
  }
  __syncthreads();
  if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
  uint16_t *list = (uint16_t *)(fsm + table_lds_bytes(P, lslots, sizeof(KS)));
  const uint32_t n_hash = build_list<KS>(lkeys, lcap, list, &sh[4]);
  const uint32_t hs = sh[0] ? 1u : 0u, hn = sh[1] ? 1u : 0u;
  const uint32_t n = n_hash + hs + hn;
  if (threadIdx.x == 0) {
    if (hs) list[n_hash] = (uint16_t)lcap;
    if (hn) list[n_hash + hs] = (uint16_t)(lcap + 1);
    sh[5] = n ? atomicAdd(&A.flags[FLAG_NGROUPS], n) : 0;
  }
  __syncthreads();
  const uint64_t gbase = sh[5];
  if (gbase + n > A.out_cap) { // the host sized the outputs from the table capacities: cannot happen unless they overflowed
    if (threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
    return;
  }
  const OutPlan &O = *Op;
  for (uint32_t k0 = 0; k0 < n; k0 += blockDim.x) { // a wave covers 64 consecutive output rows
    const uint32_t k = k0 + threadIdx.x;
    const bool livek = k < n;
    const uint32_t s = livek ? list[k] : 0;
    const uint64_t i = gbase + k;
    const uint64_t packed = livek ? (uint64_t)lkeys[s] : 0;
    for (int o = 0; o < O.n_outs; o++) {
      const OutSpec &S = O.outs[o];
      bool valid = false;
      if (livek) {
        if (S.kind == OUT_KEY) valid = finalize_key(P, S, packed, s == lcap + 1, s == lcap, i);
        else {
          uint64_t acc = 0;
          if (S.acc >= 0) acc = P.accs[S.acc].width == 8 ? ((const uint64_t *)(fsm + loff[S.acc]))[s] : ((const uint32_t *)(fsm + loff[S.acc]))[s];
          const uint32_t cnt_all = S.cnt_all >= 0 ? ((const uint32_t *)(fsm + loff[S.cnt_all]))[s] : 0;
          const uint32_t cnt_valid = S.cnt_valid >= 0 ? ((const uint32_t *)(fsm + loff[S.cnt_valid]))[s] : cnt_all;
          valid = finalize_value(S, acc, cnt_all, cnt_valid, i);
        }
      }
      if (S.validity) { // bits [i0, i0 + 64) of a zeroed bitmap; other workgroups own the neighbouring bits
        const unsigned long long mk = __ballot(valid);
        if (mk) {
          const uint64_t i0 = gbase + (k0 + (threadIdx.x & ~63u));
          const uint32_t sft = (uint32_t)(i0 & 31);
          const unsigned long long lo = mk << sft, hi = sft ? mk >> (64 - sft) : 0ull;
          uint32_t word = lane == 0 ? (uint32_t)lo : lane == 1 ? (uint32_t)(lo >> 32) : (uint32_t)hi;
          if (lane < 3 && word) atomicOr((unsigned int *)S.validity + (i0 >> 5) + lane, word);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// sampled cardinality (the analogue of estimate_unique_count, group_by_partitioned.rs:115-157): one
// workgroup inserts 32-bit key fingerprints of a strided sample into an LDS set and counts them
// ---------------------------------------------------------------------------------------------------------
#define EST_BITS 15
__global__ void __launch_bounds__(1024) k_estimate(Plan P, int64_t n_sample, int64_t stride, uint32_t *flags) {
  __shared__ uint32_t set[1 << EST_BITS];
  __shared__ uint32_t cnt;
  for (uint32_t i = threadIdx.x; i < (1u << EST_BITS); i += blockDim.x) set[i] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (int64_t i = threadIdx.x; i < n_sample; i += blockDim.x) {
    bool isn;
    const uint64_t key = row_packed_key(P, i * stride, isn);
    uint32_t f = isn ? 0x7FFFFFFEu : (uint32_t)(hash_key(key) >> 32);
    if (f == 0xFFFFFFFFu) f = 0x7FFFFFFFu;
    uint32_t s = f & ((1u << EST_BITS) - 1);
    for (int it = 0; it < (1 << EST_BITS); it++) {
      uint32_t k = set[s];
      if (k == f) break;
      if (k == 0xFFFFFFFFu) {
        k = atomicCAS(&set[s], 0xFFFFFFFFu, f);
        if (k == 0xFFFFFFFFu) { mine++; break; }
        if (k == f) break;
      }
      s = (s + 1) & ((1u << EST_BITS) - 1);
    }
  }
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) flags[FLAG_NGROUPS] = cnt;
}

template <typename F>
void allow_big_lds(F f) {
  // static LDS (flags, histogram) comes on top of the dynamic part: leave 8 KB for it
  if (hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError();
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
size_t fast_table_bytes(const Plan &p, uint32_t lbits, int key_bytes) {
  const size_t lslots = ((size_t)1 << lbits) + 2;
  size_t off = (lslots * key_bytes + 7) & ~(size_t)7;
  for (int a = 0; a < p.n_accs; a++) off += (lslots * p.accs[a].width + 7) & ~(size_t)7;
  return off + lslots * 2 + 16; // + the u16 slot list of the dump / emit step
}

uint64_t fast_dump_stride(const Plan &p, uint32_t cap, int key_bytes) {
  uint64_t off = ((uint64_t)cap * key_bytes + 15) & ~15ull;
  for (int a = 0; a < p.n_accs; a++) off += ((uint64_t)cap * p.accs[a].width + 15) & ~15ull;
  return (off + 255) & ~255ull;
}

void launch_fold_rows(const Plan &p, const FoldRowsArgs &a, int key_bytes, bool quad, uint32_t n_items, uint32_t block, hipStream_t s) {
  static bool once = false;
  if (!once) {
    allow_big_lds(k_fold_rows<uint32_t, true>); allow_big_lds(k_fold_rows<uint32_t, false>);
    allow_big_lds(k_fold_rows<uint64_t, true>); allow_big_lds(k_fold_rows<uint64_t, false>);
    once = true;
  }
  const size_t bytes = fast_table_bytes(p, a.lbits, key_bytes);
  if (key_bytes == 4) {
    if (quad) hipLaunchKernelGGL((k_fold_rows<uint32_t, true>), dim3(n_items), dim3(block), bytes, s, p, a);
    else hipLaunchKernelGGL((k_fold_rows<uint32_t, false>), dim3(n_items), dim3(block), bytes, s, p, a);
  } else {
    if (quad) hipLaunchKernelGGL((k_fold_rows<uint64_t, true>), dim3(n_items), dim3(block), bytes, s, p, a);
    else hipLaunchKernelGGL((k_fold_rows<uint64_t, false>), dim3(n_items), dim3(block), bytes, s, p, a);
  }
}

void launch_tile_sort(const Plan &p, const PartOut &o, uint32_t pbits, uint32_t n_tiles, uint32_t *offs, int keymode, uint64_t kxor,
                      int key_bytes, hipStream_t s) {
  static bool once = false;
  if (!once) { allow_big_lds(k_tile_sort<uint32_t>); allow_big_lds(k_tile_sort<uint64_t>); once = true; }
  const size_t bytes = (size_t)TS_TILE * 8;
  if (key_bytes == 4) hipLaunchKernelGGL(k_tile_sort<uint32_t>, dim3(n_tiles), dim3(TS_BLOCK), bytes, s, p, o, pbits, n_tiles, offs, keymode, kxor);
  else hipLaunchKernelGGL(k_tile_sort<uint64_t>, dim3(n_tiles), dim3(TS_BLOCK), bytes, s, p, o, pbits, n_tiles, offs, keymode, kxor);
}

void launch_fold_runs(const Plan &p2, const FoldRunsArgs &a, int key_bytes, uint32_t block, hipStream_t s) {
  static bool once = false;
  if (!once) { allow_big_lds(k_fold_runs<uint32_t>); allow_big_lds(k_fold_runs<uint64_t>); once = true; }
  const size_t bytes = fast_table_bytes(p2, a.lbits, key_bytes);
  const uint32_t grid = a.n_pg * a.S;
  if (key_bytes == 4) hipLaunchKernelGGL(k_fold_runs<uint32_t>, dim3(grid), dim3(block), bytes, s, p2, a);
  else hipLaunchKernelGGL(k_fold_runs<uint64_t>, dim3(grid), dim3(block), bytes, s, p2, a);
}

void launch_merge_emit(const Plan &p, const MergeArgs &a, const OutPlan *o_dev, int key_bytes, uint32_t n_pg, uint32_t block, hipStream_t s) {
  static bool once = false;
  if (!once) { allow_big_lds(k_merge_emit<uint32_t>); allow_big_lds(k_merge_emit<uint64_t>); once = true; }
  const size_t bytes = fast_table_bytes(p, a.lbits, key_bytes);
  const uint32_t grid = n_pg << a.mbits;
  if (key_bytes == 4) hipLaunchKernelGGL(k_merge_emit<uint32_t>, dim3(grid), dim3(block), bytes, s, p, a, o_dev);
  else hipLaunchKernelGGL(k_merge_emit<uint64_t>, dim3(grid), dim3(block), bytes, s, p, a, o_dev);
}

void launch_estimate(const Plan &p, int64_t n_sample, int64_t stride, uint32_t *flags, hipStream_t s) {
  hipLaunchKernelGGL(k_estimate, dim3(1), dim3(1024), 0, s, p, n_sample, stride, flags);
}
