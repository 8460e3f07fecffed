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
#include "shapes.hpp"

extern __shared__ __align__(16) uint8_t fsm[];

namespace {

template <typename KS> struct KE;
template <> struct KE<uint32_t> { static constexpr uint32_t v = 0xFFFFFFFFu; };
template <> struct KE<uint64_t> { static constexpr uint64_t v = 0xFFFFFFFFFFFFFFFFull; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load
// and store of the wave (s_waitcnt vmcnt(0)), which serialises a kernel that streams through LDS.
DEVINL void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

DEVINL uint32_t al8(uint32_t x) { return (x + 7u) & ~7u; }
DEVINL uint64_t al16(uint64_t x) { return (x + 15ull) & ~15ull; }

DEVINL uint32_t cas_key(uint32_t *p, uint32_t expect, uint32_t val) { return atomicCAS(p, expect, val); }
DEVINL uint64_t cas_key(uint64_t *p, uint64_t expect, uint64_t val) {
  return atomicCAS((unsigned long long *)p, (unsigned long long)expect, (unsigned long long)val);
}

// used[0]: sentinel slot taken, used[1]: null slot taken, used[2]: table full
#define FP_PROBE_LIMIT 96
// Home slot: the bits of the Fibonacci product right below the `pshift` bits the tile sort partitions on
// (pshift = 0 when the rows were not partitioned): one multiply and one bit-field extract per row.
DEVINL uint32_t mix_key(uint32_t k) { return k * 0x9E3779B1u; }
DEVINL uint32_t mix_key(uint64_t k) { return mix_a(k); }
DEVINL uint32_t home_slot(uint32_t a, uint32_t lbits, uint32_t pshift) { return (a << pshift) >> (32 - lbits); }

template <typename KS>
DEVINL uint32_t ltab_slot(KS *lkeys, uint32_t lbits, uint32_t pshift, uint32_t *used, KS k0) {
  const uint32_t lcap = 1u << lbits;
  if (k0 == KE<KS>::v) { used[0] = 1; return lcap; }
  const uint32_t mask = lcap - 1;
  uint32_t s = home_slot(mix_key(k0), lbits, pshift);
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

// LDS table: lcap hashed slots, then the slot of the key equal to the empty sentinel, the slot of the null
// key, and 64 trash slots (one per lane: no bank conflicts, no same-address serialisation) that rows which do
// not count (tail lanes, null values) are steered to, so that the accumulator updates need no predication.
#define FP_EXTRA_SLOTS 66u
DEVINL uint32_t lslots_of(uint32_t lbits) { return (1u << lbits) + FP_EXTRA_SLOTS; }

// Folds B rows per lane into the workgroup's LDS table.  row[] are in-bounds for every lane (dead lanes are
// clamped by the caller); bit b of `live` says whether row[b] counts.  All global loads of the batch (keys
// and the first NPRE value columns) are issued before the first LDS probe, so a wave exposes one memory
// round trip per batch; the B probes of a lane advance together, one batched LDS read per probe step.
template <typename KS, int B, bool QUAD, int NPRE, typename SH>
DEVINL void fold_batch(const Plan &P, uint8_t *sm, uint32_t lbits, uint32_t *used, const uint32_t (&row)[B], uint32_t live,
                       int keymode, const uint8_t *kvals, uint64_t kxor, bool force_null, uint32_t pshift) {
  const uint32_t lcap = 1u << lbits, lslots = lcap + FP_EXTRA_SLOTS, trash = lcap + 2 + (threadIdx.x & 63), mask = lcap - 1;
  KS *lkeys = (KS *)sm;
  KS key[B];
  uint32_t nullm = 0;
  if (!force_null) {
    if (keymode == 0) {
#pragma unroll
      for (int b = 0; b < B; b++) {
        bool isn = false;
        key[b] = (KS)row_packed_key(P, (int64_t)row[b], isn);
        if (isn) nullm |= 1u << b;
      }
    } else {
      uint64_t kraw[B];
      load_raw<B, QUAD>(kvals, keymode == 1 ? 4u : 8u, row, kraw);
#pragma unroll
      for (int b = 0; b < B; b++) key[b] = (KS)(kraw[b] ^ kxor);
    }
  }
  uint64_t pre[NPRE ? NPRE : 1][B];
#pragma unroll
  for (int c = 0; c < NPRE; c++)
    if (c < SH::n_vals(P)) load_raw<B, QUAD>(P.vals[c].values, wbytes(SH::val_dtype(P, c)), row, pre[c]);

  // ---- slots -----------------------------------------------------------------------------------------------
  // Every row probes, dead rows too: the caller clamps them to a real row of the same table, so at worst they
  // look up a key that is (or will be) there anyway; they are steered to the trash slot afterwards.  That
  // keeps the common path free of per-row branches: hash, one batched LDS read, compare.
  uint32_t slot[B];
  if (force_null) {
    if (live) used[1] = 1;
#pragma unroll
    for (int b = 0; b < B; b++) slot[b] = lcap + 1;
  } else {
    bool any_sent = false;
#pragma unroll
    for (int b = 0; b < B; b++) {
      slot[b] = home_slot(mix_key(key[b]), lbits, pshift);
      any_sent |= key[b] == KE<KS>::v;
    }
    KS k[B];
#pragma unroll
    for (int b = 0; b < B; b++) k[b] = lkeys[slot[b]];
    uint32_t pend = 0;
#pragma unroll
    for (int b = 0; b < B; b++) pend |= (k[b] != key[b]) ? 1u << b : 0u;
    if (any_sent || nullm) { // rare: rows that live in the reserved slots
#pragma unroll
      for (int b = 0; b < B; b++) {
        if ((nullm >> b) & 1) { slot[b] = lcap + 1; pend &= ~(1u << b); if ((live >> b) & 1) used[1] = 1; }
        else if (key[b] == KE<KS>::v) { slot[b] = lcap; pend &= ~(1u << b); if ((live >> b) & 1) used[0] = 1; }
      }
    }
    // the rest (new keys, displaced keys): every trip takes ONE pending row of each lane through the probe
    // loop, so the trip count is the largest number of such rows any lane of the wave holds
    while (pend) {
      const int bsel = __ffs((int)pend) - 1;
      pend &= pend - 1;
      KS kk = key[0];
      uint32_t ss = slot[0];
#pragma unroll
      for (int b = 1; b < B; b++) if (bsel == b) { kk = key[b]; ss = slot[b]; }
      for (int step = 0;; step++) {
        KS kb = lkeys[ss];
        if (kb == KE<KS>::v) kb = cas_key(&lkeys[ss], KE<KS>::v, kk); // returns EMPTY when the claim won
        if (kb == kk || kb == KE<KS>::v) break;
        ss = (ss + 1) & mask;
        if (step >= FP_PROBE_LIMIT) { used[2] = 1; ss = trash; break; } // table full: the host redoes the query
      }
#pragma unroll
      for (int b = 0; b < B; b++) if (bsel == b) slot[b] = ss;
    }
  }
#pragma unroll
  for (int b = 0; b < B; b++) slot[b] = ((live >> b) & 1) ? slot[b] : trash;

  // ---- accumulators: unconditional LDS atomics, rows that do not count go to the trash slot ----------------
  uint32_t off = al8(lslots * (uint32_t)sizeof(KS));
  int cur_src = -1;
  uint64_t raw[B];
  uint32_t vslot[B]; // slot[] with the null values of the current source column steered to the trash slot
#pragma unroll
  for (int b = 0; b < B; b++) { raw[b] = 0; vslot[b] = slot[b]; }
  auto body = [&](int a) {
    const uint32_t op = SH::acc_op(P, a), width = SH::acc_width(P, a);
    const int src = SH::acc_src(P, a);
    uint8_t *lb = sm + off;
    off += al8(lslots * width);
    if (op == ACC_CNT_ALL) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicAdd((unsigned int *)(lb + slot[b] * 4), 1u);
      return;
    }
    if (op == ACC_FIRST_ROW) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicMin((unsigned int *)(lb + slot[b] * 4), P.rowidx ? P.rowidx[row[b]] : row[b]);
      return;
    }
    if (src != cur_src) { // accumulators are ordered by source column: one fetch serves all of them
      cur_src = src;
      if (src < NPRE) {
        // wave-uniform select of the prefetched column (register arrays cannot be indexed dynamically)
#pragma unroll
        for (int c = 0; c < NPRE; c++)
          if (src == c) {
#pragma unroll
            for (int b = 0; b < B; b++) raw[b] = pre[c][b];
          }
      } else {
        load_raw<B, QUAD>(P.vals[src].values, wbytes(SH::val_dtype(P, src)), row, raw);
      }
#pragma unroll
      for (int b = 0; b < B; b++) vslot[b] = slot[b];
      if (SH::val_nullable(P, src)) {
        const ValSpec V = P.vals[src];
#pragma unroll
        for (int b = 0; b < B; b++) if (!val_valid(V, (int64_t)row[b])) vslot[b] = trash;
      }
    }
    if (op == ACC_CNT_VALID) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicAdd((unsigned int *)(lb + vslot[b] * 4), 1u);
      return;
    }
    uint64_t opnd[B];
    raw_to_operands<B>(SH::val_dtype(P, src), SH::acc_conv(P, a), raw, opnd);
    if (op == ACC_ADD_U64) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicAdd((unsigned long long *)(lb + vslot[b] * 8), (unsigned long long)opnd[b]);
    } else if (op == ACC_ADD_F64) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicAdd((double *)(lb + vslot[b] * 8), __longlong_as_double((long long)opnd[b]));
    } else if (op == ACC_MIN_I64) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicMin((long long *)(lb + vslot[b] * 8), (long long)opnd[b]);
    } else if (op == ACC_MAX_I64) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicMax((long long *)(lb + vslot[b] * 8), (long long)opnd[b]);
    } else if (op == ACC_MIN_U64) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicMin((unsigned long long *)(lb + vslot[b] * 8), (unsigned long long)opnd[b]);
    } else if (op == ACC_MAX_U64) {
#pragma unroll
      for (int b = 0; b < B; b++) atomicMax((unsigned long long *)(lb + vslot[b] * 8), (unsigned long long)opnd[b]);
    }
  };
  if constexpr (SH::STATIC) static_for<0, SH::NA>(body);
  else for (int a = 0; a < SH::n_accs(P); a++) body(a);
}

DEVINL uint32_t merge_bucket(uint64_t key, uint32_t mbits) {
  return mbits ? (mix_a(key) * 0x85EBCA77u) >> (32 - mbits) : 0u;
}

// Occupied slots -> list[] (u16 slot numbers) grouped by merge bucket; bk[0..nb] receives the bucket
// offsets (bk[nb] = number of ordinary entries).  Every thread of the workgroup must call it.
template <typename KS>
DEVINL uint32_t build_list(const KS *lkeys, uint32_t lcap, uint32_t mbits, uint16_t *list, uint32_t *bk /* [2 * 65] */) {
  const uint32_t nb = 1u << mbits;
  uint32_t *cur = bk + 65;
  for (uint32_t b = threadIdx.x; b <= nb; b += blockDim.x) { bk[b] = 0; cur[b] = 0; }
  __syncthreads();
  if (mbits) {
    for (uint32_t i = threadIdx.x; i < lcap; i += blockDim.x) {
      const KS k = lkeys[i];
      if (k != KE<KS>::v) atomicAdd(&bk[merge_bucket((uint64_t)k, mbits) + 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) for (uint32_t b = 0; b < nb; b++) { bk[b + 1] += bk[b]; }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) cur[b] = bk[b];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lcap; i += blockDim.x) {
      const KS k = lkeys[i];
      if (k != KE<KS>::v) list[atomicAdd(&cur[merge_bucket((uint64_t)k, mbits)], 1u)] = (uint16_t)i;
    }
  } else {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t i0 = 0; i0 < lcap; i0 += blockDim.x) {
      const uint32_t i = i0 + threadIdx.x;
      const bool occ = i < lcap && lkeys[i] != KE<KS>::v;
      const unsigned long long m = __ballot(occ);
      if (m) {
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&bk[1], (uint32_t)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (occ) list[wbase + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)i;
      }
    }
  }
  __syncthreads();
  return bk[nb];
}

// LDS table -> compact dump of item `item`, entries grouped by merge bucket
template <typename KS>
DEVINL void dump_table(const Plan &P, uint8_t *sm, uint32_t lbits, uint32_t mbits, uint32_t *used, uint32_t *bk, uint16_t *list,
                       const DumpRef &D, uint32_t item) {
  const uint32_t lcap = 1u << lbits, lslots = lcap + FP_EXTRA_SLOTS;
  const KS *lkeys = (const KS *)sm;
  const uint32_t n_hash = build_list<KS>(lkeys, lcap, mbits, list, bk);
  const uint32_t hs = used[0] ? 1u : 0u, hn = used[1] ? 1u : 0u;
  DumpHdr *H = &D.hdr[item];
  if (threadIdx.x == 0) {
    if (hs) list[n_hash] = (uint16_t)lcap;
    if (hn) list[n_hash + hs] = (uint16_t)(lcap + 1);
    H->n_hash = n_hash; H->has_sent = hs; H->has_null = hn; H->pad = 0;
  }
  for (uint32_t b = threadIdx.x; b <= (1u << mbits); b += blockDim.x) H->boff[b] = bk[b];
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

// LDS table -> output columns.  A workgroup takes a block of output rows with one global atomic; inside it
// thread k writes row base + k of every column (coalesced); validity bits go out as three 32-bit ORs per wave
// into a zeroed bitmap.
template <typename KS>
DEVINL void emit_table(const Plan &P, uint8_t *sm, uint32_t lbits, uint32_t *used, uint32_t *bk, uint16_t *list,
                       const OutPlan &O, uint32_t *flags, uint64_t out_cap) {
  __shared__ uint32_t e_loff[PLHIP_MAX_ACCS];
  __shared__ uint32_t e_base;
  const uint32_t lcap = 1u << lbits, lslots = lcap + FP_EXTRA_SLOTS;
  const KS *lkeys = (const KS *)sm;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t n_hash = build_list<KS>(lkeys, lcap, 0, list, bk);
  const uint32_t hs = used[0] ? 1u : 0u, hn = used[1] ? 1u : 0u;
  const uint32_t n = n_hash + hs + hn;
  if (threadIdx.x == 0) {
    if (hs) list[n_hash] = (uint16_t)lcap;
    if (hn) list[n_hash + hs] = (uint16_t)(lcap + 1);
    uint32_t off = al8(lslots * (uint32_t)sizeof(KS));
    for (int a = 0; a < P.n_accs; a++) { e_loff[a] = off; off += al8(lslots * P.accs[a].width); }
    e_base = n ? atomicAdd(&flags[FLAG_NGROUPS], n) : 0;
  }
  __syncthreads();
  const uint64_t gbase = e_base;
  if (gbase + n > out_cap) { // the outputs are sized from the table capacities: only after an overflow elsewhere
    if (threadIdx.x == 0) { flags[FLAG_OVERFLOW] = 1; used[2] = 1; }
    return;
  }
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
          if (S.acc >= 0) acc = P.accs[S.acc].width == 8 ? ((const uint64_t *)(sm + e_loff[S.acc]))[s] : ((const uint32_t *)(sm + e_loff[S.acc]))[s];
          const uint32_t cnt_all = S.cnt_all >= 0 ? ((const uint32_t *)(sm + e_loff[S.cnt_all]))[s] : 0;
          const uint32_t cnt_valid = S.cnt_valid >= 0 ? ((const uint32_t *)(sm + e_loff[S.cnt_valid]))[s] : cnt_all;
          valid = finalize_value(S, acc, cnt_all, cnt_valid, i);
        }
      }
      if (S.validity) {
        const unsigned long long mk = __ballot(valid);
        if (mk) {
          const uint64_t i0 = gbase + (k0 + (threadIdx.x & ~63u));
          const uint32_t sft = (uint32_t)(i0 & 31);
          const unsigned long long lo = mk << sft, hi = sft ? mk >> (64 - sft) : 0ull;
          const uint32_t word = lane == 0 ? (uint32_t)lo : lane == 1 ? (uint32_t)(lo >> 32) : (uint32_t)hi;
          if (lane < 3 && word) atomicOr((unsigned int *)S.validity + (i0 >> 5) + lane, word);
        }
      }
    }
  }
}

// exclusive scan of one value per thread over the workgroup (blockDim.x a multiple of 64, <= 1024)
DEVINL uint32_t block_scan_excl(uint32_t x, uint32_t *tmp /* [16] */) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t inc = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= (uint32_t)off) inc += t;
  }
  __syncthreads(); // tmp may still be read from a previous call
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const uint32_t wv = lane < nw ? tmp[lane] : 0;
    uint32_t winc = wv;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t t = __shfl_up(winc, off);
      if (lane >= (uint32_t)off) winc += t;
    }
    if (lane < nw) tmp[lane] = winc - wv;
  }
  __syncthreads();
  return tmp[wave] + inc - x;
}

// the same with LDS-only barriers (tile sort: global loads / stores stay in flight)
// exclusive scan of one value per thread over the workgroup (blockDim.x a multiple of 64, <= 1024)
DEVINL uint32_t block_scan_excl_lds(uint32_t x, uint32_t *tmp /* [16] */) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t inc = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= (uint32_t)off) inc += t;
  }
  lds_barrier(); // tmp may still be read from a previous call
  if (lane == 63) tmp[wave] = inc;
  lds_barrier();
  if (wave == 0) {
    const uint32_t wv = lane < nw ? tmp[lane] : 0;
    uint32_t winc = wv;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t t = __shfl_up(winc, off);
      if (lane >= (uint32_t)off) winc += t;
    }
    if (lane < nw) tmp[lane] = winc - wv;
  }
  lds_barrier();
  return tmp[wave] + inc - x;
}

// ---------------------------------------------------------------------------------------------------------
// few groups: row slices of the input columns
// ---------------------------------------------------------------------------------------------------------
#define ROWS_NPRE 3
template <typename KS, bool QUAD, typename SH>
__global__ void __launch_bounds__(1024) k_fold_rows(Plan P, FoldRowsArgs A) {
  __shared__ __align__(16) uint32_t sh[8]; // [0..2] used flags
  __shared__ __align__(16) uint32_t bk[136];
  const uint32_t lslots = lslots_of(A.lbits);
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  ltab_init<KS>(P, fsm, lslots);
  __syncthreads();
  const int64_t row0 = (int64_t)blockIdx.x * A.rows_per_item;
  const int64_t row1 = row0 + A.rows_per_item < P.n_rows ? row0 + A.rows_per_item : P.n_rows;
  constexpr int ROWS_B = SH::STATIC ? 8 : 4; // rows per lane per batch (static shapes are light on registers)
  const int64_t step = (int64_t)blockDim.x * ROWS_B;
  for (int64_t base = row0; base < row1; base += step) {
    uint32_t row[ROWS_B];
    uint32_t live = 0;
#pragma unroll
    for (int g = 0; g < ROWS_B / 4; g++) { // quad g of this lane: a wave reads 1 KB of a 4-byte column per instruction
      const int64_t q = base + ((int64_t)g * blockDim.x + threadIdx.x) * 4;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int64_t r = q + b;
        if (r < row1) live |= 1u << (4 * g + b);
        row[4 * g + b] = (uint32_t)(r < row1 ? r : row0);
      }
    }
    if (!live) continue;
    // a batch that holds the end of the column is read with scalar loads: its last quad may be cut
    if (QUAD && live != (1u << ROWS_B) - 1u) fold_batch<KS, ROWS_B, false, ROWS_NPRE, SH>(P, fsm, A.lbits, sh, row, live, A.keymode, P.keys[0].values, A.kxor, false, 0u);
    else fold_batch<KS, ROWS_B, QUAD, ROWS_NPRE, SH>(P, fsm, A.lbits, sh, row, live, A.keymode, P.keys[0].values, A.kxor, false, 0u);
  }
  __syncthreads();
  if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
  uint16_t *list = (uint16_t *)(fsm + table_lds_bytes(P, lslots, sizeof(KS)));
  dump_table<KS>(P, fsm, A.lbits, A.mbits, sh, bk, list, A.D, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// many groups, pass 1: sort every tile by hash partition (in LDS) and write it back tile-contiguous
// ---------------------------------------------------------------------------------------------------------
// one column of the tile: registers -> sorted position in LDS -> coalesced 16-byte stores
template <typename T>
DEVINL void stage_and_store(uint8_t *stage, const T (&x)[TS_R], const uint32_t (&pos)[TS_R], uint32_t live, uint8_t *dst, uint32_t nt) {
#pragma unroll
  for (int r = 0; r < TS_R; r++) if ((live >> r) & 1) ((T *)stage)[pos[r]] = x[r];
  lds_barrier();
  const uint32_t nchunk = (nt * (uint32_t)sizeof(T) + 15u) >> 4; // the destination is padded to 16 bytes
  for (uint32_t c = threadIdx.x; c < nchunk; c += TS_BLOCK) ((uint4 *)dst)[c] = ((const uint4 *)stage)[c];
  lds_barrier();
}

// scatter variant (level 1 of the two-level pipeline): the sorted tile goes to partition-contiguous copies instead of
// back to its own place; element j of the sorted tile lands at dest[j] (consecutive inside a run)
template <typename T>
DEVINL void stage_and_scatter(uint8_t *stage, const T (&x)[TS_R], const uint32_t (&pos)[TS_R], uint32_t live, uint8_t *dst_base,
                              const uint32_t (&dest)[TS_R], uint32_t nt) {
#pragma unroll
  for (int r = 0; r < TS_R; r++) if ((live >> r) & 1) ((T *)stage)[pos[r]] = x[r];
  lds_barrier();
#pragma unroll
  for (int i = 0; i < TS_R; i++) {
    const uint32_t j = (uint32_t)i * TS_BLOCK + threadIdx.x;
    if (j < nt) ((T *)dst_base)[dest[i]] = ((const T *)stage)[j];
  }
  lds_barrier();
}

#define TS_NPRE 2 // value columns requested together with the keys
// SCATTER: `offs` is the per-(partition, tile) exclusive row prefix of k_part_hist + k_part_scan (read, not written),
// L1.part_start the partitions' first rows in the output; partition = top bits of mix_a(key), as k_part_hist has it.
template <typename KS, bool SCATTER>
__global__ void __launch_bounds__(TS_BLOCK) k_tile_sort(Plan P, PartOut O, uint32_t pbits, uint32_t n_tiles, uint32_t *__restrict__ offs,
                                                        int keymode, uint64_t kxor, TileSortLevel1 L1) {
  __shared__ __align__(16) uint32_t hist[(1 << TS_MAX_PBITS) + 4];
  __shared__ __align__(16) uint32_t tmp[TS_BLOCK / 64];
  __shared__ __align__(16) uint32_t gdelta[SCATTER ? (1 << TS_MAX_PBITS) + 4 : 1]; // output row of a run's first element - its sorted position
  __shared__ __align__(16) uint16_t pid[SCATTER ? TS_TILE : 1];                    // partition of the row at sorted position j
  const uint32_t np = 1u << pbits, nh = np + 1; // partition np holds the rows with a null key
  const uint32_t tid = threadIdx.x;
  uint32_t tile = blockIdx.x;
  if (SCATTER) {
    // XCD-aware tile order (speed only: every tile is still taken exactly once).  Workgroups b, b + 8, ... share an XCD (observed
    // round-robin dispatch): they take CONSECUTIVE tiles, so the runs that neighbouring tiles add to a partition -- adjacent in
    // the output, 17 rows = 68 bytes each at 490 partitions -- meet in ONE L2 within microseconds and leave it as whole lines
    // instead of as partial lines from eight L2s.
    const uint32_t per = (n_tiles + 7u) / 8u;
    tile = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (blockIdx.x / 8u >= per || tile >= n_tiles) return;
  }
  for (uint32_t p = tid; p < nh; p += TS_BLOCK) hist[p] = 0;
  lds_barrier();
  const int64_t r0 = (int64_t)tile * TS_TILE;
  uint32_t nt = (uint32_t)((P.n_rows - r0) < TS_TILE ? (P.n_rows - r0) : TS_TILE);
  uint32_t l1p = 0; // level 2: the level-1 partition of this tile
  if (!SCATTER && L1.part_start) { // level 2: the tile lies inside one tile-aligned level-1 partition; rows behind the partition's end are padding
    uint32_t lo = 0, hi = L1.np1; // largest p with part_start[p] <= r0
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)L1.part_start[mid] <= r0) lo = mid; else hi = mid; }
    l1p = lo;
    const int64_t left = (int64_t)L1.part_start[lo] + (int64_t)L1.part_rows[lo] - r0;
    nt = left <= 0 ? 0u : (left < TS_TILE ? (uint32_t)left : (uint32_t)TS_TILE);
  }
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
  // the first value columns are requested now and land under the histogram / scan / key phases
  constexpr int NPRE_ = SCATTER ? 1 : TS_NPRE; // (the scatter variant keeps its destinations in registers: one column ahead is what fits without spills)
  uint64_t pre[NPRE_ ? NPRE_ : 1][TS_R];
#pragma unroll
  for (int c = 0; c < NPRE_; c++)
    if (c < P.n_vals) load_raw<TS_R, false>(P.vals[c].values, wbytes(P.vals[c].dtype), row, pre[c]);
#pragma unroll
  for (int r = 0; r < TS_R; r++) {
    pos[r] = 0;
    if (!((live >> r) & 1)) continue;
    const uint32_t p = ((nullm >> r) & 1) ? np
                       : (SCATTER ? (L1.range_gpr ? range_part((uint32_t)key[r] - L1.range_bias, L1.range_gpr, L1.range_recip, L1.range_np)
                                                  : part_of(mix_a((uint64_t)key[r]), pbits))
                                  : part_of(mix_key(key[r]) << L1.pshift, pbits));
    pos[r] = p | (atomicAdd(&hist[p], 1u) << 12);
  }
  lds_barrier();
  const uint32_t mine = tid < nh ? hist[tid] : 0;
  if (!SCATTER && L1.hot_rows && tid < nh && mine >= L1.hot_run) { // a heavy hitter's leaf (level 2: no null partition)
    atomicAdd(&L1.hot_rows[l1p * np + tid], mine);
    if (L1.hot_any) atomicOr(L1.hot_any, 1u);
  }
  const uint32_t excl = block_scan_excl_lds(mine, tmp);
  if (tid < nh) {
    hist[tid] = excl;
    if (SCATTER) gdelta[tid] = tid < np ? L1.part_start[tid] + offs[(uint64_t)tid * n_tiles + tile] - excl : 0u;
    else offs[(uint64_t)tid * n_tiles + tile] = excl;
  }
  if (!SCATTER && tid == 0) offs[(uint64_t)(np + 1) * n_tiles + tile] = nt;
  lds_barrier();
  uint32_t dest[TS_R]; // SCATTER: output row of sorted element i * TS_BLOCK + tid
  if (SCATTER && (L1.range_gpr || L1.out_rows)) {
    // which partition's run holds sorted position j: every row writes its partition at its sorted position (the binary search over
    // the run starts below -- nine dependent, bank-conflicting LDS reads per element -- was a fifth of the kernel at 491 range
    // partitions: q3 at 3e8 rows 4.42 -> 3.97 ms).  Hash partitions take this placement when the host states the output's row
    // capacity: the destination is then clamped to it, so whatever the placement computes stays inside the output columns.
#pragma unroll
    for (int r = 0; r < TS_R; r++)
      if ((live >> r) & 1) {
        const uint32_t p = pos[r] & 0xFFFu;
        pos[r] = hist[p] + (pos[r] >> 12);
        pid[pos[r]] = (uint16_t)(p < np ? p : np - 1u);
      }
    lds_barrier();
    const uint32_t last_row = L1.out_rows ? L1.out_rows - 1u : 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < TS_R; i++) {
      const uint32_t j = (uint32_t)i * TS_BLOCK + tid;
      const uint32_t d = j < nt ? j + gdelta[pid[j] & ((1u << TS_MAX_PBITS) - 1u)] : 0u;
      dest[i] = d < last_row ? d : last_row;
    }
  } else {
#pragma unroll
    for (int r = 0; r < TS_R; r++) if ((live >> r) & 1) pos[r] = hist[pos[r] & 0xFFFu] + (pos[r] >> 12);
    if (SCATTER) {
#pragma unroll
      for (int i = 0; i < TS_R; i++) {
        const uint32_t j = (uint32_t)i * TS_BLOCK + tid;
        uint32_t lo = 0, hi = np; // the partition whose run holds sorted position j: last p with hist[p] <= j
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (hist[mid] <= j) lo = mid; else hi = mid; }
        dest[i] = j + gdelta[lo];
      }
    }
  }
#define TS_STORE(T, x, dstcol)                                                                              \
  do {                                                                                                      \
    if (SCATTER) stage_and_scatter<T>(fsm, x, pos, live, dstcol, dest, nt);                                 \
    else stage_and_store<T>(fsm, x, pos, live, (dstcol) + (size_t)r0 * sizeof(T), nt);                      \
  } while (0)

  TS_STORE(KS, key, O.keys);
  if (SCATTER) {
    // one workgroup per CU here (the destinations live in registers), so nothing else hides a column's load: column v + 1 is
    // requested before column v is staged and scattered, and lands under it
    uint64_t nxt[TS_R];
#pragma unroll
    for (int r = 0; r < TS_R; r++) nxt[r] = pre[0][r];
    for (int v = 0; v < P.n_vals; v++) {
      const ValSpec V = P.vals[v];
      uint64_t cur[TS_R];
#pragma unroll
      for (int r = 0; r < TS_R; r++) cur[r] = nxt[r];
      if (v + 1 < P.n_vals) load_raw<TS_R, false>(P.vals[v + 1].values, wbytes(P.vals[v + 1].dtype), row, nxt);
      switch (wbytes(V.dtype)) {
#define SCAT_COL(T)                                                                                     \
  {                                                                                                     \
    T x[TS_R];                                                                                          \
    _Pragma("unroll") for (int r = 0; r < TS_R; r++) x[r] = (T)cur[r];                                  \
    TS_STORE(T, x, O.vals[v]);                                                                          \
  }                                                                                                     \
  break;
        case 1: SCAT_COL(uint8_t)
        case 2: SCAT_COL(uint16_t)
        case 4: SCAT_COL(uint32_t)
        default: SCAT_COL(uint64_t)
#undef SCAT_COL
      }
      if (O.valid[v]) {
        uint8_t x[TS_R];
#pragma unroll
        for (int r = 0; r < TS_R; r++) x[r] = val_valid(V, (int64_t)row[r]) ? 1 : 0;
        TS_STORE(uint8_t, x, O.valid[v]);
      }
    }
    return;
  }
  for (int v = 0; v < P.n_vals; v++) {
    const ValSpec V = P.vals[v];
    if (v < NPRE_) {
      uint64_t raw[TS_R];
#pragma unroll
      for (int c = 0; c < NPRE_; c++)
        if (v == c) {
#pragma unroll
          for (int r = 0; r < TS_R; r++) raw[r] = pre[c][r];
        }
      switch (wbytes(V.dtype)) {
#define SORT_PRE(T)                                                                                     \
  {                                                                                                     \
    T x[TS_R];                                                                                          \
    _Pragma("unroll") for (int r = 0; r < TS_R; r++) x[r] = (T)raw[r];                                  \
    TS_STORE(T, x, O.vals[v]);                                                                          \
  }                                                                                                     \
  break;
        case 1: SORT_PRE(uint8_t)
        case 2: SORT_PRE(uint16_t)
        case 4: SORT_PRE(uint32_t)
        default: SORT_PRE(uint64_t)
#undef SORT_PRE
      }
    } else {
      switch (wbytes(V.dtype)) {
#define SORT_COL(T)                                                                                     \
  {                                                                                                     \
    T x[TS_R];                                                                                          \
    _Pragma("unroll") for (int r = 0; r < TS_R; r++) x[r] = ((const T *)V.values)[row[r]];              \
    TS_STORE(T, x, O.vals[v]);                                                                          \
  }                                                                                                     \
  break;
        case 1: SORT_COL(uint8_t)
        case 2: SORT_COL(uint16_t)
        case 4: SORT_COL(uint32_t)
        default: SORT_COL(uint64_t)
#undef SORT_COL
      }
    }
    if (O.valid[v]) {
      uint8_t x[TS_R];
#pragma unroll
      for (int r = 0; r < TS_R; r++) x[r] = val_valid(V, (int64_t)row[r]) ? 1 : 0;
      TS_STORE(uint8_t, x, O.valid[v]);
    }
  }
#undef TS_STORE
}

// ---------------------------------------------------------------------------------------------------------
// many groups, pass 2: one (partition group, tile segment) per workgroup.  A wave takes RUNS_B tiles at a
// time; batch element b of every lane is a row of tile b's run, so all loads of RUNS_B runs are in flight
// together.  With one segment per group the workgroup writes the output columns itself.
// ---------------------------------------------------------------------------------------------------------
#define RUNS_NPRE 3
// one compact dump (dump_table) -> the workgroup's LDS table, by threads tid of nthreads (a wave per dump: several dumps in flight)
template <typename KS>
DEVINL void merge_dump(const Plan &P, uint8_t *sm, uint32_t lbits, uint32_t pshift, uint32_t *used, const DumpRef &D, uint32_t item, uint32_t tid,
                       uint32_t nthreads) {
  const uint32_t lcap = 1u << lbits, lslots = lcap + FP_EXTRA_SLOTS;
  KS *lkeys = (KS *)sm;
  const DumpHdr *H = &D.hdr[item];
  const uint32_t nh = H->n_hash, hs = H->has_sent, hn = H->has_null, n = nh + hs + hn;
  const uint8_t *base = D.base + (uint64_t)item * D.item_stride;
  constexpr int U = 4; // entries in flight per thread: the loads of a batch are independent
  for (uint32_t k0 = tid; k0 < n; k0 += nthreads * U) {
    uint32_t k[U], slot[U];
    KS key[U];
    bool lv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      k[u] = k0 + (uint32_t)u * nthreads;
      lv[u] = k[u] < n;
      key[u] = lv[u] && k[u] < nh ? ((const KS *)base)[k[u]] : (KS)0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      slot[u] = 0;
      if (!lv[u]) continue;
      if (k[u] < nh) slot[u] = ltab_slot<KS>(lkeys, lbits, pshift, used, key[u]);
      else if (k[u] == nh && hs) { slot[u] = lcap; used[0] = 1; }
      else { slot[u] = lcap + 1; used[1] = 1; }
      if (slot[u] == ~0u) { lv[u] = false; slot[u] = 0; } // (table full: used[2] is set, the query is redone)
    }
    uint32_t loff = al8(lslots * (uint32_t)sizeof(KS));
    uint64_t goff = al16((uint64_t)D.cap * sizeof(KS));
    for (int a = 0; a < P.n_accs; a++) {
      const AccSpec Ac = P.accs[a];
      uint64_t part[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        part[u] = !lv[u] ? Ac.init : (Ac.width == 8 ? ((const uint64_t *)(base + goff))[k[u]] : (uint64_t)((const uint32_t *)(base + goff))[k[u]]);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (!lv[u] || part[u] == Ac.init) continue; // identity: nothing to fold
        uint8_t *p = sm + loff + slot[u] * Ac.width;
        switch (Ac.op) {
          case ACC_ADD_U64: atomicAdd((unsigned long long *)p, (unsigned long long)part[u]); break;
          case ACC_ADD_F64: atomicAdd((double *)p, __longlong_as_double((long long)part[u])); break;
          case ACC_MIN_I64: atomicMin((long long *)p, (long long)part[u]); break;
          case ACC_MAX_I64: atomicMax((long long *)p, (long long)part[u]); break;
          case ACC_MIN_U64: atomicMin((unsigned long long *)p, (unsigned long long)part[u]); break;
          case ACC_MAX_U64: atomicMax((unsigned long long *)p, (unsigned long long)part[u]); break;
          case ACC_CNT_ALL: case ACC_CNT_VALID: atomicAdd((unsigned int *)p, (unsigned int)part[u]); break;
          case ACC_FIRST_ROW: atomicMin((unsigned int *)p, (unsigned int)part[u]); break;
        }
      }
      loff += al8(lslots * Ac.width);
      goff += al16((uint64_t)D.cap * Ac.width);
    }
  }
}

// One workgroup: the leaves with enough rows in long runs become HotState entries (in leaf order: thread t owns a
// contiguous range of leaves, two block scans place its entries), hot_rows[leaf] becomes entry index + 1 or 0.
__global__ void __launch_bounds__(1024) k_hot_plan(uint32_t *__restrict__ hot_rows, uint32_t n_leaves, uint32_t n_pg, const uint32_t *__restrict__ seg_start,
                                                   uint32_t min_rows, uint32_t chunk_rows, uint32_t max_extra, HotState *__restrict__ H) {
  __shared__ __align__(16) uint32_t tmp[16];
  const uint32_t per = (n_leaves + blockDim.x - 1) / blockDim.x;
  const uint32_t l0 = threadIdx.x * per < n_leaves ? threadIdx.x * per : n_leaves, l1 = l0 + per < n_leaves ? l0 + per : n_leaves;
  auto chunks_of = [&](uint32_t leaf, uint32_t rows) -> uint32_t {
    if (rows < min_rows) return 1u;
    const uint32_t seg = leaf / n_pg, tiles = (seg_start[seg + 1] - seg_start[seg]) / TS_TILE;
    uint32_t c = (rows + chunk_rows - 1) / chunk_rows;
    c = c < HOT_MAX_CHUNKS ? c : HOT_MAX_CHUNKS;
    return c < tiles ? c : tiles;
  };
  // too many hot leaves for the entry table or the extra workgroups: the thresholds double until the largest ones fit
  __shared__ uint32_t s_nh, s_ne, s_th, s_te;
  if (threadIdx.x == 0) { s_nh = 0; s_ne = 0; }
  uint32_t nh = 0, ne = 0;
  for (int round = 0; round < 12; round++) {
    if (threadIdx.x == 0) { s_th = 0; s_te = 0; }
    __syncthreads();
    nh = 0; ne = 0;
    for (uint32_t l = l0; l < l1; l++) {
      const uint32_t c = chunks_of(l, hot_rows[l]);
      if (c > 1) { nh++; ne += c - 1; }
    }
    if (nh) { atomicAdd(&s_th, nh); atomicAdd(&s_te, ne); }
    __syncthreads();
    const bool fits = s_th <= HOT_MAX_LEAVES && s_te <= max_extra;
    __syncthreads();
    if (fits) break;
    min_rows *= 2; chunk_rows *= 2;
  }
  uint32_t h = block_scan_excl(nh, tmp);
  __syncthreads();
  uint32_t e = block_scan_excl(ne, tmp);
  // h, e run over ALL hot leaves in leaf order; both are monotone, so the leaves inside both limits are a prefix and
  // the accepted entries are dense in H->e[] (past the limits a leaf stays with its one workgroup)
  for (uint32_t l = l0; l < l1; l++) {
    const uint32_t c = chunks_of(l, hot_rows[l]);
    uint32_t mark = 0;
    if (c > 1) {
      if (h < HOT_MAX_LEAVES && e + (c - 1) <= max_extra) {
        H->e[h] = HotEntry{l, c, e, 0u};
        mark = h + 1;
        atomicMax(&s_nh, h + 1);
        atomicMax(&s_ne, e + (c - 1));
      }
      h++; e += c - 1;
    }
    hot_rows[l] = mark;
  }
  __syncthreads();
  if (threadIdx.x == 0) { H->n_hot = s_nh; H->n_extra = s_ne; H->pad[0] = 0; H->pad[1] = 0; }
}

// The group count reaches the host without a device-to-host copy: every workgroup of the query's last kernel
// counts itself done (after marking a table overflow, fenced: rare); the last one writes the flags the host needs into pinned
// host memory as ONE 64-bit word (seq << 32 | overflow << 31 | groups) that the host is spinning on.  Only atomics carry the values (they
// execute memory-side, no fence needed); the host still waits for the kernel's end before it hands the result
// buffers on, so the output stores need no fence here either.
DEVINL void publish_done(uint32_t *flags, const uint32_t *sh, uint32_t *host_sig, uint32_t seq) {
  if (!host_sig) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh[2]) { atomicOr(&flags[FLAG_OVERFLOW], 1u); __threadfence(); } // (rare) the mark is in place before the count
    const uint32_t old = atomicAdd(&flags[FLAG_DONE], 1u);
    if (old == gridDim.x - 1) {
      const uint32_t o = atomicAdd(&flags[FLAG_OVERFLOW], 0u); // this kernel's workgroups and earlier kernels'
      const uint32_t g = atomicAdd(&flags[FLAG_NGROUPS], 0u);
      // every other workgroup has made its last flag access: a clean run leaves the block zeroed for the next query (api.hip
      // skips its memset then); after an overflow the host zeroes it
      if (!o) { flags[FLAG_NGROUPS] = 0u; flags[FLAG_DONE] = 0u; }
      // one 64-bit store carries everything: no ordering between host words to arrange
      __hip_atomic_store((uint64_t *)host_sig, ((uint64_t)seq << 32) | (o ? 0x80000000u : 0u) | (g & 0x7FFFFFFFu), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <typename KS, typename SH>
__global__ void __launch_bounds__(1024) k_fold_runs(Plan P, FoldRunsArgs A, OutPlan O) { // ~3.7 KB of kernel arguments: no plan upload per query
  __shared__ __align__(16) uint32_t sh[8];
  __shared__ __align__(16) uint32_t bk[136];
  const uint32_t lslots = lslots_of(A.lbits);
  const uint32_t hot_any = A.self_hot ? A.self_hot[-2] : 0u; // (requested before the table is initialised, needed after it)
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  // (an extra workgroup of the heavy-hitter split initialises its table only once it knows it has work)
  const bool extra_wg = A.hot && blockIdx.x >= A.n_pg * A.S;
  if (A.skip && !A.self_hot && blockIdx.x < A.n_pg * A.S) { // two-level pipeline: the leaf may have been folded by k_leaf_sort already
    uint32_t it = blockIdx.x;
    const uint32_t n8s = (A.n_pg * A.S) & ~7u;
    if (it < n8s) it = (it & 7u) * (n8s >> 3) + (it >> 3);
    if (A.skip[it]) { __syncthreads(); publish_done(A.flags, sh, A.host_sig, A.seq); return; }
  }
  if (!extra_wg) ltab_init<KS>(P, fsm, lslots);
  __syncthreads();
  // Neighbouring partitions share the cache lines at their run boundaries.  Workgroups b, b + 8, b + 16, ...
  // are observed to land on one XCD, so they are given neighbouring partitions: the second reader of a line
  // then hits in that XCD's L2 instead of fetching it over the fabric again (speed only, never correctness).
  const bool self = A.self_hot != nullptr;
  const uint32_t n_main = (A.hot || self) ? A.n_pg * A.S : gridDim.x;
  // single level: A.self_hot[-2] != 0 says that some partition has long runs at all; then a workgroup that has finished its own
  // partition takes chunks of the split ones from a queue (A.self_hot[-1]) until none is left -- no extra workgroups,
  // nothing but one load when the keys are not skewed
  uint32_t vitem = blockIdx.x, over_acc = 0;
  for (;;) {
  uint32_t item = vitem;
  uint32_t chunk = 0, chunks = 1, hidx = 0, hslot = 0; // a split leaf (HotState): this workgroup folds chunk `chunk` of `chunks`
  uint32_t slot0 = 0;                                  // ... and chunk 0's dump item (chunk c >= 1: hslot + c)
  if (self) { // single level: the workgroups plan for themselves (fast.hpp)
    __shared__ __align__(16) uint32_t scan_tmp[16];
    uint32_t chunk_rows = A.hot_chunk_rows, min_rows = A.hot_min_rows;
    auto chunks_of = [&](uint32_t rows) -> uint32_t {
      if (rows < min_rows) return 1u;
      uint32_t c = (rows + chunk_rows - 1) / chunk_rows;
      c = c < HOT_MAX_CHUNKS ? c : HOT_MAX_CHUNKS;
      return c < A.n_tiles ? c : A.n_tiles;
    };
    bool extra = item >= n_main;
    if (!extra) {
      const uint32_t n8 = n_main & ~7u;
      if (item < n8) item = (item & 7u) * (n8 >> 3) + (item >> 3);
      if (hot_any) chunks = chunks_of(A.self_hot[item]);
    }
    if (extra || chunks > 1) { // (uniform over the workgroup)
      const uint32_t t = threadIdx.x;
      const uint32_t rows_t = t < n_main ? A.self_hot[t] : 0u;
      uint32_t x = 0, excl = 0;
      for (int round = 0; round < 8; round++) { // more chunks than extra workgroups: the thresholds double (the same in every workgroup)
        x = chunks_of(rows_t) - 1u;
        excl = block_scan_excl(x, scan_tmp);
        if (t == blockDim.x - 1) sh[3] = excl + x;
        __syncthreads();
        const bool fits = sh[3] <= A.max_extra;
        __syncthreads();
        if (fits) break;
        chunk_rows *= 2; min_rows *= 2;
      }
      if (!extra) chunks = chunks_of(A.self_hot[item]);
      const bool acc = x > 0 && excl + x <= A.max_extra; // (a prefix, if eight doublings were not enough: the rest keep their one workgroup)
      if (!extra) { if (t == item) { sh[5] = excl; sh[6] = acc ? 1u : 0u; } }
      else { const uint32_t e = item - n_main; if (acc && excl <= e && e < excl + x) { sh[5] = excl; sh[6] = 1u; sh[7] = t; sh[3] = x + 1u; } }
      __syncthreads();
      if (extra) {
        if (!sh[6]) break; // the queue is empty
        chunks = sh[3]; chunk = item - n_main - sh[5] + 1u; item = sh[7];
      } else if (!sh[6]) chunks = 1;
      hslot = sh[5] - 1u; // chunk c >= 1 -> item first_extra + c - 1
      slot0 = A.max_extra + item;
      hidx = item;
      __syncthreads();
      if (threadIdx.x < 8 && threadIdx.x >= 3) sh[threadIdx.x] = 0;
      if (extra) ltab_init<KS>(P, fsm, lslots);
      __syncthreads();
    }
  } else if (item < n_main) {
    const uint32_t n8 = n_main & ~7u;
    if (item < n8) item = (item & 7u) * (n8 >> 3) + (item >> 3);
    if (A.hot) {
      const uint32_t h1 = A.hot_of_leaf[item];
      if (h1 && h1 <= HOT_MAX_LEAVES) { hidx = h1 - 1; chunks = A.hot->e[hidx].chunks; hslot = A.hot->e[hidx].first_extra + hidx; }
    }
  } else { // one of the extra workgroups: chunk >= 1 of a split leaf, or nothing to do
    const uint32_t e = item - n_main;
    const uint32_t n_hot = A.hot->n_hot;
    if (e >= A.hot->n_extra || n_hot == 0 || n_hot > HOT_MAX_LEAVES) { publish_done(A.flags, sh, A.host_sig, A.seq); return; }
    uint32_t lo = 0, hi = n_hot; // the last entry with first_extra <= e
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (A.hot->e[mid].first_extra <= e) lo = mid; else hi = mid; }
    hidx = lo; chunks = A.hot->e[lo].chunks; chunk = e - A.hot->e[lo].first_extra + 1; hslot = A.hot->e[lo].first_extra + lo;
    item = A.hot->e[lo].leaf;
    ltab_init<KS>(P, fsm, lslots);
    __syncthreads();
  }
  if (!self) slot0 = hslot;
  // two-level pipeline: neighbouring sub-partitions of ONE level-1 partition (their runs are neighbours in memory) are neighbours here
  const uint32_t pg = A.disjoint ? item % A.n_pg : item / A.S, seg = A.disjoint ? item / A.n_pg : item % A.S;
  const bool null_pg = A.has_null_pg && pg == A.n_pg - 1;
  const uint32_t p0 = null_pg ? A.np : pg * A.w, p1 = null_pg ? A.np + 1 : p0 + A.w;
  const uint32_t per = (A.n_tiles + A.S - 1) / A.S;
  uint32_t t0 = seg * per, t1 = t0 + per < A.n_tiles ? t0 + per : A.n_tiles;
  if (A.seg_start) { t0 = A.seg_start[seg] / TS_TILE; t1 = A.seg_start[seg + 1] / TS_TILE; }
  if (chunks > 1) {
    const uint32_t per_c = (t1 - t0 + chunks - 1) / chunks;
    t0 = t0 + chunk * per_c < t1 ? t0 + chunk * per_c : t1;
    t1 = t0 + per_c < t1 ? t0 + per_c : t1;
  }
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t *o0 = A.offs + (uint64_t)p0 * A.n_tiles, *o1 = A.offs + (uint64_t)p1 * A.n_tiles;
  constexpr int RUNS_B = SH::STATIC ? 8 : 4; // tiles per wave batch
  const int keymode = sizeof(KS) == 4 ? 1 : 2; // the sorted copies hold packed keys of sizeof(KS) bytes
  // lanes 0..B-1 hold the run bounds of the wave's next batch of tiles (fetched one batch ahead)
  // A chunk of a split leaf has few tiles with long runs: a wave then takes ONE tile per batch and cuts its run into
  // RUNS_B segments (the batch elements), so that all waves work and a 2 000-row run takes 4 trips instead of 32.
  const bool cut = chunks > 1 && (t1 - t0) * 2u <= nw * (uint32_t)RUNS_B;
  // Otherwise a wave takes the runs of 64 tiles at once (lane l: the bounds of tile t + l) and lays their rows out FLAT:
  // batch element f of the concatenated runs goes to lane f % 64, whatever run it belongs to.  Runs are short when there are
  // many partitions (8192-row tiles over 512 partitions: 16 rows), and a lane-per-row-of-ONE-run mapping would leave
  // three lanes of four idle on every load, probe and atomic.
  const uint32_t TB = cut ? 1u : 64u; // tiles per wave batch
  uint32_t t = t0 + wave * TB;
  uint32_t st = 0, len = 0;
  if (lane < TB && t + lane < t1) { st = o0[t + lane]; len = o1[t + lane] - st; }
  while (t < t1) {
    const uint32_t tn = t + nw * TB;
    uint32_t stn = 0, lenn = 0;
    if (lane < TB && tn + lane < t1) { stn = o0[tn + lane]; lenn = o1[tn + lane] - stn; }
    if (cut) {
      uint32_t s_[RUNS_B], n_[RUNS_B], maxlen = 0;
      const uint32_t st0 = (uint32_t)__builtin_amdgcn_readlane((int)st, 0) + t * (uint32_t)TS_TILE;
      const uint32_t len0 = (uint32_t)__builtin_amdgcn_readlane((int)len, 0), sg = (len0 + RUNS_B - 1) / RUNS_B;
#pragma unroll
      for (int b = 0; b < RUNS_B; b++) {
        const uint32_t off = (uint32_t)b * sg;
        s_[b] = st0 + (off < len0 ? off : 0u);
        n_[b] = off < len0 ? (len0 - off < sg ? len0 - off : sg) : 0u;
        maxlen = n_[b] > maxlen ? n_[b] : maxlen;
      }
      // dead lanes still look their (clamped) row's key up: an empty segment must not point them at another
      // partition's rows, so it borrows the start of a segment that has rows
      uint32_t s_any = s_[0];
#pragma unroll
      for (int b = 0; b < RUNS_B; b++) if (n_[b]) s_any = s_[b];
#pragma unroll
      for (int b = 0; b < RUNS_B; b++) if (!n_[b]) s_[b] = s_any;
      for (uint32_t i0 = 0; i0 < maxlen; i0 += 64) {
        uint32_t row[RUNS_B];
        uint32_t live = 0;
        const uint32_t i = i0 + lane;
#pragma unroll
        for (int b = 0; b < RUNS_B; b++) {
          if (i < n_[b]) live |= 1u << b;
          row[b] = s_[b] + (i < n_[b] ? i : 0); // dead lanes read the segment's first row
        }
        fold_batch<KS, RUNS_B, false, RUNS_NPRE, SH>(P, fsm, A.lbits, sh, row, live, keymode, A.pkeys, 0ull, null_pg, A.pbits);
      }
    } else {
      const uint32_t astart = st + (t + lane) * (uint32_t)TS_TILE; // first row of lane's run in the sorted copy
      uint32_t incl = len;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= (uint32_t)off) incl += up;
      }
      const uint32_t excl = incl - len;
      const uint32_t total = (uint32_t)__shfl((int)incl, 63);
      // the run that holds flat element f: r = the number of runs that end at or before f (f < total: r <= 63)
      auto run_of = [&](uint32_t f) -> uint32_t {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {
          const uint32_t v = (uint32_t)__shfl((int)incl, (int)(lo + step - 1));
          if (v <= f) lo += step;
        }
        return lo;
      };
      if (total) {
        // dead lanes still look a row's key up: they take the batch's first row (a row of THIS partition)
        const uint32_t s_any = (uint32_t)__shfl((int)astart, (int)run_of(0u));
        for (uint32_t i0 = 0; i0 < total; i0 += 64u * RUNS_B) {
          uint32_t row[RUNS_B];
          uint32_t live = 0;
#pragma unroll
          for (int b = 0; b < RUNS_B; b++) {
            const uint32_t f = i0 + (uint32_t)b * 64u + lane;
            const uint32_t r = run_of(f) & 63u;
            const uint32_t rs = (uint32_t)__shfl((int)astart, (int)r), re = (uint32_t)__shfl((int)excl, (int)r);
            const bool lv = f < total;
            if (lv) live |= 1u << b;
            row[b] = lv ? rs + (f - re) : s_any;
          }
          fold_batch<KS, RUNS_B, false, RUNS_NPRE, SH>(P, fsm, A.lbits, sh, row, live, keymode, A.pkeys, 0ull, null_pg, A.pbits);
        }
      }
    }
    t = tn; st = stn; len = lenn;
  }
  __syncthreads();
  if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
  uint16_t *list = (uint16_t *)(fsm + table_lds_bytes(P, lslots, sizeof(KS)));
  if (A.S == 1 || A.disjoint) {
    bool emit = true;
    if (chunks > 1) { // a split leaf: every chunk dumps its table; the workgroup that finishes last merges them
      dump_table<KS>(P, fsm, A.lbits, 0, sh, bk, list, A.D, chunk ? hslot + chunk : slot0);
      __threadfence(); // the dump is visible device-wide before this chunk counts as done
      __syncthreads();
      if (threadIdx.x == 0) sh[4] = atomicAdd(self ? &A.self_hot[n_main + hidx] : &A.hot->e[hidx].done, 1u);
      __syncthreads();
      emit = sh[4] == chunks - 1;
      if (emit) {
        __threadfence(); // the other chunks' dumps, not stale lines of this XCD's L2
        const uint32_t over = sh[2];
        __syncthreads();
        if (threadIdx.x < 8) sh[threadIdx.x] = threadIdx.x == 2 ? over : 0u;
        ltab_init<KS>(P, fsm, lslots);
        __syncthreads();
        for (uint32_t c = wave; c < chunks; c += nw) merge_dump<KS>(P, fsm, A.lbits, A.pbits, sh, A.D, c ? hslot + c : slot0, lane, 64u); // a wave per dump
        __syncthreads();
        if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
      }
    }
    if (emit) emit_table<KS>(P, fsm, A.lbits, sh, bk, list, O, A.flags, A.out_cap);
    if (!hot_any) { publish_done(A.flags, sh, A.host_sig, A.seq); return; }
  } else { dump_table<KS>(P, fsm, A.lbits, 0, sh, bk, list, A.D, item); return; }
  // the next chunk from the queue (self mode with skewed keys only)
  __syncthreads();
  over_acc |= sh[2];
  if (threadIdx.x == 0) sh[3] = atomicAdd(&A.self_hot[-1], 1u);
  __syncthreads();
  vitem = n_main + sh[3];
  __syncthreads();
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  __syncthreads();
  }
  if (threadIdx.x == 0 && over_acc) sh[2] = 1;
  publish_done(A.flags, sh, A.host_sig, A.seq);
}

// ---------------------------------------------------------------------------------------------------------
// merge the dumps of one partition group (split over 2^mbits workgroups by key hash) and write the outputs
// ---------------------------------------------------------------------------------------------------------
#define MG_MAX_S 2048
#define MG_U 4
template <typename KS>
__global__ void __launch_bounds__(1024) k_merge_emit(Plan P, MergeArgs A, OutPlan O) {
  __shared__ __align__(16) uint32_t sh[8];
  __shared__ __align__(16) uint32_t bk[136];
  __shared__ __align__(16) uint32_t tmp[16];
  __shared__ __align__(16) uint32_t pre[MG_MAX_S + 1]; // flat index of each dump's first entry of this bucket
  __shared__ __align__(16) uint32_t bst[MG_MAX_S];     // entry index where the bucket starts inside the dump
  __shared__ uint32_t loff[PLHIP_MAX_ACCS];
  const uint32_t lcap = 1u << A.lbits, lslots = lcap + FP_EXTRA_SLOTS;
  KS *lkeys = (KS *)fsm;
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    uint32_t off = al8(lslots * (uint32_t)sizeof(KS));
    for (int a = 0; a < P.n_accs; a++) { loff[a] = off; off += al8(lslots * P.accs[a].width); }
  }
  ltab_init<KS>(P, fsm, lslots);
  const uint32_t pg = blockIdx.x >> A.mbits, m = blockIdx.x & ((1u << A.mbits) - 1);
  // bucket extents of every dump of the group, then their prefix sums
  const uint32_t chunk = (A.S + blockDim.x - 1) / blockDim.x;
  const uint32_t s_lo = threadIdx.x * chunk, s_hi = s_lo + chunk < A.S ? s_lo + chunk : A.S;
  uint32_t mine = 0;
  for (uint32_t s = s_lo; s < s_hi; s++) {
    const DumpHdr *H = &A.D.hdr[pg * A.S + s];
    const uint32_t b0 = H->boff[m], b1 = H->boff[m + 1];
    bst[s] = b0;
    pre[s] = b1 - b0;
    mine += b1 - b0;
  }
  uint32_t run = block_scan_excl(mine, tmp);
  for (uint32_t s = s_lo; s < s_hi; s++) { const uint32_t c = pre[s]; pre[s] = run; run += c; }
  if (threadIdx.x == blockDim.x - 1) pre[A.S] = run;
  __syncthreads();
  const uint32_t total = pre[A.S];
  const uint64_t koff = al16((uint64_t)A.D.cap * sizeof(KS));
  for (uint32_t f0 = 0; f0 < total; f0 += blockDim.x * MG_U) {
    const uint8_t *base[MG_U];
    uint32_t k[MG_U], slot[MG_U];
    KS key[MG_U];
    bool lv[MG_U];
#pragma unroll
    for (int u = 0; u < MG_U; u++) {
      const uint32_t f = f0 + u * blockDim.x + threadIdx.x;
      lv[u] = f < total;
      uint32_t lo = 0, hi = A.S; // largest s with pre[s] <= f
      const uint32_t ff = lv[u] ? f : 0;
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= ff) lo = mid; else hi = mid; }
      base[u] = A.D.base + (uint64_t)(pg * A.S + lo) * A.D.item_stride;
      k[u] = bst[lo] + (ff - pre[lo]);
    }
#pragma unroll
    for (int u = 0; u < MG_U; u++) key[u] = lv[u] ? ((const KS *)base[u])[k[u]] : (KS)0;
#pragma unroll
    for (int u = 0; u < MG_U; u++) {
      slot[u] = 0;
      if (!lv[u]) continue;
      slot[u] = ltab_slot<KS>(lkeys, A.lbits, A.pshift, sh, key[u]);
      if (slot[u] == ~0u) { lv[u] = false; slot[u] = 0; }
    }
    uint64_t goff = koff;
    for (int a = 0; a < P.n_accs; a++) {
      const AccSpec Ac = P.accs[a];
      uint64_t part[MG_U];
#pragma unroll
      for (int u = 0; u < MG_U; u++)
        part[u] = !lv[u] ? Ac.init : (Ac.width == 8 ? ((const uint64_t *)(base[u] + goff))[k[u]] : (uint64_t)((const uint32_t *)(base[u] + goff))[k[u]]);
      goff += al16((uint64_t)A.D.cap * Ac.width);
#pragma unroll
      for (int u = 0; u < MG_U; u++) {
        if (!lv[u] || part[u] == Ac.init) continue; // identity: nothing to fold
        uint8_t *p = fsm + loff[a] + slot[u] * Ac.width;
        switch (Ac.op) {
          case ACC_ADD_U64: atomicAdd((unsigned long long *)p, (unsigned long long)part[u]); break;
          case ACC_ADD_F64: atomicAdd((double *)p, __longlong_as_double((long long)part[u])); break;
          case ACC_MIN_I64: atomicMin((long long *)p, (long long)part[u]); break;
          case ACC_MAX_I64: atomicMax((long long *)p, (long long)part[u]); break;
          case ACC_MIN_U64: atomicMin((unsigned long long *)p, (unsigned long long)part[u]); break;
          case ACC_MAX_U64: atomicMax((unsigned long long *)p, (unsigned long long)part[u]); break;
          case ACC_CNT_ALL: case ACC_CNT_VALID: atomicAdd((unsigned int *)p, (unsigned int)part[u]); break;
          case ACC_FIRST_ROW: atomicMin((unsigned int *)p, (unsigned int)part[u]); break;
        }
      }
    }
  }
  if (m == 0) { // the reserved entries (sentinel key, null key) sit behind the ordinary ones
    for (uint32_t s = threadIdx.x; s < A.S; s += blockDim.x) {
      const uint32_t item = pg * A.S + s;
      const DumpHdr *H = &A.D.hdr[item];
      const uint32_t nh = H->n_hash, hs = H->has_sent, hn = H->has_null;
      const uint8_t *base = A.D.base + (uint64_t)item * A.D.item_stride;
      for (uint32_t e = 0; e < hs + hn; e++) {
        const bool is_sent = e == 0 && hs;
        const uint32_t slot = is_sent ? lcap : lcap + 1;
        sh[is_sent ? 0 : 1] = 1;
        uint64_t goff = koff;
        for (int a = 0; a < P.n_accs; a++) {
          const AccSpec Ac = P.accs[a];
          const uint64_t part = Ac.width == 8 ? ((const uint64_t *)(base + goff))[nh + e] : (uint64_t)((const uint32_t *)(base + goff))[nh + e];
          goff += al16((uint64_t)A.D.cap * Ac.width);
          if (part == Ac.init) continue;
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
  }
  __syncthreads();
  if (sh[2] && threadIdx.x == 0) A.flags[FLAG_OVERFLOW] = 1;
  uint16_t *list = (uint16_t *)(fsm + table_lds_bytes(P, lslots, sizeof(KS)));
  emit_table<KS>(P, fsm, A.lbits, sh, bk, list, O, A.flags, A.out_cap);
  publish_done(A.flags, sh, A.host_sig, A.seq);
}

// ---------------------------------------------------------------------------------------------------------
// sampled cardinality (the analogue of estimate_unique_count, group_by_partitioned.rs:115-157): one key per
// thread from 256-row blocks spread evenly over the column; 32-bit fingerprints go into a scratch set in HBM
// (one CAS per key: ~1e5 global atomics, a few microseconds) and the new ones are counted, together with the
// keys seen exactly once (two state bits per slot), which is what tells skew from many groups.
// ---------------------------------------------------------------------------------------------------------
#define EST_BLOCK 1024
#define EST_PER 2                      // samples per thread
#define EST_CHUNK (EST_BLOCK * EST_PER) // samples per workgroup
#define EST_LSLOTS 4096u               // LDS pre-aggregation table (load <= 1/2)
__global__ void __launch_bounds__(EST_BLOCK) k_estimate(Plan P, int64_t n_sample, int64_t stride, uint32_t *set, uint32_t set_bits, uint32_t *flags) {
  // The workgroup first folds its 2048 samples into an LDS set (hash + sightings): a hot key reaches the global set
  // once per workgroup instead of once per sample (atomics and memory-side loads on a handful of addresses serialise).
  __shared__ unsigned long long lh[EST_LSLOTS]; // full hash of the key in the slot, 0 = empty
  __shared__ uint32_t lc[EST_LSLOTS];           // sightings inside this workgroup
  __shared__ int s_d, s_f1;
  for (uint32_t j = threadIdx.x; j < EST_LSLOTS; j += EST_BLOCK) { lh[j] = 0ull; lc[j] = 0u; }
  if (threadIdx.x == 0) { s_d = 0; s_f1 = 0; }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EST_PER; r++) {
    const int64_t i = (int64_t)blockIdx.x * EST_CHUNK + (int64_t)r * EST_BLOCK + threadIdx.x;
    if (i >= n_sample) continue;
    bool isn;
    const uint64_t key = row_packed_key(P, (i >> 8) * stride + (i & 255), isn);
    unsigned long long h = isn ? 0x9E3779B97F4A7C15ull : hash_key(key);
    if (h == 0ull) h = 1ull;
    uint32_t s = (uint32_t)(h >> 17) & (EST_LSLOTS - 1);
    for (;;) { // the table holds twice the samples: a free slot always turns up
      const unsigned long long k = atomicCAS(&lh[s], 0ull, h);
      if (k == 0ull || k == h) { atomicAdd(&lc[s], 1u); break; }
      s = (s + 1) & (EST_LSLOTS - 1);
    }
  }
  __syncthreads();
  const uint32_t mask = (1u << set_bits) - 1;
  uint32_t *seen = set + ((size_t)1 << set_bits); // two bits per slot of the global set: seen once, seen twice or more
  int d = 0, f1 = 0;
  for (uint32_t j = threadIdx.x; j < EST_LSLOTS; j += EST_BLOCK) {
    const unsigned long long h = lh[j];
    if (h == 0ull) continue;
    const uint32_t c = lc[j];
    uint32_t f = (uint32_t)(h >> 32);
    if (f == 0u) f = 1u; // 0 = empty slot
    uint32_t s = (uint32_t)h & mask;
    for (uint32_t it = 0; it <= mask; it++) {
      const uint32_t k = atomicCAS(&set[s], 0u, f);
      if (k == 0u || k == f) {
        if (k == 0u) d++;
        const uint32_t w = s >> 4, sh = (s & 15u) * 2u;
        const uint32_t prev = (atomicOr(&seen[w], (c >= 2 ? 3u : 1u) << sh) >> sh) & 3u;
        if (c >= 2) { if (prev == 1u) f1--; }            // was "once", is "twice or more" now
        else if (prev == 0u) f1++;                       // first sighting anywhere
        else if (prev == 1u && !((atomicOr(&seen[w], 2u << sh) >> sh) & 2u)) f1--; // second sighting
        break;
      }
      s = (s + 1) & mask;
    }
  }
  if (d) atomicAdd(&s_d, d);
  if (f1) atomicAdd(&s_f1, f1);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_d) atomicAdd(&flags[FLAG_NGROUPS], (uint32_t)s_d);
    if (s_f1) atomicAdd((int *)&flags[FLAG_F1], s_f1);
  }
}

template <typename F>
void allow_big_lds(F f) {
  // static LDS (flags, histogram, merge prefix arrays) comes on top of the dynamic part
  if (hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess) (void)hipGetLastError();
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
size_t fast_table_bytes(const Plan &p, uint32_t lbits, int key_bytes) {
  const size_t lslots = ((size_t)1 << lbits) + FP_EXTRA_SLOTS;
  size_t off = (lslots * key_bytes + 7) & ~(size_t)7;
  for (int a = 0; a < p.n_accs; a++) off += (lslots * p.accs[a].width + 7) & ~(size_t)7;
  return off + lslots * 2 + 16; // + the u16 slot list of the dump / emit step
}

uint64_t fast_dump_stride(const Plan &p, uint32_t cap, int key_bytes) {
  uint64_t off = ((uint64_t)cap * key_bytes + 15) & ~15ull;
  for (int a = 0; a < p.n_accs; a++) off += ((uint64_t)cap * p.accs[a].width + 15) & ~15ull;
  return (off + 255) & ~255ull;
}

template <typename KS, bool QUAD, typename SH>
void launch_fold_rows_t(const Plan &p, const FoldRowsArgs &a, size_t bytes, uint32_t n_items, uint32_t block, hipStream_t s) {
  static bool once = false;
  if (!once) { allow_big_lds(k_fold_rows<KS, QUAD, SH>); once = true; }
  hipLaunchKernelGGL((k_fold_rows<KS, QUAD, SH>), dim3(n_items), dim3(block), bytes, s, p, a);
}

int fast_shape_id(const Plan &p) { return find_shape(p); }

void launch_fold_rows(const Plan &p, const FoldRowsArgs &a, int key_bytes, bool quad, uint32_t n_items, uint32_t block, hipStream_t s) {
  const size_t bytes = fast_table_bytes(p, a.lbits, key_bytes);
  const int sid = quad ? find_shape(p) : -1; // static shapes are built for the aligned (vector load) case only
  switch (sid) {
#define X(i, SH)                                                                                   \
  case i:                                                                                          \
    if (key_bytes == 4) launch_fold_rows_t<uint32_t, true, SH>(p, a, bytes, n_items, block, s);    \
    else launch_fold_rows_t<uint64_t, true, SH>(p, a, bytes, n_items, block, s);                   \
    return;
    FP_SHAPES(X)
#undef X
    default: break;
  }
  if (key_bytes == 4) {
    if (quad) launch_fold_rows_t<uint32_t, true, RtShape>(p, a, bytes, n_items, block, s);
    else launch_fold_rows_t<uint32_t, false, RtShape>(p, a, bytes, n_items, block, s);
  } else {
    if (quad) launch_fold_rows_t<uint64_t, true, RtShape>(p, a, bytes, n_items, block, s);
    else launch_fold_rows_t<uint64_t, false, RtShape>(p, a, bytes, n_items, block, s);
  }
}

void launch_tile_sort(const Plan &p, const PartOut &o, uint32_t pbits, uint32_t n_tiles, uint32_t *offs, int keymode, uint64_t kxor,
                      int key_bytes, hipStream_t s, const TileSortLevel1 *lvl1, bool scatter) {
  const TileSortLevel1 l1 = lvl1 ? *lvl1 : TileSortLevel1{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u, nullptr};
  static bool once = false;
  if (!once) {
    allow_big_lds(k_tile_sort<uint32_t, false>); allow_big_lds(k_tile_sort<uint64_t, false>);
    allow_big_lds(k_tile_sort<uint32_t, true>); allow_big_lds(k_tile_sort<uint64_t, true>);
    once = true;
  }
  const size_t bytes = (size_t)TS_TILE * 8;
  if (scatter) { // level 1 of the two-level pipeline: `offs` = k_part_scan's prefixes, lvl1->part_start = the partitions' first rows
    const uint32_t grid = ((n_tiles + 7u) / 8u) * 8u; // (the kernel deals consecutive tiles to the workgroups of one XCD)
    if (key_bytes == 4) hipLaunchKernelGGL((k_tile_sort<uint32_t, true>), dim3(grid), dim3(TS_BLOCK), bytes, s, p, o, pbits, n_tiles, offs, keymode, kxor, l1);
    else hipLaunchKernelGGL((k_tile_sort<uint64_t, true>), dim3(grid), dim3(TS_BLOCK), bytes, s, p, o, pbits, n_tiles, offs, keymode, kxor, l1);
    return;
  }
  if (key_bytes == 4) hipLaunchKernelGGL((k_tile_sort<uint32_t, false>), dim3(n_tiles), dim3(TS_BLOCK), bytes, s, p, o, pbits, n_tiles, offs, keymode, kxor, l1);
  else hipLaunchKernelGGL((k_tile_sort<uint64_t, false>), dim3(n_tiles), dim3(TS_BLOCK), bytes, s, p, o, pbits, n_tiles, offs, keymode, kxor, l1);
}

template <typename KS, typename SH>
void launch_fold_runs_t(const Plan &p2, const FoldRunsArgs &a, const OutPlan &o, size_t bytes, uint32_t block, hipStream_t s) {
  static bool once = false;
  if (!once) { allow_big_lds(k_fold_runs<KS, SH>); once = true; }
  hipLaunchKernelGGL((k_fold_runs<KS, SH>), dim3(a.n_pg * a.S + (a.hot ? a.max_extra : 0u)), dim3(block), bytes, s, p2, a, o);
}

void launch_fold_runs(const Plan &p2, const FoldRunsArgs &a, const OutPlan &o, int key_bytes, uint32_t block, hipStream_t s) {
  const size_t bytes = fast_table_bytes(p2, a.lbits, key_bytes);
  switch (find_shape(p2)) {
#define X(i, SH)                                                                              \
  case i:                                                                                     \
    if (key_bytes == 4) launch_fold_runs_t<uint32_t, SH>(p2, a, o, bytes, block, s);      \
    else launch_fold_runs_t<uint64_t, SH>(p2, a, o, bytes, block, s);                     \
    return;
    FP_SHAPES(X)
#undef X
    default: break;
  }
  if (key_bytes == 4) launch_fold_runs_t<uint32_t, RtShape>(p2, a, o, bytes, block, s);
  else launch_fold_runs_t<uint64_t, RtShape>(p2, a, o, bytes, block, s);
}

void launch_merge_emit(const Plan &p, const MergeArgs &a, const OutPlan &o, int key_bytes, uint32_t n_pg, uint32_t block, hipStream_t s) {
  static bool once = false;
  if (!once) { allow_big_lds(k_merge_emit<uint32_t>); allow_big_lds(k_merge_emit<uint64_t>); once = true; }
  const size_t bytes = fast_table_bytes(p, a.lbits, key_bytes);
  const uint32_t grid = n_pg << a.mbits;
  if (key_bytes == 4) hipLaunchKernelGGL(k_merge_emit<uint32_t>, dim3(grid), dim3(block), bytes, s, p, a, o);
  else hipLaunchKernelGGL(k_merge_emit<uint64_t>, dim3(grid), dim3(block), bytes, s, p, a, o);
}

void launch_estimate(const Plan &p, int64_t n_sample, int64_t stride, uint32_t *set, uint32_t set_bits, uint32_t *flags, hipStream_t s) {
  hipLaunchKernelGGL(k_estimate, dim3((unsigned)((n_sample + EST_CHUNK - 1) / EST_CHUNK)), dim3(EST_BLOCK), 0, s, p, n_sample, stride, set, set_bits, flags);
}

void launch_hot_plan(uint32_t *hot_rows, uint32_t n_leaves, uint32_t n_pg, const uint32_t *seg_start, uint32_t min_rows, uint32_t chunk_rows,
                     uint32_t max_extra, HotState *H, hipStream_t s) {
  hipLaunchKernelGGL(k_hot_plan, dim3(1), dim3(1024), 0, s, hot_rows, n_leaves, n_pg, seg_start, min_rows, chunk_rows, max_extra, H);
}
