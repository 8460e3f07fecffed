// leaf_path.hip — the leaf kernel of the two-level pipeline for the one-row-per-group regime (BASELINE config 3: 1e8 rows,
// ~1e8 groups).  After the two partition levels (fast_path.hip: k_tile_sort scatter + k_tile_sort) a leaf holds a few
// hundred rows, nearly all with distinct keys.  A hash table per leaf is then all overhead (k_fold_runs: init, probe, scan,
// one global atomic per leaf: 8.0 of that query's 14 ms).  Here ONE WAVE takes a leaf:
//   gather the leaf's runs (keys + their positions in the partitioned copy) into the wave's 6 KB of LDS,
//   sort (key << 9 | slot) composites in registers -- bitonic over 512 elements, 8 per lane, xor-shuffles between lanes --,
//   mark the run heads, and fold every run straight from the partitioned value columns into its output row
//   (a lane per run; the first element of eight runs per lane is gathered in one round trip, longer runs walk on).
// The waves of a workgroup (LS_WAVES leaves) share ONE global atomic for their block of output rows (a returning atomic
// on one address retires at ~88 per us: one per leaf would cost 3 ms at 262 144 leaves).  Leaves that do not fit (more than
// 512 rows: a heavy hitter, skew) are left to k_fold_runs, which skips the leaves marked here.
// The reference builds these groups with a hash map per partition (crates/polars-core/src/frame/group_by/hashing.rs:380-468);
// sorting a leaf instead changes nothing observable: group order is unspecified there as well.
// Integer / f64 work bound by HBM gathers and VALU: no MFMA.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "fast.hpp"
#include "shapes.hpp"

extern __shared__ __align__(16) uint8_t lsm[];

namespace {

#define LS_WCAP 512u  // rows a wave sorts
#define LS_IDXB 9u    // log2(LS_WCAP): low bits of a composite = the row's slot in the wave's staging area
#ifndef LS_BLOCK
#define LS_BLOCK 256
#endif
#define LS_WAVES (LS_BLOCK / 64)
#define LS_WAVE_BYTES (LS_WCAP * 8u + LS_WCAP * 4u + (LS_WCAP + 8u) * 2u) // composites | positions | run starts (u16)
#define LS_MAX_ACCS 8
#define LS_NPRE 2     // value columns gathered ahead for a lane's eight runs

DEVINL void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

DEVINL uint32_t wave_incl_scan(uint32_t x, uint32_t lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(x, off);
    if (lane >= (uint32_t)off) x += t;
  }
  return x;
}

DEVINL uint64_t shfl_xor64(uint64_t v, int m) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// ascending bitonic sort of 512 composites: element i = lane * 8 + r lives in c[r] of lane `lane`
DEVINL void bitonic512(uint64_t (&c)[8], uint32_t lane) {
#pragma unroll
  for (int k = 2; k <= 512; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 8) { // partner = same register of lane ^ (j / 8)
        const bool lower = (lane & (uint32_t)(j >> 3)) == 0;
        const bool up = ((lane * 8u) & (uint32_t)k) == 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const uint64_t o = shfl_xor64(c[r], j >> 3);
          const bool take_min = lower == up;
          c[r] = take_min ? (o < c[r] ? o : c[r]) : (o > c[r] ? o : c[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
          if ((r & j) == 0) {
            // direction: (i & k) == 0 with i = lane * 8 + r
            const bool up = k >= 8 ? ((lane * 8u) & (uint32_t)k) == 0 : ((r & k) == 0);
            const uint64_t a = c[r], b = c[r | j];
            const bool sw = up ? (b < a) : (b > a);
            c[r] = sw ? b : a;
            c[r | j] = sw ? a : b;
          }
        }
      }
    }
  }
}

// st[i] for a wave-uniform i (an OutSpec field: a kernel argument): a scalar branch, not a chain of 16 v_cndmask
DEVINL uint64_t sel8(const uint64_t (&st)[LS_MAX_ACCS], int i) {
  switch (__builtin_amdgcn_readfirstlane(i)) {
    case 0: return st[0];
    case 1: return st[1];
    case 2: return st[2];
    case 3: return st[3];
    case 4: return st[4];
    case 5: return st[5];
    case 6: return st[6];
    default: return st[7];
  }
}

DEVINL void fold_state(uint32_t op, uint64_t &state, uint64_t operand) {
  switch (op) {
    case ACC_ADD_F64: state = (uint64_t)__double_as_longlong(__longlong_as_double((long long)state) + __longlong_as_double((long long)operand)); break;
    case ACC_MIN_I64: state = (long long)operand < (long long)state ? operand : state; break;
    case ACC_MAX_I64: state = (long long)operand > (long long)state ? operand : state; break;
    case ACC_MIN_U64: state = operand < state ? operand : state; break;
    case ACC_MAX_U64: state = operand > state ? operand : state; break;
    default: state += operand; break; // ADD_U64, counters
  }
}

// one row (position `pos` in the partitioned copy; raw bits of the first LS_NPRE value columns already here) into the states.
// SH: the query's shape (shapes.hpp): a static shape unrolls the accumulator list into straight-line code on registers
template <typename SH>
DEVINL void fold_row(const Plan &P, uint64_t (&st)[LS_MAX_ACCS], uint32_t pos, const uint64_t (&pre)[LS_NPRE]) {
  int cur_src = -2;
  uint64_t raw[1] = {0};
  bool vok = true;
  auto body = [&](int a) {
    const uint32_t op = SH::acc_op(P, a);
    const int src = SH::acc_src(P, a);
    if (op == ACC_CNT_ALL) { st[a] += 1; return; }
    if (src != cur_src) { // accumulators are ordered by source column
      cur_src = src;
      vok = SH::val_nullable(P, src) ? val_valid(P.vals[src], (int64_t)pos) : true;
      if (src < LS_NPRE) {
#pragma unroll
        for (int c = 0; c < LS_NPRE; c++) if (src == c) raw[0] = pre[c];
      } else {
        const uint32_t row1[1] = {pos};
        load_raw<1, false>(P.vals[src].values, wbytes(SH::val_dtype(P, src)), row1, raw);
      }
    }
    if (!vok) return;
    if (op == ACC_CNT_VALID) { st[a] += 1; return; }
    uint64_t opnd[1];
    raw_to_operands<1>(SH::val_dtype(P, src), SH::acc_conv(P, a), raw, opnd);
    fold_state(op, st[a], opnd[0]);
  };
  if constexpr (SH::STATIC) static_for<0, SH::NA>(body);
  else {
#pragma unroll
    for (int a = 0; a < LS_MAX_ACCS; a++) if (a < P.n_accs) body(a);
  }
}

template <typename KS, typename SH>
__global__ void __launch_bounds__(LS_BLOCK) k_leaf_sort(Plan P, LeafSortArgs A, OutPlan O) {
  __shared__ uint32_t s_cnt[LS_WAVES], s_pre[LS_WAVES], s_ok[LS_WAVES];
  __shared__ uint32_t s_base, s_over;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t *wk = (uint64_t *)(lsm + (size_t)wave * LS_WAVE_BYTES); // keys, then the sorted composites
  uint32_t *wg = (uint32_t *)(wk + LS_WCAP);                        // position of every staged row in the partitioned copy
  uint16_t *wr = (uint16_t *)(wg + LS_WCAP);                        // first sorted element of every run (+ sentinel)
  const uint32_t leaf = blockIdx.x * LS_WAVES + wave;
  const bool in_grid = leaf < A.n_leaves;
  // ---- the leaf's runs: one per tile of its level-1 partition ------------------------------------------------------------
  uint32_t st = 0, len = 0, nt = 0;
  if (in_grid) {
    const uint32_t pg = leaf % A.n_pg, seg = leaf / A.n_pg;
    const uint32_t t0 = A.seg_start[seg] / TS_TILE, t1 = A.seg_start[seg + 1] / TS_TILE;
    nt = t1 - t0;
    if (lane < nt && nt <= 64) {
      const uint32_t t = t0 + lane;
      st = A.offs[(uint64_t)pg * A.n_tiles + t];
      len = A.offs[(uint64_t)(pg + 1) * A.n_tiles + t] - st;
      st += t * (uint32_t)TS_TILE;
    }
  }
  const uint32_t incl = wave_incl_scan(len, lane);
  const uint32_t total = (uint32_t)__shfl((int)incl, 63);
  const uint32_t excl = incl - len;
  const bool ok = in_grid && nt <= 64 && total <= LS_WCAP; // (wave-uniform) this wave folds the leaf
  uint32_t n_runs = 0;
  uint64_t c[8];
  if (ok && total) {
    // ---- keys and positions: the runs laid out flat over the lanes (element r * 64 + lane), eight coalesced loads in flight; the
    // composites are built in registers right away -- a sort does not care where its input sits -- and only the positions go to LDS
    auto run_of = [&](uint32_t f) -> uint32_t { // the run that holds flat element f: the number of runs that end at or before f
      uint32_t lo = 0;
#pragma unroll
      for (uint32_t step = 32; step; step >>= 1) {
        const uint32_t v = (uint32_t)__shfl((int)incl, (int)(lo + step - 1));
        if (v <= f) lo += step;
      }
      return lo & 63u;
    };
    const KS *pk = (const KS *)A.pkeys;
    const uint32_t p_any = (uint32_t)__shfl((int)st, (int)run_of(0u)); // dead lanes read the leaf's first row
    uint32_t posr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint32_t e = (uint32_t)r * 64u + lane;
      const uint32_t run = run_of(e);
      const uint32_t rs = (uint32_t)__shfl((int)st, (int)run), rx = (uint32_t)__shfl((int)excl, (int)run);
      posr[r] = e < total ? rs + (e - rx) : p_any;
    }
    KS kk[8];
#pragma unroll
    for (int r = 0; r < 8; r++) kk[r] = pk[posr[r]];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint32_t e = (uint32_t)r * 64u + lane;
      if (e < total) wg[e] = posr[r];
      c[r] = e < total ? (((uint64_t)kk[r] << LS_IDXB) | (uint64_t)e) : ~0ull;
    }
    bitonic512(c, lane);
    // ---- run heads ----------------------------------------------------------------------------------------------------------
    uint64_t up_lo = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)c[7], 1), up_hi = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(c[7] >> 32), 1);
    const uint64_t before = up_lo | (up_hi << 32); // last element of the previous lane
    uint32_t heads = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint32_t e = lane * 8u + (uint32_t)r;
      const uint64_t pv = r ? c[r - 1] : before;
      const bool head = e < total && (e == 0 || (c[r] >> LS_IDXB) != (pv >> LS_IDXB));
      if (head) heads |= 1u << r;
      wk[e] = c[r];
    }
    const uint32_t hc = (uint32_t)__popc(heads);
    const uint32_t hincl = wave_incl_scan(hc, lane);
    n_runs = (uint32_t)__shfl((int)hincl, 63);
    uint32_t ri = hincl - hc;
#pragma unroll
    for (int r = 0; r < 8; r++)
      if ((heads >> r) & 1u) wr[ri++] = (uint16_t)(lane * 8u + (uint32_t)r);
    if (lane == 0) wr[n_runs] = (uint16_t)total;
    wave_lds_fence();
  }
  // ---- one block of output rows for the workgroup's sixteen leaves ------------------------------------------------------------
  if (lane == 0) {
    s_cnt[wave] = n_runs;
    s_ok[wave] = (!in_grid || ok) ? 1u : 0u;
    if (in_grid) A.handled[leaf] = ok ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0, unhandled = 0;
    for (uint32_t w = 0; w < LS_WAVES; w++) { s_pre[w] = sum; sum += s_cnt[w]; }
    for (uint32_t w = 0; w < LS_WAVES; w++) if (!s_ok[w]) unhandled++;
    const uint32_t base = sum ? atomicAdd(&A.flags[FLAG_NGROUPS], sum) : 0u;
    s_base = base;
    s_over = ((uint64_t)base + sum > A.out_cap) ? 1u : 0u;
    if (s_over) atomicOr(&A.flags[FLAG_OVERFLOW], 1u);
    if (unhandled) atomicAdd(&A.flags[11], unhandled);
  }
  __syncthreads();
  // ---- fold every run into its output row (a lane per run) -----------------------------------------------------------------
  if (ok && n_runs && !s_over) {
    const uint64_t gbase = (uint64_t)s_base + s_pre[wave];
    uint32_t rs[8], re[8], pos[8];
    uint64_t key[8], pre[LS_NPRE][8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t run = (uint32_t)u * 64u + lane;
      const bool live = run < n_runs;
      rs[u] = live ? wr[run] : 0u;
      re[u] = live ? wr[run + 1] : 0u;
      const uint64_t comp = wk[rs[u]];
      key[u] = comp >> LS_IDXB;
      pos[u] = wg[(uint32_t)comp & (LS_WCAP - 1u)];
#pragma unroll
      for (int v = 0; v < LS_NPRE; v++) {
        pre[v][u] = 0;
        if (live && v < P.n_vals) {
          const uint32_t row1[1] = {pos[u]};
          uint64_t raw1[1];
          load_raw<1, false>(P.vals[v].values, wbytes(P.vals[v].dtype), row1, raw1);
          pre[v][u] = raw1[0];
        }
      }
    }
    auto fold_run = [&](const uint32_t run, const uint32_t rs_u, const uint32_t re_u, const uint32_t pos_u, const uint64_t key_u, const uint64_t pre0,
                        const uint64_t pre1) {
      if (run >= n_runs) return;
      uint64_t stt[LS_MAX_ACCS];
#pragma unroll
      for (int a = 0; a < LS_MAX_ACCS; a++) stt[a] = a < P.n_accs ? P.accs[a].init : 0ull;
      const uint64_t p1[LS_NPRE] = {pre0, pre1};
      fold_row<SH>(P, stt, pos_u, p1);
      for (uint32_t el = rs_u + 1; el < re_u; el++) { // (rare here: more than one row in the group)
        const uint32_t pp = wg[(uint32_t)wk[el] & (LS_WCAP - 1u)];
        uint64_t p2[LS_NPRE];
#pragma unroll
        for (int v = 0; v < LS_NPRE; v++) {
          p2[v] = 0;
          if (v < P.n_vals) {
            const uint32_t row1[1] = {pp};
            uint64_t raw1[1];
            load_raw<1, false>(P.vals[v].values, wbytes(P.vals[v].dtype), row1, raw1);
            p2[v] = raw1[0];
          }
        }
        fold_row<SH>(P, stt, pp, p2);
      }
      const uint64_t g = gbase + run;
      for (int o = 0; o < O.n_outs; o++) {
        const OutSpec &S = O.outs[o];
        bool valid;
        if (S.kind == OUT_KEY) valid = finalize_key(P, S, key_u, false, false, g);
        else {
          const uint64_t acc = S.acc >= 0 ? sel8(stt, S.acc) : 0ull;
          const uint32_t cnt_all = S.cnt_all >= 0 ? (uint32_t)sel8(stt, S.cnt_all) : 0u;
          const uint32_t cnt_valid = S.cnt_valid >= 0 ? (uint32_t)sel8(stt, S.cnt_valid) : cnt_all;
          valid = finalize_value(S, acc, cnt_all, cnt_valid, g);
        }
        if (S.validity && valid) atomicOr((unsigned int *)S.validity + (g >> 5), 1u << (g & 31)); // zeroed bitmap
      }
    };
    static_assert(LS_NPRE == 2, "fold_run takes two prefetched columns");
    // ONE copy of the fold's code: the lane's eight runs pass through element 0 of the register arrays, which shift down
    // after every run (eight inlined copies needed 2 KB of scratch per lane)
#pragma unroll 1
    for (uint32_t it = 0; it < 8; it++) {
      fold_run(it * 64u + lane, rs[0], re[0], pos[0], key[0], pre[0][0], pre[1][0]);
#pragma unroll
      for (int q = 0; q < 7; q++) {
        rs[q] = rs[q + 1]; re[q] = re[q + 1]; pos[q] = pos[q + 1]; key[q] = key[q + 1];
        pre[0][q] = pre[0][q + 1]; pre[1][q] = pre[1][q + 1];
      }
    }
  }
  // ---- the last workgroup tells the host: groups so far, and how many leaves are left for k_fold_runs --------------------------
  if (A.host_sig) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t old = atomicAdd(&A.flags[10], 1u);
      if (old == gridDim.x - 1) {
        const uint32_t o = atomicAdd(&A.flags[FLAG_OVERFLOW], 0u);
        const uint32_t g = atomicAdd(&A.flags[FLAG_NGROUPS], 0u);
        const uint32_t un = atomicAdd(&A.flags[11], 0u);
        __hip_atomic_store(A.host_sig + 2, un, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // read by the host after the stream has drained
        __hip_atomic_store((uint64_t *)A.host_sig, ((uint64_t)A.seq << 32) | (o ? 0x80000000u : 0u) | (g & 0x7FFFFFFFu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

} // namespace

template <typename KS, typename SH>
static void launch_leaf_sort_t(const Plan &p2, const LeafSortArgs &a, const OutPlan &o, size_t lds, uint32_t grid, hipStream_t s) {
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void *)k_leaf_sort<KS, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess) (void)hipGetLastError();
    once = true;
  }
  hipLaunchKernelGGL((k_leaf_sort<KS, SH>), dim3(grid), dim3(LS_BLOCK), lds, s, p2, a, o);
}

void launch_leaf_sort(const Plan &p2, const LeafSortArgs &a, const OutPlan &o, int key_bytes, hipStream_t s) {
  const size_t lds = (size_t)LS_WAVES * LS_WAVE_BYTES;
  const uint32_t grid = (a.n_leaves + LS_WAVES - 1) / LS_WAVES;
  // the one-row-per-group shapes that have a static form (BASELINE config 3 and its f64-first twin); everything else interprets the plan
  const int sid = find_shape(p2);
#define LS_GO(SH)                                                                         \
  do {                                                                                    \
    if (key_bytes == 4) launch_leaf_sort_t<uint32_t, SH>(p2, a, o, lds, grid, s);         \
    else launch_leaf_sort_t<uint64_t, SH>(p2, a, o, lds, grid, s);                        \
  } while (0)
  if (sid == 3) LS_GO(ShWide);
  else if (sid == 6) LS_GO(ShWideF);
  else LS_GO(RtShape);
#undef LS_GO
}
