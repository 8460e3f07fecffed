// api.hip — the C ABI of libplhip.so (include/plhip.h): argument checking, staging of host columns,
// lowering of a query to a Plan, strategy choice, kernel launches, result frames.
//
// This is the host layer that replaces polars-lazy's GroupByExec / PartitionGroupByExec for the path
// (crates/polars-lazy/src/physical_plan/executors/group_by.rs:57-148, group_by_partitioned.rs:161-361).
// There is no CPU compute path in this file: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/plhip.h"
#include "dense.hpp"
#include "comm.hpp"
#include "fast.hpp"
#include "kernels.hpp"
#include "utf8.hpp"
#include "join.hpp"

// Scratch of plhip_group_by_agg_noalloc (flag block, dump headers, table dumps): a file-scope device array, so that the
// entry point never calls hipMalloc.  Every GPU the library runs on holds its own copy until the process ends.  In host
// code this NAME is only a stand-in for hipGetSymbolAddress (resolve_scratch below): kernels, memsets and copies get the
// resolved device pointer, never the name.
static __device__ __align__(256) unsigned char scratch[PLHIP_NOALLOC_SCRATCH_BYTES];

// ---------------------------------------------------------------------------------------------------------
// per-thread context
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Scratch { // grow-only device buffer reused across calls (hipMalloc/hipFree are slow and synchronising)
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) hipFree(p);
    p = nullptr; bytes = 0;
    size_t want = n + n / 4;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { e = hipMalloc(&p, n); want = n; }
    if (e == hipSuccess) bytes = want;
    return e;
  }
  void release() { if (p) hipFree(p); p = nullptr; bytes = 0; }
};

// A result frame may be handed to the caller while the tail of its last kernel is still retiring (wait_signal, `defer`): whoever
// touches its buffers next from outside the producing stream waits through this token.  The token outlives the thread's stream: a
// thread that ends drains its stream and clears `s`, and later waits become no-ops.
struct StreamToken {
  std::mutex mu;
  hipStream_t s = nullptr;
};
using TokenRef = std::shared_ptr<StreamToken>;
inline hipError_t settle_token(TokenRef &t) {
  hipError_t e = hipSuccess;
  if (t) {
    std::lock_guard<std::mutex> g(t->mu);
    if (t->s) e = hipStreamSynchronize(t->s);
  }
  t.reset();
  return e;
}

struct Ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  TokenRef token;             // this thread's stream, as results and pooled buffers refer to it
  bool allow_defer = false;   // set by plhip_group_by_agg_opts around a DEVICE-output query: the last wait may skip the stream's retirement
  bool deferred = false;      // ... and did: the result frame carries the token
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::string err;
  std::string strategy = "";
  float total_ms = 0.f, dom_ms = 0.f, part_ms = 0.f, agg_ms = 0.f;
  std::string dom_kernel = "";
  int forced_strategy = -1; // -1: not set by the API (env / auto)
  Scratch table, slots, sortkeys, flags, misc, rank, plans, items, part, dump, hdr, offs;
  float k_ms[3] = {0.f, 0.f, 0.f}; // fast path: tile sort, fold, merge+emit
  bool signal_seen = false;        // wait_signal: the last query's count came through the pinned signal word (not the flag block)
  int flags_clean = 0;             // 1: the 64-byte flag block is zero (re-zeroed behind every read, off the next query's critical path);
                                   // 2: zero but for FLAG_NGROUPS (k_dense_merge's last workgroup cleans up after itself: good for the next dense query)
  uint32_t *host_flags = nullptr; // pinned, 64 bytes
  // fast path: the last kernel publishes the flag block here (pinned, coherent, device-visible) and the host
  // spins on its first 64-bit word instead of queueing a device-to-host copy and waiting for the stream
  uint32_t *host_sig = nullptr, *host_sig_dev = nullptr;
  uint64_t *host_hdr = nullptr, *host_hdr_dev = nullptr; // pinned, device-mapped: the slot headers of the sharded path's exchange
  uint32_t seq = 0;
  bool hot_clean = false;  // the heavy-hitter words behind the flag block (FLAGS_HOT_OFF ..) are zero
  int timing_every = 1, timing_tick = 0; // plhip_set_timing_interval: the fast paths record their events on every Nth query (0: never)
  bool timed() { if (timing_every <= 0) return false; if (++timing_tick >= timing_every) { timing_tick = 0; return true; } return false; }
  int timing_pending = 0; // 1: direct, 2: tilesort (S == 1), 3: tilesort + merge -- event times not read yet
  Scratch xsend, xrecv, xcnt; // the sharded path's exchange buffers
  Scratch str_a, str_b;       // plhip_utf8_encode: hashes + row ids; lookup table + dictionary lengths + scan space
  plhip_comm *dense_comm = nullptr; // set by sharded_impl around its dense attempt (group_by_agg_impl reads it)
  plhip_sharded_stats_t shard_stats{};
  uint8_t *na_scratch = nullptr; // device address of this GPU's static scratch array (plhip_group_by_agg_noalloc)
  // the grow-only workspace of this thread (roughly the largest query's working set): back to the driver
  void release_workspace() {
    if (device < 0) return;
    if (hipSetDevice(device) != hipSuccess) return;
    if (stream) hipStreamSynchronize(stream);
    for (Scratch *b : {&table, &slots, &sortkeys, &flags, &misc, &rank, &plans, &items, &part, &dump, &hdr, &offs, &xsend, &xrecv, &xcnt, &str_a, &str_b}) b->release();
    flags_clean = false; hot_clean = false; timing_pending = 0;
  }
  // A thread that ends gives back what it holds: workspace, stream, events, pinned words.  (Threads still alive when the
  // process exits are not unwound; the main thread's context is, before the HIP runtime's own static teardown.)
  ~Ctx() {
    if (device < 0) return;
    release_workspace();
    for (auto &e : ev) if (e) hipEventDestroy(e);
    if (host_flags) hipHostFree(host_flags);
    if (host_sig) hipHostFree(host_sig);
    if (host_hdr) hipHostFree(host_hdr);
    if (token) { std::lock_guard<std::mutex> g(token->mu); token->s = nullptr; } // (drained by release_workspace above)
    if (stream) hipStreamDestroy(stream);
    device = -1;
  }
};

// the flag block: 64 bytes of flags, then the single-level tile sort's heavy-hitter words (rows in long runs and done counters
// per partition, fast.hpp FoldRunsArgs::self_hot); one allocation, never regrown
constexpr size_t FLAGS_BYTES = 8192, FLAGS_HOT_OFF = 64, FLAGS_HOT_WORDS = 2 + 2 * ((1u << TS_MAX_PBITS) + 1);
static_assert(FLAGS_HOT_OFF + FLAGS_HOT_WORDS * 4 <= FLAGS_BYTES, "flag block too small");

thread_local Ctx g_ctx;

int fail(int status, const std::string &msg) {
  g_ctx.err = msg;
  return status;
}
int fail_hip(hipError_t e, const char *what) {
  return fail(e == hipErrorOutOfMemory ? PLHIP_ERR_OOM : PLHIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return fail_hip(_e, #expr);  \
  } while (0)

void resolve_timing();
void resolve_timing_if_ready();
int resolve_scratch(int dev, uint8_t **p);

int ensure_ctx() {
  Ctx &c = g_ctx;
  if (c.device >= 0) {
    HIP_TRY(hipSetDevice(c.device));
    resolve_timing_if_ready(); // (never waits: a query that records the events again resolves the pending ones first, below)
    return PLHIP_OK;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(PLHIP_ERR_NO_DEVICE, "no HIP device visible; libplhip.so has no CPU path (gfx950 required)");
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) dev = 0;
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  c.token = std::make_shared<StreamToken>();
  c.token->s = c.stream;
  for (auto &ev : c.ev) HIP_TRY(hipEventCreate(&ev));
  HIP_TRY(hipHostMalloc((void **)&c.host_flags, 64, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&c.host_sig, 64, hipHostMallocMapped | hipHostMallocCoherent));
  memset(c.host_sig, 0, 64);
  HIP_TRY(hipHostGetDevicePointer((void **)&c.host_sig_dev, c.host_sig, 0));
  HIP_TRY(hipHostMalloc((void **)&c.host_hdr, EX_MAX_SLOTS * 16, hipHostMallocMapped | hipHostMallocCoherent));
  memset(c.host_hdr, 0, EX_MAX_SLOTS * 16);
  HIP_TRY(hipHostGetDevicePointer((void **)&c.host_hdr_dev, c.host_hdr, 0));
  { const int rs = resolve_scratch(dev, &c.na_scratch); if (rs) return rs; } // `dev` is current
  c.device = dev;
  return PLHIP_OK;
}

int dtype_width(int dt) {
  switch (dt) {
    case PLHIP_I8: case PLHIP_U8: return 1;
    case PLHIP_I16: case PLHIP_U16: return 2;
    case PLHIP_I32: case PLHIP_U32: case PLHIP_F32: return 4;
    case PLHIP_I64: case PLHIP_U64: case PLHIP_F64: return 8;
    default: return 0;
  }
}
bool dt_is_float(int dt) { return dt == PLHIP_F32 || dt == PLHIP_F64; }
bool dt_is_signed(int dt) { return dt == PLHIP_I8 || dt == PLHIP_I16 || dt == PLHIP_I32 || dt == PLHIP_I64; }

// A column as the kernels see it: pointers to element 0 of the logical slice, validity bit offset.
struct DevCol {
  const uint8_t *values = nullptr;
  const uint8_t *validity = nullptr;
  int64_t boff = 0;
  int64_t length = 0;
  int dtype = 0;
};

struct Staging { // owns device copies of host columns for the duration of one call
  std::vector<void *> owned;
  ~Staging() { for (void *p : owned) hipFree(p); }
};

// Declared AFTER the owners of pooled buffers in a function: it is destroyed first, so the stream has retired every kernel that
// still touches those buffers before an early return hands them back to the pool (where the next query would reuse them).
struct StreamDrain {
  hipStream_t s;
  explicit StreamDrain(hipStream_t st) : s(st) {}
  ~StreamDrain() { if (s && hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError(); }
};

int stage_column(const plhip_column_t &c, DevCol &d, Staging &st, hipStream_t s) {
  const int w = dtype_width(c.dtype);
  if (!w) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "unsupported dtype code " + std::to_string(c.dtype));
  if (c.length < 0 || c.offset < 0) return fail(PLHIP_ERR_INVALID, "negative column offset/length");
  if (c.length > 0 && !c.values) return fail(PLHIP_ERR_INVALID, "column values pointer is NULL");
  d.length = c.length;
  d.dtype = c.dtype;
  if (c.location == PLHIP_DEVICE) {
    d.values = (const uint8_t *)c.values + c.offset * w;
    d.validity = c.validity;
    d.boff = c.offset;
    return PLHIP_OK;
  }
  if (c.location != PLHIP_HOST) return fail(PLHIP_ERR_INVALID, "bad column location");
  void *dv = nullptr;
  const size_t vbytes = (size_t)c.length * w;
  HIP_TRY(hipMalloc(&dv, vbytes ? vbytes : 8));
  st.owned.push_back(dv);
  if (vbytes) HIP_TRY(hipMemcpyAsync(dv, (const uint8_t *)c.values + c.offset * w, vbytes, hipMemcpyHostToDevice, s));
  d.values = (const uint8_t *)dv;
  if (c.validity) {
    const int64_t b0 = c.offset >> 3, b1 = (c.offset + c.length + 7) >> 3;
    void *db = nullptr;
    HIP_TRY(hipMalloc(&db, (size_t)(b1 - b0) + 8));
    st.owned.push_back(db);
    if (b1 > b0) HIP_TRY(hipMemcpyAsync(db, c.validity + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, s));
    d.validity = (const uint8_t *)db;
    d.boff = c.offset & 7;
  }
  return PLHIP_OK;
}

uint64_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}
int bit_length(uint64_t x) {
  int b = 0;
  while (x) { b++; x >>= 1; }
  return b;
}

int add_acc(Plan &p, uint32_t op, uint32_t conv, int src, uint32_t width, uint64_t init) {
  for (int i = 0; i < p.n_accs; i++)
    if (p.accs[i].op == op && p.accs[i].conv == conv && p.accs[i].src == src) return i;
  if (p.n_accs >= PLHIP_MAX_ACCS) return -1;
  AccSpec &a = p.accs[p.n_accs];
  a.op = op; a.conv = conv; a.src = src; a.width = width; a.init = init; a.offset = 0;
  return p.n_accs++;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// result frames
// ---------------------------------------------------------------------------------------------------------
struct ResultCol {
  int dtype = 0;
  void *values = nullptr;   // device or host (malloc)
  uint8_t *validity = nullptr;
  size_t vcap = 0, bcap = 0; // != 0: the device buffer came from the pool and goes back to it
};

namespace {
// Device buffers of result frames are recycled: hipMalloc / hipFree cost more than a whole small query.
// Entries are keyed by (device, size): a thread bound to one GPU must never be handed another GPU's memory.
struct DevPool {
  struct Entry { int dev; size_t cap; void *p; TokenRef busy; }; // busy: the stream whose last kernel may still be writing the buffer
  std::mutex mu;
  std::vector<Entry> free_list;
  size_t pooled = 0;
  static size_t round_up(size_t b) { size_t c = 4096; while (c < b) c <<= 1; return c; }
  // a buffer of the CURRENT device `dev` (the calling thread's Ctx::device).  on_own_stream: the caller only touches the buffer from
  // work queued on its thread's stream, so a buffer that stream is still finishing with needs no wait (stream order); anyone else
  // waits for the producer first.
  hipError_t get(int dev, size_t bytes, void **p, size_t *cap, bool on_own_stream = false) {
    const size_t c = round_up(bytes);
    TokenRef busy;
    bool hit = false;
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < free_list.size(); i++)
        if (free_list[i].dev == dev && free_list[i].cap == c) {
          *p = free_list[i].p; *cap = c; pooled -= c;
          busy = std::move(free_list[i].busy);
          free_list[i] = std::move(free_list.back()); free_list.pop_back();
          hit = true;
          break;
        }
    }
    if (hit) {
      if (busy && !(on_own_stream && busy == g_ctx.token)) {
        const hipError_t e = settle_token(busy);
        if (e != hipSuccess) { hipFree(*p); *p = nullptr; return e; }
      }
      return hipSuccess;
    }
    *cap = c;
    return hipMalloc(p, c);
  }
  void put(int dev, void *p, size_t cap, const TokenRef &busy = nullptr) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> g(mu);
      // sized for 288 GB of HBM: a one-row-per-group result at 1e8 rows is 7 GB of output columns, and handing them to
      // hipFree / hipMalloc every query costs more than a millisecond
      if (pooled + cap <= ((size_t)48 << 30) && free_list.size() < 256) { free_list.push_back(Entry{dev, cap, p, busy}); pooled += cap; return; }
    }
    hipFree(p); // (waits for the device)
  }
  // test hook / trim: how many pooled buffers belong to `dev`
  int64_t count(int dev) {
    std::lock_guard<std::mutex> g(mu);
    int64_t k = 0;
    for (auto &e : free_list) k += e.dev == dev;
    return k;
  }
  void trim(int dev) { // hands every pooled buffer of `dev` back to the driver
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < free_list.size();)
        if (free_list[i].dev == dev) { drop.push_back(free_list[i].p); pooled -= free_list[i].cap; free_list[i] = std::move(free_list.back()); free_list.pop_back(); }
        else i++;
    }
    for (void *q : drop) hipFree(q);
  }
};
DevPool g_pool;

// Host-resident result buffers (PLHIP_HOST output).  A large frame copied into freshly malloc'ed pageable memory pays a page fault per
// 4 KB and the runtime's staged copy -- measured on the reference harness's q10 (1e7 groups x 8 columns = 360 MB): 90 ms, 4 GB/s,
// against 1.7 ms of kernels.  Buffers of 32 KiB and more are therefore page-locked (hipHostMalloc) and recycled: the copy is one DMA
// at the link's rate and the pages are touched once per process, not once per query.  Smaller buffers stay malloc'ed (cap == 0).
// PLHIP_NO_PINNED_RESULTS=1 switches it off; plhip_release_workspace hands the pooled buffers back.
struct HostPool {
  struct Entry { size_t cap; void *p; };
  std::mutex mu;
  std::vector<Entry> free_list;
  size_t pooled = 0;
  static constexpr size_t PIN_FROM = (size_t)32 << 10, KEEP_MAX = (size_t)2 << 30;
  void *get(size_t bytes, size_t *cap) {
    *cap = 0;
    static const bool off = getenv("PLHIP_NO_PINNED_RESULTS") != nullptr;
    if (bytes < PIN_FROM || off) return malloc(bytes ? bytes : 8);
    const size_t c = DevPool::round_up(bytes);
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < free_list.size(); i++)
        if (free_list[i].cap == c) {
          void *p = free_list[i].p;
          pooled -= c;
          free_list[i] = free_list.back(); free_list.pop_back();
          *cap = c;
          return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, c, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return malloc(bytes); } // (no pinned memory left: pageable)
    *cap = c;
    return p;
  }
  void put(void *p, size_t cap) {
    if (!p) return;
    if (!cap) { free(p); return; }
    {
      std::lock_guard<std::mutex> g(mu);
      if (pooled + cap <= KEEP_MAX && free_list.size() < 64) { free_list.push_back(Entry{cap, p}); pooled += cap; return; }
    }
    (void)hipHostFree(p);
  }
  void trim() {
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> g(mu);
      for (auto &e : free_list) drop.push_back(e.p);
      free_list.clear(); pooled = 0;
    }
    for (void *q : drop) (void)hipHostFree(q);
  }
};
HostPool g_hpool;
} // namespace

struct plhip_result {
  std::vector<ResultCol> cols;
  int64_t n_rows = 0;
  int64_t row_offset = 0; // a sliced frame (plhip_options_t::slice_*): the columns' Arrow offset
  bool on_device = false;
  int device = 0;
  // set while the producing stream may still be retiring the frame's last kernel (the row count is already final): cleared by the
  // first access to the buffers from outside that stream (settle_result)
  mutable TokenRef pending;
  ~plhip_result() {
    for (auto &c : cols) {
      if (on_device) {
        if (c.vcap) g_pool.put(device, c.values, c.vcap, pending); else if (c.values) hipFree(c.values);
        if (c.bcap) g_pool.put(device, c.validity, c.bcap, pending); else if (c.validity) hipFree(c.validity);
      } else { g_hpool.put(c.values, c.vcap); g_hpool.put(c.validity, c.bcap); } // (host: caps are the host pool's, 0 = malloc)
    }
  }
};

struct plhip_groups {
  uint32_t *first = nullptr, *offsets = nullptr, *indices = nullptr;
  size_t cap_first = 0, cap_offsets = 0, cap_indices = 0; // != 0: the buffer came from the (device / host) pool and goes back to it
  int device = 0;
  int64_t n_groups = 0, n_rows = 0;
  bool on_device = false;
  ~plhip_groups() {
    if (on_device) {
      if (cap_first) g_pool.put(device, first, cap_first); else hipFree(first);
      if (cap_offsets) g_pool.put(device, offsets, cap_offsets); else hipFree(offsets);
      if (cap_indices) g_pool.put(device, indices, cap_indices); else hipFree(indices);
    } else { g_hpool.put(first, cap_first); g_hpool.put(offsets, cap_offsets); g_hpool.put(indices, cap_indices); }
  }
};

// ---------------------------------------------------------------------------------------------------------
// lowering
// ---------------------------------------------------------------------------------------------------------
namespace {

struct KeyCol { // a key column on its way into the packed key
  DevCol col;
  uint64_t lo = 0, hi = 0; // range of the normalised bits over valid rows
  uint32_t bits = 64;
  bool nullable = false;
};

void set_keyspec(KeySpec &k, const KeyCol &kc) {
  k.values = kc.col.values; k.validity = kc.col.validity; k.boff = kc.col.boff; k.dtype = (uint32_t)kc.col.dtype;
  k.minv = kc.lo; k.shift = 0; k.bits = kc.bits; k.nullable = kc.nullable ? 1 : 0;
}

int read_flags(TableRef t, uint32_t *out8);

// min/max pass over `cols` (k_key_minmax), then bits = bit_length(hi - lo + nullable)
int measure_key_columns(std::vector<KeyCol> &cols, int64_t n) {
  Ctx &c = g_ctx;
  Plan M{};
  M.n_rows = n; M.n_keys = (int)cols.size();
  for (size_t i = 0; i < cols.size(); i++) { cols[i].lo = 0; cols[i].bits = 64; set_keyspec(M.keys[i], cols[i]); }
  // mm[2i] = ~min, mm[2i + 1] = max over the valid rows (0, 0 when there is none); the result comes back through PINNED host
  // memory (pageable async copies are staged: ~100 us each way)
  const size_t mm_bytes = cols.size() * 2 * sizeof(uint64_t);
  HIP_TRY(c.misc.reserve(256 + key_minmax_scratch_bytes((int)cols.size())));
  uint64_t *mm = c.host_hdr; // (EX_MAX_SLOTS * 16 bytes >= PLHIP_MAX_KEYS * 16; the sharded path reads it only inside its own call)
  static_assert(EX_MAX_SLOTS * 16 >= PLHIP_MAX_KEYS * 16, "the pinned block holds a min / max pair per key column");
  uint64_t mm_local[2];
  static const bool no_word = getenv("PLHIP_MINMAX_COPY") != nullptr;
  if (cols.size() == 1 && n > 0 && dtype_width(cols[0].col.dtype) <= 4 && !no_word) {
    // One key column of at most 32 bits (the compact-range test of q5's id6 under AUTO): the pair comes back as ONE pinned 64-bit
    // word the host spins on -- no device-to-host copy, no stream wait (~12 us of a 0.15 ms query).  The word starts as all ones;
    // should the answer BE all ones (every key 0xFFFFFFFF), the stream going idle settles it.
    __atomic_store_n(&c.host_hdr[0], ~0ull, __ATOMIC_RELEASE);
    launch_key_minmax(M, (uint64_t *)c.misc.p, (uint64_t *)((uint8_t *)c.misc.p + 256), c.stream, c.host_hdr_dev);
    HIP_TRY(hipGetLastError());
    uint64_t v;
    for (uint64_t spin = 1;; spin++) {
      if ((v = __atomic_load_n(&c.host_hdr[0], __ATOMIC_ACQUIRE)) != ~0ull) break;
      if ((spin & 2047) == 0) {
        const hipError_t q = hipStreamQuery(c.stream);
        if (q == hipSuccess) { v = __atomic_load_n(&c.host_hdr[0], __ATOMIC_ACQUIRE); break; }
        if (q != hipErrorNotReady) return fail_hip(q, "k_key_minmax");
      }
      __builtin_ia32_pause();
    }
    mm_local[0] = v >> 32; mm_local[1] = v & 0xFFFFFFFFull; // (min > max: no valid row)
    mm = mm_local;
  } else {
    HIP_TRY(hipMemsetAsync(c.misc.p, 0, mm_bytes, c.stream));
    if (n > 0) launch_key_minmax(M, (uint64_t *)c.misc.p, (uint64_t *)((uint8_t *)c.misc.p + 256), c.stream);
    HIP_TRY(hipMemcpyAsync(mm, c.misc.p, mm_bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    for (size_t i = 0; i < cols.size(); i++) mm[2 * i] = ~mm[2 * i];
  }
  for (size_t i = 0; i < cols.size(); i++) {
    KeyCol &k = cols[i];
    const uint64_t lo = mm[2 * i], hi = mm[2 * i + 1];
    const uint64_t span = hi >= lo ? hi - lo : 0; // no valid row: only the null code
    k.lo = hi >= lo ? lo : 0;
    k.hi = hi >= lo ? hi : 0;
    k.bits = (k.nullable && span == ~0ull) ? 65u : (uint32_t)bit_length(span + (k.nullable ? 1 : 0));
  }
  return PLHIP_OK;
}

// Fills P.keys from packable columns (sum of bits <= 64): key 0 in the most significant bits.
void pack_keys(Plan &P, const std::vector<KeyCol> &cols) {
  P.n_keys = (int)cols.size();
  int shift = 0;
  for (auto &k : cols) shift += (int)k.bits;
  for (size_t i = 0; i < cols.size(); i++) {
    set_keyspec(P.keys[i], cols[i]);
    shift -= (int)cols[i].bits;
    P.keys[i].shift = (uint32_t)shift;
  }
}

// Dense ids of the distinct tuples of `sub` (whose bits fit one word): ids[row] in [0, n_distinct).
// The wide-tuple analogue of the reference hashing rows first (vector_hasher.rs:467-483) and comparing
// keys on equal hashes (hashing.rs:334-378) - here the tuple is made narrow instead, so equality is exact.
int dense_ids(const std::vector<KeyCol> &sub, int64_t n, Staging &st, KeyCol &out) {
  Ctx &c = g_ctx;
  Plan K{};
  K.n_rows = n;
  K.single_key = 0; // nulls are codes: a null tuple member must get an id like any other value
  if (sub.size() == 1 && sub[0].bits > 64) { // a nullable column spanning all 2^64 values: no code is left for its null,
    std::vector<KeyCol> one = sub;           // so it is encoded alone with the single-key rules (null -> the reserved slot)
    one[0].lo = 0; one[0].bits = 64; one[0].nullable = false;
    K.single_key = 1;
    pack_keys(K, one);
  } else pack_keys(K, sub);
  const uint64_t cap = next_pow2(std::max<uint64_t>(2 * (uint64_t)n, 1024));
  plan_layout(K, cap + 2);
  HIP_TRY(c.table.reserve(table_bytes(K, cap + 2)));
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  HIP_TRY(c.slots.reserve((cap + 2) * 8));
  HIP_TRY(c.rank.reserve((cap + 2) * 4));
  void *ids = nullptr;
  HIP_TRY(hipMalloc(&ids, (size_t)std::max<int64_t>(n, 1) * 4));
  st.owned.push_back(ids);
  TableRef t{(uint8_t *)c.table.p, cap, (uint32_t *)c.flags.p};
  launch_table_init(K, t, c.stream);
  launch_sample_keys(K, t, n, 1, c.stream);
  launch_compact(K, t, (uint64_t *)c.slots.p, c.stream);
  uint32_t f[16];
  int rc = read_flags(t, f);
  if (rc) return rc;
  if (f[FLAG_OVERFLOW]) return fail(PLHIP_ERR_COMPUTE, "hash table overflow while encoding wide keys");
  const uint64_t g = f[FLAG_NGROUPS];
  launch_rank_scatter((const uint64_t *)c.slots.p, g, (uint32_t *)c.rank.p, c.stream);
  launch_encode_rows(K, t, (const uint32_t *)c.rank.p, (uint32_t *)ids, c.stream);
  HIP_TRY(hipGetLastError());
  out = KeyCol{};
  out.col.values = (const uint8_t *)ids; out.col.validity = nullptr; out.col.boff = 0; out.col.length = n;
  out.col.dtype = PLHIP_U32;
  out.lo = 0; out.hi = g ? g - 1 : 0; out.bits = (uint32_t)bit_length(out.hi); out.nullable = false;
  return PLHIP_OK;
}

// Lowers the key columns to the packed-key plan.  `KG.enabled` is set when the tuple had to be reduced
// to dense ids (then output keys are gathered at the first row).
int build_keys(Plan &P, const std::vector<DevCol> &keys, Staging &st, KeyGather &KG) {
  KG.enabled = 0; KG.first_acc = -1;
  P.single_key = keys.size() == 1;
  std::vector<KeyCol> cols(keys.size());
  for (size_t i = 0; i < keys.size(); i++) {
    cols[i].col = keys[i];
    cols[i].nullable = keys[i].validity != nullptr;
    KeySpec tmp{};
    set_keyspec(tmp, cols[i]);
    tmp.minv = 0;
    KG.okeys[i] = tmp;
  }
  if (P.single_key) { // normalised bits as they are; null -> reserved slot
    cols[0].lo = 0; cols[0].bits = 64; cols[0].nullable = false;
    pack_keys(P, cols);
    return PLHIP_OK;
  }
  int rc = measure_key_columns(cols, P.n_rows);
  if (rc) return rc;
  auto total_bits = [&]() { int t = 0; for (auto &k : cols) t += (int)k.bits; return t; };
  while (total_bits() > 64) {
    KG.enabled = 1;
    std::stable_sort(cols.begin(), cols.end(), [](const KeyCol &a, const KeyCol &b) { return a.bits > b.bits; });
    std::vector<KeyCol> sub;
    size_t take = 0;
    if (cols[0].bits > 32) take = 1; // (65 bits = a nullable column over the full 64-bit range: dense_ids encodes it alone)
    else { int t = 0; while (take < cols.size() && t + (int)cols[take].bits <= 64) { t += (int)cols[take].bits; take++; } }
    sub.assign(cols.begin(), cols.begin() + take);
    KeyCol enc;
    if ((rc = dense_ids(sub, P.n_rows, st, enc))) return rc;
    cols.erase(cols.begin(), cols.begin() + take);
    cols.push_back(enc);
  }
  pack_keys(P, cols);
  return PLHIP_OK;
}

struct Lowered {
  Plan P;
  KeyGather KG;
  int n_out_keys = 0;
  std::vector<OutSpec> outs;
  int first_acc = -1;
};

// Internal to the library (never set by callers: the entry points mask it): the partial sums of 4-byte integer columns
// only have to be right modulo 2^32, because the caller -- the sharded path -- wraps them into the column's type at the end.
constexpr uint32_t FLAG_INTERNAL_WRAP32 = 1u << 30;

int build_aggs(Lowered &L, const std::vector<DevCol> &vals, const plhip_agg_t *aggs, int n_aggs, uint32_t flags) {
  Plan &P = L.P;
  P.n_vals = (int)vals.size();
  for (size_t i = 0; i < vals.size(); i++) {
    ValSpec &v = P.vals[i];
    v.values = vals[i].values; v.validity = vals[i].validity; v.boff = vals[i].boff; v.dtype = (uint32_t)vals[i].dtype; v.valid_bytes = 0;
  }
  for (int k = 0; k < L.n_out_keys; k++) { // output keys are the ORIGINAL key columns, in order
    OutSpec o{};
    o.kind = OUT_KEY; o.dtype = L.KG.okeys[k].dtype; o.acc = k; o.cnt_valid = -1; o.cnt_all = -1;
    o.validity = L.KG.okeys[k].validity ? (uint8_t *)1 : nullptr; // placeholder: "needs a bitmap"
    L.outs.push_back(o);
  }
  const bool partial = flags & PLHIP_FLAG_PARTIAL;
  const bool exact = (flags & PLHIP_FLAG_EXACT_FOLD) && !partial;
  for (int i = 0; i < n_aggs; i++) {
    const int kind = aggs[i].kind;
    OutSpec o{};
    o.cnt_valid = -1; o.cnt_all = -1; o.acc = -1;
    if (kind == PLHIP_AGG_COUNT) {
      o.kind = partial ? OUT_RAW_U32_AS_U64 : OUT_COUNT;
      o.dtype = partial ? PLHIP_U64 : PLHIP_U32;
      o.acc = add_acc(P, ACC_CNT_ALL, CONV_NONE, -1, 4, 0);
      if (o.acc < 0) return fail(PLHIP_ERR_INVALID, "too many accumulators");
      L.outs.push_back(o);
      continue;
    }
    if (kind < 0 || kind > PLHIP_AGG_COUNT) return fail(PLHIP_ERR_INVALID, "bad aggregation kind");
    const int ci = aggs[i].column;
    if (ci < 0 || ci >= (int)vals.size()) return fail(PLHIP_ERR_INVALID, "aggregation column index out of range");
    const int dt = vals[ci].dtype;
    const bool nullable = vals[ci].validity != nullptr;
    const bool is_f = dt_is_float(dt), is_s = dt_is_signed(dt);
    const int w = dtype_width(dt);
    const int cnt_valid = nullable ? add_acc(P, ACC_CNT_VALID, CONV_NONE, ci, 4, 0) : -1;
    if (kind == PLHIP_AGG_SUM) {
      if (is_f) { o.kind = OUT_SUM_F64; o.dtype = dt; o.acc = add_acc(P, (exact && dt == PLHIP_F32) ? ACC_ADD_F32 : ACC_ADD_F64, CONV_F64, ci, 8, 0); }
      else {
        o.kind = OUT_SUM_INT; o.dtype = w < 4 ? PLHIP_I64 : dt; // series/implementations/mod.rs:167-174
        o.acc = add_acc(P, ACC_ADD_U64, CONV_INT, ci, 8, 0);
      }
      if (partial && is_f) { o.kind = OUT_RAW_U64; o.dtype = PLHIP_F64; }
      if (partial && !is_f) {
        o.kind = ((flags & FLAG_INTERNAL_WRAP32) && w == 4) ? OUT_RAW_U64_WRAP32 : OUT_RAW_U64;
        o.dtype = (is_s || w < 4) ? PLHIP_I64 : PLHIP_U64;
      }
      o.validity = nullptr;
    } else if (kind == PLHIP_AGG_MEAN) {
      o.cnt_valid = cnt_valid;
      o.cnt_all = add_acc(P, ACC_CNT_ALL, CONV_NONE, -1, 4, 0);
      if (!is_f && w <= 4) {
        o.kind = is_s ? OUT_MEAN_INT : OUT_MEAN_UINT; o.dtype = PLHIP_F64;
        o.acc = add_acc(P, ACC_ADD_U64, CONV_INT, ci, 8, 0);
      } else {
        o.kind = OUT_MEAN_F64; o.dtype = dt == PLHIP_F32 ? PLHIP_F32 : PLHIP_F64;
        o.acc = add_acc(P, (exact && dt == PLHIP_F32) ? ACC_ADD_F32 : ACC_ADD_F64, CONV_F64, ci, 8, 0);
      }
      o.validity = nullable ? (uint8_t *)1 : nullptr;
      if (partial) { // (sum, valid count) as in evaluate_partitioned (aggregation.rs:297-322)
        OutSpec s = o;
        s.kind = OUT_RAW_U64; s.validity = nullptr;
        s.dtype = (o.kind == OUT_MEAN_F64) ? PLHIP_F64 : (is_s ? PLHIP_I64 : PLHIP_U64);
        L.outs.push_back(s);
        o.kind = OUT_RAW_U32_AS_U64; o.dtype = PLHIP_U64; o.acc = nullable ? cnt_valid : o.cnt_all; o.validity = nullptr;
      }
    } else { // MIN / MAX
      const bool is_min = kind == PLHIP_AGG_MIN;
      o.dtype = dt;
      o.cnt_valid = cnt_valid;
      if (is_f) {
        o.kind = is_min ? OUT_FMIN : OUT_FMAX;
        o.cnt_all = add_acc(P, ACC_CNT_ALL, CONV_NONE, -1, 4, 0);
        o.acc = is_min ? add_acc(P, ACC_MIN_U64, CONV_FMIN, ci, 8, ~0ull) : add_acc(P, ACC_MAX_U64, CONV_FMAX, ci, 8, 0);
      } else {
        o.kind = OUT_MIN_INT;
        if (is_s) o.acc = is_min ? add_acc(P, ACC_MIN_I64, CONV_INT, ci, 8, (uint64_t)INT64_MAX)
                                 : add_acc(P, ACC_MAX_I64, CONV_INT, ci, 8, (uint64_t)INT64_MIN);
        else o.acc = is_min ? add_acc(P, ACC_MIN_U64, CONV_INT, ci, 8, ~0ull) : add_acc(P, ACC_MAX_U64, CONV_INT, ci, 8, 0);
      }
      o.validity = nullable ? (uint8_t *)1 : nullptr;
    }
    if (o.acc < 0 || (kind == PLHIP_AGG_MEAN && o.cnt_all < 0)) return fail(PLHIP_ERR_INVALID, "too many accumulators");
    L.outs.push_back(o);
  }
  if ((flags & PLHIP_FLAG_MAINTAIN_ORDER) || L.KG.enabled) {
    L.first_acc = add_acc(P, ACC_FIRST_ROW, CONV_NONE, -1, 4, 0xFFFFFFFFull);
    if (L.first_acc < 0) return fail(PLHIP_ERR_INVALID, "too many accumulators");
    L.KG.first_acc = L.first_acc;
  }
  if ((int)L.outs.size() > PLHIP_MAX_OUTS) return fail(PLHIP_ERR_INVALID, "too many output columns");
  return PLHIP_OK;
}

int env_strategy() {
  const char *e = getenv("PLHIP_STRATEGY");
  if (!e) return PLHIP_STRATEGY_AUTO;
  if (!strcmp(e, "global")) return PLHIP_STRATEGY_GLOBAL;
  if (!strcmp(e, "lds")) return PLHIP_STRATEGY_LDS;
  if (!strcmp(e, "partitioned")) return PLHIP_STRATEGY_PARTITIONED;
  if (!strcmp(e, "direct")) return PLHIP_STRATEGY_DIRECT;
  if (!strcmp(e, "tilesort")) return PLHIP_STRATEGY_TILESORT;
  if (!strcmp(e, "sort")) return PLHIP_STRATEGY_SORT;
  if (!strcmp(e, "tilesort2")) return PLHIP_STRATEGY_TILESORT2;
  if (!strcmp(e, "dense")) return PLHIP_STRATEGY_DENSE;
  return PLHIP_STRATEGY_AUTO;
}

// Sampled cardinality estimate, the analogue of estimate_unique_count
// (crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:115-157): insert every
// (n/s)-th row's key into a scratch HBM table and count the distinct keys d, then invert
// E[d] = G (1 - exp(-s/G)).
int estimate_groups(const Plan &P, uint64_t &g_est);

} // namespace

// kernels.hip: a keys-only pass over a strided sample (declared here to keep kernels.hpp minimal)
void launch_sample_keys(const Plan &p, TableRef t, int64_t n_sample, int64_t stride, hipStream_t s);

namespace {

int read_flags(TableRef t, uint32_t *out8) {
  Ctx &c = g_ctx;
  c.flags_clean = false; // whoever reads the flags has used them
  HIP_TRY(hipMemcpyAsync(c.host_flags, t.flags, 32, hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  memcpy(out8, c.host_flags, 32); // the first 8 flags are the ones the host reads
  return PLHIP_OK;
}

// Waits for publish_done (fast_path.hip) of the query tagged `seq`.  The stream is polled now and then so
// that a failed launch cannot leave the host spinning; if the stream drained without the signal the flags are
// fetched the ordinary way.
int wait_signal(TableRef t, uint32_t seq, uint32_t *out8, bool defer = false) {
  Ctx &c = g_ctx;
  c.flags_clean = false;
  c.signal_seen = false;
  const uint64_t *sig = (const uint64_t *)c.host_sig;
  uint64_t v;
  for (uint64_t spin = 1;; spin++) {
    if (((v = __atomic_load_n(sig, __ATOMIC_ACQUIRE)) >> 32) == seq) break;
    if ((spin & 2047) == 0) {
      const hipError_t q = hipStreamQuery(c.stream);
      if (q == hipSuccess) {
        if (((v = __atomic_load_n(sig, __ATOMIC_ACQUIRE)) >> 32) == seq) break;
        // (never seen: a kernel that retired without its signal.  The flag block may have been zeroed by the kernel's last
        // workgroup already, so a count of zero cannot be told from "no result": an error, not an empty frame)
        const int frc = read_flags(t, out8);
        if (!frc && !out8[FLAG_OVERFLOW] && out8[FLAG_NGROUPS] == 0) return fail(PLHIP_ERR_HIP, "the kernels retired without publishing their result");
        return frc;
      }
      if (q != hipErrorNotReady) return fail_hip(q, "group-by kernels");
    }
    __builtin_ia32_pause();
  }
  c.signal_seen = true;
  memset(out8, 0, 32);
  out8[FLAG_OVERFLOW] = (uint32_t)(v >> 31) & 1u;
  out8[FLAG_NGROUPS] = (uint32_t)v & 0x7FFFFFFFu;
  // the values are here; the result buffers may be handed on only once the kernel has retired (the signal leaves when the last
  // workgroup has RESERVED its rows, not when every workgroup has stored them).  That wait is ~7 us of host time per query: a
  // DEVICE-output query leaves it to the first reader of the frame (plhip_result::pending), so a caller that chains queries or only
  // wants the row count does not pay it per call.
  static const bool no_defer = getenv("PLHIP_NO_DEFERRED_SYNC") != nullptr;
  if (defer && !no_defer && !out8[FLAG_OVERFLOW]) { c.deferred = true; return PLHIP_OK; } // (an overflowed query is redone: its buffers go back first)
  HIP_TRY(hipStreamSynchronize(c.stream));
  return PLHIP_OK;
}

// event times of the last fast-path query (read lazily: the host returns before the stream has retired the
// trailing event records)
// ensure_ctx's variant: the next call usually comes while the previous query's last kernel is still retiring (deferred
// completion); waiting for its trailing event there would put the stream wait back into every sampled step
void resolve_timing_if_ready() {
  Ctx &c = g_ctx;
  if (!c.timing_pending) return;
  const bool merged = c.timing_pending != 2;
  const hipError_t q = hipEventQuery(merged ? c.ev[2] : c.ev[5]);
  if (q == hipSuccess) resolve_timing();
  else if (q != hipErrorNotReady) (void)hipGetLastError();
}

void resolve_timing() {
  Ctx &c = g_ctx;
  if (!c.timing_pending) return;
  const bool dense_part = c.timing_pending == 5; // range partition, dense fold, dense merge
  const bool dense = c.timing_pending == 4 || dense_part;
  const bool direct = c.timing_pending == 1 || c.timing_pending == 4, merged = c.timing_pending != 2;
  c.timing_pending = 0;
  if (hipEventSynchronize(merged ? c.ev[2] : c.ev[5]) != hipSuccess) { (void)hipGetLastError(); return; }
  hipEventElapsedTime(&c.total_ms, c.ev[1], merged ? c.ev[2] : c.ev[5]);
  c.k_ms[0] = c.k_ms[2] = 0.f;
  if (!direct) hipEventElapsedTime(&c.k_ms[0], c.ev[1], c.ev[4]);
  hipEventElapsedTime(&c.k_ms[1], direct ? c.ev[1] : c.ev[4], c.ev[5]);
  if (merged) hipEventElapsedTime(&c.k_ms[2], c.ev[5], c.ev[2]);
  const bool leaf = c.strategy == "leafsort";
  const char *names[3] = {dense_part ? "k_dense_part_hist+k_tile_sort(scatter)" : "k_tile_sort", dense ? "k_dense_fold" : (direct ? "k_fold_rows" : (leaf ? "k_leaf_sort" : "k_fold_runs")),
                          dense ? "k_dense_merge" : "k_merge_emit"};
  int dom = 0;
  for (int i = 1; i < 3; i++) if (c.k_ms[i] > c.k_ms[dom]) dom = i;
  c.dom_ms = c.k_ms[dom]; c.dom_kernel = names[dom]; c.part_ms = c.k_ms[0]; c.agg_ms = c.k_ms[1];
}

int estimate_groups(const Plan &P, uint64_t &g_est) {
  Ctx &c = g_ctx;
  const int64_t n = P.n_rows;
  const int64_t s = std::min<int64_t>(n, 1 << 16);
  if (s <= 0) { g_est = 0; return PLHIP_OK; }
  Plan K = P;
  K.n_accs = 0;
  const uint64_t cap = 1u << 18;
  HIP_TRY(c.table.reserve(table_bytes(K, cap + 2)));
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  HIP_TRY(c.slots.reserve((cap + 2) * 8));
  TableRef t{(uint8_t *)c.table.p, cap, (uint32_t *)c.flags.p};
  plan_layout(K, cap + 2);
  launch_table_init(K, t, c.stream);
  launch_sample_keys(K, t, s, n / s, c.stream);
  launch_compact(K, t, (uint64_t *)c.slots.p, c.stream);
  uint32_t f[16];
  int rc = read_flags(t, f);
  if (rc) return rc;
  const double d = f[FLAG_NGROUPS];
  if (d >= 0.9 * (double)s) { g_est = (uint64_t)n; return PLHIP_OK; }
  // solve d = G (1 - exp(-s/G)) by bisection on G in [d, n]
  double lo = d, hi = (double)n;
  for (int it = 0; it < 60; it++) {
    const double g = 0.5 * (lo + hi);
    const double e = g * (1.0 - std::exp(-(double)s / g));
    if (e < d) lo = g; else hi = g;
  }
  g_est = (uint64_t)std::min<double>((double)n, std::ceil(hi));
  return PLHIP_OK;
}

int copy_out(const void *dev, size_t bytes, void **host, size_t *cap) { // *cap: what the owner hands back to g_hpool.put
  *host = g_hpool.get(bytes, cap);
  if (!*host) return fail(PLHIP_ERR_OOM, "host allocation failed");
  if (bytes) HIP_TRY(hipMemcpy(*host, dev, bytes, hipMemcpyDeviceToHost));
  return PLHIP_OK;
}

} // namespace


namespace {

// bytes of the packed key as stored in LDS tables and partitioned intermediates
int packed_key_bytes(const Plan &P, const std::vector<DevCol> &dk) {
  if (P.single_key) return dtype_width(dk[0].dtype) <= 4 ? 4 : 8;
  int total = 0;
  for (int i = 0; i < P.n_keys; i++) total += (int)P.keys[i].bits;
  return total <= 32 ? 4 : 8;
}

struct PartitionSetup {
  uint32_t pbits = 0, lbits = 0;
  int key_bytes = 8;
};

// Chooses the partition fan-out and the per-partition LDS table so that an average partition fills its
// table to <= 25 % (hash skew head-room; a table that still overflows spills rows to the HBM table).
bool plan_partitions(const Plan &P, uint64_t g_est, int key_bytes, uint32_t lbits2, uint32_t lbits1, PartitionSetup &PS) {
  (void)P;
  PS.key_bytes = key_bytes;
  for (uint32_t lb : {lbits2, lbits1}) {
    const uint64_t per_part = (1ull << lb) / 4;
    uint32_t pb = 4;
    while (pb < 12 && (g_est >> pb) > per_part) pb++;
    if ((g_est >> pb) <= per_part) {
      PS.pbits = pb;
      uint32_t fit = 6;
      while ((1ull << fit) < 4 * std::max<uint64_t>(g_est >> pb, 1) && fit < lb) fit++;
      PS.lbits = fit;
      return true;
    }
  }
  return false;
}

int run_partitioned(const Plan &P, TableRef T, const PartitionSetup &PS, uint64_t dense_cap, uint64_t *slots_dev, bool need_rowidx) {
  Ctx &c = g_ctx;
  const int64_t n = P.n_rows;
  const uint32_t np = 1u << PS.pbits;
  const int64_t tile_rows = (int64_t)PART_SCATTER_R * 1024; // the scatter kernel sorts one tile in LDS
  const uint32_t n_tiles = (uint32_t)std::max<int64_t>(1, (n + tile_rows - 1) / tile_rows);
  const uint32_t split_rows = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, std::max<uint64_t>(4ull * (uint64_t)n / np, 1 << 16));
  const uint64_t max_items = np + (uint64_t)n / split_rows + 2;
  // scratch: counts | totals | part_start | n_items,next_item | items
  size_t off_counts = 0, off_totals = off_counts + (size_t)np * n_tiles * 4, off_start = off_totals + (size_t)np * 4;
  size_t off_cnt = (off_start + (size_t)(np + 1) * 4 + 63) & ~(size_t)63, off_items = off_cnt + 64;
  HIP_TRY(c.items.reserve(off_items + max_items * sizeof(WorkItem)));
  uint8_t *sb = (uint8_t *)c.items.p;
  uint32_t *counts = (uint32_t *)(sb + off_counts), *totals = (uint32_t *)(sb + off_totals), *pstart = (uint32_t *)(sb + off_start);
  uint32_t *cnt = (uint32_t *)(sb + off_cnt);
  WorkItem *items = (WorkItem *)(sb + off_items);
  // intermediates: packed keys | row ids | value columns | validity bytes
  PartOut O{};
  O.key_bytes = (uint32_t)PS.key_bytes;
  size_t bytes = ((size_t)n * PS.key_bytes + 255) & ~(size_t)255;
  size_t off_row = bytes;
  if (need_rowidx) bytes += ((size_t)n * 4 + 255) & ~(size_t)255;
  size_t off_val[PLHIP_MAX_VALS], off_valid[PLHIP_MAX_VALS];
  for (int v = 0; v < P.n_vals; v++) {
    off_val[v] = bytes;
    bytes += ((size_t)n * dtype_width((int)P.vals[v].dtype) + 255) & ~(size_t)255;
    off_valid[v] = bytes;
    if (P.vals[v].validity) bytes += ((size_t)n + 255) & ~(size_t)255;
  }
  HIP_TRY(c.part.reserve(bytes + 256));
  uint8_t *pb = (uint8_t *)c.part.p;
  O.keys = pb;
  O.rowidx = need_rowidx ? (uint32_t *)(pb + off_row) : nullptr;
  for (int v = 0; v < P.n_vals; v++) {
    O.vals[v] = pb + off_val[v];
    O.valid[v] = P.vals[v].validity ? pb + off_valid[v] : nullptr;
  }
  launch_part_hist(P, PS.pbits, tile_rows, n_tiles, counts, c.stream);
  launch_part_scan(counts, np, n_tiles, totals, c.stream);
  launch_part_plan(totals, np, pstart, items, cnt, cnt + 1, split_rows, c.stream);
  launch_part_scatter(P, O, PS.pbits, tile_rows, n_tiles, counts, pstart, c.stream);
  HIP_TRY(hipEventRecord(c.ev[4], c.stream));
  // the aggregation pass reads the partitioned copies
  Plan P2 = P;
  P2.prepacked = 1;
  P2.single_key = 0;
  P2.keys[0].values = O.keys; P2.keys[0].validity = nullptr; P2.keys[0].boff = 0;
  P2.keys[0].dtype = PS.key_bytes == 4 ? PLHIP_U32 : PLHIP_U64;
  P2.rowidx = O.rowidx;
  for (int v = 0; v < P.n_vals; v++) {
    P2.vals[v].values = O.vals[v]; P2.vals[v].validity = O.valid[v]; P2.vals[v].boff = 0; P2.vals[v].valid_bytes = 1;
  }
  launch_lds_agg2(P2, T, items, cnt, (uint32_t)max_items, cnt + 1, PS.lbits, PS.key_bytes, dense_cap, slots_dev, c.stream);
  HIP_TRY(hipEventRecord(c.ev[5], c.stream));
  HIP_TRY(hipGetLastError());
  return PLHIP_OK;
}

} // namespace


namespace {

// Hands the device-resident frame R (G rows) to the caller, copied to host memory unless
// PLHIP_FLAG_OUTPUT_DEVICE is set.
int finish_result(plhip_result *R, uint64_t G, uint32_t flags, plhip_result_t **out) {
  Ctx &c = g_ctx;
  if (!(flags & PLHIP_FLAG_OUTPUT_DEVICE)) {
    plhip_result *H = new (std::nothrow) plhip_result();
    if (!H) { delete R; return fail(PLHIP_ERR_OOM, "host allocation failed"); }
    H->n_rows = R->n_rows; H->on_device = false; H->device = c.device;
    // all columns' copies are queued, then waited for once: page-locked destinations (HostPool) let the copy engines overlap them
    hipError_t e = settle_token(R->pending);
    R->pending = nullptr;
    for (auto &col : R->cols) {
      if (e != hipSuccess) break;
      ResultCol h;
      h.dtype = col.dtype;
      const size_t vb = col.dtype == PLHIP_BOOL ? (size_t)((G + 63) / 64) * 8 : (size_t)G * dtype_width(col.dtype), bb = (size_t)(G / 64 + 1) * 8;
      h.values = g_hpool.get(vb, &h.vcap);
      if (col.validity) h.validity = (uint8_t *)g_hpool.get(bb, &h.bcap);
      H->cols.push_back(h);
      if (!h.values || (col.validity && !h.validity)) { (void)hipStreamSynchronize(c.stream); delete R; delete H; return fail(PLHIP_ERR_OOM, "host allocation failed"); }
      if (vb) e = hipMemcpyAsync(h.values, col.values, vb, hipMemcpyDeviceToHost, c.stream);
      if (e == hipSuccess && col.validity) e = hipMemcpyAsync(h.validity, col.validity, bb, hipMemcpyDeviceToHost, c.stream);
    }
    const hipError_t e2 = hipStreamSynchronize(c.stream); // (also on the error path: the host buffers go back to the pool below)
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) { delete R; delete H; return fail_hip(e, "result copy"); }
    delete R;
    R = H;
  }
  *out = R;
  return PLHIP_OK;
}

// Orders the accumulators by source column (stable) and renumbers the outputs' references.
void order_accs_by_source(Lowered &L) {
  Plan &P = L.P;
  for (int i = 1; i < P.n_accs; i++)
    for (int j = i; j > 0 && P.accs[j].src < P.accs[j - 1].src; j--) {
      std::swap(P.accs[j], P.accs[j - 1]);
      for (auto &o : L.outs) {
        auto fix = [&](int32_t &x) { if (x == j) x = j - 1; else if (x == j - 1) x = j; };
        if (o.kind != OUT_KEY) fix(o.acc);
        fix(o.cnt_valid); fix(o.cnt_all);
      }
      if (L.first_acc == j) L.first_acc = j - 1; else if (L.first_acc == j - 1) L.first_acc = j;
      if (L.KG.first_acc == j) L.KG.first_acc = j - 1; else if (L.KG.first_acc == j - 1) L.KG.first_acc = j;
    }
}

int ceil_log2(uint64_t x) { int b = 0; while ((1ull << b) < x) b++; return b; }

// Sampled cardinality for the LDS pipelines (fast_path.hip k_estimate): g_est = estimated number of groups.
int estimate_groups_sampled(const Plan &P, uint64_t &g_est) {
  Ctx &c = g_ctx;
  const int64_t n = P.n_rows;
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  uint32_t *dflags = (uint32_t *)c.flags.p;
  // 131 072 sampled keys resolve up to ~1e6 groups under the uniform model E[d] = G (1 - exp(-s / G)).  Two things
  // send the estimate back for a second, larger sample when the column is long enough for it to matter: a sample
  // that is (almost) all distinct, and skew -- heavy hitters eat the sample and the uniform model then reads far too
  // few groups; it shows as more once-seen keys than the model allows, and then the estimate is the larger of the
  // model's and GEE's (sqrt(n / s) * f1 + (d - f1), Charikar et al.), which is built for that case.
  int64_t ns = n <= (1 << 17) ? n : (1 << 17); // whole 256-row blocks when sampling
  g_est = (uint64_t)n;
  static const int cap_log2 = getenv("PLHIP_EST_CAP_LOG2") ? std::max(18, std::min(24, atoi(getenv("PLHIP_EST_CAP_LOG2")))) : 22;
  const int64_t ns2_cap = (int64_t)1 << cap_log2; // the second, larger sample
  for (int round = 0; round < 2; round++) {
    const uint32_t set_bits = (uint32_t)std::max(10, ceil_log2((uint64_t)ns * 4));
    const size_t set_bytes = ((size_t)4 << set_bits) + ((size_t)4 << set_bits) / 16; // fingerprints + 2 state bits per slot
    HIP_TRY(c.table.reserve(set_bytes));
    HIP_TRY(hipMemsetAsync(c.table.p, 0, set_bytes, c.stream));
    HIP_TRY(hipMemsetAsync(dflags, 0, 64, c.stream));
    c.flags_clean = false;
    launch_estimate(P, ns, ns == n ? 256 : n / (ns >> 8), (uint32_t *)c.table.p, set_bits, dflags, c.stream);
    HIP_TRY(hipGetLastError());
    TableRef t{nullptr, 0, dflags};
    uint32_t f[16];
    int rc = read_flags(t, f);
    if (rc) return rc;
    const double d = f[FLAG_NGROUPS], f1 = (double)(int32_t)f[FLAG_F1];
    if (getenv("PLHIP_DEBUG_EST")) fprintf(stderr, "estimate: round %d n %lld ns %lld d %.0f f1 %.0f flags %u %u %u %u %u %u %u %u\n", round, (long long)n, (long long)ns, d, f1, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    const bool more = round == 0 && n >= 32 * ns; // a second sample is possible and worth its time
    const bool saturated = d >= 0.95 * (double)ns;
    // (an all-distinct first sample means no heavy hitters: 2^20 keys then settle G up to ~5e8 -- 1 024 repeats -- to a few per
    // cent, a quarter of the full second sample's time: 0.38 -> 0.1 ms on config 3.  The skew branch below keeps the full sample:
    // config 5 at 2^20 read 4x too few groups and paid an overflow retry, 41 -> 152 ms)
    if (saturated && more) { ns = std::min<int64_t>((n / 16) & ~(int64_t)255, std::min<int64_t>(ns2_cap, (int64_t)1 << 20)); continue; }
    // (almost) all distinct: the model still reads G ~ s^2 / (2 * repeats) while the sample holds enough repeated keys
    // for that count to mean something (4M samples of 1e9 rows over 1e8 ids: 8e4 repeats, +-0.4 %)
    if (saturated && (double)ns - d < 1024.0) {
      g_est = (uint64_t)n;
    } else { // invert E[d] = G (1 - exp(-s / G))
      double lo = d, hi = (double)n;
      for (int it = 0; it < 60; it++) {
        const double g = 0.5 * (lo + hi);
        if (g * (1.0 - std::exp(-(double)ns / g)) < d) lo = g; else hi = g;
      }
      double est = hi;
      const double f1_uniform = (double)ns * std::exp(-(double)ns / hi);
      if (ns < n && f1 > 1.25 * f1_uniform + 64) { // skewed
        if (more) { ns = std::min<int64_t>((n / 16) & ~(int64_t)255, ns2_cap); continue; }
        est = std::max(est, std::sqrt((double)n / (double)ns) * f1 + (d - f1));
      }
      g_est = (uint64_t)std::min<double>((double)n, std::ceil(est * 1.1));
    }
    break;
  }
  return PLHIP_OK;
}

// Shape of the DIRECT strategy (row slices -> per-workgroup LDS tables -> merge): ok = the groups fit one LDS table.
struct DirectShape {
  bool ok = false;
  uint32_t lbits = 0, block = 1024, n_items = 0, mbits = 0;
  int64_t rows_per_item = 0;
};
DirectShape plan_direct(const Plan &P, uint64_t g_est, int key_bytes, int64_t n, int cus) {
  DirectShape D;
  const size_t LDS_MAX = 136 * 1024; // fast_path.hip allows 140 KB of dynamic LDS per workgroup
  // sparse tables (<= 1/8 full when LDS allows) keep most probes at the home slot
  // densest table accepted: 3/4 full (one pass at 2 500 groups and load 0.61 measured 0.081 ms against 0.136 ms
  // for the three-pass tile sort, scripts/sweep_groups.py)
  const uint64_t min_slots = (4 * g_est + 2) / 3 + 2;
  uint32_t lb = (uint32_t)std::max(6, ceil_log2(min_slots));
  if (lb > 15 || fast_table_bytes(P, lb, key_bytes) > LDS_MAX) return D;
  D.ok = true;
  while (lb < 15 && (1ull << lb) < 8 * g_est && fast_table_bytes(P, lb + 1, key_bytes) <= 64 * 1024) lb++;
  D.lbits = lb;
  // the fold kernel holds ~112 VGPRs: 4 waves per SIMD = 1024 threads per CU, as few workgroups as the
  // table size allows (each workgroup dumps its table once)
  const size_t tb = fast_table_bytes(P, D.lbits, key_bytes);
  // pre-instantiated shapes need < 64 VGPRs (8 waves per SIMD), the run-time shape ~112 (4 waves per SIMD)
  const bool st = fast_shape_id(P) >= 0;
  int wg_per_cu;
  // few, large workgroups: every workgroup initialises and dumps a table of its own
  if (tb <= 72 * 1024 && st) { D.block = 1024; wg_per_cu = 2; }
  else if (tb <= 72 * 1024) { D.block = 512; wg_per_cu = 2; }
  else { D.block = 1024; wg_per_cu = 1; }
  const int64_t step = (int64_t)D.block * (st ? 8 : 4); // rows one workgroup folds per loop trip
  const int64_t items = std::max<int64_t>(1, std::min<int64_t>((int64_t)cus * wg_per_cu, (n + step - 1) / step));
  D.rows_per_item = ((n + items - 1) / items + step - 1) / step * step;
  D.n_items = (uint32_t)((n + D.rows_per_item - 1) / D.rows_per_item);
  const uint64_t entries = (uint64_t)D.n_items * std::min<uint64_t>(g_est, 1ull << D.lbits);
  while (D.mbits < FP_MAX_MBITS && (entries >> D.mbits) > 4096) D.mbits++;
  return D;
}

// The LDS-only pipeline (fast_path.hip).  Sets `handled` when it produced the result; otherwise the caller
// runs the general path (stable order, wide key tuples, tiny inputs, more groups than one partition level
// brings into LDS, or a cardinality estimate that turned out too low).
int run_fast(Lowered &L, const std::vector<DevCol> &dk, uint32_t flags, int64_t groups_hint, int strategy,
             plhip_result_t **out, bool &handled, uint64_t &g_seen) {
  handled = false;
  g_seen = 0;
  Ctx &c = g_ctx;
  Plan &P = L.P;
  const int64_t n = P.n_rows;
  if ((flags & PLHIP_FLAG_MAINTAIN_ORDER) || L.KG.enabled || L.first_acc >= 0) return PLHIP_OK;
  const bool forced = strategy == PLHIP_STRATEGY_DIRECT || strategy == PLHIP_STRATEGY_TILESORT;
  if (n < 1 || (!forced && n < (1 << 15))) return PLHIP_OK;
  order_accs_by_source(L); // a source column is fetched once for consecutive accumulators
  const int key_bytes = packed_key_bytes(P, dk);
  int keymode = 0;
  uint64_t kxor = 0;
  if (P.single_key && !dk[0].validity) {
    const int w = dtype_width(dk[0].dtype);
    if (w == 4) { keymode = 1; kxor = dk[0].dtype == PLHIP_I32 ? 0x80000000ull : 0ull; }
    if (w == 8) { keymode = 2; kxor = dk[0].dtype == PLHIP_I64 ? 0x8000000000000000ull : 0ull; }
  }
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  uint32_t *dflags = (uint32_t *)c.flags.p;
  (void)hipGetLastError(); // launches below are checked with hipGetLastError: start clean

  // ---- cardinality: the caller's hint (e.g. the category count of a categorical key) or a sample ----------
  uint64_t g_est;
  if (groups_hint > 0) g_est = (uint64_t)std::min<int64_t>(groups_hint, n);
  else {
    int rc = estimate_groups_sampled(P, g_est);
    if (rc) return rc;
  }
  g_est = std::max<uint64_t>(g_est, 1);
  g_seen = g_est;

  // ---- shape of the pipeline -----------------------------------------------------------------------------
  const int cus = plhip_num_cus();
  const size_t LDS_MAX = 136 * 1024; // fast_path.hip allows 140 KB of dynamic LDS per workgroup
  const DirectShape DS = plan_direct(P, g_est, key_bytes, n, cus);
  bool direct = DS.ok;
  uint32_t lbits = DS.lbits, block = 1024;
  if (strategy == PLHIP_STRATEGY_TILESORT) direct = false;
  if (strategy == PLHIP_STRATEGY_DIRECT && !direct) return PLHIP_OK;
  uint32_t pbits = 8, np = 1u << pbits; // 9 when that is what brings the groups of a partition into LDS
  uint32_t w = 1, S = 1, n_pg = 1, has_null_pg = 0, n_items = 0, mbits = 0;
  int64_t rows_per_item = 0;
  const uint32_t n_tiles = (uint32_t)((n + TS_TILE - 1) / TS_TILE);
  if (direct) {
    block = DS.block; rows_per_item = DS.rows_per_item; n_items = DS.n_items; mbits = DS.mbits;
    S = n_items;
  } else {
    // groups per hash partition, with head-room for the sampling error and uneven partitions
    const double slack = groups_hint > 0 ? 1.12 : 1.3; // a hint is a bound: only the partitions' unevenness is left
    double gf = (double)g_est / np * slack + 8;
    uint32_t lb_max = 15;
    while (lb_max > 6 && fast_table_bytes(P, lb_max, key_bytes) > LDS_MAX) lb_max--;
    if (gf > 0.45 * (double)(1u << lb_max) && n >= (1 << 22)) { // shorter runs (16 rows per tile), twice the partitions
      pbits = TS_MAX_PBITS; np = 1u << pbits;
      gf = (double)g_est / np * slack + 8;
    }
    if (gf > 0.62 * (double)(1u << lb_max)) return PLHIP_OK; // needs a second partition level: the sort strategy
    lbits = 6;
    while (lbits < lb_max && gf * w > 0.45 * (double)(1u << lbits)) lbits++;
    while (lbits < lb_max && gf * w > 0.125 * (double)(1u << lbits) && fast_table_bytes(P, lbits + 1, key_bytes) <= 100 * 1024) lbits++;
    has_null_pg = (P.single_key && dk[0].validity) ? 1 : 0;
    n_pg = np / w + has_null_pg;
    S = 1; // one workgroup per partition: it writes the output columns itself
    n_items = n_pg * S;
    block = 1024;
  }
  const uint32_t lslots = (1u << lbits) + 66; // hashed slots + sentinel + null + 64 trash slots (fast_path.hip)
  const uint64_t out_cap = direct ? (uint64_t)lslots << mbits : (uint64_t)n_pg * lslots;
  if (out_cap >= (1ull << 31)) return PLHIP_OK; // the host signal word carries the group count in 31 bits: decline, the sort strategy / general path take it

  // ---- scratch ---------------------------------------------------------------------------------------------
  DumpRef D{};
  D.cap = lslots;
  D.item_stride = fast_dump_stride(P, lslots, key_bytes);
  // single level: partitions that hold a heavy hitter (or the null keys) are folded by several workgroups (fast.hpp, self_hot)
  const bool hot_self = !direct && S == 1 && n_pg <= block && !getenv("PLHIP_NOHOT");
  const uint32_t hot_extra = hot_self ? 256u : 0u; // chunks beyond the partitions' own (taken from a queue by workgroups that are done)
  const size_t dump_items = hot_self ? (size_t)hot_extra + n_pg : n_items;
  HIP_TRY(c.dump.reserve(D.item_stride * dump_items + 256));
  HIP_TRY(c.hdr.reserve(dump_items * sizeof(DumpHdr)));
  D.base = (uint8_t *)c.dump.p;
  D.hdr = (DumpHdr *)c.hdr.p;
  PartOut O{};
  if (!direct) {
    O.key_bytes = (uint32_t)key_bytes;
    size_t bytes = ((size_t)n * key_bytes + 511) & ~(size_t)255;
    size_t off_val[PLHIP_MAX_VALS], off_valid[PLHIP_MAX_VALS];
    for (int v = 0; v < P.n_vals; v++) {
      off_val[v] = bytes;
      bytes += ((size_t)n * dtype_width((int)P.vals[v].dtype) + 511) & ~(size_t)255;
      off_valid[v] = bytes;
      if (P.vals[v].validity) bytes += ((size_t)n + 511) & ~(size_t)255;
    }
    HIP_TRY(c.part.reserve(bytes + 256));
    HIP_TRY(c.offs.reserve((size_t)(np + 2) * n_tiles * 4));
    uint8_t *pb = (uint8_t *)c.part.p;
    O.keys = pb;
    for (int v = 0; v < P.n_vals; v++) {
      O.vals[v] = pb + off_val[v];
      O.valid[v] = P.vals[v].validity ? pb + off_valid[v] : nullptr;
    }
  }

  // ---- outputs (sized by the table capacities; G is only known at the end) ---------------------------------
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->on_device = true;
  R->device = c.device;
  OutPlan OP{};
  OP.n_outs = (int)L.outs.size();
  if (c.flags_clean != 1) HIP_TRY(hipMemsetAsync(dflags, 0, 64, c.stream));
  c.flags_clean = false;
  uint32_t *self_hot = (uint32_t *)((uint8_t *)dflags + FLAGS_HOT_OFF) + 2; // [-2] any, [-1] queue, rows[n_pg], done[n_pg]
  if (hot_self) {
    if (!c.hot_clean) HIP_TRY(hipMemsetAsync(self_hot - 2, 0, FLAGS_HOT_WORDS * 4, c.stream));
    c.hot_clean = false;
  }
  for (size_t i = 0; i < L.outs.size(); i++) {
    OutSpec o = L.outs[i];
    ResultCol col;
    col.dtype = (int)o.dtype;
    hipError_t e = g_pool.get(g_ctx.device, (size_t)out_cap * dtype_width((int)o.dtype), &col.values, &col.vcap, true);
    if (e == hipSuccess && o.validity) {
      e = g_pool.get(g_ctx.device, (size_t)(out_cap / 64 + 2) * 8, (void **)&col.validity, &col.bcap, true);
      if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (size_t)(out_cap / 64 + 2) * 8, c.stream);
    }
    R->cols.push_back(col);
    if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
    o.values = (uint8_t *)col.values;
    o.validity = col.validity;
    OP.outs[i] = o;
  }
  hipError_t e = hipSuccess;

  // ---- launches (events: [1] start, [4] tile sort done, [5] fold done, [2] merge done) ------------------------
  if (++c.seq == 0) c.seq = 1;
  const uint32_t seq = c.seq;
  const bool timed = c.timed();
  if (timed) { resolve_timing(); hipEventRecord(c.ev[1], c.stream); } // (the previous sample is read before its events are recorded again)
  if (direct) {
    bool quad = true;
    auto aligned = [](const uint8_t *p, int wd) { return ((uintptr_t)p % (size_t)(4 * wd)) == 0; };
    if (keymode && !aligned(dk[0].values, dtype_width(dk[0].dtype))) quad = false;
    for (int v = 0; v < P.n_vals; v++) if (!aligned(P.vals[v].values, dtype_width((int)P.vals[v].dtype))) quad = false;
    FoldRowsArgs A{};
    A.D = D; A.flags = dflags; A.rows_per_item = rows_per_item; A.kxor = kxor; A.lbits = lbits; A.keymode = keymode; A.mbits = mbits;
    launch_fold_rows(P, A, key_bytes, quad, n_items, block, c.stream);
    if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "k_fold_rows launch"); }
    c.strategy = "direct";
  } else {
    // runs of 4x the mean length and more are added up per partition
    TileSortLevel1 HL{nullptr, nullptr, 0u, 0u, hot_self ? self_hot : nullptr, 4u * (uint32_t)TS_TILE / np, 0u, hot_self ? self_hot - 2 : nullptr};
    launch_tile_sort(P, O, pbits, n_tiles, (uint32_t *)c.offs.p, keymode, kxor, key_bytes, c.stream, &HL);
    if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "k_tile_sort launch"); }
    if (timed) hipEventRecord(c.ev[4], c.stream);
    Plan P2 = P; // the fold pass reads the sorted copies
    for (int v = 0; v < P.n_vals; v++) { // the key specs stay: k_fold_runs reads the packed keys from A.pkeys
      P2.vals[v].values = O.vals[v]; P2.vals[v].validity = O.valid[v]; P2.vals[v].boff = 0; P2.vals[v].valid_bytes = 1;
    }
    FoldRunsArgs A{};
    A.pkeys = O.keys; A.out_cap = out_cap;
    A.D = D; A.flags = dflags; A.offs = (const uint32_t *)c.offs.p; A.lbits = lbits; A.np = np; A.w = w; A.n_pg = n_pg;
    A.has_null_pg = has_null_pg; A.S = S; A.n_tiles = n_tiles; A.pbits = pbits;
    A.host_sig = c.host_sig_dev; A.seq = seq;
    if (hot_self) {
      const char *dbg_hc = getenv("PLHIP_HOT_CHUNK"); // (tests: split at small sizes)
      A.self_hot = self_hot; A.max_extra = hot_extra;
      A.hot_chunk_rows = dbg_hc ? (uint32_t)std::max(256, atoi(dbg_hc)) : (uint32_t)std::max<int64_t>(8192, n / np / 4);
      A.hot_min_rows = 4 * A.hot_chunk_rows;
    }
    launch_fold_runs(P2, A, OP, key_bytes, block, c.stream);
    if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "k_fold_runs launch"); }
    c.strategy = "tilesort";
  }
  if (timed) hipEventRecord(c.ev[5], c.stream);
  MergeArgs M{};
  M.D = D; M.flags = dflags; M.out_cap = out_cap; M.lbits = lbits; M.S = S; M.mbits = mbits; M.pshift = direct ? 0u : pbits;
  M.host_sig = c.host_sig_dev; M.seq = seq;
  const bool merged = direct || S > 1;
  if (merged) {
    launch_merge_emit(P, M, OP, key_bytes, direct ? 1u : n_pg, 1024, c.stream);
    if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "k_merge_emit launch"); }
    if (timed) hipEventRecord(c.ev[2], c.stream);
  }
  e = hipGetLastError();
  uint32_t fl[16] = {0};
  if (e == hipSuccess) {
    TableRef t{nullptr, 0, dflags};
    int rc = wait_signal(t, seq, fl, c.allow_defer);
    if (rc) { return rc; }
    // for the next query: the last workgroup has zeroed the words it used when the run was clean (publish_done); otherwise, and
    // for the heavy-hitter words, one memset
    if (c.signal_seen && !fl[FLAG_OVERFLOW] && !hot_self) c.flags_clean = true;
    else if (hipMemsetAsync(dflags, 0, hot_self ? FLAGS_HOT_OFF + FLAGS_HOT_WORDS * 4 : 64, c.stream) == hipSuccess) {
      c.flags_clean = true;
      if (hot_self) c.hot_clean = true;
    }
  } else { return fail_hip(e, "group-by kernels"); }
  if (timed) c.timing_pending = direct ? 1 : (merged ? 3 : 2);
  if (fl[FLAG_OVERFLOW]) { return PLHIP_OK; } // estimate too low: the general path redoes the query
  const uint64_t G = fl[FLAG_NGROUPS];
  R->n_rows = (int64_t)G;
  if (c.deferred) R->pending = c.token;
  handled = true;
  return finish_result(R_own.release(), G, flags, out);
}


// ---------------------------------------------------------------------------------------------------------
// plhip_group_by_agg_noalloc: the DIRECT strategy for callers that cannot allocate (HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc).  Everything the kernels need beyond the caller's inputs and
// outputs -- the flag block, the dump headers and the compact table dumps that k_fold_rows hands to k_merge_emit --
// lives in ONE file-scope device array.  Host code never passes the array's name to a kernel: the name is a
// host-side stand-in; the device address comes from hipGetSymbolAddress, once per device (each GPU holds its
// own copy at its own address), and that pointer is what kernels, memsets and copies get.
// ---------------------------------------------------------------------------------------------------------
std::mutex g_scratch_mu;
uint8_t *g_scratch_ptr[64] = {nullptr}; // per device

// with `dev` current: the device address of this GPU's copy of scratch[]
int resolve_scratch(int dev, uint8_t **p) {
  if (dev < 0 || dev >= 64) return fail(PLHIP_ERR_INVALID, "device id out of range");
  std::lock_guard<std::mutex> g(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void *q = nullptr;
    const hipError_t e = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !q) return fail_hip(e == hipSuccess ? hipErrorInvalidSymbol : e, "hipGetSymbolAddress(scratch)");
    g_scratch_ptr[dev] = (uint8_t *)q;
  }
  *p = g_scratch_ptr[dev];
  return PLHIP_OK;
}

int run_direct_noalloc(Lowered &L, const std::vector<DevCol> &dk, int64_t groups_hint, const plhip_column_t *out_cols, int32_t n_out_cols,
                       int64_t *n_groups_out) {
  Ctx &c = g_ctx;
  Plan &P = L.P;
  const int64_t n = P.n_rows;
  if ((int)L.outs.size() != n_out_cols)
    return fail(PLHIP_ERR_SHAPE_MISMATCH, "the query has " + std::to_string(L.outs.size()) + " output columns, the caller passed " + std::to_string(n_out_cols));
  int64_t cap_rows = INT64_MAX;
  for (int i = 0; i < n_out_cols; i++) {
    const plhip_column_t &oc = out_cols[i];
    const OutSpec &o = L.outs[i];
    if (oc.location != PLHIP_DEVICE || oc.offset != 0 || oc.length < 0 || (oc.length > 0 && !oc.values))
      return fail(PLHIP_ERR_INVALID, "output columns must be device buffers (offset 0, length = capacity in rows)");
    if (oc.dtype != (int32_t)o.dtype) return fail(PLHIP_ERR_SHAPE_MISMATCH, "output column " + std::to_string(i) + " has the wrong dtype");
    if (o.validity && !oc.validity) return fail(PLHIP_ERR_INVALID, "output column " + std::to_string(i) + " can hold nulls: it needs a validity bitmap");
    cap_rows = std::min<int64_t>(cap_rows, oc.length);
  }
  if (n == 0) { *n_groups_out = 0; return PLHIP_OK; }
  order_accs_by_source(L);
  const int key_bytes = packed_key_bytes(P, dk);
  int keymode = 0;
  uint64_t kxor = 0;
  if (!dk[0].validity) {
    const int w = dtype_width(dk[0].dtype);
    if (w == 4) { keymode = 1; kxor = dk[0].dtype == PLHIP_I32 ? 0x80000000ull : 0ull; }
    if (w == 8) { keymode = 2; kxor = dk[0].dtype == PLHIP_I64 ? 0x8000000000000000ull : 0ull; }
  }
  const uint64_t g_est = (uint64_t)std::max<int64_t>(1, std::min<int64_t>(groups_hint, n));
  const DirectShape DS = plan_direct(P, g_est, key_bytes, n, plhip_num_cus());
  if (!DS.ok) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc: " + std::to_string(g_est) + " groups do not fit one LDS table (the DIRECT strategy); use plhip_group_by_agg_opts");
  const uint32_t lslots = (1u << DS.lbits) + 66;
  // scratch layout: flag block | dump headers | dumps
  DumpRef D{};
  D.cap = lslots;
  D.item_stride = fast_dump_stride(P, lslots, key_bytes);
  const size_t off_hdr = FLAGS_BYTES, off_dump = (off_hdr + (size_t)DS.n_items * sizeof(DumpHdr) + 255) & ~(size_t)255;
  const size_t need = off_dump + (size_t)D.item_stride * DS.n_items + 256;
  if (need > (size_t)PLHIP_NOALLOC_SCRATCH_BYTES)
    return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc: the query needs " + std::to_string(need) + " bytes of scratch, the static buffer holds " +
                                       std::to_string((size_t)PLHIP_NOALLOC_SCRATCH_BYTES) + "; use plhip_group_by_agg_opts");
  uint8_t *sp = c.na_scratch; // the device address (hipGetSymbolAddress), never the array's name
  uint32_t *dflags = (uint32_t *)sp;
  D.hdr = (DumpHdr *)(sp + off_hdr);
  D.base = sp + off_dump;
  const uint64_t out_cap = (uint64_t)std::min<int64_t>(cap_rows, (int64_t)((uint64_t)lslots << DS.mbits));
  (void)hipGetLastError();
  // the array keeps the last call's contents: what must start at zero is cleared here (the dumps and their headers
  // are written whole by k_fold_rows before k_merge_emit reads them)
  HIP_TRY(hipMemsetAsync(dflags, 0, 64, c.stream));
  OutPlan OP{};
  OP.n_outs = n_out_cols;
  for (int i = 0; i < n_out_cols; i++) {
    OutSpec o = L.outs[i];
    o.values = (uint8_t *)out_cols[i].values;
    o.validity = o.validity ? (uint8_t *)out_cols[i].validity : nullptr;
    if (o.validity) HIP_TRY(hipMemsetAsync(o.validity, 0, (size_t)(out_cols[i].length / 64 + 1) * 8, c.stream));
    OP.outs[i] = o;
  }
  if (++c.seq == 0) c.seq = 1;
  const uint32_t seq = c.seq;
  bool quad = true;
  auto aligned = [](const uint8_t *p, int wd) { return ((uintptr_t)p % (size_t)(4 * wd)) == 0; };
  if (keymode && !aligned(dk[0].values, dtype_width(dk[0].dtype))) quad = false;
  for (int v = 0; v < P.n_vals; v++) if (!aligned(P.vals[v].values, dtype_width((int)P.vals[v].dtype))) quad = false;
  resolve_timing();
  hipEventRecord(c.ev[1], c.stream);
  FoldRowsArgs A{};
  A.D = D; A.flags = dflags; A.rows_per_item = DS.rows_per_item; A.kxor = kxor; A.lbits = DS.lbits; A.keymode = keymode; A.mbits = DS.mbits;
  launch_fold_rows(P, A, key_bytes, quad, DS.n_items, DS.block, c.stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "k_fold_rows launch");
  hipEventRecord(c.ev[5], c.stream);
  MergeArgs M{};
  M.D = D; M.flags = dflags; M.out_cap = out_cap; M.lbits = DS.lbits; M.S = DS.n_items; M.mbits = DS.mbits; M.pshift = 0u;
  M.host_sig = c.host_sig_dev; M.seq = seq;
  launch_merge_emit(P, M, OP, key_bytes, 1u, 1024, c.stream);
  if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "k_merge_emit launch");
  hipEventRecord(c.ev[2], c.stream);
  uint32_t fl[16] = {0};
  TableRef t{nullptr, 0, dflags};
  const int rc = wait_signal(t, seq, fl);
  c.flags_clean = false;
  if (rc) return rc;
  c.timing_pending = 1;
  c.strategy = "direct";
  if (fl[FLAG_OVERFLOW])
    return fail(PLHIP_ERR_COMPUTE, "plhip_group_by_agg_noalloc: more groups than n_groups_hint promised, or than the output buffers hold");
  *n_groups_out = (int64_t)fl[FLAG_NGROUPS];
  return PLHIP_OK;
}

// The dense-code path (dense_path.hip, dense.hpp): group_tuples_perfect's idea (crates/polars-core/src/frame/group_by/perfect.rs:
// 24-185, 190-227) -- a Categorical key's physical codes ARE the group indices -- on LDS tables.  Taken when the caller
// declares the single u32 key column dense (PLHIP_FLAG_KEYS_DENSE: codes < n_groups_hint = the rev-map's length).  A code
// outside the promise raises the overflow flag and the hashed strategies redo the query: time, never correctness.
// key_bias: a plain 32-bit integer key whose measured range [min, max] is compact is a dense code after subtracting its minimum
// (group_by_agg_impl): codes = raw bits - key_bias, groups_hint = max - min + 1.
// With `comm` (the sharded dense path, dense.hpp DenseXchg): every rank of the communicator makes this call with the same schema
// and hint.  Whatever depends on the schema alone declines as usual (handled = false on EVERY rank, nothing was exchanged);
// what depends on this rank's data (alignment, an empty shard, a code beyond the hint) must not leave the peers waiting: the
// rank takes part in the exchange with a flagged header, and all ranks decline together afterwards.
int run_dense(const Lowered &L0, const std::vector<DevCol> &dk, uint32_t flags, int64_t groups_hint, bool forced, plhip_result_t **out, bool &handled,
              uint32_t key_bias = 0, plhip_comm *comm = nullptr) {
  handled = false;
  Ctx &c = g_ctx;
  const int64_t n = L0.P.n_rows;
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  auto lap = [&](float &slot) { const auto t1 = clk::now(); slot += std::chrono::duration<float, std::milli>(t1 - t0).count(); t0 = t1; };
  if (comm) forced = true;
  bool local_bad = false; // (comm only)
  if ((flags & PLHIP_FLAG_MAINTAIN_ORDER) || L0.KG.enabled || L0.first_acc >= 0) return PLHIP_OK;
  if (!L0.P.single_key || (dk[0].dtype != PLHIP_U32 && dk[0].dtype != PLHIP_I32) || groups_hint < 1 || groups_hint > 0x7FFFFFF0ll) return PLHIP_OK;
  if (!comm && (n < 1 || (!forced && n < (1 << 15)))) return PLHIP_OK;
  bool misaligned = ((uintptr_t)dk[0].values & 15u) != 0; // the row loop reads aligned quads
  for (int v = 0; v < L0.P.n_vals; v++)
    if (((uintptr_t)L0.P.vals[v].values % (size_t)(4 * dtype_width((int)L0.P.vals[v].dtype))) != 0) misaligned = true;
  if (misaligned && n > 0) { if (!comm) return PLHIP_OK; local_bad = true; }
  Lowered L = L0; // (a count is added below: the hashed strategies keep the query as it was lowered)
  Plan &P = L.P;
  const int cnt_acc = add_acc(P, ACC_CNT_ALL, CONV_NONE, -1, 4, 0); // a code that never occurs is not a group
  if (cnt_acc < 0) return PLHIP_OK;
  order_accs_by_source(L);
  DenseArgs A{};
  A.hint = (uint32_t)groups_hint; A.n_slots = A.hint + 1u;
  // ---- parts: one per value column; the group count rides with the cheapest one ----------------------------------
  int cnt_acc_idx = 0;
  for (int a = 0; a < P.n_accs; a++) if (P.accs[a].op == ACC_CNT_ALL) cnt_acc_idx = a;
  int part_of_val[PLHIP_MAX_VALS];
  for (int v = 0; v < P.n_vals; v++) part_of_val[v] = -1;
  uint32_t row_bytes[DENSE_MAX_PARTS] = {0};
  bool has_valued_acc = false;
  for (int a = 0; a < P.n_accs; a++) {
    const int src = P.accs[a].src;
    if (src < 0) continue;
    has_valued_acc = true;
    if (part_of_val[src] < 0) {
      if (A.n_parts >= DENSE_MAX_PARTS) return PLHIP_OK;
      part_of_val[src] = (int)A.n_parts;
      A.part[A.n_parts].src = src;
      row_bytes[A.n_parts] = 4u + (uint32_t)dtype_width((int)P.vals[src].dtype);
      A.n_parts++;
    }
  }
  if (!has_valued_acc) { A.part[0].src = -1; row_bytes[0] = 4u; A.n_parts = 1; }
  int cheapest = 0;
  for (uint32_t p = 1; p < A.n_parts; p++) if (row_bytes[p] < row_bytes[cheapest]) cheapest = (int)p;
  auto narrow32 = [&](int a) { // an integer sum may live in 32 bits when every output that reads it wraps in 32 bits anyway
    if (P.accs[a].op != ACC_ADD_U64) return false;
    bool used = false;
    for (auto &o : L.outs) {
      if (o.kind == OUT_KEY) continue;
      if (o.acc == a) { used = true; if (o.kind != OUT_RAW_U64_WRAP32 && (o.kind != OUT_SUM_INT || dtype_width((int)o.dtype) != 4)) return false; }
    }
    return used;
  };
  auto referenced = [&](int a) { // (a nullable column's valid count is lowered for every aggregation; SUM never reads it)
    if (a == cnt_acc_idx) return true;
    for (auto &o : L.outs) {
      if (o.kind == OUT_KEY) continue;
      if (o.acc == a || o.cnt_valid == a || o.cnt_all == a) return true;
    }
    return false;
  };
  for (int a = 0; a < P.n_accs; a++) {
    const int src = P.accs[a].src;
    const uint32_t op = P.accs[a].op;
    if (op == ACC_FIRST_ROW) return PLHIP_OK;
    if (!referenced(a)) continue;
    // (the row count rides with the cheapest part.  A part of its own for it -- so that a lone f64 sum keeps its straight-line kind --
    // was measured: the count-only part runs the generic kind and is instruction-bound, sum(v3) by id3 took 0.181 ms against 0.126 ms
    // for the hashed tile sort, which is where AUTO now sends such queries)
    DensePart &D = A.part[src < 0 ? cheapest : part_of_val[src]];
    if (D.n_accs >= DENSE_MAX_PART_ACCS) return PLHIP_OK;
    D.acc[D.n_accs] = (uint8_t)a;
    D.lw[D.n_accs] = (op == ACC_CNT_ALL || op == ACC_CNT_VALID || narrow32(a)) ? 4 : 8;
    D.n_accs++;
  }
  A.cnt_acc = (uint32_t)cnt_acc_idx;
  A.key_bias = key_bias;
  // A nullable f64 column's [CNT_VALID, ADD_F64] (its mean) is 12 bytes of state per group.  When that is what sets the number of
  // ranges, the count moves into a part of its own that reads the key and the column's validity bits only (DENSE_KIND_CNTV):
  // every part is back to 8 bytes per group (fewer ranges for ALL parts) and the counting part streams 4 bytes a row.
  if (!getenv("PLHIP_DENSE_NO_SPLIT")) {
    uint32_t st_max = 0, st_second = 0;
    auto state_of = [&](const DensePart &D) { uint32_t st = 0; for (uint32_t i = 0; i < D.n_accs; i++) st += D.lw[i]; return st; };
    for (uint32_t p = 0; p < A.n_parts; p++) { const uint32_t st = state_of(A.part[p]); if (st > st_max) { st_second = st_max; st_max = st; } else st_second = std::max(st_second, st); }
    const uint32_t n_parts0 = A.n_parts;
    for (uint32_t p = 0; p < n_parts0 && A.n_parts < DENSE_MAX_PARTS; p++) {
      DensePart &D = A.part[p];
      if (D.src < 0 || D.n_accs != 2 || state_of(D) != st_max || st_max != 12) continue;
      const ValSpec &V = P.vals[D.src];
      int ic = -1, is = -1;
      for (int i = 0; i < 2; i++) {
        if (P.accs[D.acc[i]].op == ACC_CNT_VALID && D.lw[i] == 4) ic = i;
        if (P.accs[D.acc[i]].op == ACC_ADD_F64 && D.lw[i] == 8) is = i;
      }
      if (ic < 0 || is < 0 || V.dtype != PLHIP_F64 || !V.validity) continue;
      const uint64_t lds_cap = 160 * 1024 - 512;
      if (((uint64_t)A.n_slots * 12 + lds_cap - 1) / lds_cap <= ((uint64_t)A.n_slots * 8 + lds_cap - 1) / lds_cap) continue; // no range saved
      DensePart &C = A.part[A.n_parts];
      C = DensePart{};
      C.src = D.src; C.n_accs = 1; C.acc[0] = D.acc[ic]; C.lw[0] = 4;
      row_bytes[A.n_parts] = 4u;
      const uint8_t sa = D.acc[is];
      D.n_accs = 1; D.acc[0] = sa; D.lw[0] = 8;
      A.n_parts++;
    }
    (void)st_second;
  }
  // ---- ranges: the largest part's arrays cut into pieces of one CU's LDS ------------------------------------------
  const size_t LDS_CAP = 160 * 1024 - 512;
  uint32_t state_max = 0;
  for (uint32_t p = 0; p < A.n_parts; p++) {
    uint32_t st = 0;
    for (uint32_t i = 0; i < A.part[p].n_accs; i++) st += A.part[p].lw[i];
    state_max = std::max(state_max, st);
  }
  const char *dbg_r = getenv("PLHIP_DENSE_R");
  uint32_t R = (uint32_t)(((uint64_t)A.n_slots * state_max + LDS_CAP - 1) / LDS_CAP);
  if (dbg_r) R = std::max<uint32_t>(R, (uint32_t)atoi(dbg_r));
  uint32_t rep = dk[0].validity ? 64u : 0u; // a nullable key: 64 replicas of the null slot behind a range's entries (dense_path.hip)
  // More groups than DENSE_MAX_RANGES ranges hold (q3 at 1e8 .. 1e9 rows: 1e6 .. 1e7 ids): PARTITION MODE.  The rows are scattered
  // by code range first -- exact and hash-free: partition = code / gpr, sized by one histogram pass over the key column -- and
  // every partition is then a dense problem of ONE range: read, write partitioned, read partitioned; three passes over the row
  // bytes where the hashed two-level pipeline makes five.  Needs a key without nulls and up to 2^TS_MAX_PBITS partitions.
  bool pmode = false;
  uint32_t np_pow2 = 0, pbits = 0;
  const uint64_t n_pad = (uint64_t)n + ((uint64_t)1 << TS_MAX_PBITS) * TS_TILE; // every partition starts on a tile boundary
  for (;; R++) {
    if (R > DENSE_MAX_RANGES) { // too many groups: every row would be read R times
      // (up to ~1.2e6 groups the hashed single-level tile sort is the faster pipeline -- measured, q3 at 1e8 rows / 1e6 ids: 1.20 ms
      // against 1.55 here, the scatter's returning LDS atomics meet on 50 counters -- beyond it the alternative makes two partition
      // levels: q3 at 1e9 rows / 1e7 ids 23.3 -> 18.8 ms)
      static const bool part_any = getenv("PLHIP_DENSE_PARTITION_ANY") != nullptr; // (tests: the mode at small sizes)
      // (sharded: whether the mode applies is a property of the schema and the hint, so every rank decides alike; what only THIS
      // rank's rows rule out -- a nullable key, too many rows -- is flagged in its slot headers and all ranks fall back together)
      if (getenv("PLHIP_NO_DENSE_PARTITION")) return PLHIP_OK;
      if (!comm && !part_any && !forced && (n < (1 << 22) || A.n_slots <= 1200000u)) return PLHIP_OK;
      A.gpr = (uint32_t)((LDS_CAP - 16 * DENSE_MAX_PART_ACCS) / state_max) & ~3u;
      const uint64_t np = ((uint64_t)A.n_slots + A.gpr - 1) / A.gpr;
      if (np > (1u << TS_MAX_PBITS)) return PLHIP_OK;
      if (dk[0].validity || n_pad > 0xFFFFFFFFull) { if (!comm) return PLHIP_OK; local_bad = true; }
      rep = 0; // (no null replicas in this mode: a nullable key has just been flagged)
      R = (uint32_t)np;
      while ((1u << pbits) < R) pbits++;
      np_pow2 = 1u << pbits;
      pmode = true;
      break;
    }
    A.gpr = (((A.n_slots + R - 1) / R) + 3u) & ~3u;
    const bool fits = (uint64_t)(A.gpr + rep) * state_max + 16 * DENSE_MAX_PART_ACCS <= LDS_CAP;
    if (fits && (uint64_t)(R - 1) * A.gpr < A.n_slots) break;
    if (fits && (uint64_t)(R - 1) * A.gpr >= A.n_slots) { R = (A.n_slots + A.gpr - 1) / A.gpr; break; }
  }
  A.R = R;
  // a handful of groups in ONE range: the hashed DIRECT strategy is as fast there (one LDS table either way) and its merge is
  // cheaper than adding up a partial array per workgroup (measured, q1 at 1e7 rows: 0.062 against 0.070 ms per query)
  if (!forced && R == 1 && A.n_slots <= 4096) return PLHIP_OK;
  size_t lds_bytes = 0;
  for (uint32_t p = 0; p < A.n_parts; p++) {
    DensePart &D = A.part[p];
    uint32_t lo = 0;
    uint64_t dof = 0;
    for (uint32_t i = 0; i < D.n_accs; i++) {
      D.loff[i] = lo; lo += ((A.gpr + rep) * D.lw[i] + 15u) & ~15u;
      D.doff[i] = (uint32_t)dof; dof += ((uint64_t)(pmode ? 1u : R) * A.gpr * D.lw[i] + 15u) & ~15ull; // (partition mode: a slice holds ONE range)
    }
    if (dof > 0xFFFFFFFFull) return PLHIP_OK;
    D.slice_stride = (dof + 255) & ~255ull;
    lds_bytes = std::max<size_t>(lds_bytes, lo);
    // straight-line kinds; validity bitmaps ride along when their bit offsets are multiples of 4 (a quad's bits in one byte)
    D.kind = DENSE_KIND_RT;
    D.pad0 = 0;
    const bool kv_ok = !dk[0].validity || (dk[0].boff & 3) == 0;
    if (D.src >= 0 && kv_ok && !getenv("PLHIP_DENSE_RT")) {
      const ValSpec &V = P.vals[D.src];
      const bool vv = V.validity != nullptr;
      const uint32_t vd = V.dtype;
      const bool i4 = vd == PLHIP_I32 || vd == PLHIP_U32;
      auto is = [&](uint32_t i, uint32_t op, uint32_t lw) { return P.accs[D.acc[i]].op == op && D.lw[i] == lw; };
      if (!vv || (!pmode && !V.valid_bytes && (V.boff & 3) == 0)) { // (the partitioned copies carry a validity BYTE per row: run-time kind)
        if (i4 && D.n_accs == 2 && is(0, ACC_CNT_ALL, 4) && is(1, ACC_ADD_U64, 4)) D.kind = DENSE_KIND_CNT_SUM32;
        else if (i4 && D.n_accs == 1 && is(0, ACC_ADD_U64, 4)) D.kind = DENSE_KIND_SUM32;
        else if (vd == PLHIP_F64 && D.n_accs == 1 && is(0, ACC_ADD_F64, 8)) D.kind = DENSE_KIND_SUMF64;
        else if (vd == PLHIP_F64 && vv && D.n_accs == 2 && is(0, ACC_CNT_VALID, 4) && is(1, ACC_ADD_F64, 8)) D.kind = DENSE_KIND_CNTV_SUMF64;
        else if (vv && D.n_accs == 1 && is(0, ACC_CNT_VALID, 4)) D.kind = DENSE_KIND_CNTV;
        if (D.kind != DENSE_KIND_RT) D.pad0 = vv ? 1u : 0u;
      }
    }
  }
  if (lds_bytes > LDS_CAP) return PLHIP_OK;
  // A value column whose accumulators have no straight-line kind (min / max, 64-bit integer sums, means of integers) is folded by
  // the run-time kind: one generic, branchy update per accumulator and row.  Measured on the harness's q7 (max(v1), min(v2) by id3,
  // 1e7 rows): 0.259 ms here against 0.142 ms for the hashed tile sort -- so unless the caller forces the strategy (or the rows are
  // sharded / range-partitioned, where the alternatives cost more) such a query is left to the hashed pipelines.
  if (!forced && !pmode && !getenv("PLHIP_DENSE_ALLOW_RT"))
    for (uint32_t p = 0; p < A.n_parts; p++)
      if (A.part[p].src >= 0 && A.part[p].kind == DENSE_KIND_RT) return PLHIP_OK;
  // ---- workgroups: per XCD `slots` of them, cut into groups of R (the R ranges of one row slice) ------------------
  const int cus = plhip_num_cus();
  A.xcds = 8;
  const uint32_t wg_per_cu = lds_bytes <= 76 * 1024 ? 2u : 1u;
  const char *dbg_slots = getenv("PLHIP_DENSE_SLOTS");
  const uint32_t slots = dbg_slots ? (uint32_t)atoi(dbg_slots) : (uint32_t)std::max(1, cus / 8) * wg_per_cu;
  // group slots: every XCD lane holds slots / R of them; they are dealt to the parts in proportion to the bytes a part
  // reads per row, so that every workgroup streams about the same number of bytes
  const uint32_t n_gs = pmode ? A.n_parts : (slots / R) * A.xcds;
  if (n_gs < A.n_parts) return PLHIP_OK;
  uint32_t nsl[DENSE_MAX_PARTS];
  for (uint32_t p = 0; p < A.n_parts; p++) nsl[p] = 1;
  if (getenv("PLHIP_DENSE_EQ")) for (uint32_t p = 0; p < A.n_parts; p++) row_bytes[p] = 8;
  {
    // With validity bitmaps in play a row costs instructions more than bytes (bit extraction, null slots, masked atomics), and
    // about the same in every part: the slices are then dealt out nearly evenly (measured, q3's NA variant: fold 139 -> 119 us;
    // without nulls the byte-proportional split stays ahead: 61 against 65 us)
    bool nullable = dk[0].validity != nullptr;
    for (uint32_t p = 0; p < A.n_parts; p++) if (A.part[p].src >= 0 && P.vals[A.part[p].src].validity) nullable = true;
    const char *rc_env = getenv("PLHIP_DENSE_ROWCOST");
    const uint32_t row_cost = rc_env ? (uint32_t)atoi(rc_env) : (nullable ? 32u : 0u);
    for (uint32_t p = 0; p < A.n_parts; p++) row_bytes[p] += row_cost;
  }
  for (uint32_t gi = A.n_parts; gi < n_gs; gi++) { // the next slot goes to the part with the most bytes per slice
    uint32_t best = 0;
    for (uint32_t p = 1; p < A.n_parts; p++)
      if ((uint64_t)row_bytes[p] * nsl[best] > (uint64_t)row_bytes[best] * nsl[p]) best = p;
    nsl[best]++;
  }
  // small inputs: no more slices than whole batches
  const int64_t batches = (n + DENSE_BATCH - 1) / DENSE_BATCH;
  uint32_t slot = 0;
  uint64_t dump_bytes = 0;
  for (uint32_t p = 0; p < A.n_parts; p++) {
    DensePart &D = A.part[p];
    D.n_slices = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(nsl[p], batches));
    D.slot0 = slot;
    slot += D.n_slices;
    const int64_t per = (n + D.n_slices - 1) / D.n_slices;
    D.rows_per_slice = (per + DENSE_BATCH - 1) / DENSE_BATCH * DENSE_BATCH;
    if (!pmode && (uint64_t)D.rows_per_slice * 8u >= 0xFFFFFFFFull) { // the row loop addresses a slice with 32-bit byte offsets
      if (!comm) return PLHIP_OK;
      local_bad = true;
    }
    D.dump_off = dump_bytes;
    dump_bytes += D.slice_stride * D.n_slices;
  }
  const uint32_t q = ((slot + A.xcds - 1) / A.xcds) * R; // positions per XCD lane
  uint32_t n_wgs = q * A.xcds;
  if (pmode) { // S slices per (part, partition): about two workgroups per CU and part in flight, whatever the partition count
    A.pmode = 1;
    A.S = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, ((uint64_t)2 * cus * wg_per_cu + (uint64_t)R * A.n_parts - 1) / ((uint64_t)R * A.n_parts)));
    dump_bytes = 0;
    for (uint32_t p = 0; p < A.n_parts; p++) {
      DensePart &D = A.part[p];
      D.n_slices = A.S; D.slot0 = 0; D.rows_per_slice = 0;
      D.dump_off = dump_bytes;
      dump_bytes += D.slice_stride * A.S * R;
    }
    n_wgs = A.n_parts * R * A.S;
  }
  HIP_TRY(c.dump.reserve(dump_bytes + 256));
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  A.dump = (uint8_t *)c.dump.p;
  uint32_t *dflags = (uint32_t *)c.flags.p;
  A.flags = dflags;
  uint64_t out_cap = A.n_slots; // (< 2^31: groups_hint is bounded in run_dense's first line, and the signal word holds 31 bits)
  // ---- sharded: the owners' slots and the view k_dense_merge gets of the received ones ---------------------------------
  DenseXchg X{};
  DenseArgs A2{};
  const bool fold_runs = n > 0 && !local_bad;
  if (comm) {
    const uint32_t world = (uint32_t)comm->world;
    X.world = world;
    X.per = (((A.n_slots + world - 1) / world) + 3u) & ~3u;
    uint64_t off = DENSE_SLOT_HEADER;
    for (uint32_t p = 0; p < A.n_parts; p++)
      for (uint32_t i = 0; i < A.part[p].n_accs; i++) { X.xoff[p][i] = (uint32_t)off; off += ((uint64_t)X.per * A.part[p].lw[i] + 15u) & ~15ull; }
    if (off > 0xFFFFFF00ull) return PLHIP_OK; // (a property of the schema and the hint: every rank declines)
    X.slot_bytes = (off + 255) & ~255ull;
    X.local_bad = local_bad ? 1u : 0u;
    HIP_TRY(c.xsend.reserve(X.slot_bytes * world + 256));
    HIP_TRY(c.xrecv.reserve(X.slot_bytes * world + 256));
    X.send = (uint8_t *)c.xsend.p;
    if (!fold_runs) for (uint32_t p = 0; p < A.n_parts; p++) A.part[p].n_slices = 0; // nothing folded: the slots carry the accumulators' initial values
    A2 = A;
    A2.dump = (uint8_t *)c.xrecv.p;
    A2.R = 1; A2.gpr = X.per;
    A2.pmode = 0; A2.S = 0; // (the received slots are plain dense arrays, whatever the local pass looked like)
    A2.slot_base = (uint32_t)comm->rank * X.per;
    A2.n_slots = A2.slot_base >= A.n_slots ? 0u : std::min<uint32_t>(X.per, A.n_slots - A2.slot_base);
    A2.n_hdr = world;
    for (uint32_t p = 0; p < A2.n_parts; p++) {
      DensePart &D = A2.part[p];
      D.dump_off = 0; D.slice_stride = X.slot_bytes; D.n_slices = world;
      for (uint32_t i = 0; i < D.n_accs; i++) D.doff[i] = X.xoff[p][i];
    }
    out_cap = X.per;
  }
  // ---- outputs --------------------------------------------------------------------------------------------------
  std::unique_ptr<plhip_result> R_(new (std::nothrow) plhip_result());
  if (!R_) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R_->on_device = true; R_->device = c.device;
  // every early return below (a failed launch, a failed exchange, a failed wait) drains the stream before R_ hands its pooled
  // buffers back: kernels already queued may still write them (disarmed once wait_signal has synchronised the stream)
  StreamDrain drain(c.stream);
  OutPlan OP{};
  OP.n_outs = (int)L.outs.size();
  (void)hipGetLastError();
  if (c.flags_clean == 0) HIP_TRY(hipMemsetAsync(dflags, 0, 64, c.stream));
  c.flags_clean = false;
  for (size_t i = 0; i < L.outs.size(); i++) {
    OutSpec o = L.outs[i];
    ResultCol col;
    col.dtype = (int)o.dtype;
    hipError_t e = g_pool.get(g_ctx.device, (size_t)out_cap * dtype_width((int)o.dtype), &col.values, &col.vcap, true);
    if (e == hipSuccess && o.validity) {
      e = g_pool.get(g_ctx.device, (size_t)(out_cap / 64 + 2) * 8, (void **)&col.validity, &col.bcap, true);
      if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (size_t)(out_cap / 64 + 2) * 8, c.stream);
    }
    R_->cols.push_back(col);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(output)");
    o.values = (uint8_t *)col.values; o.validity = col.validity;
    OP.outs[i] = o;
  }
  if (++c.seq == 0) c.seq = 1;
  const uint32_t seq = c.seq;
  const bool timed = c.timed();
  if (timed) { resolve_timing(); hipEventRecord(c.ev[1], c.stream); } // (the previous sample is read before its events are recorded again)
  Plan P2 = P; // partition mode: the fold and the merge read the partitioned copies
  if (pmode && fold_runs) {
    const uint32_t n_tiles = (uint32_t)((n + TS_TILE - 1) / TS_TILE);
    size_t off_totals = (size_t)np_pow2 * n_tiles * 4, off_start = off_totals + (size_t)np_pow2 * 4;
    size_t off_cnt = (off_start + (size_t)(np_pow2 + 1) * 4 + 63) & ~(size_t)63;
    HIP_TRY(c.items.reserve(off_cnt + 64));
    uint8_t *sb = (uint8_t *)c.items.p;
    uint32_t *counts = (uint32_t *)sb, *totals = (uint32_t *)(sb + off_totals), *pstart = (uint32_t *)(sb + off_start), *cnt = (uint32_t *)(sb + off_cnt);
    PartOut PO{};
    PO.key_bytes = 4;
    size_t bytes = ((size_t)n_pad * 4 + 511) & ~(size_t)255;
    size_t off_val[PLHIP_MAX_VALS], off_valid[PLHIP_MAX_VALS];
    for (int v = 0; v < P.n_vals; v++) {
      off_val[v] = bytes;
      bytes += ((size_t)n_pad * dtype_width((int)P.vals[v].dtype) + 511) & ~(size_t)255;
      off_valid[v] = bytes;
      if (P.vals[v].validity) bytes += ((size_t)n_pad + 511) & ~(size_t)255;
    }
    HIP_TRY(c.part.reserve(bytes + 256));
    uint8_t *pa = (uint8_t *)c.part.p;
    PO.keys = pa;
    for (int v = 0; v < P.n_vals; v++) { PO.vals[v] = pa + off_val[v]; PO.valid[v] = P.vals[v].validity ? pa + off_valid[v] : nullptr; }
    const uint32_t recip = (uint32_t)((((uint64_t)1 << 32) + A.gpr - 1) / A.gpr);
    launch_dense_part_hist(P, key_bias, A.gpr, recip, R, np_pow2, A.hint, n_tiles, counts, dflags, c.stream);
    launch_part_scan(counts, np_pow2, n_tiles, totals, c.stream);
    launch_part_plan(totals, np_pow2, pstart, nullptr, cnt, cnt + 1, 1u << 30, c.stream, (uint32_t)TS_TILE);
    TileSortLevel1 S1{pstart, totals, np_pow2, 0u};
    S1.range_gpr = A.gpr; S1.range_bias = key_bias; S1.range_recip = recip; S1.range_np = R;
    launch_tile_sort(P, PO, pbits, n_tiles, counts, 1 /* the key's raw 32 bits */, 0ull, 4, c.stream, &S1, true);
    hipError_t pe = hipGetLastError();
    if (pe != hipSuccess) return fail_hip(pe, "range partition launch");
    if (timed) hipEventRecord(c.ev[4], c.stream);
    P2.n_rows = (int64_t)n_pad;
    P2.keys[0].values = PO.keys; P2.keys[0].validity = nullptr; P2.keys[0].boff = 0;
    for (int v = 0; v < P.n_vals; v++) { P2.vals[v].values = PO.vals[v]; P2.vals[v].validity = PO.valid[v]; P2.vals[v].boff = 0; P2.vals[v].valid_bytes = 1; }
    A.part_start = pstart; A.part_rows = totals;
  }
  if (fold_runs) launch_dense_fold(P2, A, n_wgs, lds_bytes, c.stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "k_dense_fold launch");
  if (timed) hipEventRecord(c.ev[5], c.stream);
  if (comm) {
    plhip_sharded_stats_t &S = c.shard_stats;
    lap(S.partial_ms);
    launch_dense_scatter(P, A, X, c.stream);
    if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "k_dense_scatter launch");
    lap(S.pack_ms);
    std::string err;
    const int xrc = comm_all_to_all(comm, c.xsend.p, c.xrecv.p, (size_t)X.slot_bytes, c.stream, err);
    if (xrc) return fail(xrc, err);
    lap(S.exchange_ms);
    S.shuffle_bytes = (int64_t)(X.slot_bytes * X.world);
    S.host_syncs = comm->nccl ? 0 : 1; // (the host's transport drains the stream before it moves the bytes)
  }
  launch_dense_merge(P2, comm ? A2 : A, OP, out_cap, c.host_sig_dev, seq, c.stream);
  if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "k_dense_merge launch");
  if (timed) hipEventRecord(c.ev[2], c.stream);
  uint32_t fl[16] = {0};
  TableRef t{nullptr, 0, dflags};
  const int rc = wait_signal(t, seq, fl, c.allow_defer && !comm);
  if (rc) return rc;
  drain.s = nullptr; // (the stream is idle: wait_signal synchronised it -- or the frame carries the stream's token, below)
  if (comm) { c.shard_stats.host_syncs++; lap(c.shard_stats.merge_ms); }
  if (!fl[FLAG_OVERFLOW] && c.signal_seen) c.flags_clean = 2; // (the signal was seen: the kernel's last workgroup has zeroed its counters)
  else if (hipMemsetAsync(dflags, 0, 64, c.stream) == hipSuccess) c.flags_clean = true;
  if (timed) c.timing_pending = pmode ? 5 : 4;
  c.strategy = pmode ? "dense_part" : "dense";
  if (getenv("PLHIP_DENSE_VERBOSE"))
    fprintf(stderr, "dense: parts %u R %u gpr %u lds %zu B wgs %u dump %llu B slices %u %u kinds %u %u\n", A.n_parts, R, A.gpr, lds_bytes, n_wgs,
            (unsigned long long)dump_bytes, A.part[0].n_slices, A.part[1].n_slices, A.part[0].kind, A.part[1].kind);
  if (fl[FLAG_OVERFLOW]) return PLHIP_OK; // a code outside the promise: the hashed path redoes the query
  const uint64_t G = fl[FLAG_NGROUPS];
  R_->n_rows = (int64_t)G;
  if (c.deferred) R_->pending = c.token;
  handled = true;
  return finish_result(R_.release(), G, flags, out);
}

// k_leaf_sort (leaf_path.hip) can take the leaves of the two-level pipeline: the packed key leaves 9 bits of a 64-bit
// composite free, and the accumulators fit its register file
bool leaf_sort_eligible(const Plan &P) {
  if (P.single_key || P.n_accs > 8 || getenv("PLHIP_NO_LEAFSORT")) return false; // (a single key's bit range is not measured)
  int bits = 0;
  for (int i = 0; i < P.n_keys; i++) bits += (int)P.keys[i].bits;
  for (int a = 0; a < P.n_accs; a++) if (P.accs[a].op == ACC_FIRST_ROW) return false;
  return bits <= 55;
}

// Two partition levels (DESIGN.md "TILESORT2"): more groups than 512 LDS tables hold.  Level 1 = the radix
// partition of the general path (k_part_hist / scan / plan / scatter) into tile-aligned, partition-contiguous
// copies; level 2 = the tile sort inside every level-1 partition on the next hash bits; then one LDS table per
// (level-1, level-2) leaf, which writes its output rows itself (the leaves are disjoint in the keys).
int run_two_level(Lowered &L, std::vector<DevCol> &dk, uint32_t flags, uint64_t g_est, bool forced, plhip_result_t **out, bool &handled,
                  bool *overflowed = nullptr) {
  handled = false;
  if (overflowed) *overflowed = false;
  Ctx &c = g_ctx;
  Plan &P = L.P;
  const int64_t n = P.n_rows;
  if ((flags & PLHIP_FLAG_MAINTAIN_ORDER) || L.KG.enabled || L.first_acc >= 0 || n < 1) return PLHIP_OK;
  if (!forced && n < (1 << 20)) return PLHIP_OK;
  order_accs_by_source(L);
  int rc;
  if (P.single_key && dk[0].validity) { // the partitioned copies carry no null flag: nulls become code 0 of a packed key
    std::vector<KeyCol> one(1);
    one[0].col = dk[0]; one[0].nullable = true;
    if ((rc = measure_key_columns(one, n))) return rc;
    if (one[0].bits > 64) return PLHIP_OK;
    P.single_key = 0;
    pack_keys(P, one);
  }
  const int key_bytes = packed_key_bytes(P, dk);
  // ---- shape ---------------------------------------------------------------------------------------------------
  const uint32_t pbits2 = n >= (1 << 22) ? 9u : 8u, np2 = 1u << pbits2;
  auto runs_lds = [&](uint32_t lb) { return fast_table_bytes(P, lb, key_bytes); };
  // A leaf workgroup is a chain of dependent round trips (run bounds, rows, output block): ~10 us however little it
  // folds.  Small tables keep four or five leaves in flight per CU.
  uint32_t lb_max = 13;
  const char *dbg_kb = getenv("PLHIP_T2_KB"), *dbg_blk = getenv("PLHIP_T2_BLOCK"), *dbg_p1 = getenv("PLHIP_T2_PBITS1");
  // Small tables (32 KB) unless they would need 512 or more level-1 partitions (the scatter's runs are then 16 rows):
  // then 76 KB tables with a quarter of the partitions.  Measured: 256 partitions + small tables beat 128 + large ones
  // on uniform keys (3e8 rows, 3e7 ids: 14.2 against 16.7 ms) and lose 7 % on config 5 (53.5 against 49.7 ms); small
  // tables win on skewed keys at 1e8 rows / 1e7 ids (leaf fold 3.3 against 5.1 ms) and on q3 at 3e8 rows (6.7 / 7.1 ms).
  double leaf_cap = 0;
  uint32_t pbits1 = 4;
  for (int kb : {32, 76}) {
    const size_t lds_budget = (size_t)(dbg_kb ? atoi(dbg_kb) : kb) * 1024;
    lb_max = 13;
    while (lb_max > 6 && runs_lds(lb_max) > lds_budget) lb_max--;
    leaf_cap = 0.5 * (double)(1u << lb_max);
    pbits1 = 4;
    while (pbits1 < 12 && (double)g_est * 1.3 / ((double)(1u << pbits1) * np2) + 8 > leaf_cap) pbits1++;
    if (pbits1 < (getenv("PLHIP_T2_SWITCH") ? (uint32_t)atoi(getenv("PLHIP_T2_SWITCH")) : 9u) || dbg_kb) break;
  }
  if ((double)g_est * 1.3 / ((double)(1u << pbits1) * np2) + 8 > leaf_cap) return PLHIP_OK;
  // about one row per group: a wave sorts a leaf (k_leaf_sort) instead of hashing it, if the leaves are small enough
  // (<= 512 rows: ~380 on average) and a level-1 partition has at most 64 tiles
  bool leaf_sort = leaf_sort_eligible(P) && (uint64_t)n < 3 * g_est;
  if (leaf_sort) {
    uint32_t pb = pbits1;
    while (pb < TS_MAX_PBITS && (uint64_t)n > 380ull * ((uint64_t)np2 << pb)) pb++;
    if ((uint64_t)n > 400ull * ((uint64_t)np2 << pb) || ((uint64_t)n >> pb) > 60ull * TS_TILE) leaf_sort = false;
    else pbits1 = pb;
  }
  if (dbg_p1 && (uint32_t)atoi(dbg_p1) > pbits1) pbits1 = (uint32_t)atoi(dbg_p1);
  if (pbits1 + pbits2 + lb_max > 32) return PLHIP_OK; // the table's home slots come from the hash bits below the two levels
  const uint32_t np1 = 1u << pbits1;
  const double gf = (double)g_est * 1.3 / ((double)np1 * np2) + 8;
  uint32_t lbits = 6;
  while (lbits < lb_max && gf > 0.5 * (double)(1u << lbits)) lbits++;
  while (lbits < lb_max && gf > 0.25 * (double)(1u << lbits)) lbits++; // sparser when it is free
  const uint64_t n_pad = (uint64_t)n + (uint64_t)np1 * TS_TILE; // every partition rounded up to whole tiles
  if (n_pad > 0xFFFFFFFFull) return PLHIP_OK;
  const uint32_t n_tiles2 = (uint32_t)(n_pad / TS_TILE);
  const uint32_t lslots = (1u << lbits) + 66;
  const uint64_t n_items = (uint64_t)np1 * np2;
  const uint64_t out_cap = std::min<uint64_t>((uint64_t)n, n_items * lslots);
  if (out_cap >= (1ull << 31)) return PLHIP_OK; // the host signal word carries the group count in 31 bits: decline, the sort strategy / general path take it
  // ---- scratch -------------------------------------------------------------------------------------------------
  const int64_t tile_rows1 = (int64_t)PART_SCATTER_R * 1024;
  const uint32_t n_tiles1 = (uint32_t)std::max<int64_t>(1, (n + tile_rows1 - 1) / tile_rows1);
  size_t off_totals = (size_t)np1 * n_tiles1 * 4, off_start = off_totals + (size_t)np1 * 4;
  size_t off_cnt = (off_start + (size_t)(np1 + 1) * 4 + 63) & ~(size_t)63;
  HIP_TRY(c.items.reserve(off_cnt + 64));
  uint8_t *sb = (uint8_t *)c.items.p;
  uint32_t *counts = (uint32_t *)sb, *totals = (uint32_t *)(sb + off_totals), *pstart = (uint32_t *)(sb + off_start);
  uint32_t *cnt = (uint32_t *)(sb + off_cnt);
  PartOut A{}, B{};
  A.key_bytes = B.key_bytes = (uint32_t)key_bytes;
  size_t bytes = ((size_t)n_pad * key_bytes + 511) & ~(size_t)255;
  size_t off_val[PLHIP_MAX_VALS], off_valid[PLHIP_MAX_VALS];
  for (int v = 0; v < P.n_vals; v++) {
    off_val[v] = bytes;
    bytes += ((size_t)n_pad * dtype_width((int)P.vals[v].dtype) + 511) & ~(size_t)255;
    off_valid[v] = bytes;
    if (P.vals[v].validity) bytes += ((size_t)n_pad + 511) & ~(size_t)255;
  }
  HIP_TRY(c.part.reserve(bytes + 256));
  HIP_TRY(c.dump.reserve(bytes + 256));
  HIP_TRY(c.offs.reserve((size_t)(np2 + 2) * n_tiles2 * 4));
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  // heavy hitters (fast.hpp, HotState): leaves with >= hot_min rows in long runs are folded by several workgroups
  const bool hot_on = !getenv("PLHIP_T2_NOHOT") && n_items <= (1u << 22);
  const char *dbg_hc = getenv("PLHIP_T2_HOT_CHUNK"); // (tests: split at small sizes)
  const uint32_t hot_chunk = dbg_hc ? (uint32_t)std::max(1024, atoi(dbg_hc)) : 65536u, hot_min = 4 * hot_chunk;
  const uint32_t max_extra = hot_on ? (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(64, (uint64_t)n / hot_chunk / 4)) : 0u;
  const uint32_t hot_items = max_extra + HOT_MAX_LEAVES;
  const uint64_t hot_stride = fast_dump_stride(P, lslots, key_bytes);
  const size_t hoff_state = ((size_t)n_items * 4 + 255) & ~(size_t)255, hoff_hdr = hoff_state + ((sizeof(HotState) + 255) & ~(size_t)255);
  if (hot_on) {
    HIP_TRY(c.misc.reserve(hoff_hdr + (size_t)hot_items * sizeof(DumpHdr)));
    HIP_TRY(c.slots.reserve((size_t)hot_items * hot_stride));
  }
  uint32_t *hot_rows = hot_on ? (uint32_t *)c.misc.p : nullptr;
  HotState *hot_state = hot_on ? (HotState *)((uint8_t *)c.misc.p + hoff_state) : nullptr;
  uint32_t *dflags = (uint32_t *)c.flags.p;
  uint8_t *pa = (uint8_t *)c.part.p, *pb = (uint8_t *)c.dump.p;
  A.keys = pa; B.keys = pb;
  for (int v = 0; v < P.n_vals; v++) {
    A.vals[v] = pa + off_val[v]; B.vals[v] = pb + off_val[v];
    A.valid[v] = P.vals[v].validity ? pa + off_valid[v] : nullptr;
    B.valid[v] = P.vals[v].validity ? pb + off_valid[v] : nullptr;
  }
  // ---- outputs ---------------------------------------------------------------------------------------------------
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->on_device = true; R->device = c.device;
  OutPlan OP{};
  OP.n_outs = (int)L.outs.size();
  (void)hipGetLastError();
  if (c.flags_clean != 1) HIP_TRY(hipMemsetAsync(dflags, 0, 64, c.stream));
  c.flags_clean = false;
  for (size_t i = 0; i < L.outs.size(); i++) {
    OutSpec o = L.outs[i];
    ResultCol col;
    col.dtype = (int)o.dtype;
    hipError_t e = g_pool.get(g_ctx.device, (size_t)out_cap * dtype_width((int)o.dtype), &col.values, &col.vcap, true);
    if (e == hipSuccess && o.validity) {
      e = g_pool.get(g_ctx.device, (size_t)(out_cap / 64 + 2) * 8, (void **)&col.validity, &col.bcap, true);
      if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (size_t)(out_cap / 64 + 2) * 8, c.stream);
    }
    R->cols.push_back(col);
    if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
    o.values = (uint8_t *)col.values; o.validity = col.validity;
    OP.outs[i] = o;
  }
  if (++c.seq == 0) c.seq = 1;
  const uint32_t seq = c.seq;
  // ---- level 1 -----------------------------------------------------------------------------------------------------
  resolve_timing();
  hipEventRecord(c.ev[1], c.stream);
  launch_part_hist(P, pbits1, tile_rows1, n_tiles1, counts, c.stream);
  launch_part_scan(counts, np1, n_tiles1, totals, c.stream);
  launch_part_plan(totals, np1, pstart, nullptr, cnt, cnt + 1, 1u << 30, c.stream, (uint32_t)TS_TILE);
  if (pbits1 <= TS_MAX_PBITS && tile_rows1 == TS_TILE && !getenv("PLHIP_T2_LEGACY_SCATTER")) {
    // the tile sort's scatter variant (LDS-only barriers, loads in flight across them): 2x the legacy scatter's rate
    int keymode1 = 0;
    uint64_t kxor1 = 0;
    if (P.single_key && !dk[0].validity) {
      const int w = dtype_width(dk[0].dtype);
      if (w == 4) { keymode1 = 1; kxor1 = dk[0].dtype == PLHIP_I32 ? 0x80000000ull : 0ull; }
      if (w == 8) { keymode1 = 2; kxor1 = dk[0].dtype == PLHIP_I64 ? 0x8000000000000000ull : 0ull; }
    }
    TileSortLevel1 S1{pstart, totals, np1, 0u};
    // the level-1 copies hold n_pad rows: with the capacity stated the scatter places its runs through the partition-id array
    // (and clamps every destination to it) instead of a binary search per element
    if (!getenv("PLHIP_T2_SEARCH_SCATTER") && n_pad <= 0xFFFFFFFFull) S1.out_rows = (uint32_t)n_pad;
    launch_tile_sort(P, A, pbits1, n_tiles1, counts, keymode1, kxor1, key_bytes, c.stream, &S1, true);
  } else {
    launch_part_scatter(P, A, pbits1, tile_rows1, n_tiles1, counts, pstart, c.stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { return fail_hip(e, "level-1 partition launch"); }
  // ---- level 2 -----------------------------------------------------------------------------------------------------
  Plan P1 = P; // reads the level-1 copies: plain packed-key column, byte validity
  P1.n_rows = (int64_t)n_pad;
  P1.keys[0].values = A.keys; P1.keys[0].validity = nullptr; P1.keys[0].boff = 0;
  for (int v = 0; v < P.n_vals; v++) { P1.vals[v].values = A.vals[v]; P1.vals[v].validity = A.valid[v]; P1.vals[v].boff = 0; P1.vals[v].valid_bytes = 1; }
  TileSortLevel1 L1{pstart, totals, np1, pbits1, hot_rows, std::max<uint32_t>(128u, 16u * (uint32_t)TS_TILE / np2), 0u, nullptr};
  if (hot_on) HIP_TRY(hipMemsetAsync(hot_rows, 0, (size_t)n_items * 4, c.stream));
  launch_tile_sort(P1, B, pbits2, n_tiles2, (uint32_t *)c.offs.p, key_bytes == 4 ? 1 : 2, 0ull, key_bytes, c.stream, &L1);
  // (with the leaf sort in front, the heavy-hitter plan is made only if some leaves are left for k_fold_runs: it is one
  // workgroup walking every leaf, 0.45 ms at 262 144 leaves)
  if (hot_on && !leaf_sort) launch_hot_plan(hot_rows, (uint32_t)n_items, np2, pstart, hot_min, hot_chunk, max_extra, hot_state, c.stream);
  if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "level-2 k_tile_sort launch"); }
  hipEventRecord(c.ev[4], c.stream);
  Plan P2 = P; // the key specs stay: the fold decodes the packed keys with them
  for (int v = 0; v < P.n_vals; v++) { P2.vals[v].values = B.vals[v]; P2.vals[v].validity = B.valid[v]; P2.vals[v].boff = 0; P2.vals[v].valid_bytes = 1; }
  FoldRunsArgs Af{};
  Af.pkeys = B.keys; Af.out_cap = out_cap; Af.flags = dflags; Af.offs = (const uint32_t *)c.offs.p; Af.lbits = lbits; Af.np = np2; Af.w = 1;
  Af.n_pg = np2; Af.has_null_pg = 0; Af.S = np1; Af.n_tiles = n_tiles2; Af.pbits = pbits1 + pbits2;
  Af.host_sig = c.host_sig_dev; Af.seq = seq; Af.disjoint = 1; Af.seg_start = pstart;
  if (hot_on) {
    Af.hot = hot_state; Af.hot_of_leaf = hot_rows; Af.max_extra = max_extra;
    Af.D.base = (uint8_t *)c.slots.p; Af.D.hdr = (DumpHdr *)((uint8_t *)c.misc.p + hoff_hdr); Af.D.item_stride = hot_stride; Af.D.cap = lslots;
  }
  if (getenv("PLHIP_T2_VERBOSE")) fprintf(stderr, "tilesort2: np1 %u np2 %u lbits %u lb_max %u table %zu B leaves %llu rows/leaf %llu leaf_sort %d\n", np1, np2, lbits, lb_max, fast_table_bytes(P, lbits, key_bytes), (unsigned long long)n_items, (unsigned long long)((uint64_t)n / n_items), (int)leaf_sort);
  uint32_t fl[16] = {0};
  TableRef t{nullptr, 0, dflags};
  bool fold_needed = true;
  if (leaf_sort) { // a wave per leaf sorts and folds it; the leaves that do not fit (skew) are left to k_fold_runs below
    hipError_t er = c.rank.reserve((size_t)n_items + 256);
    if (er != hipSuccess) { return fail_hip(er, "leaf flags"); }
    LeafSortArgs Al{};
    Al.pkeys = B.keys; Al.offs = (const uint32_t *)c.offs.p; Al.seg_start = pstart; Al.n_pg = np2; Al.S = np1; Al.n_tiles = n_tiles2;
    Al.n_leaves = (uint32_t)n_items; Al.flags = dflags; Al.handled = (uint8_t *)c.rank.p; Al.out_cap = out_cap;
    Al.host_sig = c.host_sig_dev; Al.seq = seq;
    launch_leaf_sort(P2, Al, OP, key_bytes, c.stream);
    if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "k_leaf_sort launch"); }
    rc = wait_signal(t, seq, fl);
    if (rc) { return rc; }
    const uint32_t left = __atomic_load_n(&c.host_sig[2], __ATOMIC_ACQUIRE); // (the stream has drained: the word is final)
    fold_needed = left != 0 && !fl[FLAG_OVERFLOW];
    if (fold_needed) {
      if (hot_on) launch_hot_plan(hot_rows, (uint32_t)n_items, np2, pstart, hot_min, hot_chunk, max_extra, hot_state, c.stream);
      Af.skip = (const uint8_t *)c.rank.p;
      if (++c.seq == 0) c.seq = 1;
      Af.seq = c.seq;
    }
    c.strategy = "leafsort";
  } else c.strategy = "tilesort2";
  if (fold_needed) {
    launch_fold_runs(P2, Af, OP, key_bytes, dbg_blk ? (uint32_t)atoi(dbg_blk) : ((uint64_t)n / n_items >= 8192 ? 512u : 256u), c.stream);
    if ((e = hipGetLastError()) != hipSuccess) { return fail_hip(e, "leaf k_fold_runs launch"); }
  }
  hipEventRecord(c.ev[5], c.stream);
  if (fold_needed) {
    rc = wait_signal(t, Af.seq, fl);
    if (rc) { return rc; }
  }
  if (hipMemsetAsync(dflags, 0, 64, c.stream) == hipSuccess) c.flags_clean = true;
  c.timing_pending = 2;
  if (fl[FLAG_OVERFLOW]) { // a leaf table overflowed: the estimate was too low
    if (overflowed) *overflowed = true;
    return PLHIP_OK;
  }
  const uint64_t G = fl[FLAG_NGROUPS];
  R->n_rows = (int64_t)G;
  handled = true;
  return finish_result(R_own.release(), G, flags, out);
}

// The sort strategy (sort_path.hip): any number of groups, any skew; also serves wide key tuples (output keys
// gathered at the first row of each run).  Declines stable order (the general path sorts groups by first row).
// keys_sorted: the caller declared the single key column sorted (PLHIP_FLAG_KEYS_SORTED): no sort, groups are the
// runs of equal consecutive values in row order (which is also the stable order).
int run_sorted(Lowered &L, std::vector<DevCol> &dk, uint32_t flags, plhip_result_t **out, bool &handled, bool keys_sorted = false) {
  handled = false;
  Ctx &c = g_ctx;
  Plan &P = L.P;
  const int64_t n = P.n_rows;
  if (n < 1 || P.n_accs > 20 || n > 0xFFFFFFFEll) return PLHIP_OK;
  if (keys_sorted && !P.single_key) return PLHIP_OK;
  const bool stable = !keys_sorted && (flags & PLHIP_FLAG_MAINTAIN_ORDER); // runs come out in key order: reordered by first row below
  order_accs_by_source(L);
  int rc;
  if (!keys_sorted && P.single_key && dk[0].validity) { // nulls become code 0 of a range-compressed key
    std::vector<KeyCol> one(1);
    one[0].col = dk[0]; one[0].nullable = true;
    if ((rc = measure_key_columns(one, n))) return rc;
    if (one[0].bits > 64) return PLHIP_OK;
    P.single_key = 0;
    pack_keys(P, one);
  }
  int end_bit = 0;
  if (P.single_key) end_bit = dtype_width(dk[0].dtype) <= 4 ? 32 : 64;
  else for (int i = 0; i < P.n_keys; i++) end_bit += (int)P.keys[i].bits;
  end_bit = std::max(1, std::min(64, end_bit));
  size_t temp_bytes = 0;
  HIP_TRY(sort_path_temp_bytes((uint64_t)n, end_bit, &temp_bytes));
  const size_t n8 = ((size_t)n * 8 + 255) & ~(size_t)255, n4 = ((size_t)n * 4 + 255) & ~(size_t)255;
  if (!keys_sorted) HIP_TRY(c.table.reserve(2 * n8));
  HIP_TRY(c.slots.reserve(2 * n4));
  HIP_TRY(c.part.reserve(2 * n4));
  HIP_TRY(c.rank.reserve(n4 + 256));
  HIP_TRY(c.sortkeys.reserve(temp_bytes));
  // long_count, long_list, then (launch_fold_sorted) a done counter and the combined states per long run
  HIP_TRY(c.items.reserve(256 + ((size_t)n / 2048 + 2) * (8 + (size_t)std::max(P.n_accs, 1) * 8) + 64));
  HIP_TRY(c.flags.reserve(FLAGS_BYTES));
  HIP_TRY(c.plans.reserve(sizeof(KeyGather)));
  uint32_t *dflags = (uint32_t *)c.flags.p;
  uint64_t *keys_a = (uint64_t *)c.table.p, *keys_b = (uint64_t *)((uint8_t *)c.table.p + n8);
  uint32_t *rows_a = (uint32_t *)c.slots.p, *rows_b = (uint32_t *)((uint8_t *)c.slots.p + n4);
  uint32_t *fl_d = (uint32_t *)c.part.p, *gid = (uint32_t *)((uint8_t *)c.part.p + n4);
  uint32_t *offsets = (uint32_t *)c.rank.p;
  uint32_t *long_count = (uint32_t *)c.items.p, *long_list = long_count + 64;
  resolve_timing();
  HIP_TRY(hipEventRecord(c.ev[0], c.stream));
  HIP_TRY(hipMemsetAsync(dflags, 0, 64, c.stream));
  c.flags_clean = false;
  HIP_TRY(hipMemsetAsync(long_count, 0, 256, c.stream));
  HIP_TRY(hipEventRecord(c.ev[1], c.stream));
  if (keys_sorted) {
    keys_b = nullptr; // output keys are gathered from the key column at each run's first row
    HIP_TRY(launch_sorted_key_groups(P.keys[0], (uint64_t)n, rows_b, fl_d, gid, offsets, c.sortkeys.p, temp_bytes, dflags, c.stream));
  } else
    HIP_TRY(launch_sort_groups(P, keys_a, keys_b, rows_a, rows_b, fl_d, gid, offsets, c.sortkeys.p, temp_bytes, end_bit, dflags, c.stream));
  HIP_TRY(hipEventRecord(c.ev[4], c.stream));
  uint32_t fl[16];
  TableRef t{nullptr, 0, dflags};
  if ((rc = read_flags(t, fl))) return rc;
  const uint64_t G = fl[FLAG_NGROUPS];
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->on_device = true; R->device = c.device; R->n_rows = (int64_t)G;
  OutPlan OP{};
  OP.n_outs = (int)L.outs.size();
  for (size_t i = 0; i < L.outs.size(); i++) {
    OutSpec o = L.outs[i];
    ResultCol col;
    col.dtype = (int)o.dtype;
    hipError_t e = g_pool.get(g_ctx.device, (size_t)std::max<uint64_t>(G, 1) * dtype_width((int)o.dtype), &col.values, &col.vcap);
    if (e == hipSuccess && o.validity) {
      e = g_pool.get(g_ctx.device, (size_t)(G / 64 + 2) * 8, (void **)&col.validity, &col.bcap);
      if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (size_t)(G / 64 + 2) * 8, c.stream);
    }
    R->cols.push_back(col);
    if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
    o.values = (uint8_t *)col.values;
    o.validity = col.validity;
    OP.outs[i] = o;
  }
  const KeyGather *kg_dev = nullptr;
  hipError_t e = hipSuccess;
  KeyGather KGh = L.KG; // (the async copy below reads it until the stream is synchronised at the end of this function)
  if (keys_sorted) KGh.enabled = 1;
  if (KGh.enabled) {
    e = hipMemcpyAsync(c.plans.p, &KGh, sizeof(KeyGather), hipMemcpyHostToDevice, c.stream);
    kg_dev = (const KeyGather *)c.plans.p;
  }
  if (e == hipSuccess) {
    const uint32_t *gmap = nullptr;
    if (stable && G > 1) {
      size_t tb = 0;
      e = stable_order_temp_bytes(G, &tb);
      if (e == hipSuccess) e = c.misc.reserve((size_t)G * 20 + 256);
      if (e == hipSuccess) e = c.sortkeys.reserve(std::max(tb, temp_bytes));
      if (e == hipSuccess) {
        uint32_t *buf = (uint32_t *)c.misc.p;
        e = launch_stable_order(rows_b, offsets, G, buf, buf + 4 * G, c.sortkeys.p, std::max(tb, temp_bytes), c.stream);
        gmap = buf + 4 * G;
      }
    }
    if (e == hipSuccess) {
      launch_fold_sorted(P, OP, kg_dev, keys_b, rows_b, offsets, G, long_list, long_count, c.stream, gmap);
      e = hipGetLastError();
    }
  }
  hipEventRecord(c.ev[5], c.stream);
  hipEventRecord(c.ev[3], c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) { return fail_hip(e, "sort strategy kernels"); }
  hipEventElapsedTime(&c.total_ms, c.ev[0], c.ev[3]);
  c.k_ms[0] = c.k_ms[1] = c.k_ms[2] = 0.f;
  hipEventElapsedTime(&c.k_ms[0], c.ev[1], c.ev[4]);
  hipEventElapsedTime(&c.k_ms[1], c.ev[4], c.ev[5]);
  c.part_ms = c.k_ms[0]; c.agg_ms = c.k_ms[1];
  c.dom_ms = std::max(c.k_ms[0], c.k_ms[1]);
  c.dom_kernel = c.k_ms[0] >= c.k_ms[1] ? (keys_sorted ? "k_sorted_run_flags+run scan" : "k_pack_rows+rocprim radix sort+run scan") : "k_fold_sorted";
  c.strategy = keys_sorted ? "sorted_keys" : "sort";
  handled = true;
  return finish_result(R_own.release(), G, flags, out);
}

} // namespace

void plhip_internal_set_error(const char *msg) { g_ctx.err = msg ? msg : ""; } // for plugin.hip

// ---------------------------------------------------------------------------------------------------------
// library entry points
// ---------------------------------------------------------------------------------------------------------
extern "C" {

int32_t plhip_abi_version(void) { return PLHIP_ABI_VERSION; }

int32_t plhip_init(int32_t device_id) {
  Ctx &c = g_ctx;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(PLHIP_ERR_NO_DEVICE, "no HIP device visible; libplhip.so has no CPU path (gfx950 required)");
  if (device_id < 0 || device_id >= n) return fail(PLHIP_ERR_INVALID, "device id out of range");
  if (c.device >= 0 && c.device != device_id) return fail(PLHIP_ERR_INVALID, "this thread is already bound to another device");
  HIP_TRY(hipSetDevice(device_id));
  return ensure_ctx();
}

const char *plhip_last_error(void) { return g_ctx.err.c_str(); }

int32_t plhip_release_workspace(void) {
  Ctx &c = g_ctx;
  if (c.device < 0) return PLHIP_OK;
  resolve_timing();
  c.release_workspace();
  g_pool.trim(c.device);
  g_hpool.trim();
  return PLHIP_OK;
}

int64_t plhip_pooled_buffers(int32_t device_id) { return g_pool.count(device_id); }

int32_t plhip_set_strategy(int32_t strategy) {
  if (strategy < 0 || strategy > PLHIP_STRATEGY_DENSE) return fail(PLHIP_ERR_INVALID, "bad strategy");
  g_ctx.forced_strategy = strategy;
  return PLHIP_OK;
}

const char *plhip_last_strategy(void) { return g_ctx.strategy.c_str(); }

int32_t plhip_set_timing_interval(int32_t every) {
  if (every < 0) return fail(PLHIP_ERR_INVALID, "the timing interval is >= 0");
  g_ctx.timing_every = every; g_ctx.timing_tick = 0;
  return PLHIP_OK;
}

int32_t plhip_last_kernel_ms(float *total_ms, float *dominant_ms, const char **dominant_kernel) {
  resolve_timing();
  if (total_ms) *total_ms = g_ctx.total_ms;
  if (dominant_ms) *dominant_ms = g_ctx.dom_ms;
  if (dominant_kernel) *dominant_kernel = g_ctx.dom_kernel.c_str();
  return PLHIP_OK;
}

int32_t plhip_memcpy(void *dst, int32_t dst_location, const void *src, int32_t src_location, size_t bytes) {
  if (!bytes) return PLHIP_OK;
  if (!dst || !src) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  int rc = ensure_ctx();
  if (rc) return rc;
  const hipMemcpyKind kind = dst_location == PLHIP_DEVICE ? (src_location == PLHIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice)
                                                          : (src_location == PLHIP_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
  HIP_TRY(hipMemcpy(dst, src, bytes, kind));
  return PLHIP_OK;
}

int32_t plhip_last_kernel_breakdown(float *tile_sort_ms, float *fold_ms, float *merge_emit_ms) {
  resolve_timing();
  if (tile_sort_ms) *tile_sort_ms = g_ctx.k_ms[0];
  if (fold_ms) *fold_ms = g_ctx.k_ms[1];
  if (merge_emit_ms) *merge_emit_ms = g_ctx.k_ms[2];
  return PLHIP_OK;
}

int32_t plhip_vec_hash_seeded(const plhip_column_t *keys, int32_t n_keys, const plhip_hash_seeds_t *seeds, uint64_t *out, int32_t out_location) {
  if (!keys || n_keys < 1 || n_keys > PLHIP_MAX_KEYS) return fail(PLHIP_ERR_INVALID, "bad key columns");
  if (!seeds) return fail(PLHIP_ERR_INVALID, "seeds is NULL");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  const int64_t n = keys[0].length;
  for (int i = 0; i < n_keys; i++)
    if (keys[i].length != n) return fail(PLHIP_ERR_SHAPE_MISMATCH, "key columns differ in length");
  if (n > 0 && !out) return fail(PLHIP_ERR_INVALID, "output pointer is NULL");
  Staging st;
  std::vector<DevCol> dk(n_keys);
  Plan P{};
  P.n_rows = n; P.n_keys = n_keys;
  for (int i = 0; i < n_keys; i++) {
    if (keys[i].dtype == PLHIP_BOOL) { // an Arrow boolean array: the values buffer is a bitmap; staged as the bytes that hold the slice
      plhip_column_t bytes = keys[i];
      const int64_t b0 = keys[i].offset >> 3, b1 = (keys[i].offset + n + 7) >> 3;
      bytes.dtype = PLHIP_U8; bytes.values = keys[i].values ? (const uint8_t *)keys[i].values + b0 : nullptr; bytes.offset = 0; bytes.length = b1 - b0;
      bytes.validity = nullptr;
      DevCol vb, vd;
      if ((rc = stage_column(bytes, vb, st, c.stream))) return rc;
      if (keys[i].validity) { // (its validity bitmap: device columns stay put, host bitmaps are staged like the values')
        if (keys[i].location == PLHIP_DEVICE) { vd.validity = keys[i].validity; vd.boff = keys[i].offset; }
        else {
          plhip_column_t vv = keys[i];
          vv.dtype = PLHIP_U8; vv.values = (const uint8_t *)keys[i].validity + b0; vv.offset = 0; vv.length = b1 - b0; vv.validity = nullptr;
          DevCol t;
          if ((rc = stage_column(vv, t, st, c.stream))) return rc;
          vd.validity = t.values; vd.boff = keys[i].offset & 7;
        }
      }
      P.keys[i].values = vb.values; P.keys[i].minv = (uint64_t)(keys[i].offset & 7);
      P.keys[i].validity = vd.validity; P.keys[i].boff = vd.boff; P.keys[i].dtype = PLHIP_BOOL;
      continue;
    }
    if ((rc = stage_column(keys[i], dk[i], st, c.stream))) return rc;
    P.keys[i].values = dk[i].values; P.keys[i].validity = dk[i].validity; P.keys[i].boff = dk[i].boff;
    P.keys[i].dtype = dk[i].dtype;
  }
  if (n == 0) return PLHIP_OK;
  uint64_t *dout = out;
  if (out_location == PLHIP_HOST) {
    HIP_TRY(c.misc.reserve((size_t)n * 8));
    dout = (uint64_t *)c.misc.p;
  }
  launch_vec_hash(P, seeds->null_hash, seeds->true_hash, seeds->false_hash, dout, c.stream);
  HIP_TRY(hipGetLastError());
  if (out_location == PLHIP_HOST) HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  return PLHIP_OK;
}

int32_t plhip_vec_hash(const plhip_column_t *keys, int32_t n_keys, uint64_t null_hash, uint64_t *out, int32_t out_location) {
  if (keys && n_keys >= 1 && n_keys <= PLHIP_MAX_KEYS)
    for (int i = 0; i < n_keys; i++)
      if (keys[i].dtype == PLHIP_BOOL) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "a Boolean key hashes to seed-dependent constants: plhip_vec_hash_seeded takes them");
  const plhip_hash_seeds_t seeds{null_hash, 0, 0};
  return plhip_vec_hash_seeded(keys, n_keys, &seeds, out, out_location);
}

int32_t group_by_agg_impl(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values, int32_t n_values,
                          const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out);

int32_t plhip_group_by_agg(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                           int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                           plhip_result_t **out) {
  return plhip_group_by_agg_opts(keys, n_keys, values, n_values, aggs, n_aggs, flags, nullptr, out);
}

namespace {
// A Boolean column (values = the Arrow bitmap, host or device) becomes a device-resident column of `w`-byte integers of dtype `dt`
// holding 0 / 1 (the reference's casts: into_groups.rs:215-230 to UInt8 for keys, boolean.rs:107-109 to IdxSize for sums); the
// validity bitmap travels as it is.  The new buffers belong to `st`.
int lower_bool_column(plhip_column_t &col, uint32_t w, int dt, Staging &st) {
  Ctx &c = g_ctx;
    if (col.length < 0 || col.offset < 0) return fail(PLHIP_ERR_INVALID, "negative column offset/length");
    if (col.length > 0 && !col.values) return fail(PLHIP_ERR_INVALID, "column values pointer is NULL");
    const int64_t n = col.length, b0 = col.offset >> 3, b1 = (col.offset + n + 7) >> 3;
    const uint8_t *bits = (const uint8_t *)col.values, *valid = col.validity;
    int64_t boff = col.offset;
    if (col.location == PLHIP_HOST) { // bitmaps travel as bytes; the bit offset inside the first byte stays
      void *db = nullptr;
      HIP_TRY(hipMalloc(&db, (size_t)(b1 - b0) + 8)); st.owned.push_back(db);
      if (b1 > b0) HIP_TRY(hipMemcpyAsync(db, bits + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, c.stream));
      bits = (const uint8_t *)db; boff = col.offset & 7;
      if (valid) {
        void *dvb = nullptr;
        HIP_TRY(hipMalloc(&dvb, (size_t)(b1 - b0) + 8)); st.owned.push_back(dvb);
        if (b1 > b0) HIP_TRY(hipMemcpyAsync(dvb, valid + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, c.stream));
        valid = (const uint8_t *)dvb;
      }
    } else if (col.location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "bad column location");
    void *vals = nullptr;
    HIP_TRY(hipMalloc(&vals, n > 0 ? (size_t)n * w : 8)); st.owned.push_back(vals);
    launch_unpack_bool(bits, boff, n, vals, w, c.stream);
    HIP_TRY(hipGetLastError());
    // one Arrow offset serves values and validity: row 0 of the unpacked column must sit `boff` elements into `values`
    col.values = (const void *)((uintptr_t)vals - (uintptr_t)boff * w);
    col.validity = valid; col.offset = boff; col.dtype = dt; col.location = PLHIP_DEVICE;
    return PLHIP_OK;
}

// Boolean columns (PLHIP_BOOL): lowered at the boundary with the reference's own casts -- a key to UInt8 (into_groups.rs:215-230), a
// value column to IdxSize = u32 (boolean.rs:107-109 for sum; the f64 mean of the cast is dispatch.rs:164's) -- and the Boolean outputs
// (keys, min, max) packed back into bitmaps.
int sharded_impl(plhip_comm *comm, const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values, int32_t n_values,
                 const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out);
// comm != nullptr: the row-sharded query (plhip_group_by_agg_sharded) over the lowered columns -- every rank lowers alike, the
// schema is the same everywhere
int bool_group_by(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values, int32_t n_values, const plhip_agg_t *aggs,
                  int32_t n_aggs, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out, plhip_comm *comm = nullptr) {
  if (flags & PLHIP_FLAG_PARTIAL) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "Boolean columns have no partial (mergeable) form");
  if (n_keys < 1 || n_keys > PLHIP_MAX_KEYS || n_values < 0 || n_values > PLHIP_MAX_VALS) return fail(PLHIP_ERR_INVALID, "bad column counts");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  Staging st;
  std::vector<plhip_column_t> lk(keys, keys + n_keys), lv(values, values + n_values);
  auto lower = [&](plhip_column_t &col, uint32_t w, int dt) -> int { return lower_bool_column(col, w, dt, st); };
  std::vector<char> bool_out;
  for (int i = 0; i < n_keys; i++) {
    const bool b = keys[i].dtype == PLHIP_BOOL;
    bool_out.push_back(b);
    if (b && (rc = lower(lk[i], 1, PLHIP_U8))) return rc;
  }
  for (int i = 0; i < n_values; i++)
    if (values[i].dtype == PLHIP_BOOL && (rc = lower(lv[i], 4, PLHIP_U32))) return rc;
  for (int i = 0; i < n_aggs; i++) {
    const bool on_bool = aggs[i].kind != PLHIP_AGG_COUNT && aggs[i].column >= 0 && aggs[i].column < n_values && values[aggs[i].column].dtype == PLHIP_BOOL;
    bool_out.push_back(on_bool && (aggs[i].kind == PLHIP_AGG_MIN || aggs[i].kind == PLHIP_AGG_MAX));
  }
  plhip_result_t *R = nullptr;
  if (comm) rc = sharded_impl(comm, lk.data(), n_keys, lv.data(), n_values, aggs, n_aggs, flags | PLHIP_FLAG_OUTPUT_DEVICE, opts, &R);
  else rc = group_by_agg_impl(lk.data(), n_keys, lv.data(), n_values, aggs, n_aggs, flags | PLHIP_FLAG_OUTPUT_DEVICE, opts, &R);
  if (rc) { (void)hipStreamSynchronize(c.stream); return rc; } // (the unpacked copies are freed on the way out)
  std::unique_ptr<plhip_result> Rg(R);
  StreamDrain drain(c.stream);
  const int64_t G = R->n_rows;
  if (R->cols.size() != bool_out.size()) return fail(PLHIP_ERR_COMPUTE, "the result frame does not match the query");
  for (size_t i = 0; i < R->cols.size(); i++) {
    if (!bool_out[i]) continue;
    ResultCol &col = R->cols[i];
    const uint32_t w = (uint32_t)dtype_width(col.dtype);
    if (w != 1 && w != 4) return fail(PLHIP_ERR_COMPUTE, "a Boolean output came back with an unexpected type");
    void *bits = nullptr;
    size_t cap = 0;
    HIP_TRY(g_pool.get(c.device, (size_t)((G + 63) / 64) * 8 + 8, &bits, &cap));
    launch_pack_bool(col.values, w, G, (uint64_t *)bits, c.stream);
    const hipError_t e = hipGetLastError() == hipSuccess ? hipStreamSynchronize(c.stream) : hipErrorLaunchFailure;
    if (e != hipSuccess) { g_pool.put(c.device, bits, cap); return fail_hip(e, "k_pack_bool"); }
    if (col.vcap) g_pool.put(c.device, col.values, col.vcap); else hipFree(col.values);
    col.values = bits; col.vcap = cap; col.dtype = PLHIP_BOOL;
  }
  HIP_TRY(hipStreamSynchronize(c.stream));
  return finish_result(Rg.release(), (uint64_t)G, flags, out);
}
} // namespace

int32_t plhip_group_by_agg_opts(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                                int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                const plhip_options_t *opts, plhip_result_t **out) {
  bool any_bool = false;
  if (keys && n_keys > 0 && n_keys <= PLHIP_MAX_KEYS) for (int i = 0; i < n_keys; i++) any_bool |= keys[i].dtype == PLHIP_BOOL;
  if (values && n_values > 0 && n_values <= PLHIP_MAX_VALS) for (int i = 0; i < n_values; i++) any_bool |= values[i].dtype == PLHIP_BOOL;
  if (any_bool && out) *out = nullptr;
  // a DEVICE-output frame may come back with its last kernel still retiring (wait_signal): only from this entry point, whose
  // callers reach the buffers through plhip_result_column
  Ctx &c = g_ctx;
  c.deferred = false;
  c.allow_defer = !any_bool && (flags & PLHIP_FLAG_OUTPUT_DEVICE) != 0;
  const int rc = any_bool && out ? bool_group_by(keys, n_keys, values, n_values, aggs, n_aggs, flags & ~FLAG_INTERNAL_WRAP32, opts, out)
                                 : group_by_agg_impl(keys, n_keys, values, n_values, aggs, n_aggs, flags & ~FLAG_INTERNAL_WRAP32, opts, out);
  c.allow_defer = false;
  if (c.deferred && !(out && *out && (*out)->pending)) { // (cannot happen: a deferred wait is the last step of its path)
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    if (out && *out) (*out)->pending.reset();
  }
  c.deferred = false;
  // slice pushdown (GroupByExec's `slice`, executors/group_by.rs:79-82; offsets as slice_offsets, utils/mod.rs:199-218)
  if (!rc && out && *out && opts && opts->struct_size >= offsetof(plhip_options_t, has_slice) + sizeof(uint32_t) && opts->has_slice) {
    plhip_result *R = *out;
    const uint64_t len = (uint64_t)R->n_rows, want = opts->slice_len;
    const int64_t off = opts->slice_offset;
    const uint64_t a = off < 0 ? (uint64_t)(-(off + 1)) + 1 : (uint64_t)off; // |off| without overflow
    uint64_t start, count;
    if (off < 0) {
      if (a <= len) { start = len - a; count = std::min(want, a); } else { start = 0; count = std::min(want, len); }
    } else if (a <= len) { start = a; count = std::min(want, len - a); }
    else { start = len; count = 0; }
    R->row_offset = (int64_t)start;
    R->n_rows = (int64_t)count;
  }
  return rc;
}

namespace {
// One chunked column -> one single-chunk DEVICE column (owned by `st` unless the column had a single chunk).
int rechunk_column(const plhip_chunked_column_t &cc, plhip_column_t &out, Staging &st, hipStream_t s) {
  if (cc.n_chunks < 1 || !cc.chunks) return fail(PLHIP_ERR_INVALID, "a chunked column needs at least one chunk");
  if (cc.n_chunks == 1) { out = cc.chunks[0]; return PLHIP_OK; }
  const int dt = cc.chunks[0].dtype;
  const bool is_bool = dt == PLHIP_BOOL; // the values buffers are bitmaps: laid end to end bit by bit, like the validity below
  const int w = is_bool ? 1 : dtype_width(dt);
  if (!w) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "unsupported dtype code " + std::to_string(dt));
  int64_t total = 0;
  bool any_valid = false;
  for (int i = 0; i < cc.n_chunks; i++) {
    const plhip_column_t &c = cc.chunks[i];
    if (c.dtype != dt) return fail(PLHIP_ERR_INVALID, "the chunks of a column differ in dtype");
    if (c.length < 0 || c.offset < 0) return fail(PLHIP_ERR_INVALID, "negative column offset/length");
    if (c.length > 0 && !c.values) return fail(PLHIP_ERR_INVALID, "column values pointer is NULL");
    if (c.location != PLHIP_HOST && c.location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "bad column location");
    total += c.length;
    any_valid |= c.validity != nullptr;
  }
  void *dv = nullptr, *db = nullptr;
  const size_t bm_bytes = (((size_t)total + 31) / 32) * 4 + 8;
  HIP_TRY(hipMalloc(&dv, is_bool ? bm_bytes : (size_t)std::max<int64_t>(total, 1) * w));
  st.owned.push_back(dv);
  if (is_bool) HIP_TRY(hipMemsetAsync(dv, 0, bm_bytes, s));
  if (any_valid) {
    HIP_TRY(hipMalloc(&db, bm_bytes));
    st.owned.push_back(db);
    HIP_TRY(hipMemsetAsync(db, 0, bm_bytes, s));
  }
  int64_t pos = 0;
  for (int i = 0; i < cc.n_chunks; i++) {
    const plhip_column_t &c = cc.chunks[i];
    if (!c.length) continue;
    const hipMemcpyKind kind = c.location == PLHIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (is_bool) {
      const uint8_t *src = (const uint8_t *)c.values;
      int64_t src_off = c.offset;
      if (c.location == PLHIP_HOST) { // the bitmap's bytes that cover the chunk travel to the device first
        const int64_t b0 = c.offset >> 3, b1 = (c.offset + c.length + 7) >> 3;
        void *tmp = nullptr;
        HIP_TRY(hipMalloc(&tmp, (size_t)(b1 - b0) + 8));
        st.owned.push_back(tmp);
        HIP_TRY(hipMemcpyAsync(tmp, (const uint8_t *)c.values + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, s));
        src = (const uint8_t *)tmp; src_off = c.offset & 7;
      }
      launch_copy_bits((uint32_t *)dv, pos, src, src_off, c.length, s);
      HIP_TRY(hipGetLastError());
    } else
      HIP_TRY(hipMemcpyAsync((uint8_t *)dv + (size_t)pos * w, (const uint8_t *)c.values + (size_t)c.offset * w, (size_t)c.length * w, kind, s));
    if (any_valid) {
      const uint8_t *src = c.validity;
      int64_t src_off = c.offset;
      if (src && c.location == PLHIP_HOST) { // the bitmap's bytes that cover the chunk travel to the device first
        const int64_t b0 = c.offset >> 3, b1 = (c.offset + c.length + 7) >> 3;
        void *tmp = nullptr;
        HIP_TRY(hipMalloc(&tmp, (size_t)(b1 - b0) + 8));
        st.owned.push_back(tmp);
        HIP_TRY(hipMemcpyAsync(tmp, c.validity + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, s));
        src = (const uint8_t *)tmp; src_off = c.offset & 7;
      }
      launch_copy_bits((uint32_t *)db, pos, src, src_off, c.length, s);
      HIP_TRY(hipGetLastError());
    }
    pos += c.length;
  }
  out = plhip_column_t{};
  out.values = dv; out.validity = (const uint8_t *)db; out.offset = 0; out.length = total; out.dtype = dt; out.location = PLHIP_DEVICE;
  return PLHIP_OK;
}
} // namespace

int32_t plhip_group_by_agg_chunked(const plhip_chunked_column_t *keys, int32_t n_keys, const plhip_chunked_column_t *values,
                                   int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                   const plhip_options_t *opts, plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_keys < 1 || !keys) return fail(PLHIP_ERR_COMPUTE, "at least one key is required in a group_by operation");
  if (n_keys > PLHIP_MAX_KEYS) return fail(PLHIP_ERR_INVALID, "too many key columns (max 16)");
  if (n_values < 0 || n_values > PLHIP_MAX_VALS || (n_values > 0 && !values)) return fail(PLHIP_ERR_INVALID, "bad value columns");
  int rc = ensure_ctx();
  if (rc) return rc;
  Staging st; // the rechunked buffers live until the query has returned (its count comes from its last kernel, which no longer reads inputs; hipFree waits for the device besides)
  std::vector<plhip_column_t> k(n_keys), v(std::max(n_values, 1));
  for (int i = 0; i < n_keys; i++) if ((rc = rechunk_column(keys[i], k[i], st, g_ctx.stream))) return rc;
  for (int i = 0; i < n_values; i++) if ((rc = rechunk_column(values[i], v[i], st, g_ctx.stream))) return rc;
  rc = plhip_group_by_agg_opts(k.data(), n_keys, v.data(), n_values, aggs, n_aggs, flags, opts, out);
  if (rc) hipStreamSynchronize(g_ctx.stream); // the copies may still be queued: the buffers are freed on the way out
  return rc;
}

// page-locked host buffers for the host layer's Arrow allocator (host -> HBM copies then run as one DMA)
int32_t plhip_alloc_pinned(size_t bytes, void **ptr) {
  if (!ptr) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *ptr = nullptr;
  const int rc = ensure_ctx();
  if (rc) return rc;
  const hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault);
  if (e != hipSuccess) { *ptr = nullptr; return fail_hip(e, "hipHostMalloc"); }
  return PLHIP_OK;
}

int32_t plhip_free_pinned(void *ptr) {
  if (!ptr) return PLHIP_OK;
  const hipError_t e = hipHostFree(ptr);
  return e == hipSuccess ? PLHIP_OK : fail_hip(e, "hipHostFree");
}

// ---------------------------------------------------------------------------------------------------------
// string keys: LargeUtf8 -> dense codes + dictionary (utf8.hpp)
// ---------------------------------------------------------------------------------------------------------
struct plhip_dictionary {
  int device = 0;
  int64_t n = 0, G = 0, total = 0;
  uint32_t *codes = nullptr, *cvalid = nullptr, *dvalid = nullptr;
  int64_t *offs = nullptr;
  uint8_t *data = nullptr;
  size_t cap[5] = {0, 0, 0, 0, 0}; // pool capacities of codes, cvalid, dvalid, offs, data
  int64_t *h_offs = nullptr; uint8_t *h_data = nullptr, *h_valid = nullptr;
  ~plhip_dictionary() { // device buffers go back to the result pool (hipMalloc / hipFree cost more than the encoder's kernels)
    g_pool.put(device, codes, cap[0]); g_pool.put(device, cvalid, cap[1]); g_pool.put(device, dvalid, cap[2]);
    g_pool.put(device, offs, cap[3]); g_pool.put(device, data, cap[4]);
    free(h_offs); free(h_data); free(h_valid);
  }
};

int32_t plhip_utf8_encode(const plhip_utf8_column_t *col, plhip_dictionary_t **dict, plhip_column_t *codes) {
  if (!col || !dict || !codes) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *dict = nullptr;
  const int64_t n = col->length;
  if (n < 0 || col->offset < 0) return fail(PLHIP_ERR_INVALID, "negative column offset/length");
  if (n > 0xFFFFFFFFll) return fail(PLHIP_ERR_COMPUTE, "more than u32::MAX rows (IdxSize is u32)");
  if (!col->offsets || (n > 0 && !col->data)) return fail(PLHIP_ERR_INVALID, "offsets / data pointer is NULL");
  if (col->location != PLHIP_HOST && col->location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "bad column location");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  Staging st; // every temporary of the call
  auto dmalloc = [&](void **p, size_t bytes) -> hipError_t {
    const hipError_t e = hipMalloc(p, bytes ? bytes : 8);
    if (e == hipSuccess) st.owned.push_back(*p);
    return e;
  };
  std::unique_ptr<plhip_dictionary> D(new (std::nothrow) plhip_dictionary());
  if (!D) return fail(PLHIP_ERR_OOM, "host allocation failed");
  D->device = c.device; D->n = n;
  struct Groups { plhip_result_t *p = nullptr; ~Groups() { delete p; } } groups; // (device-resident frame: back to the pool on every path out)
  StreamDrain drain(c.stream);
  // ---- the column in HBM -------------------------------------------------------------------------------------------
  Utf8View V{};
  V.n = n;
  if (col->location == PLHIP_DEVICE) {
    V.offsets = col->offsets + col->offset; V.data = col->data; V.validity = col->validity; V.boff = col->offset;
  } else {
    const int64_t o0 = col->offsets[col->offset], o1 = col->offsets[col->offset + n];
    if (o1 < o0) return fail(PLHIP_ERR_INVALID, "offsets are not monotonic");
    void *doff = nullptr, *ddat = nullptr;
    HIP_TRY(dmalloc(&doff, (size_t)(n + 1) * 8));
    HIP_TRY(hipMemcpyAsync(doff, col->offsets + col->offset, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(dmalloc(&ddat, (size_t)(o1 - o0)));
    if (o1 > o0) HIP_TRY(hipMemcpyAsync(ddat, col->data + o0, (size_t)(o1 - o0), hipMemcpyHostToDevice, c.stream));
    V.offsets = (const int64_t *)doff;
    V.data = (const uint8_t *)ddat - o0; // the offsets index the host buffer: the staged range starts at o0
    if (col->validity) {
      const int64_t b0 = col->offset >> 3, b1 = (col->offset + n + 7) >> 3;
      void *db = nullptr;
      HIP_TRY(dmalloc(&db, (size_t)(b1 - b0) + 8));
      if (b1 > b0) HIP_TRY(hipMemcpyAsync(db, col->validity + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, c.stream));
      V.validity = (const uint8_t *)db; V.boff = col->offset & 7;
    }
  }
  // ---- hash -> distinct hashes with their first rows -> codes -> verify ---------------------------------------------
  const size_t h_bytes = ((size_t)n * 8 + 255) & ~(size_t)255, i_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
  HIP_TRY(c.str_a.reserve(h_bytes + i_bytes + 256));
  void *h = c.str_a.p, *iota = (uint8_t *)c.str_a.p + h_bytes, *bad_d = (uint8_t *)c.str_a.p + h_bytes + i_bytes;
  HIP_TRY(g_pool.get(c.device, n ? (size_t)n * 4 : 8, (void **)&D->codes, &D->cap[0]));
  uint64_t G = 0;
  static const uint64_t seeds[3] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull};
  for (int attempt = 0;; attempt++) {
    if (attempt == 3) return fail(PLHIP_ERR_COMPUTE, "string hashes collided under three seeds");
    if (groups.p) { delete groups.p; groups.p = nullptr; }
    G = 0;
    if (n == 0) break;
    launch_utf8_hash(V, seeds[attempt], (uint64_t *)h, (uint32_t *)iota, c.stream);
    HIP_TRY(hipGetLastError());
    plhip_column_t kc{}, vc{};
    kc.values = h; kc.length = n; kc.dtype = PLHIP_U64; kc.location = PLHIP_DEVICE;
    vc.values = iota; vc.length = n; vc.dtype = PLHIP_U32; vc.location = PLHIP_DEVICE;
    const plhip_agg_t first_row{0, PLHIP_AGG_MIN};
    if ((rc = group_by_agg_impl(&kc, 1, &vc, 1, &first_row, 1, PLHIP_FLAG_OUTPUT_DEVICE, nullptr, &groups.p))) return rc;
    G = (uint64_t)groups.p->n_rows;
    if (G == 0 || G > 0x7FFFFFF0ull || groups.p->cols.size() != 2 || groups.p->cols[1].dtype != PLHIP_U32)
      return fail(PLHIP_ERR_COMPUTE, "the group-by over the string hashes came back malformed");
    const uint64_t cap = next_pow2(std::max<uint64_t>(2 * G, 1024));
    HIP_TRY(c.str_b.reserve(cap * 16 + 256));
    Utf8Slot *table = (Utf8Slot *)c.str_b.p;
    uint32_t bad = 0;
    HIP_TRY(hipMemsetAsync(table, 0xFF, cap * 16, c.stream));
    HIP_TRY(hipMemsetAsync(bad_d, 0, 8, c.stream));
    launch_ht_build((const uint64_t *)groups.p->cols[0].values, (const uint32_t *)groups.p->cols[1].values, (uint32_t)G, table, cap - 1, c.stream);
    launch_ht_probe_verify(V, (const uint64_t *)h, table, cap - 1, D->codes, (uint32_t *)bad_d, c.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c.host_flags, bad_d, 4, hipMemcpyDeviceToHost, c.stream)); // (pinned)
    HIP_TRY(hipStreamSynchronize(c.stream));
    bad = c.host_flags[0];
    if (bad & 2u) return fail(PLHIP_ERR_COMPUTE, "a row's hash is missing from the lookup table");
    if (!bad) break; // every row equals its group's first row byte for byte
  }
  D->G = (int64_t)G;
  // ---- the dictionary: the strings at the groups' first rows ---------------------------------------------------------
  HIP_TRY(g_pool.get(c.device, (size_t)(G + 1) * 8, (void **)&D->offs, &D->cap[3]));
  HIP_TRY(g_pool.get(c.device, (size_t)(G / 32 + 2) * 4, (void **)&D->dvalid, &D->cap[2]));
  HIP_TRY(hipMemsetAsync(D->dvalid, 0, (size_t)(G / 32 + 2) * 4, c.stream));
  if (G > 0) {
    const uint32_t *first = (const uint32_t *)groups.p->cols[1].values;
    const size_t tb = utf8_scan_temp_bytes((uint32_t)G), l_bytes = ((size_t)(G + 1) * 8 + 255) & ~(size_t)255;
    HIP_TRY(c.str_b.reserve(l_bytes + tb + 256)); // (the lookup table is done with)
    void *lens = c.str_b.p, *tmp = (uint8_t *)c.str_b.p + l_bytes;
    launch_utf8_lengths(V, first, (uint32_t)G, (int64_t *)lens, c.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(utf8_scan(tmp, tb, (const int64_t *)lens, D->offs, (uint32_t)G, c.stream));
    HIP_TRY(hipMemcpyAsync(c.host_flags, D->offs + G, 8, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    memcpy(&D->total, c.host_flags, 8);
    if (D->total < 0) return fail(PLHIP_ERR_INVALID, "the column's offsets are not monotonic");
    HIP_TRY(g_pool.get(c.device, D->total > 0 ? (size_t)D->total : 8, (void **)&D->data, &D->cap[4]));
    launch_utf8_gather(V, first, (uint32_t)G, D->offs, D->data, D->dvalid, c.stream);
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemsetAsync(D->offs, 0, 8, c.stream));
    HIP_TRY(g_pool.get(c.device, 8, (void **)&D->data, &D->cap[4]));
  }
  if (V.validity) { // the codes' validity: the column's bitmap, re-based at bit 0
    const size_t words = (size_t)((n + 31) / 32) + 2;
    HIP_TRY(g_pool.get(c.device, words * 4, (void **)&D->cvalid, &D->cap[1]));
    HIP_TRY(hipMemsetAsync(D->cvalid, 0, words * 4, c.stream));
    if (n > 0) { launch_copy_bits(D->cvalid, 0, V.validity, V.boff, n, c.stream); HIP_TRY(hipGetLastError()); }
  }
  HIP_TRY(hipStreamSynchronize(c.stream));
  *codes = plhip_column_t{};
  codes->values = D->codes; codes->validity = (const uint8_t *)D->cvalid; codes->offset = 0; codes->length = n;
  codes->dtype = PLHIP_U32; codes->location = PLHIP_DEVICE;
  *dict = D.release();
  return PLHIP_OK;
}

int64_t plhip_dictionary_len(const plhip_dictionary_t *d) { return d ? d->G : -1; }

int32_t plhip_dictionary_strings(plhip_dictionary_t *d, int32_t location, plhip_utf8_column_t *out) {
  if (!d || !out) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *out = plhip_utf8_column_t{};
  out->length = d->G; out->location = location;
  if (location == PLHIP_DEVICE) {
    out->offsets = d->offs; out->data = d->data; out->validity = (const uint8_t *)d->dvalid;
    return PLHIP_OK;
  }
  if (location != PLHIP_HOST) return fail(PLHIP_ERR_INVALID, "bad location");
  if (!d->h_offs) {
    int rc = ensure_ctx();
    if (rc) return rc;
    if (g_ctx.device != d->device) return fail(PLHIP_ERR_INVALID, "the dictionary belongs to another device than the calling thread's");
    const size_t vb = (size_t)(d->G / 32 + 2) * 4;
    int64_t *ho = (int64_t *)malloc((size_t)(d->G + 1) * 8);
    uint8_t *hd = (uint8_t *)malloc(d->total > 0 ? (size_t)d->total : 8), *hv = (uint8_t *)malloc(vb);
    hipError_t e = (ho && hd && hv) ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(ho, d->offs, (size_t)(d->G + 1) * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && d->total > 0) e = hipMemcpy(hd, d->data, (size_t)d->total, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hv, d->dvalid, vb, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(ho); free(hd); free(hv); return fail_hip(e, "dictionary copy"); }
    d->h_offs = ho; d->h_data = hd; d->h_valid = hv;
  }
  out->offsets = d->h_offs; out->data = d->h_data; out->validity = d->h_valid;
  return PLHIP_OK;
}

void plhip_dictionary_free(plhip_dictionary_t *d) { delete d; }

// ---------------------------------------------------------------------------------------------------------
// hash join on one integer key (join.hpp)
// ---------------------------------------------------------------------------------------------------------
struct plhip_join {
  int device = 0;
  int64_t n = 0;
  bool on_device = true;
  uint32_t *left = nullptr, *right = nullptr;
  size_t capl = 0, capr = 0;
  ~plhip_join() {
    if (on_device) { g_pool.put(device, left, capl); g_pool.put(device, right, capr); }
    else { free(left); free(right); }
  }
};

int32_t plhip_hash_join(const plhip_column_t *left, const plhip_column_t *right, int32_t how, int32_t out_location, plhip_join_t **out) {
  if (!left || !right || !out) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *out = nullptr;
  if (how < PLHIP_JOIN_INNER || how > PLHIP_JOIN_OUTER) return fail(PLHIP_ERR_INVALID, "bad join kind");
  const bool outer = how == PLHIP_JOIN_OUTER;
  const bool membership = how == PLHIP_JOIN_SEMI || how == PLHIP_JOIN_ANTI;
  if (out_location != PLHIP_HOST && out_location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "bad output location");
  if (left->dtype != right->dtype) return fail(PLHIP_ERR_COMPUTE, "datatypes of join keys don't match");
  if (dt_is_float(left->dtype) || !dtype_width(left->dtype)) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "the join key must be an integer column");
  if (left->length > 0xFFFFFFFEll || right->length > 0xFFFFFFFEll) return fail(PLHIP_ERR_COMPUTE, "more than u32::MAX rows (IdxSize is u32)");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  Staging st;
  DevCol dl, dr;
  if ((rc = stage_column(*left, dl, st, c.stream)) || (rc = stage_column(*right, dr, st, c.stream))) return rc;
  // the shorter relation builds (hash_join/mod.rs:42-58); a left join always probes with the left one
  const bool swapped = (how == PLHIP_JOIN_INNER || outer) && !(left->length > right->length);
  const DevCol &dp = swapped ? dr : dl, &db = swapped ? dl : dr;
  const uint32_t w = (uint32_t)dtype_width(left->dtype);
  const int64_t n = dp.length, nb = db.length;
  // ---- the build side's key -> rows map: the group-by seam's CSR (lists ascending) ------------------------------------
  struct GroupsGuard { plhip_groups_t *p = nullptr; ~GroupsGuard() { delete p; } } gr;
  const uint32_t *first = nullptr, *offsets = nullptr, *indices = nullptr;
  uint64_t G = 0;
  if (nb > 0) {
    plhip_column_t bc{};
    bc.values = db.values - (size_t)db.boff * w; bc.validity = db.validity; bc.offset = db.boff; bc.length = nb; // (values + offset * w = element 0)
    bc.dtype = left->dtype; bc.location = PLHIP_DEVICE;
    if ((rc = plhip_group_tuples(&bc, 1, 0, PLHIP_DEVICE, &gr.p))) return rc;
    int32_t loc = 0;
    if ((rc = plhip_groups_buffers(gr.p, &first, &offsets, &indices, &loc))) return rc;
    G = (uint64_t)gr.p->n_groups;
  }
  const uint64_t cap = next_pow2(std::max<uint64_t>(2 * G, 1024));
  HIP_TRY(c.str_b.reserve(cap * 16 + 512));
  JoinTable T{};
  T.slots = (JoinSlot *)c.str_b.p; T.g_null = (uint32_t *)((uint8_t *)c.str_b.p + cap * 16); T.mask = cap - 1;
  HIP_TRY(hipMemsetAsync(T.slots, 0, cap * 16, c.stream));
  HIP_TRY(hipMemsetAsync(T.g_null, 0xFF, 256, c.stream));
  JoinKey KB{db.values, db.validity, db.boff, nb, w, 0}, KP{dp.values, dp.validity, dp.boff, n, w, 0};
  launch_join_build(KB, first, (uint32_t)G, T, c.stream);
  HIP_TRY(hipGetLastError());
  // ---- probe: count, scan, fill -----------------------------------------------------------------------------------------
  const size_t g_bytes = ((size_t)n * 4 + 255) & ~(size_t)255, c_bytes = ((size_t)(n + 1) * 8 + 255) & ~(size_t)255;
  const size_t tb = std::max(join_scan_temp_bytes(n), join_scan_temp_bytes((int64_t)G));
  // outer join: matched flags per build group, and the count / position arrays of the tail (the unmatched groups' rows)
  const size_t m_bytes = outer ? (((size_t)G * 4 + 255) & ~(size_t)255) + 256 : 0, t_bytes = outer ? (((size_t)(G + 1) * 8 + 255) & ~(size_t)255) : 0;
  HIP_TRY(c.str_a.reserve(g_bytes + 2 * c_bytes + tb + m_bytes + 2 * t_bytes + 256));
  uint32_t *gid = (uint32_t *)c.str_a.p;
  uint64_t *cnt = (uint64_t *)((uint8_t *)c.str_a.p + g_bytes), *pos = (uint64_t *)((uint8_t *)c.str_a.p + g_bytes + c_bytes);
  void *tmp = (uint8_t *)c.str_a.p + g_bytes + 2 * c_bytes;
  uint32_t *matched = (uint32_t *)((uint8_t *)tmp + tb);
  uint64_t *tcnt = (uint64_t *)((uint8_t *)matched + m_bytes), *tpos = (uint64_t *)((uint8_t *)matched + m_bytes + t_bytes);
  if (outer) HIP_TRY(hipMemsetAsync(matched, 0, m_bytes, c.stream));
  launch_join_count(KP, T, offsets, how, gid, cnt, outer ? matched : nullptr, c.stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(join_scan(tmp, tb, cnt, pos, n, c.stream));
  HIP_TRY(hipMemcpyAsync(c.host_flags, pos + n, 8, hipMemcpyDeviceToHost, c.stream));
  if (outer) {
    launch_join_tail_count((uint32_t)G, matched, offsets, tcnt, c.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(join_scan(tmp, tb, tcnt, tpos, (int64_t)G, c.stream));
    HIP_TRY(hipMemcpyAsync(c.host_flags + 2, tpos + G, 8, hipMemcpyDeviceToHost, c.stream));
  }
  HIP_TRY(hipStreamSynchronize(c.stream));
  uint64_t total = 0, head = 0;
  memcpy(&head, c.host_flags, 8);
  total = head;
  if (outer) { uint64_t tail = 0; memcpy(&tail, c.host_flags + 2, 8); total += tail; }
  if (total > 0xFFFFFFFEull) return fail(PLHIP_ERR_COMPUTE, "the join yields more than u32::MAX rows (IdxSize is u32)");
  std::unique_ptr<plhip_join> J(new (std::nothrow) plhip_join());
  if (!J) return fail(PLHIP_ERR_OOM, "host allocation failed");
  J->device = c.device; J->n = (int64_t)total;
  StreamDrain drain(c.stream);
  uint32_t *out_probe = nullptr, *out_build = nullptr;
  size_t cp = 0, cb = 0;
  HIP_TRY(g_pool.get(c.device, std::max<size_t>((size_t)total * 4, 8), (void **)&out_probe, &cp));
  (swapped ? J->right : J->left) = out_probe; (swapped ? J->capr : J->capl) = cp;
  if (!membership) {
    HIP_TRY(g_pool.get(c.device, std::max<size_t>((size_t)total * 4, 8), (void **)&out_build, &cb));
    (swapped ? J->left : J->right) = out_build; (swapped ? J->capl : J->capr) = cb;
  }
  launch_join_fill(n, gid, pos, offsets, indices, out_probe, out_build, c.stream);
  HIP_TRY(hipGetLastError());
  if (outer) { // the build rows nobody matched, behind the probe rows' pairs
    launch_join_tail_fill((uint32_t)G, matched, tpos, offsets, indices, out_probe + head, out_build + head, c.stream);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c.stream));
  c.strategy = "hash_join";
  if (out_location == PLHIP_HOST) {
    uint32_t *hl = (uint32_t *)malloc(std::max<size_t>((size_t)total * 4, 8)), *hr = membership ? nullptr : (uint32_t *)malloc(std::max<size_t>((size_t)total * 4, 8));
    hipError_t e = (hl && (hr || membership)) ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess && total) e = hipMemcpy(hl, J->left, (size_t)total * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && total && !membership) e = hipMemcpy(hr, J->right, (size_t)total * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(hl); free(hr); return fail_hip(e, "join result copy"); }
    g_pool.put(c.device, J->left, J->capl); g_pool.put(c.device, J->right, J->capr);
    J->left = hl; J->right = hr; J->capl = J->capr = 0; J->on_device = false;
  }
  *out = J.release();
  return PLHIP_OK;
}

int32_t plhip_hash_join_multi(const plhip_column_t *left_keys, const plhip_column_t *right_keys, int32_t n_keys, int32_t how,
                              int32_t out_location, plhip_join_t **out) {
  if (!left_keys || !right_keys || !out) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *out = nullptr;
  if (n_keys < 1 || n_keys > PLHIP_MAX_KEYS) return fail(PLHIP_ERR_INVALID, "bad number of key columns");
  const int64_t nl = left_keys[0].length, nr = right_keys[0].length;
  for (int i = 0; i < n_keys; i++) {
    if (left_keys[i].dtype != right_keys[i].dtype) return fail(PLHIP_ERR_COMPUTE, "datatypes of join keys don't match");
    if (!dtype_width(left_keys[i].dtype)) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "unsupported join key dtype");
    if (left_keys[i].length != nl || right_keys[i].length != nr) return fail(PLHIP_ERR_SHAPE_MISMATCH, "join key columns of one frame differ in length");
  }
  if (n_keys == 1 && !dt_is_float(left_keys[0].dtype)) return plhip_hash_join(left_keys, right_keys, how, out_location, out);
  if (nl < 0 || nr < 0 || nl + nr > 0xFFFFFFFEll) return fail(PLHIP_ERR_COMPUTE, "more than u32::MAX rows (IdxSize is u32)");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  Staging st;
  void *gid = nullptr;
  HIP_TRY(hipMalloc(&gid, (size_t)std::max<int64_t>(nl + nr, 1) * 4));
  st.owned.push_back(gid);
  if (nl + nr > 0) {
    // both relations' key tuples end to end, one dense id per distinct tuple: equal tuples of either side share an id
    std::vector<plhip_column_t> cat(n_keys);
    for (int i = 0; i < n_keys; i++) {
      const plhip_column_t two[2] = {left_keys[i], right_keys[i]};
      const plhip_chunked_column_t cc{two, 2, 0};
      if ((rc = rechunk_column(cc, cat[i], st, c.stream))) return rc;
    }
    struct GroupsGuard { plhip_groups_t *p = nullptr; ~GroupsGuard() { delete p; } } gr;
    if ((rc = plhip_group_tuples(cat.data(), n_keys, 0, PLHIP_DEVICE, &gr.p))) return rc;
    const uint32_t *first = nullptr, *offsets = nullptr, *indices = nullptr;
    int32_t loc = 0;
    if ((rc = plhip_groups_buffers(gr.p, &first, &offsets, &indices, &loc))) return rc;
    launch_csr_to_gid(offsets, indices, (uint32_t)gr.p->n_groups, nl + nr, (uint32_t *)gid, c.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c.stream)); // (the CSR goes back with `gr`)
  }
  plhip_column_t lc{}, rcol{};
  lc.values = gid; lc.length = nl; lc.dtype = PLHIP_U32; lc.location = PLHIP_DEVICE;
  rcol.values = gid; rcol.offset = nl; rcol.length = nr; rcol.dtype = PLHIP_U32; rcol.location = PLHIP_DEVICE;
  return plhip_hash_join(&lc, &rcol, how, out_location, out);
}

int64_t plhip_join_num_rows(const plhip_join_t *j) { return j ? j->n : -1; }

int32_t plhip_join_buffers(const plhip_join_t *j, const uint32_t **left_idx, const uint32_t **right_idx, int32_t *location) {
  if (!j || !left_idx || !right_idx) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *left_idx = j->left; *right_idx = j->right;
  if (location) *location = j->on_device ? PLHIP_DEVICE : PLHIP_HOST;
  return PLHIP_OK;
}

void plhip_join_free(plhip_join_t *j) { delete j; }

int32_t plhip_take(const plhip_column_t *cols, int32_t n_cols, const uint32_t *idx, int64_t n_idx, int32_t idx_location, uint32_t flags,
                   plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_cols < 1 || n_cols > PLHIP_MAX_OUTS || !cols) return fail(PLHIP_ERR_INVALID, "bad column list");
  if (n_idx < 0 || (n_idx > 0 && !idx) || n_idx > 0xFFFFFFFFll) return fail(PLHIP_ERR_INVALID, "bad index array");
  if (idx_location != PLHIP_HOST && idx_location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "bad idx_location");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  Staging st;
  const uint32_t *d_idx = idx;
  if (idx_location == PLHIP_HOST && n_idx > 0) {
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, (size_t)n_idx * 4));
    st.owned.push_back(p);
    HIP_TRY(hipMemcpyAsync(p, idx, (size_t)n_idx * 4, hipMemcpyHostToDevice, c.stream));
    d_idx = (const uint32_t *)p;
  }
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result());
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->on_device = true; R->device = c.device; R->n_rows = n_idx;
  StreamDrain drain(c.stream); // (staged inputs and pooled outputs outlive the kernels on every path out)
  for (int i = 0; i < n_cols; i++) {
    if (cols[i].dtype == PLHIP_BOOL) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "plhip_take gathers fixed-width columns (cast a Boolean payload to u8)");
    DevCol d;
    if ((rc = stage_column(cols[i], d, st, c.stream))) return rc;
    const int w = dtype_width(cols[i].dtype);
    ResultCol col;
    col.dtype = cols[i].dtype;
    hipError_t e = g_pool.get(c.device, (size_t)std::max<int64_t>(n_idx, 1) * w, &col.values, &col.vcap);
    const bool nullable = d.validity != nullptr || (flags & PLHIP_TAKE_NULLABLE_IDX);
    if (e == hipSuccess && nullable) e = g_pool.get(c.device, (size_t)(n_idx / 64 + 2) * 8, (void **)&col.validity, &col.bcap);
    R->cols.push_back(col);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(output)");
    if (cols[i].length == 0 && n_idx > 0 && !(flags & PLHIP_TAKE_NULLABLE_IDX)) return fail(PLHIP_ERR_INVALID, "indices into an empty column");
    launch_take(d.values, w, d.validity, d.boff, d_idx, n_idx, (uint8_t *)col.values, (uint64_t *)col.validity, c.stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c.stream));
  drain.s = nullptr;
  return finish_result(R_own.release(), (uint64_t)n_idx, flags & PLHIP_FLAG_OUTPUT_DEVICE, out);
}

int32_t plhip_group_by_agg_noalloc(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values, int32_t n_values,
                                   const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags, int64_t n_groups_hint,
                                   const plhip_column_t *out_cols, int32_t n_out_cols, int64_t *n_groups) {
  if (!n_groups || !out_cols) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *n_groups = 0;
  if (n_keys != 1 || !keys) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc takes exactly one key column");
  if (n_values < 0 || n_values > PLHIP_MAX_VALS || (n_values > 0 && !values)) return fail(PLHIP_ERR_INVALID, "bad value columns");
  if (n_aggs < 0 || (n_aggs > 0 && !aggs)) return fail(PLHIP_ERR_INVALID, "bad aggregation list");
  if (flags & ~(uint32_t)PLHIP_FLAG_PARTIAL) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc accepts PLHIP_FLAG_PARTIAL only");
  if (n_groups_hint < 1) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc needs n_groups_hint (an upper bound on the groups): sampling would allocate");
  const int64_t n = keys[0].length;
  for (int i = 0; i < n_values; i++)
    if (values[i].length != n) return fail(PLHIP_ERR_SHAPE_MISMATCH, "series used as keys should have the same length as the dataframe");
  if (n > 0xFFFFFFFFll) return fail(PLHIP_ERR_COMPUTE, "more than u32::MAX rows (IdxSize is u32)");
  if (keys[0].location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc takes device-resident columns (staging would allocate)");
  for (int i = 0; i < n_values; i++)
    if (values[i].location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc takes device-resident columns (staging would allocate)");
  Ctx &c = g_ctx;
  // the stream, the pinned signal word and the scratch address are set up by plhip_init: nothing is created here
  if (c.device < 0 || !c.na_scratch) return fail(PLHIP_ERR_INVALID, "plhip_group_by_agg_noalloc: call plhip_init on this thread first");
  HIP_TRY(hipSetDevice(c.device));
  resolve_timing();
  Staging st; // (device columns: nothing is staged, nothing owned)
  std::vector<DevCol> dk(1), dv(n_values);
  int rc;
  if ((rc = stage_column(keys[0], dk[0], st, c.stream))) return rc;
  for (int i = 0; i < n_values; i++) if ((rc = stage_column(values[i], dv[i], st, c.stream))) return rc;
  Lowered L{};
  L.P.n_rows = n;
  L.n_out_keys = 1;
  if ((rc = build_keys(L.P, dk, st, L.KG))) return rc; // single key: no measuring pass, no allocation
  if ((rc = build_aggs(L, dv, aggs, n_aggs, flags))) return rc;
  return run_direct_noalloc(L, dk, n_groups_hint, out_cols, n_out_cols, n_groups);
}

int32_t group_by_agg_impl(const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                          int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                          const plhip_options_t *opts, plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  // group_by/mod.rs:59-62
  if (n_keys < 1 || !keys) return fail(PLHIP_ERR_COMPUTE, "at least one key is required in a group_by operation");
  if (n_keys > PLHIP_MAX_KEYS) return fail(PLHIP_ERR_INVALID, "too many key columns (max 16)");
  if (n_values < 0 || n_values > PLHIP_MAX_VALS || (n_values > 0 && !values)) return fail(PLHIP_ERR_INVALID, "bad value columns");
  if (n_aggs < 0 || (n_aggs > 0 && !aggs)) return fail(PLHIP_ERR_INVALID, "bad aggregation list");
  const int64_t n = keys[0].length;
  // group_by/mod.rs:67-71
  for (int i = 0; i < n_keys; i++)
    if (keys[i].length != n) return fail(PLHIP_ERR_SHAPE_MISMATCH, "series used as keys should have the same length as the dataframe");
  for (int i = 0; i < n_values; i++)
    if (values[i].length != n) return fail(PLHIP_ERR_SHAPE_MISMATCH, "series used as keys should have the same length as the dataframe");
  if (n > 0xFFFFFFFFll) return fail(PLHIP_ERR_COMPUTE, "more than u32::MAX rows (IdxSize is u32)");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;

  Staging st;
  std::vector<DevCol> dk(n_keys), dv(n_values);
  for (int i = 0; i < n_keys; i++) if ((rc = stage_column(keys[i], dk[i], st, c.stream))) return rc;
  for (int i = 0; i < n_values; i++) if ((rc = stage_column(values[i], dv[i], st, c.stream))) return rc;

  Lowered L{};
  L.P.n_rows = n;
  L.n_out_keys = n_keys;
  if ((rc = build_keys(L.P, dk, st, L.KG))) return rc;
  if ((rc = build_aggs(L, dv, aggs, n_aggs, flags))) return rc;
  Plan &P = L.P;

  // ---- strategy ------------------------------------------------------------------------------------
  int strategy = c.forced_strategy >= 0 ? c.forced_strategy : env_strategy();
  if (flags & PLHIP_FLAG_EXACT_FOLD) { // bit-exact float sums: only the sort strategy folds a group in row order
    if (flags & PLHIP_FLAG_PARTIAL) return fail(PLHIP_ERR_INVALID, "PLHIP_FLAG_EXACT_FOLD does not combine with PLHIP_FLAG_PARTIAL");
    bool any_float = false;
    for (int a = 0; a < P.n_accs; a++) any_float |= P.accs[a].op == ACC_ADD_F64 || P.accs[a].op == ACC_ADD_F32;
    if (any_float && n > 0) {
      P.exact_fold = 1;
      bool handled = false;
      if ((rc = run_sorted(L, dk, flags, out, handled, (flags & PLHIP_FLAG_KEYS_SORTED) && n_keys == 1))) return rc;
      if (handled) return PLHIP_OK;
      return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "PLHIP_FLAG_EXACT_FOLD: the sort strategy declined this query (more than 20 accumulators, or a "
                                               "nullable key spanning all 2^64 values)");
    }
  }
  if ((flags & PLHIP_FLAG_KEYS_SORTED) && n_keys == 1 && n > 0) { // into_groups.rs:142-150: the flag wins over every hash path
    bool handled = false;
    if ((rc = run_sorted(L, dk, flags, out, handled, true))) return rc;
    if (handled) return PLHIP_OK;
  }
  // A plain 32-bit integer key (db-benchmark's id4 .. id6) whose values sit in a compact range IS a dense code once its minimum is
  // subtracted: one vectorised min / max pass over the key column (it replaces the cardinality sample the hashed pipelines
  // would take) and the dense-code path runs without hashing, probing or a sorted copy.  The reference has no counterpart for
  // plain integers (group_tuples_perfect serves Categoricals only, perfect.rs:190-227); the result is the same set of groups.
  int64_t range_bound = 0; // > 0: the key takes at most this many distinct values (a bound the hashed pipelines can use as a hint)
  if (strategy == PLHIP_STRATEGY_AUTO && !(flags & (PLHIP_FLAG_KEYS_DENSE | PLHIP_FLAG_MAINTAIN_ORDER | PLHIP_FLAG_KEYS_SORTED)) && n_keys == 1 &&
      (dk[0].dtype == PLHIP_I32 || dk[0].dtype == PLHIP_U32) && n >= (1 << 20) && !getenv("PLHIP_NO_RANGE_DENSE")) {
    int64_t hint = 0;
    if (opts && opts->struct_size >= offsetof(plhip_options_t, n_groups_hint) + sizeof(int64_t)) hint = opts->n_groups_hint;
    if (hint <= 0 || hint > 4096) { // (a few thousand groups: DIRECT's single table is the right tool, no pass needed)
      std::vector<KeyCol> one(1);
      one[0].col = dk[0]; one[0].nullable = dk[0].validity != nullptr;
      if ((rc = measure_key_columns(one, n))) return rc;
      const uint64_t span = one[0].hi - one[0].lo; // normalised bits: order-preserving for i32 and u32 alike
      if (span < (uint64_t)n && span < 0x7FFFFFF0ull) {
        range_bound = (int64_t)span + 1 + (one[0].nullable ? 1 : 0);
        const uint32_t raw_min = dk[0].dtype == PLHIP_I32 ? (uint32_t)one[0].lo ^ 0x80000000u : (uint32_t)one[0].lo;
        bool handled = false;
        if ((rc = run_dense(L, dk, flags, (int64_t)span + 1, false, out, handled, raw_min))) return rc;
        if (handled) return PLHIP_OK;
      }
    }
  }
  if (c.dense_comm) { // the sharded dense path (sharded_impl): the whole query or nothing -- *out stays NULL when every rank declines
    bool handled = false;
    int64_t hint = 0;
    if (opts && opts->struct_size >= offsetof(plhip_options_t, n_groups_hint) + sizeof(int64_t)) hint = opts->n_groups_hint;
    if (!(flags & PLHIP_FLAG_KEYS_DENSE) || (strategy != PLHIP_STRATEGY_AUTO && strategy != PLHIP_STRATEGY_DENSE)) return PLHIP_OK;
    return run_dense(L, dk, flags, hint, true, out, handled, 0, c.dense_comm);
  }
  if ((flags & PLHIP_FLAG_KEYS_DENSE) && (strategy == PLHIP_STRATEGY_AUTO || strategy == PLHIP_STRATEGY_DENSE)) {
    int64_t hint = 0;
    if (opts && opts->struct_size >= offsetof(plhip_options_t, n_groups_hint) + sizeof(int64_t)) hint = opts->n_groups_hint;
    bool handled = false;
    if ((rc = run_dense(L, dk, flags, hint, strategy == PLHIP_STRATEGY_DENSE, out, handled))) return rc;
    if (handled) return PLHIP_OK;
  }
  if (strategy == PLHIP_STRATEGY_DENSE) strategy = PLHIP_STRATEGY_AUTO; // (not declared dense, or declined: the hashed strategies)
  if (strategy == PLHIP_STRATEGY_AUTO || strategy == PLHIP_STRATEGY_DIRECT || strategy == PLHIP_STRATEGY_TILESORT) {
    int64_t hint = 0;
    if (opts && opts->struct_size >= offsetof(plhip_options_t, n_groups_hint) + sizeof(int64_t)) hint = opts->n_groups_hint;
    if (hint <= 0 && range_bound > 0 && range_bound <= n / 2) hint = range_bound; // the measured value range bounds the groups: no sample
    bool handled = false;
    uint64_t g_seen = 0;
    if ((rc = run_fast(L, dk, flags, hint, strategy, out, handled, g_seen))) return rc;
    if (handled) return PLHIP_OK;
    // more groups than the LDS pipeline holds (or wide key tuples): sort instead of hammering HBM with atomics
    strategy = (n >= (1 << 16) && (g_seen > 4096 || L.KG.enabled)) ? PLHIP_STRATEGY_SORT : PLHIP_STRATEGY_AUTO;
    if (strategy == PLHIP_STRATEGY_SORT && g_seen > 4096 && ((uint64_t)n / g_seen >= 2 || (leaf_sort_eligible(L.P) && !L.KG.enabled))) { // beyond one partition level: two
      // levels before the sort (nearly all-unique keys: the two are on par, and the sort is immune to skew)
      bool over = false;
      if ((rc = run_two_level(L, dk, flags, g_seen, false, out, handled, &over))) return rc;
      if (handled) return PLHIP_OK;
      if (over && g_seen * 8 <= (uint64_t)n) { // once more with room for 8x the groups (still far cheaper than the sort)
        if ((rc = run_two_level(L, dk, flags, g_seen * 8, false, out, handled))) return rc;
        if (handled) return PLHIP_OK;
      }
    }
  }
  if (strategy == PLHIP_STRATEGY_TILESORT2) {
    bool handled = false;
    uint64_t g_est = (uint64_t)n;
    int64_t hint = 0;
    if (opts && opts->struct_size >= offsetof(plhip_options_t, n_groups_hint) + sizeof(int64_t)) hint = opts->n_groups_hint;
    if (hint > 0) g_est = (uint64_t)std::min<int64_t>(hint, n);
    else if (n > 0 && (rc = estimate_groups_sampled(P, g_est))) return rc;
    bool over = false;
    if ((rc = run_two_level(L, dk, flags, std::max<uint64_t>(g_est, 1), true, out, handled, &over))) return rc;
    if (handled) return PLHIP_OK;
    strategy = over ? PLHIP_STRATEGY_SORT : PLHIP_STRATEGY_AUTO; // (declined: stable order, wide tuples -> the general path)
  }
  if (strategy == PLHIP_STRATEGY_SORT) {
    bool handled = false;
    if ((rc = run_sorted(L, dk, flags, out, handled))) return rc;
    if (handled) return PLHIP_OK;
    strategy = PLHIP_STRATEGY_AUTO;
  }
  resolve_timing();
  HIP_TRY(hipEventRecord(c.ev[0], c.stream)); // the general path's timing starts here (the LDS-only pipelines above record their own events)
  uint64_t g_est = (uint64_t)n;
  if (n >= (1 << 14) && (rc = estimate_groups(P, g_est))) return rc;
  if (strategy == PLHIP_STRATEGY_AUTO && (flags & PLHIP_FLAG_MAINTAIN_ORDER) && n >= (1 << 20) && g_est > (1u << 16)) {
    // stable order with many groups: the sort strategy (reordered by first row) instead of an HBM table full of atomics
    bool handled = false;
    if ((rc = run_sorted(L, dk, flags, out, handled))) return rc;
    if (handled) return PLHIP_OK;
  }
  const int key_bytes = packed_key_bytes(P, dk);
  // largest LDS table (log2 slots) for two / one workgroup(s) per CU
  uint32_t lbits2 = 13, lbits1 = 14;
  while (lbits2 > 6 && lds2_table_bytes(P, lbits2, key_bytes) > 78 * 1024) lbits2--;
  while (lbits1 > 6 && lds2_table_bytes(P, lbits1, key_bytes) > 150 * 1024) lbits1--;
  if (strategy == PLHIP_STRATEGY_AUTO) {
    if (n < (1 << 14)) strategy = PLHIP_STRATEGY_GLOBAL;
    else if (g_est <= (1ull << lbits1) / 2) strategy = PLHIP_STRATEGY_LDS;
    else strategy = PLHIP_STRATEGY_PARTITIONED;
  }
  PartitionSetup PS;
  if (strategy == PLHIP_STRATEGY_PARTITIONED) {
    // the partitioned intermediates carry no null flag: a nullable single key is re-lowered as a packed code
    if (P.single_key && dk[0].validity) {
      std::vector<KeyCol> one(1);
      one[0].col = dk[0]; one[0].nullable = true;
      if ((rc = measure_key_columns(one, n))) return rc;
      if (one[0].bits <= 64) { P.single_key = 0; pack_keys(P, one); }
      else strategy = PLHIP_STRATEGY_GLOBAL;
    }
    if (strategy == PLHIP_STRATEGY_PARTITIONED && !plan_partitions(P, g_est, packed_key_bytes(P, dk), lbits2, lbits1, PS))
      strategy = PLHIP_STRATEGY_GLOBAL; // more groups than one partition level can bring into LDS
  }

  uint32_t fl[16] = {0};
  uint64_t cap = 0;
  TableRef T{};
  for (int attempt = 0;; attempt++) {
    uint64_t dense_cap = 0;
    if (strategy == PLHIP_STRATEGY_PARTITIONED) {
      dense_cap = attempt == 0 ? std::min<uint64_t>((uint64_t)n, std::max<uint64_t>(4 * g_est, 1 << 16)) : (uint64_t)n;
      const uint64_t want = attempt == 0 ? std::min<uint64_t>((uint64_t)n, std::max<uint64_t>(g_est / 4, 1 << 15)) : (uint64_t)n;
      cap = next_pow2(2 * want);
    } else {
      const uint64_t want = attempt == 0 ? std::min<uint64_t>((uint64_t)n, g_est * 2 + 1024) : (uint64_t)n;
      cap = next_pow2(std::max<uint64_t>(2 * want, 1024));
    }
    const uint64_t n_slots = cap + 2 + dense_cap;
    plan_layout(P, n_slots);
    HIP_TRY(c.table.reserve(table_bytes(P, n_slots)));
    HIP_TRY(c.flags.reserve(FLAGS_BYTES));
    HIP_TRY(c.slots.reserve(n_slots * 8));
    T = TableRef{(uint8_t *)c.table.p, cap, (uint32_t *)c.flags.p};
    uint64_t *slots_dev = (uint64_t *)c.slots.p;
    launch_table_init(P, T, c.stream);
    HIP_TRY(hipEventRecord(c.ev[1], c.stream));
    if (strategy == PLHIP_STRATEGY_PARTITIONED) {
      if ((rc = run_partitioned(P, T, PS, dense_cap, slots_dev, L.first_acc >= 0))) return rc;
      c.strategy = "partitioned";
    } else if (strategy == PLHIP_STRATEGY_LDS) {
      // unpartitioned input cut into row slices; every slice merges its LDS table into the HBM table
      uint32_t lbits = lbits2;
      if (g_est < (uint64_t)n) { // size the LDS table to the estimate (4x head-room), not to the budget
        uint32_t fit = 6;
        while ((1ull << fit) < 4 * g_est && fit < lbits1) fit++;
        lbits = fit <= lbits2 ? fit : lbits1;
      }
      const size_t lbytes = lds2_table_bytes(P, lbits, key_bytes);
      const int64_t max_wgs = (int64_t)plhip_num_cus() * (lbytes <= 78 * 1024 ? 2 : 1);
      int64_t n_items = std::max<int64_t>(1, std::min<int64_t>(max_wgs, (n + 4095) / 4096));
      int64_t per = (n + n_items - 1) / n_items;
      per = (per + 1023) / 1024 * 1024;
      std::vector<WorkItem> items;
      for (int64_t r = 0; r < n; r += per) items.push_back(WorkItem{r, std::min<int64_t>(n, r + per), 0u, 0u});
      if (items.empty()) items.push_back(WorkItem{0, 0, 0u, 0u});
      const uint32_t cnt[2] = {(uint32_t)items.size(), 0u};
      HIP_TRY(c.items.reserve(items.size() * sizeof(WorkItem) + 64));
      HIP_TRY(hipMemcpyAsync(c.items.p, cnt, 8, hipMemcpyHostToDevice, c.stream));
      HIP_TRY(hipMemcpyAsync((uint8_t *)c.items.p + 64, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, c.stream));
      launch_lds_agg2(P, T, (const WorkItem *)((uint8_t *)c.items.p + 64), (const uint32_t *)c.items.p, (uint32_t)items.size(),
                      (uint32_t *)c.items.p + 1, lbits, key_bytes, 0, slots_dev, c.stream);
      c.strategy = "lds"; c.dom_kernel = "k_lds_agg2";
      HIP_TRY(hipEventRecord(c.ev[4], c.stream));
      HIP_TRY(hipEventRecord(c.ev[5], c.stream));
    } else {
      launch_agg_global(P, T, c.stream);
      c.strategy = "global"; c.dom_kernel = "k_agg_global";
      HIP_TRY(hipEventRecord(c.ev[4], c.stream));
      HIP_TRY(hipEventRecord(c.ev[5], c.stream));
    }
    HIP_TRY(hipEventRecord(c.ev[2], c.stream));
    launch_compact(P, T, slots_dev, c.stream);
    HIP_TRY(hipGetLastError());
    if ((rc = read_flags(T, fl))) return rc;
    if (!fl[FLAG_OVERFLOW]) break;
    if (attempt >= 1) return fail(PLHIP_ERR_COMPUTE, "hash table overflow at load factor <= 0.5");
  }
  const uint64_t G = fl[FLAG_NGROUPS];
  uint64_t *slots = (uint64_t *)c.slots.p;

  // ---- ordering --------------------------------------------------------------------------------------
  if ((flags & PLHIP_FLAG_MAINTAIN_ORDER) && G > 1) { // groups by first row (hashing.rs:32-79)
    HIP_TRY(c.sortkeys.reserve(G * 8));
    launch_gather_sort_keys(P, T, slots, G, L.first_acc, (uint64_t *)c.sortkeys.p, c.stream);
    HIP_TRY(sort_pairs_u64((uint64_t *)c.sortkeys.p, slots, G, 32, c.stream));
  }

  // ---- outputs ---------------------------------------------------------------------------------------
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->n_rows = (int64_t)G;
  R->on_device = true; // until copied
  R->device = c.device;
  OutPlan O{};
  O.n_outs = (int)L.outs.size();
  for (size_t i = 0; i < L.outs.size(); i++) {
    OutSpec o = L.outs[i];
    ResultCol rcnew;
    rcnew.dtype = (int)o.dtype;
    const size_t vb = (size_t)std::max<uint64_t>(G, 1) * dtype_width((int)o.dtype);
    hipError_t e = hipMalloc(&rcnew.values, vb);
    if (e == hipSuccess && o.validity) e = hipMalloc((void **)&rcnew.validity, (G / 64 + 1) * 8);
    R->cols.push_back(rcnew);
    if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
    o.values = (uint8_t *)rcnew.values;
    o.validity = rcnew.validity;
    O.outs[i] = o;
  }
  HIP_TRY(c.plans.reserve(sizeof(OutPlan) + sizeof(KeyGather)));
  HIP_TRY(hipMemcpyAsync(c.plans.p, &O, sizeof(OutPlan), hipMemcpyHostToDevice, c.stream));
  HIP_TRY(hipMemcpyAsync((uint8_t *)c.plans.p + sizeof(OutPlan), &L.KG, sizeof(KeyGather), hipMemcpyHostToDevice, c.stream));
  launch_finalize(P, T, (const OutPlan *)c.plans.p, (const KeyGather *)((uint8_t *)c.plans.p + sizeof(OutPlan)), slots, G, c.stream);
  HIP_TRY(hipEventRecord(c.ev[3], c.stream));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) { return fail_hip(e, "group-by kernels"); }
  hipEventElapsedTime(&c.total_ms, c.ev[0], c.ev[3]);
  {
    // dominant kernel: the aggregation pass, or the partition scatter when that is the longer one
    float t_all = 0.f, t_part = 0.f, t_agg = 0.f;
    hipEventElapsedTime(&t_all, c.ev[1], c.ev[2]);
    if (c.strategy == "partitioned") {
      hipEventElapsedTime(&t_part, c.ev[1], c.ev[4]);
      hipEventElapsedTime(&t_agg, c.ev[4], c.ev[5]);
      c.part_ms = t_part; c.agg_ms = t_agg;
      if (t_part > t_agg) { c.dom_ms = t_part; c.dom_kernel = "k_part_hist+k_part_scan+k_part_plan+k_part_scatter"; }
      else { c.dom_ms = t_agg; c.dom_kernel = "k_lds_agg2"; }
    } else { c.dom_ms = t_all; c.part_ms = 0.f; c.agg_ms = t_all; }
  }

  return finish_result(R_own.release(), G, flags, out);
}

int64_t plhip_result_num_rows(const plhip_result_t *r) { return r ? r->n_rows : -1; }
int32_t plhip_result_num_columns(const plhip_result_t *r) { return r ? (int32_t)r->cols.size() : -1; }
namespace {
// the frame's buffers are about to be read from outside the stream that produced them: wait for that stream's tail (once)
int settle_result(const plhip_result_t *r) {
  if (!r->pending) return PLHIP_OK;
  const hipError_t e = settle_token(r->pending);
  return e == hipSuccess ? PLHIP_OK : fail_hip(e, "group-by kernels (reported when the frame was first read)");
}
} // namespace
int32_t plhip_result_column(const plhip_result_t *r, int32_t i, plhip_column_t *col) {
  if (!r || !col || i < 0 || i >= (int32_t)r->cols.size()) return fail(PLHIP_ERR_INVALID, "bad result column index");
  { const int rc = settle_result(r); if (rc) return rc; }
  col->values = r->cols[i].values;
  col->validity = r->cols[i].validity;
  col->offset = r->row_offset;
  col->length = r->n_rows;
  col->dtype = r->cols[i].dtype;
  col->location = r->on_device ? PLHIP_DEVICE : PLHIP_HOST;
  return PLHIP_OK;
}
void plhip_result_free(plhip_result_t *r) { delete r; }

// ---- two-step API (filled in by groups.hip) --------------------------------------------------------
int64_t plhip_groups_num_groups(const plhip_groups_t *g) { return g ? g->n_groups : -1; }
int64_t plhip_groups_num_rows(const plhip_groups_t *g) { return g ? g->n_rows : -1; }
int32_t plhip_groups_buffers(const plhip_groups_t *g, const uint32_t **first, const uint32_t **offsets,
                             const uint32_t **indices, int32_t *location) {
  if (!g) return fail(PLHIP_ERR_INVALID, "groups handle is NULL");
  if (first) *first = g->first;
  if (offsets) *offsets = g->offsets;
  if (indices) *indices = g->indices;
  if (location) *location = g->on_device ? PLHIP_DEVICE : PLHIP_HOST;
  return PLHIP_OK;
}
void plhip_groups_free(plhip_groups_t *g) { delete g; }

// IntoGroupsProxy::group_tuples / group_by_threaded_multiple_keys_flat -> GroupsProxy::Idx as CSR.
// Dense group ids come from the HBM key table (one find-or-insert per row), the index lists from a stable
// radix sort of (group id, row): rows of a group stay in ascending order, as the reference builds them.
int32_t plhip_group_tuples(const plhip_column_t *keys, int32_t n_keys, int32_t sorted, int32_t out_location,
                           plhip_groups_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_keys < 1 || !keys) return fail(PLHIP_ERR_COMPUTE, "at least one key is required in a group_by operation");
  if (n_keys > PLHIP_MAX_KEYS) return fail(PLHIP_ERR_INVALID, "too many key columns (max 16)");
  if (out_location != PLHIP_HOST && out_location != PLHIP_DEVICE) return fail(PLHIP_ERR_INVALID, "bad out_location");
  const int64_t n = keys[0].length;
  for (int i = 0; i < n_keys; i++)
    if (keys[i].length != n) return fail(PLHIP_ERR_SHAPE_MISMATCH, "series used as keys should have the same length as the dataframe");
  if (n > 0xFFFFFFFFll) return fail(PLHIP_ERR_COMPUTE, "more than u32::MAX rows (IdxSize is u32)");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  {
    // Boolean keys group as their UInt8 cast (IntoGroupsProxy for BooleanChunked, into_groups.rs:215-230): unpacked on the device
    bool any_bool = false;
    for (int i = 0; i < n_keys; i++) any_bool |= keys[i].dtype == PLHIP_BOOL;
    if (any_bool) {
      Staging bst;
      std::vector<plhip_column_t> lk(keys, keys + n_keys);
      for (int i = 0; i < n_keys; i++)
        if (lk[i].dtype == PLHIP_BOOL && (rc = lower_bool_column(lk[i], 1, PLHIP_U8, bst))) { (void)hipStreamSynchronize(c.stream); return rc; }
      rc = plhip_group_tuples(lk.data(), n_keys, sorted, out_location, out);
      (void)hipStreamSynchronize(c.stream); // the unpacked copies are freed on the way out
      return rc;
    }
  }
  plhip_groups *Gp = new (std::nothrow) plhip_groups();
  if (!Gp) return fail(PLHIP_ERR_OOM, "host allocation failed");
  Gp->n_rows = n;
  Gp->on_device = true;
  uint64_t G = 0;
  Gp->device = c.device;
  // the CSR arrays come from the result-buffer pool and the scratch from the thread's workspace: at 1e7 rows the four hipMalloc, the
  // hipFree and the sort helper's own allocations were 10 of the call's 12.6 ms
  auto dev_alloc = [&](uint32_t **p, size_t *cap, size_t count) { return g_pool.get(c.device, (count ? count : 1) * 4, (void **)p, cap); };
  if (n > 0) {
    Staging st;
    std::vector<DevCol> dk(n_keys);
    for (int i = 0; i < n_keys; i++) if ((rc = stage_column(keys[i], dk[i], st, c.stream))) { delete Gp; return rc; }
    Lowered L{};
    L.P.n_rows = n;
    if ((rc = build_keys(L.P, dk, st, L.KG))) { delete Gp; return rc; }
    Plan &P = L.P;
    // the table is sized from a sampled estimate of the groups (the general path's, estimate_unique_count's analogue): a table
    // for n distinct keys is 2^25 slots at 1e7 rows, and initialising and compacting it cost 1.1 ms for 1e5 groups.  A low
    // estimate overflows (probe limit) and the second attempt takes the table every input fits.
    uint64_t g_est = (uint64_t)n;
    if (n >= (1 << 14) && (rc = estimate_groups(P, g_est))) { delete Gp; return rc; }
    const int first_acc = add_acc(P, ACC_FIRST_ROW, CONV_NONE, -1, 4, 0xFFFFFFFFull);
    TableRef T{};
    uint64_t *slots = nullptr;
    uint32_t fl[16];
    hipError_t e = hipSuccess;
    for (int attempt = 0;; attempt++) {
      const uint64_t want = attempt == 0 ? std::min<uint64_t>((uint64_t)n, g_est * 2 + 1024) : (uint64_t)n;
      const uint64_t cap = next_pow2(std::max<uint64_t>(2 * want, 1024));
      const uint64_t n_slots = cap + 2;
      plan_layout(P, n_slots);
      e = c.table.reserve(table_bytes(P, n_slots));
      if (e == hipSuccess) e = c.flags.reserve(FLAGS_BYTES);
      if (e == hipSuccess) e = c.slots.reserve(2 * n_slots * 8); // the slot list, and its copy in first-row order
      if (e == hipSuccess) e = c.rank.reserve(n_slots * 4);
      if (e != hipSuccess) { delete Gp; return fail_hip(e, "group_tuples scratch"); }
      T = TableRef{(uint8_t *)c.table.p, cap, (uint32_t *)c.flags.p};
      slots = (uint64_t *)c.slots.p;
      launch_table_init(P, T, c.stream);
      launch_agg_global(P, T, c.stream);
      launch_compact(P, T, slots, c.stream);
      if ((rc = read_flags(T, fl))) { delete Gp; return rc; }
      if (!fl[FLAG_OVERFLOW]) break;
      if (attempt >= 1) { delete Gp; return fail(PLHIP_ERR_COMPUTE, "hash table overflow at load factor <= 0.5"); }
    }
    G = fl[FLAG_NGROUPS];
    const size_t tmp_bytes = sort_pairs_tmp_bytes((uint64_t)n);
    const size_t n4 = ((size_t)n * 4 + 255) & ~(size_t)255, g8 = ((size_t)G * 8 + 255) & ~(size_t)255;
    if ((e = c.sortkeys.reserve(tmp_bytes + 2 * g8)) != hipSuccess || (e = c.part.reserve(3 * n4)) != hipSuccess) { delete Gp; return fail_hip(e, "group_tuples scratch"); }
    uint8_t *tmp = (uint8_t *)c.sortkeys.p;
    if (sorted && G > 1) { // groups ordered by their first row (hashing.rs:32-79)
      uint64_t *k_in = (uint64_t *)(tmp + tmp_bytes), *k_out = (uint64_t *)(tmp + tmp_bytes + g8), *slots_out = slots + (T.cap + 2);
      launch_gather_sort_keys(P, T, slots, G, first_acc, k_in, c.stream);
      if ((e = sort_pairs_u64_ws(k_in, k_out, slots, slots_out, G, 32, tmp, tmp_bytes, c.stream)) != hipSuccess) { delete Gp; return fail_hip(e, "sort"); }
      slots = slots_out;
    }
    uint32_t *ids = (uint32_t *)c.part.p, *ids_sorted = (uint32_t *)((uint8_t *)c.part.p + n4), *rows = (uint32_t *)((uint8_t *)c.part.p + 2 * n4);
    e = dev_alloc(&Gp->indices, &Gp->cap_indices, (size_t)n);
    if (e == hipSuccess) e = dev_alloc(&Gp->offsets, &Gp->cap_offsets, (size_t)G + 1);
    if (e == hipSuccess) e = dev_alloc(&Gp->first, &Gp->cap_first, (size_t)G);
    if (e == hipSuccess) {
      launch_rank_scatter(slots, G, (uint32_t *)c.rank.p, c.stream);
      launch_encode_rows(P, T, (const uint32_t *)c.rank.p, ids, c.stream);
      launch_iota_u32(rows, (uint64_t)n, c.stream);
      // stable: the rows of a group stay ascending (the invariant of GroupsIdx, proxy.rs:16-20)
      e = sort_pairs_u32_ws(ids, ids_sorted, rows, Gp->indices, (uint64_t)n, std::max(1, bit_length(G ? G - 1 : 0)), tmp, tmp_bytes, c.stream);
    }
    if (e == hipSuccess) {
      launch_csr(ids_sorted, Gp->indices, (uint64_t)n, G, Gp->offsets, Gp->first, c.stream);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) { delete Gp; return fail_hip(e, "group_tuples"); }
  } else {
    hipError_t e = dev_alloc(&Gp->indices, &Gp->cap_indices, 0);
    if (e == hipSuccess) e = dev_alloc(&Gp->offsets, &Gp->cap_offsets, 1);
    if (e == hipSuccess) e = dev_alloc(&Gp->first, &Gp->cap_first, 0);
    if (e == hipSuccess) e = hipMemset(Gp->offsets, 0, 4);
    if (e != hipSuccess) { delete Gp; return fail_hip(e, "group_tuples"); }
  }
  Gp->n_groups = (int64_t)G;
  if (out_location == PLHIP_HOST) {
    plhip_groups *H = new (std::nothrow) plhip_groups();
    if (!H) { delete Gp; return fail(PLHIP_ERR_OOM, "host allocation failed"); }
    H->n_groups = Gp->n_groups; H->n_rows = n; H->on_device = false;
    rc = copy_out(Gp->first, (size_t)G * 4, (void **)&H->first, &H->cap_first);
    if (!rc) rc = copy_out(Gp->offsets, ((size_t)G + 1) * 4, (void **)&H->offsets, &H->cap_offsets);
    if (!rc) rc = copy_out(Gp->indices, (size_t)n * 4, (void **)&H->indices, &H->cap_indices);
    delete Gp;
    if (rc) { delete H; return rc; }
    Gp = H;
  }
  *out = Gp;
  return PLHIP_OK;
}

static int32_t agg_groups_impl(const plhip_column_t *values, int32_t kind, const plhip_groups_t *groups, uint32_t flags, double quantile,
                               int interp, plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!groups) return fail(PLHIP_ERR_INVALID, "groups handle is NULL");
  if (kind < 0 || kind > AGG_EXT_QUANTILE) return fail(PLHIP_ERR_INVALID, "bad aggregation kind");
  if (kind != PLHIP_AGG_COUNT && !values) return fail(PLHIP_ERR_INVALID, "values is NULL");
  if (values && kind != PLHIP_AGG_COUNT && values->length != groups->n_rows)
    return fail(PLHIP_ERR_SHAPE_MISMATCH, "the column and the groups differ in length");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  const uint64_t G = (uint64_t)groups->n_groups;
  const int64_t n = groups->n_rows;
  if (kind >= PLHIP_AGG_FIRST) { // first / last / var / std: their own kernel
    Staging st;
    DevCol dv;
    if ((rc = stage_column(*values, dv, st, c.stream))) return rc;
    const uint32_t *offsets = groups->offsets, *indices = groups->indices;
    if (!groups->on_device) {
      void *d_off = nullptr, *d_idx = nullptr;
      HIP_TRY(hipMalloc(&d_off, (G + 1) * 4));
      st.owned.push_back(d_off);
      HIP_TRY(hipMalloc(&d_idx, (size_t)std::max<int64_t>(n, 1) * 4));
      st.owned.push_back(d_idx);
      HIP_TRY(hipMemcpyAsync(d_off, groups->offsets, (G + 1) * 4, hipMemcpyHostToDevice, c.stream));
      if (n) HIP_TRY(hipMemcpyAsync(d_idx, groups->indices, (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
      offsets = (const uint32_t *)d_off; indices = (const uint32_t *)d_idx;
    }
    const bool moment = kind == PLHIP_AGG_VAR || kind == PLHIP_AGG_STD || kind == PLHIP_AGG_MEDIAN || kind == AGG_EXT_QUANTILE; // float-valued results
    std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
    plhip_result *R = R_own.get();
    if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
    R->n_rows = (int64_t)G; R->on_device = true; R->device = c.device;
    ResultCol col;
    col.dtype = moment ? (dv.dtype == PLHIP_F32 ? PLHIP_F32 : PLHIP_F64) : dv.dtype; // aggregations/mod.rs:781-803, 1018-1035
    if (kind == PLHIP_AGG_N_UNIQUE) col.dtype = PLHIP_U32; // IdxSize (dispatch.rs:82-107)
    hipError_t e = hipMalloc(&col.values, (size_t)std::max<uint64_t>(G, 1) * dtype_width(col.dtype));
    if (e == hipSuccess) e = hipMalloc((void **)&col.validity, (G / 64 + 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (G / 64 + 1) * 8, c.stream);
    R->cols.push_back(col);
    if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
    AggExtArgs A{};
    A.values = dv.values; A.validity = dv.validity; A.boff = dv.boff; A.dtype = (uint32_t)dv.dtype; A.width = (uint32_t)dtype_width(dv.dtype);
    A.kind = (uint32_t)kind; A.ddof = (flags >> 8) & 0xFFu; A.out_dtype = (uint32_t)col.dtype;
    A.out = (uint8_t *)col.values; A.out_validity = col.validity;
    if (kind == PLHIP_AGG_MEDIAN || kind == PLHIP_AGG_N_UNIQUE || kind == AGG_EXT_QUANTILE) {
      if (n > 0x7FFFFFFFll) { return fail(PLHIP_ERR_INVALID, "too many rows for the segmented sort"); }
      size_t tb = 0;
      e = median_temp_bytes((uint64_t)n, G, &tb);
      const size_t n8 = ((size_t)std::max<int64_t>(n, 1) * 8 + 255) & ~(size_t)255;
      const size_t n4 = ((size_t)std::max<int64_t>(n, 1) * 4 + 255) & ~(size_t)255;
      if (e == hipSuccess) e = c.table.reserve(2 * n8 + 2 * n4);
      uint32_t *gid_a = (uint32_t *)((uint8_t *)c.table.p + 2 * n8), *gid_b = (uint32_t *)((uint8_t *)c.table.p + 2 * n8 + n4);
      if (e == hipSuccess) e = c.sortkeys.reserve(tb + 256);
      if (e == hipSuccess && kind == PLHIP_AGG_N_UNIQUE) e = c.rank.reserve((G / 16 + 1) * 4);
      if (e == hipSuccess && kind == PLHIP_AGG_N_UNIQUE)
        e = launch_n_unique(A, offsets, indices, (uint64_t)n, G, (uint64_t *)c.table.p, (uint64_t *)((uint8_t *)c.table.p + n8), gid_a, gid_b, (uint32_t *)c.rank.p,
                            c.sortkeys.p, tb, c.stream);
      else if (e == hipSuccess)
        e = launch_quantile(A, offsets, indices, (uint64_t)n, G, (uint64_t *)c.table.p, (uint64_t *)((uint8_t *)c.table.p + n8), gid_a, gid_b, c.sortkeys.p, tb,
                            quantile, interp, c.stream);
    } else {
      e = c.misc.reserve(agg_long_scratch_bytes((uint64_t)n));
      if (e == hipSuccess) {
        launch_agg_groups_ext(A, offsets, indices, (uint64_t)n, G, moment && G > 0 && (uint64_t)n / G >= 64, c.misc.p, c.stream);
        e = hipGetLastError();
      }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) { return fail_hip(e, "agg_groups"); }
    return finish_result(R_own.release(), G, flags & PLHIP_FLAG_OUTPUT_DEVICE, out);
  }
  Staging st;
  std::vector<DevCol> dv;
  if (kind != PLHIP_AGG_COUNT) {
    dv.resize(1);
    if ((rc = stage_column(*values, dv[0], st, c.stream))) return rc;
  }
  Lowered L{};
  L.P.n_rows = n;
  L.n_out_keys = 0;
  plhip_agg_t a{0, kind};
  if ((rc = build_aggs(L, dv, &a, 1, 0))) return rc;
  if (L.P.n_accs > 4 || L.outs.size() != 1) return fail(PLHIP_ERR_INVALID, "unexpected accumulator layout");
  const uint32_t *offsets = groups->offsets, *indices = groups->indices;
  if (!groups->on_device) { // host-side handle: the lists travel to HBM for the call
    void *d_off = nullptr, *d_idx = nullptr;
    HIP_TRY(hipMalloc(&d_off, (G + 1) * 4));
    st.owned.push_back(d_off);
    HIP_TRY(hipMalloc(&d_idx, (size_t)std::max<int64_t>(n, 1) * 4));
    st.owned.push_back(d_idx);
    HIP_TRY(hipMemcpyAsync(d_off, groups->offsets, (G + 1) * 4, hipMemcpyHostToDevice, c.stream));
    if (n) HIP_TRY(hipMemcpyAsync(d_idx, groups->indices, (size_t)n * 4, hipMemcpyHostToDevice, c.stream));
    offsets = (const uint32_t *)d_off; indices = (const uint32_t *)d_idx;
  }
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->n_rows = (int64_t)G; R->on_device = true; R->device = c.device;
  OutPlan OP{};
  OP.n_outs = 1;
  OutSpec o = L.outs[0];
  ResultCol col;
  col.dtype = (int)o.dtype;
  hipError_t e = hipMalloc(&col.values, (size_t)std::max<uint64_t>(G, 1) * dtype_width((int)o.dtype));
  if (e == hipSuccess && o.validity) {
    e = hipMalloc((void **)&col.validity, (G / 64 + 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (G / 64 + 1) * 8, c.stream);
  }
  R->cols.push_back(col);
  if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
  o.values = (uint8_t *)col.values; o.validity = col.validity;
  OP.outs[0] = o;
  e = c.misc.reserve(agg_long_scratch_bytes((uint64_t)n));
  if (e != hipSuccess) return fail_hip(e, "agg_groups scratch");
  launch_agg_groups(L.P, OP, offsets, indices, (uint64_t)n, G, G > 0 && (uint64_t)n / G >= 64, c.misc.p, c.stream);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) { return fail_hip(e, "agg_groups"); }
  return finish_result(R_own.release(), G, flags & PLHIP_FLAG_OUTPUT_DEVICE, out);
}

// ---- exchange step of the row-sharded group-by (exchange.hip) ----------------------------------------------
namespace {
// record layout of a frame: per column the value bytes, then one validity byte if the column has a bitmap
int rec_layout(RecPlan &R, const int32_t *dtypes, const uint8_t *nullable, int n_cols) {
  if (n_cols < 1 || n_cols > PLHIP_MAX_OUTS) return fail(PLHIP_ERR_INVALID, "bad column count");
  uint32_t off = 0;
  for (int i = 0; i < n_cols; i++) {
    const int w = dtype_width(dtypes[i]);
    if (!w) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "unsupported dtype code " + std::to_string(dtypes[i]));
    R.dtype[i] = (uint32_t)dtypes[i]; R.width[i] = (uint32_t)w; R.off[i] = off; R.has_valid[i] = nullable[i] ? 1 : 0;
    off += (uint32_t)w + (nullable[i] ? 1u : 0u);
  }
  R.n_cols = n_cols; R.rec_bytes = off;
  return PLHIP_OK;
}
int frame_layout(RecPlan &R, const plhip_result_t *r) {
  int32_t dt[PLHIP_MAX_OUTS];
  uint8_t nu[PLHIP_MAX_OUTS];
  if (!r || r->cols.size() > PLHIP_MAX_OUTS) return fail(PLHIP_ERR_INVALID, "bad frame");
  if (r->row_offset) return fail(PLHIP_ERR_INVALID, "a sliced frame cannot enter the exchange");
  for (size_t i = 0; i < r->cols.size(); i++) { dt[i] = r->cols[i].dtype; nu[i] = r->cols[i].validity ? 1 : 0; }
  int rc = rec_layout(R, dt, nu, (int)r->cols.size());
  if (rc) return rc;
  for (size_t i = 0; i < r->cols.size(); i++) {
    R.values[i] = (const uint8_t *)r->cols[i].values; R.validity[i] = r->cols[i].validity; R.boff[i] = 0;
  }
  R.n_rows = r->n_rows;
  return PLHIP_OK;
}
} // namespace

int64_t plhip_result_record_bytes(const plhip_result_t *r) {
  RecPlan R{};
  return frame_layout(R, r) ? -1 : (int64_t)R.rec_bytes;
}

namespace {
// counts == nullptr (slotted layout only): nobody waits for the device here, the receivers read the counts in the slot headers
int pack_impl(const plhip_result_t *r, int32_t n_keys, uint64_t null_hash, int32_t n_parts, int64_t slot_rows, void *records, int64_t *counts) {
  if (!r || (!counts && !slot_rows)) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  if (!r->on_device) return fail(PLHIP_ERR_INVALID, "the frame must be device-resident (PLHIP_FLAG_OUTPUT_DEVICE)");
  if (n_parts < 1 || n_parts > (slot_rows ? EX_MAX_SLOTS : EX_MAX_PARTS)) return fail(PLHIP_ERR_INVALID, "owner count out of range");
  if (n_keys < 1 || n_keys > (int)r->cols.size() || n_keys > PLHIP_MAX_KEYS) return fail(PLHIP_ERR_INVALID, "bad key count");
  if (slot_rows < 0 || slot_rows > 0x7FFFFFFFll) return fail(PLHIP_ERR_INVALID, "bad slot capacity");
  int rc = ensure_ctx();
  if (rc) return rc;
  if ((rc = settle_result(r))) return rc;
  Ctx &c = g_ctx;
  RecPlan R{};
  if ((rc = frame_layout(R, r))) return rc;
  R.n_keys = n_keys; R.n_parts = (uint32_t)n_parts;
  R.slot_rows = (uint32_t)slot_rows;
  R.slot_stride = slot_rows ? (uint64_t)EX_SLOT_HDR + (uint64_t)slot_rows * R.rec_bytes : 0;
  if (counts) for (int p = 0; p < n_parts; p++) counts[p] = 0;
  const int64_t n = r->n_rows;
  if (n == 0 && !slot_rows) return PLHIP_OK;
  if (!records) return fail(PLHIP_ERR_INVALID, "records buffer is NULL");
  if (n > 0xFFFFFFFFll) return fail(PLHIP_ERR_INVALID, "too many rows for one exchange");
  // scratch: counts[n_parts] | cursor[n_parts] | owner[n] (u16)
  const size_t head = (size_t)2 * n_parts * 4;
  HIP_TRY(c.misc.reserve(head + (size_t)n * 2 + 64));
  uint32_t *d_counts = (uint32_t *)c.misc.p, *d_cursor = d_counts + n_parts;
  uint16_t *d_owner = (uint16_t *)((uint8_t *)c.misc.p + head);
  HIP_TRY(hipMemsetAsync(d_counts, 0, head, c.stream));
  if (n) {
    launch_owner_count(R, null_hash, d_owner, d_counts, c.stream);
    HIP_TRY(hipGetLastError());
  }
  launch_pack_records(R, d_owner, d_counts, d_cursor, (uint8_t *)records, c.stream);
  HIP_TRY(hipGetLastError());
  if (!counts) return PLHIP_OK;
  std::vector<uint32_t> h(n_parts);
  HIP_TRY(hipMemcpyAsync(h.data(), d_counts, (size_t)n_parts * 4, hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  for (int p = 0; p < n_parts; p++) counts[p] = h[p];
  return PLHIP_OK;
}
} // namespace

// Series::agg_sum / agg_mean / agg_min / agg_max and GroupsProxy::group_count over the CSR groups; first / last / var / std /
// median / n_unique through their own kernels.
namespace {
// A Boolean column over CSR groups: the reference casts -- sum over the IdxSize (u32) cast (aggregations/boolean.rs:107-109), mean over
// the Float64 cast (dispatch.rs:164: the same value as the u32 cast's f64 mean, the terms being 0 and 1) -- and keeps min / max
// (boolean.rs:20-106; an all-null group is null) and first / last Boolean.  var / std / median / quantile / n_unique of a Boolean
// column are not on this path.
int32_t bool_agg_groups(const plhip_column_t *values, int32_t kind, const plhip_groups_t *groups, uint32_t flags, plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  const bool as_u32 = kind == PLHIP_AGG_SUM || kind == PLHIP_AGG_MEAN;
  const bool as_bool = kind == PLHIP_AGG_MIN || kind == PLHIP_AGG_MAX || kind == PLHIP_AGG_FIRST || kind == PLHIP_AGG_LAST;
  if (!as_u32 && !as_bool) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "Boolean columns aggregate by sum, mean, min, max, first, last or count over groups");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  Staging st;
  StreamDrain drain(c.stream); // (declared after st: the stream retires the kernels that read the unpacked copy before it is freed)
  plhip_column_t lv = *values;
  if ((rc = lower_bool_column(lv, as_u32 ? 4u : 1u, as_u32 ? PLHIP_U32 : PLHIP_U8, st))) return rc;
  plhip_result_t *R = nullptr;
  if ((rc = agg_groups_impl(&lv, kind, groups, (flags & 0xFF00u) | PLHIP_FLAG_OUTPUT_DEVICE, 0.5, 4, &R))) return rc;
  std::unique_ptr<plhip_result> Rg(R);
  const int64_t G = R->n_rows;
  if (as_bool) {
    if (R->cols.size() != 1 || R->cols[0].dtype != PLHIP_U8) return fail(PLHIP_ERR_COMPUTE, "a Boolean output came back with an unexpected type");
    ResultCol &col = R->cols[0];
    void *bits = nullptr;
    size_t cap = 0;
    HIP_TRY(g_pool.get(c.device, (size_t)((G + 63) / 64) * 8 + 8, &bits, &cap));
    launch_pack_bool(col.values, 1, G, (uint64_t *)bits, c.stream);
    const hipError_t e = hipGetLastError() == hipSuccess ? hipStreamSynchronize(c.stream) : hipErrorLaunchFailure;
    if (e != hipSuccess) { g_pool.put(c.device, bits, cap); return fail_hip(e, "k_pack_bool"); }
    if (col.vcap) g_pool.put(c.device, col.values, col.vcap); else hipFree(col.values);
    col.values = bits; col.vcap = cap; col.dtype = PLHIP_BOOL;
  }
  return finish_result(Rg.release(), (uint64_t)G, flags & PLHIP_FLAG_OUTPUT_DEVICE, out);
}
} // namespace

int32_t plhip_agg_groups(const plhip_column_t *values, int32_t kind, const plhip_groups_t *groups, uint32_t flags,
                         plhip_result_t **out) {
  if (kind < 0 || kind > PLHIP_AGG_N_UNIQUE) { if (out) *out = nullptr; return fail(PLHIP_ERR_INVALID, "bad aggregation kind"); }
  if (values && values->dtype == PLHIP_BOOL && kind != PLHIP_AGG_COUNT) {
    if (groups && values->length != groups->n_rows) { if (out) *out = nullptr; return fail(PLHIP_ERR_SHAPE_MISMATCH, "the column and the groups differ in length"); }
    return bool_agg_groups(values, kind, groups, flags, out);
  }
  return agg_groups_impl(values, kind, groups, flags, 0.5, 4 /* Linear: agg_median */, out);
}

// Series::agg_quantile (dispatch.rs:115-140 -> agg_quantile_generic, aggregations/mod.rs:333-414): a quantile outside [0, 1]
// answers an all-null column of the INPUT dtype (mod.rs:346-349).
int32_t plhip_agg_groups_quantile(const plhip_column_t *values, const plhip_groups_t *groups, double quantile, int32_t interpolation,
                                  uint32_t flags, plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!values || !groups) return fail(PLHIP_ERR_INVALID, "values / groups is NULL");
  if (interpolation < 0 || interpolation > PLHIP_QUANTILE_LINEAR) return fail(PLHIP_ERR_INVALID, "bad interpolation");
  if (!(quantile >= 0.0 && quantile <= 1.0)) {
    if (values->dtype < 0 || values->dtype > PLHIP_F64) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "unsupported dtype");
    int rc = ensure_ctx();
    if (rc) return rc;
    const uint64_t G = (uint64_t)groups->n_groups;
    std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result()); // freed on every early return
    plhip_result *R = R_own.get();
    if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
    R->n_rows = (int64_t)G; R->on_device = true; R->device = g_ctx.device;
    ResultCol col;
    col.dtype = values->dtype;
    hipError_t e = hipMalloc(&col.values, (size_t)std::max<uint64_t>(G, 1) * dtype_width(col.dtype));
    if (e == hipSuccess) e = hipMemsetAsync(col.values, 0, (size_t)std::max<uint64_t>(G, 1) * dtype_width(col.dtype), g_ctx.stream);
    if (e == hipSuccess) e = hipMalloc((void **)&col.validity, (G / 64 + 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(col.validity, 0, (G / 64 + 1) * 8, g_ctx.stream);
    R->cols.push_back(col);
    if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
    if (e != hipSuccess) { return fail_hip(e, "hipMalloc(output)"); }
    return finish_result(R_own.release(), G, flags & PLHIP_FLAG_OUTPUT_DEVICE, out);
  }
  return agg_groups_impl(values, AGG_EXT_QUANTILE, groups, flags, quantile, interpolation, out);
}

int32_t plhip_result_pack_by_owner(const plhip_result_t *r, int32_t n_keys, uint64_t null_hash, int32_t n_parts, void *records,
                                   int64_t *counts) {
  if (!counts) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  return pack_impl(r, n_keys, null_hash, n_parts, 0, records, counts);
}

int64_t plhip_slot_bytes(int64_t record_bytes, int64_t slot_rows) { return EX_SLOT_HDR + record_bytes * slot_rows; }

int32_t plhip_result_pack_slots(const plhip_result_t *r, int32_t n_keys, uint64_t null_hash, int32_t n_parts, int64_t slot_rows,
                                void *records, int64_t *counts) {
  if (slot_rows < 1) return fail(PLHIP_ERR_INVALID, "bad slot capacity");
  if (!counts) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  return pack_impl(r, n_keys, null_hash, n_parts, slot_rows, records, counts);
}

namespace {
int records_to_frame_impl(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable, int32_t n_cols, bool sync,
                          plhip_result **out, int32_t n_slots = 0, int64_t slot_rows = 0, const int64_t *slot_counts = nullptr) {
  *out = nullptr;
  if (!dtypes || !nullable || n_rows < 0 || (n_rows > 0 && !records)) return fail(PLHIP_ERR_INVALID, "bad arguments");
  int rc = ensure_ctx();
  if (rc) return rc;
  Ctx &c = g_ctx;
  RecPlan R{};
  if ((rc = rec_layout(R, dtypes, nullable, n_cols))) return rc;
  R.n_rows = n_rows;
  if (n_slots) { // slotted layout: the rows of slot s come out at [slot_first[s], slot_first[s + 1])
    if (n_slots > EX_MAX_SLOTS || slot_rows < 1 || !slot_counts) return fail(PLHIP_ERR_INVALID, "bad slot description");
    uint64_t run = 0;
    for (int sidx = 0; sidx < n_slots; sidx++) {
      if (slot_counts[sidx] < 0 || slot_counts[sidx] > slot_rows) return fail(PLHIP_ERR_INVALID, "a slot holds more rows than its capacity");
      R.slot_first[sidx] = (uint32_t)run; run += (uint64_t)slot_counts[sidx];
    }
    R.slot_first[n_slots] = (uint32_t)run;
    if ((int64_t)run != n_rows || run > 0xFFFFFFFFull) return fail(PLHIP_ERR_INVALID, "slot counts do not add up to n_rows");
    R.n_parts = (uint32_t)n_slots; R.slot_rows = (uint32_t)slot_rows;
    R.slot_stride = (uint64_t)EX_SLOT_HDR + (uint64_t)slot_rows * R.rec_bytes;
  }
  plhip_result *F = new (std::nothrow) plhip_result();
  if (!F) return fail(PLHIP_ERR_OOM, "host allocation failed");
  F->on_device = true; F->device = c.device; F->n_rows = n_rows;
  for (int i = 0; i < n_cols; i++) {
    ResultCol col;
    col.dtype = dtypes[i];
    hipError_t e = g_pool.get(g_ctx.device, (size_t)std::max<int64_t>(n_rows, 1) * R.width[i], &col.values, &col.vcap);
    if (e == hipSuccess && nullable[i]) e = g_pool.get(g_ctx.device, (size_t)(n_rows / 64 + 2) * 8, (void **)&col.validity, &col.bcap);
    F->cols.push_back(col);
    if (e != hipSuccess) { delete F; return fail_hip(e, "hipMalloc(frame)"); }
    R.values[i] = (const uint8_t *)col.values; R.validity[i] = col.validity; R.boff[i] = 0;
  }
  if (n_rows > 0) {
    launch_unpack_records(R, (const uint8_t *)records, c.stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && sync) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) { delete F; return fail_hip(e, "k_unpack"); }
  }
  *out = F;
  return PLHIP_OK;
}
} // namespace

int32_t plhip_records_to_frame(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable, int32_t n_cols,
                               plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  return records_to_frame_impl(records, n_rows, dtypes, nullable, n_cols, true, out);
}

namespace {
int merge_impl(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable, int32_t n_cols, int32_t n_keys,
               const int32_t *merge_kinds, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out, int32_t n_slots,
               int64_t slot_rows, const int64_t *slot_counts);
}

int32_t plhip_merge_records(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable, int32_t n_cols,
                            int32_t n_keys, const int32_t *merge_kinds, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out) {
  return merge_impl(records, n_rows, dtypes, nullable, n_cols, n_keys, merge_kinds, flags, opts, out, 0, 0, nullptr);
}

int32_t plhip_merge_slots(const void *slots, int32_t n_slots, int64_t slot_rows, const int64_t *slot_counts, const int32_t *dtypes,
                          const uint8_t *nullable, int32_t n_cols, int32_t n_keys, const int32_t *merge_kinds, uint32_t flags,
                          const plhip_options_t *opts, plhip_result_t **out) {
  if (n_slots < 1 || !slot_counts) return fail(PLHIP_ERR_INVALID, "bad slot description");
  int64_t n_rows = 0;
  for (int i = 0; i < n_slots; i++) n_rows += slot_counts[i];
  return merge_impl(slots, n_rows, dtypes, nullable, n_cols, n_keys, merge_kinds, flags, opts, out, n_slots, slot_rows, slot_counts);
}

namespace {
int merge_impl(const void *records, int64_t n_rows, const int32_t *dtypes, const uint8_t *nullable, int32_t n_cols, int32_t n_keys,
               const int32_t *merge_kinds, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out, int32_t n_slots,
               int64_t slot_rows, const int64_t *slot_counts) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_keys < 1 || n_keys > n_cols || !merge_kinds) return fail(PLHIP_ERR_INVALID, "bad key count / merge kinds");
  plhip_result *F = nullptr;
  int rc = records_to_frame_impl(records, n_rows, dtypes, nullable, n_cols, false, &F, n_slots, slot_rows, slot_counts); // same stream: no sync in between
  if (rc) return rc;
  std::vector<plhip_column_t> cols(n_cols);
  for (int i = 0; i < n_cols; i++) {
    cols[i].values = F->cols[i].values; cols[i].validity = F->cols[i].validity; cols[i].offset = 0; cols[i].length = n_rows;
    cols[i].dtype = F->cols[i].dtype; cols[i].location = PLHIP_DEVICE;
  }
  const int nv = n_cols - n_keys;
  std::vector<plhip_agg_t> aggs(nv);
  for (int i = 0; i < nv; i++) {
    if (merge_kinds[i] != PLHIP_AGG_SUM && merge_kinds[i] != PLHIP_AGG_MIN && merge_kinds[i] != PLHIP_AGG_MAX) {
      hipStreamSynchronize(g_ctx.stream); // k_unpack is still writing F's columns
      delete F;
      return fail(PLHIP_ERR_INVALID, "partials merge by sum, min or max");
    }
    aggs[i].column = i; aggs[i].kind = merge_kinds[i];
  }
  rc = plhip_group_by_agg_opts(cols.data(), n_keys, cols.data() + n_keys, nv, aggs.data(), nv, flags, opts, out);
  if (rc) hipStreamSynchronize(g_ctx.stream); // a failed group-by may have left k_unpack / its own kernels queued: F's buffers go back to the pool
  delete F; // (on success the count has come from the query's LAST kernel: every kernel that read F's columns retired before it started)
  return rc;
}
} // namespace

int32_t plhip_finalize_partial(const plhip_result_t *merged, int32_t n_keys, const plhip_final_t *finals, int32_t n_finals,
                               const plhip_column_t *out_cols) {
  if (!merged || !finals || !out_cols) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  if (!merged->on_device) return fail(PLHIP_ERR_INVALID, "the frame must be device-resident (PLHIP_FLAG_OUTPUT_DEVICE)");
  if (merged->row_offset) return fail(PLHIP_ERR_INVALID, "a sliced frame cannot be finalized");
  if (n_keys < 1 || n_finals < 0 || n_keys + n_finals > PLHIP_MAX_OUTS) return fail(PLHIP_ERR_INVALID, "bad column counts");
  int rc = ensure_ctx();
  if (rc) return rc;
  if ((rc = settle_result(merged))) return rc;
  Ctx &c = g_ctx;
  const int64_t G = merged->n_rows;
  FinPlan F{};
  F.n_rows = G; F.n_outs = n_keys + n_finals;
  size_t src = 0;
  auto next_col = [&](const ResultCol *&col) -> bool { if (src >= merged->cols.size()) return false; col = &merged->cols[src++]; return true; };
  for (int o = 0; o < F.n_outs; o++) {
    FinOut &X = F.outs[o];
    const plhip_column_t &out = out_cols[o];
    if (out.location != PLHIP_DEVICE || out.length != G || out.offset != 0 || (G > 0 && !out.values))
      return fail(PLHIP_ERR_INVALID, "output columns must be device buffers of the frame's row count");
    const ResultCol *a = nullptr;
    if (!next_col(a)) return fail(PLHIP_ERR_SHAPE_MISMATCH, "the frame has fewer columns than the query describes");
    X.a = (const uint8_t *)a->values; X.a_valid = nullptr; X.cnt = nullptr;
    X.spec = OutSpec{};
    X.spec.values = (uint8_t *)out.values; X.spec.validity = (uint8_t *)out.validity; X.spec.acc = -1; X.spec.cnt_valid = 0; X.spec.cnt_all = 0;
    const int kind = o < n_keys ? -1 : finals[o - n_keys].kind;
    const int sdt = o < n_keys ? a->dtype : finals[o - n_keys].src_dtype;
    int want = a->dtype;
    if (kind == -1 || kind == PLHIP_AGG_MIN || kind == PLHIP_AGG_MAX) {
      X.mode = FIN_COPY; X.width = (uint32_t)dtype_width(a->dtype); X.a_valid = a->validity;
      if (a->validity && !out.validity) return fail(PLHIP_ERR_INVALID, "a nullable column needs an output bitmap");
    } else {
      X.mode = FIN_VALUE;
      if (dtype_width(a->dtype) != 8) return fail(PLHIP_ERR_SHAPE_MISMATCH, "partial sums and counts are 64-bit columns");
      if (kind == PLHIP_AGG_COUNT) { X.spec.kind = OUT_COUNT; want = PLHIP_U32; }
      else {
        if (!dtype_width(sdt)) return fail(PLHIP_ERR_UNSUPPORTED_DTYPE, "unsupported source dtype");
        const bool is_f = dt_is_float(sdt);
        if (kind == PLHIP_AGG_SUM) {
          X.spec.kind = is_f ? OUT_SUM_F64 : OUT_SUM_INT;
          want = is_f ? sdt : (dtype_width(sdt) < 4 ? PLHIP_I64 : sdt);
        } else if (kind == PLHIP_AGG_MEAN) {
          X.spec.kind = a->dtype == PLHIP_F64 ? OUT_MEAN_F64 : (a->dtype == PLHIP_I64 ? OUT_MEAN_INT : OUT_MEAN_UINT);
          want = sdt == PLHIP_F32 ? PLHIP_F32 : PLHIP_F64;
          const ResultCol *cnt = nullptr;
          if (!next_col(cnt) || dtype_width(cnt->dtype) != 8) return fail(PLHIP_ERR_SHAPE_MISMATCH, "mean needs a (sum, count) pair of columns");
          X.cnt = (const uint8_t *)cnt->values;
        } else return fail(PLHIP_ERR_INVALID, "bad aggregation kind");
      }
    }
    if (out.dtype != want) return fail(PLHIP_ERR_SHAPE_MISMATCH, "output column " + std::to_string(o) + " has the wrong dtype");
    X.spec.dtype = (uint32_t)want;
  }
  if (src != merged->cols.size()) return fail(PLHIP_ERR_SHAPE_MISMATCH, "the frame has more columns than the query describes");
  if (G == 0) return PLHIP_OK;
  launch_finalize_partial(F, c.stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c.stream));
  return PLHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// the sharded path (comm.hip carries the transport)
// ---------------------------------------------------------------------------------------------------------
int32_t plhip_comm_unique_id(uint8_t id[PLHIP_COMM_ID_BYTES]) {
  if (!id) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  std::string err;
  const int rc = comm_unique_id(id, err);
  return rc ? fail(rc, err) : PLHIP_OK;
}

int32_t plhip_comm_init_rccl(const uint8_t id[PLHIP_COMM_ID_BYTES], int32_t world, int32_t rank, plhip_comm_t **comm) {
  if (!id || !comm) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *comm = nullptr;
  if (world < 1 || world > EX_MAX_PARTS || rank < 0 || rank >= world) return fail(PLHIP_ERR_INVALID, "bad world size / rank");
  int rc = ensure_ctx(); // the calling thread's device is current: RCCL binds the communicator to it
  if (rc) return rc;
  std::string err;
  rc = comm_init_rccl(id, world, rank, g_ctx.device, comm, err);
  return rc ? fail(rc, err) : PLHIP_OK;
}

int32_t plhip_comm_init_host(int32_t world, int32_t rank, plhip_alltoall_fn all_to_all, plhip_alltoallv_fn all_to_all_v, void *user,
                             plhip_comm_t **comm) {
  if (!comm || !all_to_all || !all_to_all_v) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *comm = nullptr;
  if (world < 1 || world > EX_MAX_PARTS || rank < 0 || rank >= world) return fail(PLHIP_ERR_INVALID, "bad world size / rank");
  int rc = ensure_ctx();
  if (rc) return rc;
  plhip_comm *c = new (std::nothrow) plhip_comm();
  if (!c) return fail(PLHIP_ERR_OOM, "host allocation failed");
  c->world = world; c->rank = rank; c->device = g_ctx.device; c->a2a = all_to_all; c->a2av = all_to_all_v; c->user = user;
  *comm = c;
  return PLHIP_OK;
}

void plhip_comm_free(plhip_comm_t *comm) { comm_free(comm); }

int32_t plhip_comm_info(plhip_comm_t *comm, int32_t *world, int32_t *rank, int32_t *transport) {
  if (!comm || !world || !rank || !transport) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  std::string err;
  int w = 0, r = 0, t = 0;
  const int rc = comm_info(comm, &w, &r, &t, err);
  if (rc) return fail(rc, err);
  *world = w; *rank = r; *transport = t;
  return PLHIP_OK;
}

int32_t plhip_sharded_last_stats(plhip_sharded_stats_t *stats) {
  if (!stats) return fail(PLHIP_ERR_INVALID, "NULL pointer");
  *stats = g_ctx.shard_stats;
  return PLHIP_OK;
}

namespace {
constexpr uint64_t EXCHANGE_NULL_HASH = 0x9E3779B97F4A7C15ull; // any constant: null keys only need to agree on an owner

struct FrameGuard { // a device-resident intermediate frame: back to the pool on every path out
  plhip_result_t *p = nullptr;
  ~FrameGuard() { if (p) delete p; }
};

int sharded_impl(plhip_comm *comm, const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values, int32_t n_values,
                 const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags, const plhip_options_t *opts, plhip_result_t **out) {
  using clk = std::chrono::steady_clock;
  Ctx &c = g_ctx;
  plhip_sharded_stats_t &S = c.shard_stats;
  S = plhip_sharded_stats_t{};
  auto t0 = clk::now();
  auto lap = [&](float &slot) { const auto t1 = clk::now(); slot += std::chrono::duration<float, std::milli>(t1 - t0).count(); t0 = t1; };
  const int world = comm->world;
  const int64_t hint = (opts && opts->struct_size >= 16 && opts->n_groups_hint > 0) ? opts->n_groups_hint : 0;
  if (opts && opts->struct_size >= sizeof(plhip_options_t) && opts->has_slice) return fail(PLHIP_ERR_INVALID, "a slice cannot be pushed into the sharded path");
  // ---- dense codes: owners by code range, one exchange of raw dense arrays, no regroup (dense.hpp) -----------------------
  if ((flags & PLHIP_FLAG_KEYS_DENSE) && hint > 0 && !(flags & PLHIP_FLAG_EXACT_EXCHANGE) && n_keys == 1 && !getenv("PLHIP_NO_DENSE_EXCHANGE")) {
    c.dense_comm = comm;
    const int drc = group_by_agg_impl(keys, n_keys, values, n_values, aggs, n_aggs, flags & (PLHIP_FLAG_KEYS_DENSE | PLHIP_FLAG_OUTPUT_DEVICE), opts, out);
    c.dense_comm = nullptr;
    if (drc) return drc;
    if (*out) {
      resolve_timing();
      S.partial_kernel_ms = c.total_ms; S.partial_dominant_ms = c.dom_ms;
      S.partial_breakdown_ms[0] = c.k_ms[0]; S.partial_breakdown_ms[1] = c.k_ms[1]; S.partial_breakdown_ms[2] = c.k_ms[2];
      snprintf(S.partial_strategy, sizeof(S.partial_strategy), "%s", c.strategy.c_str());
      snprintf(S.partial_dominant_kernel, sizeof(S.partial_dominant_kernel), "%s", c.dom_kernel.c_str());
      S.partial_groups = -1; // (the local pass never counts its groups: its arrays stay dense until the owners have merged them)
      S.protocol = 3;
      return PLHIP_OK;
    }
    S = plhip_sharded_stats_t{}; // every rank declined together (schema, or a flagged header): the hashed protocols below
    t0 = clk::now();
  }
  // ---- local pass: mergeable partials, device-resident -------------------------------------------------------------
  FrameGuard part;
  const uint32_t pflags = PLHIP_FLAG_PARTIAL | PLHIP_FLAG_OUTPUT_DEVICE | (flags & PLHIP_FLAG_KEYS_DENSE) | FLAG_INTERNAL_WRAP32;
  int rc = group_by_agg_impl(keys, n_keys, values, n_values, aggs, n_aggs, pflags, opts, &part.p);
  if (rc) return rc;
  resolve_timing();
  S.partial_kernel_ms = c.total_ms; S.partial_dominant_ms = c.dom_ms;
  S.partial_breakdown_ms[0] = c.k_ms[0]; S.partial_breakdown_ms[1] = c.k_ms[1]; S.partial_breakdown_ms[2] = c.k_ms[2];
  snprintf(S.partial_strategy, sizeof(S.partial_strategy), "%s", c.strategy.c_str());
  snprintf(S.partial_dominant_kernel, sizeof(S.partial_dominant_kernel), "%s", c.dom_kernel.c_str());
  S.partial_groups = part.p->n_rows;
  S.host_syncs = 1; // the local pass ends with the host waiting for its group count
  lap(S.partial_ms);
  // ---- the record layout: a property of the query's schema (build_aggs under PLHIP_FLAG_PARTIAL), the same on every rank
  const int n_cols = (int)part.p->cols.size();
  if (n_cols > PLHIP_MAX_OUTS) return fail(PLHIP_ERR_INVALID, "too many partial columns");
  int32_t dts[PLHIP_MAX_OUTS];
  uint8_t nul[PLHIP_MAX_OUTS];
  int32_t kinds[PLHIP_MAX_OUTS];
  {
    int o = 0, nm = 0;
    for (int k = 0; k < n_keys; k++, o++) { dts[o] = keys[k].dtype; nul[o] = keys[k].validity ? 1 : 0; }
    for (int i = 0; i < n_aggs; i++) {
      const int kind = aggs[i].kind;
      if (kind == PLHIP_AGG_COUNT) { dts[o] = PLHIP_U64; nul[o++] = 0; kinds[nm++] = PLHIP_AGG_SUM; continue; }
      const int dt = values[aggs[i].column].dtype;
      const bool is_f = dt_is_float(dt), is_s = dt_is_signed(dt), nullable = values[aggs[i].column].validity != nullptr;
      const int w = dtype_width(dt);
      const int sum_dt = is_f ? PLHIP_F64 : ((is_s || w < 4) ? PLHIP_I64 : PLHIP_U64);
      if (kind == PLHIP_AGG_SUM) { dts[o] = sum_dt; nul[o++] = 0; kinds[nm++] = PLHIP_AGG_SUM; }
      else if (kind == PLHIP_AGG_MEAN) {
        dts[o] = (!is_f && w <= 4) ? (is_s ? PLHIP_I64 : PLHIP_U64) : PLHIP_F64; nul[o++] = 0; kinds[nm++] = PLHIP_AGG_SUM;
        dts[o] = PLHIP_U64; nul[o++] = 0; kinds[nm++] = PLHIP_AGG_SUM;
      } else { dts[o] = dt; nul[o++] = nullable ? 1 : 0; kinds[nm++] = kind == PLHIP_AGG_MIN ? PLHIP_AGG_MIN : PLHIP_AGG_MAX; }
      if (o > PLHIP_MAX_OUTS - 2) return fail(PLHIP_ERR_INVALID, "too many partial columns");
    }
    if (o != n_cols) return fail(PLHIP_ERR_COMPUTE, "the partial frame does not match the query's schema");
    for (int i = 0; i < n_cols; i++)
      if (part.p->cols[i].dtype != dts[i] || (part.p->n_rows > 0 && (part.p->cols[i].validity != nullptr) != (nul[i] != 0)))
        return fail(PLHIP_ERR_COMPUTE, "the partial frame does not match the query's schema (column " + std::to_string(i) + ")");
  }
  RecPlan RL{};
  if ((rc = rec_layout(RL, dts, nul, n_cols))) return rc;
  const uint64_t rec = RL.rec_bytes;
  std::string err;
  FrameGuard merged;
  plhip_options_t mo{};
  mo.struct_size = 16;
  bool done = false;
  if (hint > 0 && world <= EX_MAX_SLOTS && !(flags & PLHIP_FLAG_EXACT_EXCHANGE)) {
    // ---- slots: 1.25x an even share per owner; one equal-split exchange ---------------------------------------------
    const int64_t slot_rows = (5 * hint + 4 * world - 1) / (4 * world) + 512;
    const size_t stride = (size_t)plhip_slot_bytes((int64_t)rec, slot_rows);
    HIP_TRY(c.xsend.reserve(stride * world + 256));
    HIP_TRY(c.xrecv.reserve(stride * world + 256));
    if (part.p->n_rows == 0) { // the packer would find no bitmaps to read a layout from: write empty slot headers directly
      HIP_TRY(hipMemsetAsync(c.xsend.p, 0, stride * world, c.stream));
    } else if ((rc = pack_impl(part.p, n_keys, EXCHANGE_NULL_HASH, world, slot_rows, c.xsend.p, nullptr))) return rc;
    lap(S.pack_ms);
    if ((rc = comm_all_to_all(comm, c.xsend.p, c.xrecv.p, stride, c.stream, err))) return fail(rc, err);
    // the headers of the received slots: a one-workgroup kernel writes them into pinned host memory (a strided 2-D copy into
    // pageable memory came back with stale bytes under rocprofv3's copy interception, and costs more than a launch anyway)
    launch_slot_headers((const uint8_t *)c.xrecv.p, (uint64_t)stride, (uint32_t)world, c.host_hdr_dev, c.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c.stream)); // the exchange has landed; the headers say how many records every peer sent
    uint64_t hdr[EX_MAX_SLOTS][2];
    for (int p = 0; p < world; p++) { hdr[p][0] = __atomic_load_n(&c.host_hdr[2 * p], __ATOMIC_ACQUIRE); hdr[p][1] = __atomic_load_n(&c.host_hdr[2 * p + 1], __ATOMIC_ACQUIRE); }
    S.host_syncs++;
    S.shuffle_bytes = (int64_t)(stride * world);
    S.protocol = 1;
    lap(S.exchange_ms);
    bool overflow = false;
    int64_t rcnt[EX_MAX_SLOTS], m = 0;
    for (int p = 0; p < world; p++) { overflow |= hdr[p][1] != 0; rcnt[p] = (int64_t)hdr[p][0]; m += rcnt[p]; }
    // an overflowing bucket anywhere is seen by its receiver only: all ranks must agree on the protocol, so the flag is
    // part of what every rank sends to EVERY peer (pack sets it in all of a rank's headers when any of its buckets overflows)
    if (!overflow) {
      mo.n_groups_hint = std::max<int64_t>(std::min<int64_t>(m, hint), 1);
      rc = merge_impl(c.xrecv.p, m, dts, nul, n_cols, n_keys, kinds, PLHIP_FLAG_OUTPUT_DEVICE, &mo, &merged.p, world, slot_rows, rcnt);
      if (rc) return rc;
      S.host_syncs++;
      lap(S.merge_ms);
      done = true;
    }
  }
  if (!done) {
    // ---- exact protocol: counts, then a variable-size exchange -----------------------------------------------------------
    const int64_t g = part.p->n_rows;
    int64_t sc[EX_MAX_PARTS], rcv[EX_MAX_PARTS];
    HIP_TRY(c.xsend.reserve((size_t)std::max<int64_t>(g, 1) * rec + 256));
    if (g > 0) { if ((rc = pack_impl(part.p, n_keys, EXCHANGE_NULL_HASH, world, 0, c.xsend.p, sc))) return rc; S.host_syncs++; }
    else for (int p = 0; p < world; p++) sc[p] = 0;
    lap(S.pack_ms);
    HIP_TRY(c.xcnt.reserve((size_t)world * 16 + 64));
    int64_t *d_sc = (int64_t *)c.xcnt.p, *d_rc = d_sc + world;
    HIP_TRY(hipMemcpyAsync(d_sc, sc, (size_t)world * 8, hipMemcpyHostToDevice, c.stream));
    if ((rc = comm_all_to_all(comm, d_sc, d_rc, 8, c.stream, err))) return fail(rc, err);
    HIP_TRY(hipMemcpyAsync(rcv, d_rc, (size_t)world * 8, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    S.host_syncs++;
    int64_t m = 0, sb[EX_MAX_PARTS], rb[EX_MAX_PARTS];
    for (int p = 0; p < world; p++) {
      if (rcv[p] < 0 || rcv[p] > 0xFFFFFFFFll) return fail(PLHIP_ERR_COMPUTE, "a peer announced an impossible record count");
      m += rcv[p]; sb[p] = sc[p] * (int64_t)rec; rb[p] = rcv[p] * (int64_t)rec;
    }
    HIP_TRY(c.xrecv.reserve((size_t)std::max<int64_t>(m, 1) * rec + 256));
    if ((rc = comm_all_to_all_v(comm, c.xsend.p, sb, c.xrecv.p, rb, c.stream, err))) return fail(rc, err);
    S.shuffle_bytes = g * (int64_t)rec + (int64_t)world * 8;
    S.protocol = 2;
    lap(S.exchange_ms);
    mo.n_groups_hint = std::max<int64_t>(hint > 0 ? std::min<int64_t>(m, hint) : m, 1);
    rc = merge_impl(m ? c.xrecv.p : nullptr, m, dts, nul, n_cols, n_keys, kinds, PLHIP_FLAG_OUTPUT_DEVICE, &mo, &merged.p, 0, 0, nullptr);
    if (rc) return rc;
    S.host_syncs++;
    lap(S.merge_ms);
  }
  // ---- finalize: (sum, count) -> mean, 64-bit partial sums -> the column's type (aggregation.rs:367-483) -------------
  const int64_t G = merged.p->n_rows;
  std::unique_ptr<plhip_result> R_own(new (std::nothrow) plhip_result());
  plhip_result *R = R_own.get();
  if (!R) return fail(PLHIP_ERR_OOM, "host allocation failed");
  R->on_device = true; R->device = c.device; R->n_rows = G;
  plhip_column_t ocols[PLHIP_MAX_OUTS];
  plhip_final_t fin[PLHIP_MAX_OUTS];
  if (n_keys + n_aggs > PLHIP_MAX_OUTS) return fail(PLHIP_ERR_INVALID, "too many output columns");
  for (int o = 0; o < n_keys + n_aggs; o++) {
    int dt; bool nullable;
    if (o < n_keys) { dt = keys[o].dtype; nullable = keys[o].validity != nullptr; }
    else {
      const plhip_agg_t &a = aggs[o - n_keys];
      const int sdt = a.kind == PLHIP_AGG_COUNT ? PLHIP_U32 : values[a.column].dtype;
      const bool src_null = a.kind != PLHIP_AGG_COUNT && values[a.column].validity != nullptr;
      fin[o - n_keys].kind = a.kind; fin[o - n_keys].src_dtype = sdt;
      if (a.kind == PLHIP_AGG_COUNT) { dt = PLHIP_U32; nullable = false; }
      else if (a.kind == PLHIP_AGG_SUM) { dt = dt_is_float(sdt) ? sdt : (dtype_width(sdt) < 4 ? PLHIP_I64 : sdt); nullable = false; }
      else if (a.kind == PLHIP_AGG_MEAN) { dt = sdt == PLHIP_F32 ? PLHIP_F32 : PLHIP_F64; nullable = src_null; }
      else { dt = sdt; nullable = src_null; }
    }
    ResultCol col;
    col.dtype = dt;
    hipError_t e = g_pool.get(c.device, (size_t)std::max<int64_t>(G, 1) * dtype_width(dt), &col.values, &col.vcap);
    if (e == hipSuccess && nullable) e = g_pool.get(c.device, (size_t)(G / 64 + 2) * 8, (void **)&col.validity, &col.bcap);
    R->cols.push_back(col);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(output)");
    ocols[o] = plhip_column_t{};
    ocols[o].values = col.values; ocols[o].validity = col.validity; ocols[o].offset = 0; ocols[o].length = G; ocols[o].dtype = dt;
    ocols[o].location = PLHIP_DEVICE;
  }
  if ((rc = plhip_finalize_partial(merged.p, n_keys, fin, n_aggs, ocols))) return rc;
  if (G > 0) S.host_syncs++;
  lap(S.finalize_ms);
  return finish_result(R_own.release(), (uint64_t)G, flags & PLHIP_FLAG_OUTPUT_DEVICE, out);
}
} // namespace

int32_t plhip_group_by_agg_sharded(plhip_comm_t *comm, const plhip_column_t *keys, int32_t n_keys, const plhip_column_t *values,
                                   int32_t n_values, const plhip_agg_t *aggs, int32_t n_aggs, uint32_t flags,
                                   const plhip_options_t *opts, plhip_result_t **out) {
  if (!out) return fail(PLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!comm) return fail(PLHIP_ERR_INVALID, "comm is NULL");
  if (flags & ~(PLHIP_FLAG_OUTPUT_DEVICE | PLHIP_FLAG_EXACT_EXCHANGE | PLHIP_FLAG_KEYS_DENSE))
    return fail(PLHIP_ERR_INVALID, "the sharded path takes PLHIP_FLAG_OUTPUT_DEVICE, PLHIP_FLAG_EXACT_EXCHANGE and PLHIP_FLAG_KEYS_DENSE only");
  if (n_keys < 1) return fail(PLHIP_ERR_COMPUTE, "at least one key is required in a group_by operation");
  if (!keys || n_keys > PLHIP_MAX_KEYS || n_values < 0 || n_aggs < 0 || (n_aggs > 0 && !aggs) || (n_values > 0 && !values))
    return fail(PLHIP_ERR_INVALID, "bad arguments");
  for (int i = 0; i < n_aggs; i++)
    if (aggs[i].kind != PLHIP_AGG_COUNT && (aggs[i].kind < 0 || aggs[i].kind > PLHIP_AGG_COUNT || aggs[i].column < 0 || aggs[i].column >= n_values))
      return fail(PLHIP_ERR_INVALID, "bad aggregation");
  int rc = ensure_ctx();
  if (rc) return rc;
  if (comm->device != g_ctx.device) return fail(PLHIP_ERR_INVALID, "the communicator belongs to another device than the calling thread's");
  bool any_bool = false; // Boolean columns: the reference's casts at the boundary, as in plhip_group_by_agg (bool_group_by)
  for (int i = 0; i < n_keys; i++) any_bool |= keys[i].dtype == PLHIP_BOOL;
  for (int i = 0; i < n_values && i < PLHIP_MAX_VALS; i++) any_bool |= values[i].dtype == PLHIP_BOOL;
  if (any_bool) return bool_group_by(keys, n_keys, values, n_values, aggs, n_aggs, flags, opts, out, comm);
  return sharded_impl(comm, keys, n_keys, values, n_values, aggs, n_aggs, flags, opts, out);
}

} // extern "C"
