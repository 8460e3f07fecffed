"""bench.py — db-benchmark-shaped hash group-by on MI355X: rows/s + fraction of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload q3|q1|q5|c3|c5] [--rows R] [--total-rows T] [--na]

A "step" is one pass of the hot path (plhip_group_by_agg_opts through the C ABI) over one batch of synthetic
rows that is already resident in HBM, up to and including the device-resident result frame and its row count
on the host.  Default workload = BASELINE.json configs[1]: db-benchmark q3, sum(v1), mean(v3) by id3,
1e7 rows, 1e5 groups, one MI355X.  Other workloads: q1 / q5 (configs[0], [1]), c3 (configs[2]: 1e8 rows, three i64
keys, i64 sum/min/max + f64 sum/mean + count), c5 (configs[4]: 1e9 rows, 1e8 ids, skewed; see skewed_keys()).
For N > 1 every rank holds its own shard (weak scaling: --rows per GPU; strong scaling = configs[3]: --total-rows T
gives every rank T / N rows): local partial aggregation, ONE all-to-all of the partial records (RCCL), regroup on the
owner rank (polars_amd/distributed.py).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (keys, value columns, aggs [(value idx, kind)], wide dtypes)
    "q1": (["id1"], ["v1"], [(0, "sum")], False),
    "q3": (["id3"], ["v1", "v3"], [(0, "sum"), (1, "mean")], False),
    "q5": (["id6"], ["v1", "v2", "v3"], [(0, "sum"), (1, "sum"), (2, "sum")], False),
    # the other fused queries of the reference's harness (run_h2oai_benchmark.py:66-160; q7's subtraction of the two G-row columns is
    # the host layer's, q6 = median / std goes through the CSR seams, q8 / q9 are not group-by aggregations of this path)
    "q2": (["id1", "id2"], ["v1"], [(0, "sum")], False),
    "q4": (["id4"], ["v1", "v2", "v3"], [(0, "mean"), (1, "mean"), (2, "mean")], False),
    "q7": (["id3"], ["v1", "v2"], [(0, "max"), (1, "min")], False),
    "q10": (["id1", "id2", "id3", "id4", "id5", "id6"], ["v3", "v1"], [(0, "sum"), (1, "count")], False),
    "c3": (["id1", "id2", "id3"], ["v1", "v3"], [(0, "sum"), (0, "min"), (0, "max"), (1, "sum"), (1, "mean"), (0, "count")], True),
    # one skewed i64 key, f64 sum/mean + i64 sum/min/max + count
    "c5": (["k"], ["v3", "v1"], [(0, "sum"), (0, "mean"), (1, "sum"), (1, "min"), (1, "max"), (0, "count")], True),
}
DEFAULT_ROWS = {"c3": 100_000_000, "c5": 1_000_000_000}
C5_IDS_PER_ROW = 0.1  # 1e8 ids at 1e9 rows


def skewed_keys(n, ids, seed, device):
    """BASELINE config 5's keys (the exact sampler): with probability 1/2 a Zipf(s = 1.1) draw by inverse CDF,
    X = floor(U^(-1/(s-1))) = floor(U^-10) (P(X >= k) = k^-0.1: the top key holds ~6.7 % of these draws), folded onto the
    ids as (X - 1) mod ids; otherwise a uniform id, so that (nearly) every id occurs.  torch.Generator(seed) on the
    device: u (f64) for the Zipf draw, a f32 coin, u (f64) for the uniform id, then v3 (f64 uniform * 100) and v1
    (int64 in 1..5), in this order."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand(n, device=device, dtype=torch.float64, generator=g).clamp_min(1e-300)
    z = torch.floor(u.pow(-10.0)).clamp_max(float(ids) * 1e6)
    del u
    coin = torch.rand(n, device=device, generator=g) < float(os.environ.get("PLHIP_BENCH_ZIPF_SHARE", "0.5"))  # (experiments: 0 = uniform ids)
    uni = torch.floor(torch.rand(n, device=device, dtype=torch.float64, generator=g) * ids)
    k = torch.where(coin, (z - 1) % ids, uni).to(torch.int64)
    del z, coin, uni
    v3 = torch.rand(n, device=device, dtype=torch.float64, generator=g) * 100
    v1 = torch.randint(1, 6, (n,), device=device, dtype=torch.int64, generator=g)
    return {"k": k, "v3": v3, "v1": v1}

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel reaches


def algorithmic_bytes(n_rows, n_groups, key_cols, val_cols, out_widths):
    """every distinct input column read once, every output written once; hash / partition
    scratch, table traffic and re-reads are overhead, not credit."""
    from polars_amd._ffi import code_np_dtype

    w = lambda c: code_np_dtype(c.dtype).itemsize
    b = n_rows * (sum(w(c) for c in key_cols) + sum(w(c) for c in val_cols))
    b += sum((n_rows + 7) // 8 for c in list(key_cols) + list(val_cols) if c.validity is not None)
    b += n_groups * sum(out_widths)
    return b


def _host_columns(cols):
    """Device columns -> host columns (values + bool mask) for the oracle."""
    import numpy as np

    from polars_amd.column import Column

    out = []
    for c in cols:
        a = c.to_numpy()
        out.append(Column(c.name, np.ascontiguousarray(a), c.valid_mask()))
    return out


def cpu_baseline(tensors, keys, vals, aggs, gpu_frame, budget_s=10.0, path_b=True):
    """The oracle (C restatement of the reference algorithm: P = pow2 <= threads partitions, every thread scans
    all keys, per-group index lists, gather-fold aggregation, crates/polars-core/src/frame/group_by/hashing.rs:
    182-243 + aggregations/mod.rs) timed on this box's host cores: whole passes over the same rows for about
    `budget_s` seconds."""
    from oracle import oracle as orc

    hk, hv = _host_columns(keys), _host_columns(vals)
    threads = min(os.cpu_count() or 1, 16)  # the box's CPU share for one GPU
    n = len(hk[0])
    orc.lib()
    passes, t0 = 0, time.perf_counter()
    while True:
        want, _ = orc.group_by_agg(hk, hv, aggs, n_threads=threads)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 200:
            break
    base = {"value": n * passes / dt, "unit": "rows/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{passes} whole passes over the same {n} rows ({dt:.1f} s)",
            "path": "A: partitioned hash maps of row-index lists, then gather-fold per group (hashing.rs:182-243 + aggregations/mod.rs)"}
    if len(hk) == 1 and path_b:  # path B: the streaming sink's accumulators, timed on the same rows
        p2, t1 = 0, time.perf_counter()
        while True:
            g2 = orc.sink_group_by(hk[0], hv, aggs, n_threads=threads, outputs=False)
            p2 += 1
            d2 = time.perf_counter() - t1
            if d2 >= budget_s / 2 or p2 >= 200:
                break
        base["path_b"] = {"value": n * p2 / d2, "unit": "rows/s", "cores": threads, "groups": int(g2),
                          "groups_match_path_a": int(g2) == len(want[0][1]),
                          "what": "PrimitiveGroupbySink restated (polars-pipe sinks/group_by/primitive/mod.rs:359-454): per-thread "
                                  "partitioned hash maps of accumulators, 50 000-row morsels, sinks combined at the end",
                          "sample": f"{p2} whole passes over the same {n} rows ({d2:.1f} s)"}
    return base, parity_check(hk, hv, aggs, want, gpu_frame)


def cpu_baseline_sample(tensors, keys_n, vals_n, aggs, gpu_frame, sample_rows=50_000_000):
    """Workloads beyond 1e8 rows (config 5): the oracle is timed on the first `sample_rows` rows (a bounded sample of the
    same columns), and the GPU result of the FULL input is checked through size-independent properties:
    counts add up to N, the integer checksum of sums is exact, the f64 checksum within 1e-9, one row per distinct key."""
    import numpy as np
    import torch

    from oracle import oracle as orc
    from polars_amd.column import Column

    n = tensors[keys_n[0]].numel()
    m = min(n, sample_rows)
    hk = [Column(k, tensors[k][:m].cpu().numpy()) for k in keys_n]
    hv = [Column(v, tensors[v][:m].cpu().numpy()) for v in vals_n]
    threads = min(os.cpu_count() or 1, 16)
    orc.lib()
    t0 = time.perf_counter()
    orc.group_by_agg(hk, hv, aggs, n_threads=threads)
    dt = time.perf_counter() - t0
    base = {"value": m / dt, "unit": "rows/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"one pass over the first {m} of {n} rows ({dt:.1f} s)"}
    got = gpu_frame()
    nk = len(keys_n)
    checks = {}
    for j, (ci, kind) in enumerate(aggs):
        col = got[nk + j][0]
        src = tensors[vals_n[ci]]
        if kind == "count":
            checks["counts_add_up_to_n"] = int(col.astype(np.int64).sum()) == n
        elif kind == "sum" and src.dtype in (torch.int64, torch.int32):
            checks["integer_checksum_exact"] = int(col.astype(np.int64).sum()) == int(src.sum().item())
        elif kind == "sum":
            tot = float(src.sum().item())
            checks["f64_checksum_1e-9"] = abs(float(col.sum()) - tot) <= 1e-9 * abs(tot)
    distinct = int(torch.unique(tensors[keys_n[0]]).numel()) if nk == 1 else None
    if distinct is not None:
        checks["one_row_per_distinct_key"] = len(got[0][0]) == distinct and len(np.unique(got[0][0])) == distinct
    return base, {"ok": all(checks.values()), "properties": checks, "groups": int(len(got[0][0])),
                  "against": "size-independent properties of the full input (the oracle would need minutes at this size)"}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parity_check(keys, vals, aggs, want, gpu_frame):
    """SURVEY.md section 8(d) "parity check in the bench": the timed configuration's GPU result (one more, untimed call,
    copied to the host) against the oracle's on the same rows after the canonical sort on the keys: keys, integer
    sums, counts and validity bit-exact, f64 within 1e-12 relative."""
    import numpy as np

    got = gpu_frame()
    nk = len(keys)
    if len(got) != len(want) or len(got[0][0]) != len(want[0][1]):
        return {"ok": False, "why": f"shape: {len(got)} x {len(got[0][0])} vs {len(want)} x {len(want[0][1])}"}
    og = np.lexsort([got[j][0] for j in reversed(range(nk))])
    ow = np.lexsort([want[j][1] for j in reversed(range(nk))])
    worst = 0.0
    for j, ((gv, gm), (_, wv, wm)) in enumerate(zip(got, want)):
        if gv.dtype != wv.dtype:
            return {"ok": False, "why": f"column {j}: dtype {gv.dtype} vs {wv.dtype}"}
        gv, wv = gv[og], wv[ow]
        gm = np.ones(len(gv), bool) if gm is None else gm[og]
        if not np.array_equal(gm, wm[ow]):
            return {"ok": False, "why": f"column {j}: validity differs"}
        a, b = gv[gm], wv[gm]
        if gv.dtype.kind == "f" and j >= nk:
            a, b = a.astype(np.float64), b.astype(np.float64)
            fin = ~(np.isnan(a) & np.isnan(b))
            rel = np.abs(a[fin] - b[fin]) / np.maximum(np.maximum(np.abs(a[fin]), np.abs(b[fin])), 1e-300)
            worst = max(worst, float(rel.max()) if len(rel) else 0.0)
            if worst > 1e-12:
                return {"ok": False, "why": f"column {j}: f64 relative error {worst:.3e} > 1e-12"}
        elif not np.array_equal(a, b):
            return {"ok": False, "why": f"column {j}: integer / key values differ"}
    return {"ok": True, "groups": int(len(og)), "columns": len(got), "f64_max_rel_err": worst, "f64_rtol": 1e-12,
            "against": "oracle (C restatement), same rows, canonical key sort"}


def measured_traffic(key):
    """HBM bytes of the WHOLE query (summed over its kernels, per launch) from the rocprofv3 --pmc passes kept under
    profiles/ (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; scripts/prof_summary.py
    writes the entry together with the per-kernel figures); None if no such measurement is on file for this
    workload and strategy."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None, None
    try:
        t = json.loads(f.read_text()).get(key)
        if not t:
            return None, None
        return t["hbm_bytes"], {k: int(v["fetch_bytes"] + v["write_bytes"]) for k, v in t.get("kernels", {}).items()}
    except (ValueError, OSError, KeyError):
        return None, None


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process starts the N ranks itself, one per
    GPU, and never touches the GPU (no torch import, no HIP call here: a process that has initialised the GPU must not
    start replacements of itself, and this one has not).  Every child is this same file with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, which is what torch.distributed.run would have set; rank 0's one JSON line
    is relayed on stdout, everything else goes to stderr, and the exit status is non-zero if any rank's is.
    PLHIP_BENCH_SHARE_GPU=1 puts all ranks on device 0 with the host (gloo) transport: the launcher and the whole sharded
    step can then be tested on a one-GPU box."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PLHIP_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p_ in procs[1:]:
        try:
            rcs.append(p_.wait(timeout=120 if rcs[0] else None))
        except subprocess.TimeoutExpired:  # rank 0 failed and a peer is stuck in a collective: end exactly that child
            p_.kill()
            rcs.append(p_.wait())
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if any(rcs) or len(lines) != 1:
        sys.stderr.write(f"bench.py launcher: rank exit codes {rcs}, {len(lines)} JSON line(s) from rank 0\n")
        sys.stderr.write(out0.decode(errors="replace"))
        return max([abs(rc) for rc in rcs] + [1])
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


class Ctx:
    """What every workload of one bench.py process shares: the process group, this rank's device, the engine."""
    world = 1
    rank = 0
    dev = None
    dist = None
    share_gpu = False
    rehearse = False
    eng = None

    @property
    def sharded(self):
        return self.world > 1 or self.rehearse

    def reduce(self, values, op):
        """all_reduce of a few f64 numbers over the ranks (on the device under RCCL, on the host under gloo)."""
        import torch

        t = torch.tensor(values, dtype=torch.float64, device="cpu" if self.share_gpu else self.dev)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return t.tolist()

    def sync(self):
        import torch

        torch.cuda.synchronize()
        if self.sharded:
            self.dist.barrier()
            torch.cuda.synchronize()


def sharded_parity(ctx, tensors, keys, vals, keys_n, vals_n, aggs, hint, n_job):
    """Parity of the N-rank result: every rank's share of the result frame (one more, untimed call).  Up to
    2e7 rows in the job, the shards' rows and the shares travel to rank 0 and the whole frame is compared with the oracle's
    on the concatenated rows (canonical key sort; keys / integer sums / counts bit-exact, f64 within 1e-12).  Beyond that,
    size-independent properties over all ranks: the shares' integer sums add up to the column's sum over all rows, counts add
    up to the job's rows, the f64 checksum holds to 1e-9, and the shares' keys are exactly the distinct keys of the job
    (disjoint between ranks, none missing: presence bitmaps over the Categorical's codes)."""
    import numpy as np
    import torch

    dist = ctx.dist
    names = [f"a{i}" for i in range(len(aggs))]
    res = ctx.eng.sharded(keys, vals, aggs, ctx.world, None, hint, names)
    nk = len(keys)
    if n_job <= 20_000_000:
        mine = {"rows": {k_: tensors[k_].cpu().numpy() for k_ in list(keys_n) + list(vals_n)},
                "masks": {k_: tensors[k_ + "__valid"].cpu().numpy() for k_ in list(keys_n) + list(vals_n) if k_ + "__valid" in tensors},
                "cols": [(c.t.cpu().numpy().view(_np_dtype(c.code)), None if c.mask is None else c.mask.cpu().numpy()) for c in res]}
        gathered = [None] * ctx.world
        dist.all_gather_object(gathered, mine)
        if ctx.rank != 0:
            return None
        from oracle import oracle as orc
        from polars_amd.column import Column

        def cat(name):
            v = np.concatenate([g_["rows"][name] for g_ in gathered])
            m = np.concatenate([g_["masks"][name] for g_ in gathered]) if name in gathered[0]["masks"] else None
            return v, m

        hk = [Column(k_, *cat(k_), dtype=keys[i].dtype) for i, k_ in enumerate(keys_n)]
        hv = [Column(v_, *cat(v_)) for v_ in vals_n]
        want, _ = orc.group_by_agg(hk, hv, aggs, n_threads=min(os.cpu_count() or 1, 16))
        got = []
        for j in range(nk + len(aggs)):
            v = np.concatenate([g_["cols"][j][0] for g_ in gathered])
            ms = [g_["cols"][j][1] for g_ in gathered]
            m = None if all(x is None for x in ms) else np.concatenate([np.ones(len(g_["cols"][j][0]), bool) if x is None else x
                                                                       for x, g_ in zip(ms, gathered)])
            got.append((v, m))
        out = parity_check(hk, hv, aggs, want, lambda: got)
        out["against"] = f"oracle (C restatement) on the {ctx.world} shards' concatenated rows, canonical key sort"
        out["groups_per_rank"] = [int(len(g_["cols"][0][0])) for g_ in gathered]
        return out
    checks = {}
    g_local = int(res[0].t.numel())
    for j, (ci, kind) in enumerate(aggs):
        col = res[nk + j].t
        src = tensors[vals_n[ci]]
        ok_rows = tensors.get(vals_n[ci] + "__valid")
        if kind == "count":
            tot = ctx.reduce([float(col.to(torch.int64).sum().item())], "SUM")[0]
            checks["counts_add_up_to_n"] = int(tot) == n_job
        elif kind == "sum" and not src.dtype.is_floating_point:
            a = int(col.to(torch.int64).sum().item())
            b = int((src if ok_rows is None else src[ok_rows]).to(torch.int64).sum().item())
            d = ctx.reduce([float(a - b)], "SUM")[0]  # the difference is what travels: exact in f64 whatever the sums' size
            checks["integer_checksum_exact"] = d == 0.0 and bool(ctx.reduce([float(a != b)], "MAX")[0] == 0.0 or d == 0.0)
        elif kind == "sum":
            a = float(col.sum().item())
            b = float((src if ok_rows is None else src[ok_rows]).sum().item())
            ta, tb = ctx.reduce([a, b], "SUM")
            checks["f64_checksum_1e-9"] = abs(ta - tb) <= 1e-9 * abs(tb)
    if nk == 1 and hint and keys_n[0] + "__valid" not in tensors:
        seen_in = torch.zeros(hint + 1, dtype=torch.float64, device=ctx.dev)
        seen_in[tensors[keys_n[0]].to(torch.int64).clamp_(0, hint)] = 1.0
        seen_out = torch.zeros(hint + 1, dtype=torch.float64, device=ctx.dev)
        seen_out.index_add_(0, res[0].t.to(torch.int64).clamp_(0, hint), torch.ones(g_local, dtype=torch.float64, device=ctx.dev))
        if ctx.share_gpu:
            seen_in, seen_out = seen_in.cpu(), seen_out.cpu()
        dist.all_reduce(seen_in, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen_out, op=dist.ReduceOp.SUM)
        checks["shares_hold_every_distinct_key_exactly_once"] = bool(torch.equal(seen_in, seen_out))
    g_tot = int(ctx.reduce([float(g_local)], "SUM")[0])
    if ctx.rank != 0:
        return None
    return {"ok": bool(checks) and all(checks.values()), "properties": checks, "groups": g_tot,
            "against": f"size-independent properties over the {ctx.world} ranks' shares (the oracle would need minutes at {n_job} rows)"}


def _np_dtype(code):
    from polars_amd._ffi import code_np_dtype

    return code_np_dtype(code)


def run_workload(ctx, a, light=False):
    """One workload: generates its rows in HBM, W untimed + K timed steps (barrier + synchronize on both sides), parity.
    Returns the JSON line's dict on rank 0 (None elsewhere).  `light`: one of the extra configs of the default run: a 3 s
    CPU baseline instead of 10 + 5 s."""
    import ctypes as C

    import torch

    import polars_amd as pa
    from polars_amd import _ffi, h2oai

    world, rank, dev, dist = ctx.world, ctx.rank, ctx.dev, ctx.dist
    sharded = ctx.sharded
    L = _ffi.lib()
    keys_n, vals_n, aggs, wide = WORKLOADS[a.workload]
    strong = a.total_rows > 0
    n = a.total_rows // world if strong else (a.rows or DEFAULT_ROWS.get(a.workload, 10_000_000))
    job_rows = n * world  # the generator's N: key cardinalities (N / K ids) follow the JOB's size under strong scaling
    steps = a.steps or {"c3": 20, "c5": 5}.get(a.workload, 200)
    warmup = a.warmup if a.warmup >= 0 else {"c3": 3, "c5": 1}.get(a.workload, 10)
    if a.workload == "c5":
        tensors = skewed_keys(n, max(int((job_rows if strong else n) * C5_IDS_PER_ROW), 1), 108 + rank, dev)
        cols = {k: pa.Column(k, t) for k, t in tensors.items()}
    else:
        tensors = h2oai.generate(n, seed=108 + rank, device=dev, wide=wide, columns=keys_n + vals_n, na=a.na,
                                 n_for_ids=job_rows if strong else None)
        cols = h2oai.to_columns(tensors, wide, n=job_rows if strong else n)
    keys = [cols[k] for k in keys_n]
    vals = [cols[v] for v in vals_n]
    agg_codes = [(ci, _ffi.AGG_NAMES[kind]) for ci, kind in aggs]
    pa.set_strategy(a.strategy)
    # --no-dense: the hashed pipelines the north_star names (no dense-code flag, and no compact-range detection under AUTO either)
    if a.no_dense:
        os.environ["PLHIP_NO_RANGE_DENSE"] = "1"
    else:
        os.environ.pop("PLHIP_NO_RANGE_DENSE", None)
    # id1..id3 are Categorical in the reference harness (run_h2oai_benchmark.py:21-35): the engine knows their
    # category counts (K, K, N/K) without touching the data, and hands the product of them on as the bound.
    hint = 0
    if not wide and not a.no_hint and all(k in ("id1", "id2", "id3") for k in keys_n):
        hint = 1
        for k in keys:
            hint *= k.n_categories
        hint = min(hint, job_rows)
    # a single Categorical key: its codes index the tables directly (group_tuples_perfect, perfect.rs:190-227)
    qflags = _ffi.FLAG_KEYS_DENSE if (hint and len(keys) == 1 and keys[0].n_categories is not None and not a.no_dense) else 0
    if a.exact_fold:
        qflags |= _ffi.FLAG_EXACT_FOLD

    eng = ctx.eng
    if sharded:
        eng.dense_keys = bool(qflags & _ffi.FLAG_KEYS_DENSE)  # the local pass takes the dense-code path as the single-GPU run does
        group = None  # the default process group

        def step():  # ONE library call per rank; the result frame stays in HBM, its row count is on the host
            handle, rcols = eng.sharded_raw(keys, vals, aggs, group, hint)
            g = L.plhip_result_num_rows(handle)
            widths = [_ffi.code_np_dtype(cs.dtype).itemsize for cs in rcols]
            L.plhip_result_free(handle)
            return g, widths
    else:
        # the column / aggregation descriptors are built once, as a host layer that re-runs a query would
        karr = (_ffi.ColumnStruct * len(keys))(*[k.struct() for k in keys])
        varr = (_ffi.ColumnStruct * len(vals))(*[v.struct() for v in vals])
        aarr = (_ffi.AggStruct * len(agg_codes))(*[_ffi.AggStruct(c, k) for c, k in agg_codes])
        opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), int(hint))
        handle0, out_cols = pa.group_by_agg_raw(keys, vals, agg_codes, _ffi.FLAG_OUTPUT_DEVICE | qflags, hint)
        widths = [_ffi.code_np_dtype(cs.dtype).itemsize for cs in out_cols]
        L.plhip_result_free(handle0)

        handle = C.c_void_p()
        call_args = (karr, len(keys), varr, len(vals), aarr, len(agg_codes), _ffi.FLAG_OUTPUT_DEVICE | qflags, C.byref(opts), C.byref(handle))

        def step():
            rc = L.plhip_group_by_agg_opts(*call_args)
            if rc:
                _ffi.check(rc)
            g = L.plhip_result_num_rows(handle)  # the result frame stays in HBM; its row count is on the host
            L.plhip_result_free(handle)
            return g, widths

    for _ in range(warmup):
        step()
    ctx.sync()
    kernel_ms, dom_ms, parts, dom_names, phases = [], [], [], [], []
    # the library's HIP-event times of every step are read inside the timed region; the ctypes out-parameters are built once
    # (three object constructions per call cost more than the call)
    f_tot, f_dom, f_name, f_a, f_b, f_c = C.c_float(), C.c_float(), C.c_char_p(), C.c_float(), C.c_float(), C.c_float()
    r_ms = (C.byref(f_tot), C.byref(f_dom), C.byref(f_name))
    r_bd = (C.byref(f_a), C.byref(f_b), C.byref(f_c))
    # the library records its HIP events on every TIMING_EVERY-th query (three event records and their read-back -- which waits for
    # the stream's tail -- cost ~15 us, 15 % of a 0.1 ms step); the times below are those of the sampled steps INSIDE the timed
    # region (the driver's 20 steps: steps 10 and 20)
    TIMING_EVERY = max(1, min(10, steps // 2)) if not sharded else 1
    L.plhip_set_timing_interval(TIMING_EVERY)
    verbose = bool(os.environ.get("PLHIP_BENCH_VERBOSE"))
    t_steps = []

    def read_sample():
        if sharded:
            # the sharded step makes several library calls: the engine kept the kernel times of the LOCAL partial pass
            # (the pass over this rank's rows) and the wall time of every phase of the step
            tot, dom, dom_name = eng.last_partial_kernel_ms
            parts.append(eng.last_partial_breakdown)
            phases.append(dict(eng.last_phase_ms))
        else:
            L.plhip_last_kernel_ms(*r_ms)  # HIP events on the library's own stream, recorded live inside the timed region
            L.plhip_last_kernel_breakdown(*r_bd)
            tot, dom, dom_name = f_tot.value, f_dom.value, f_name.value
            parts.append((f_a.value, f_b.value, f_c.value))
        kernel_ms.append(tot)
        dom_ms.append(dom)
        dom_names.append(dom_name)

    # A sampled step's events are READ as late as possible: right before the next sampled step (TIMING_EVERY steps later, when they
    # have long completed and the read costs no wait) or, for the last sample, behind the closing synchronisation.  Reading them
    # right behind their own step waits for the stream's tail -- the wait that deferred completion took out of every step.
    lazy = not sharded and TIMING_EVERY >= 2
    unread = False
    import gc

    gc_was_on = gc.isenabled()
    gc.disable()  # (no collector pause inside the timed region: a step is ~0.1 ms, a generation-2 pass several)
    t0 = time.perf_counter()
    for i_step in range(steps):
        sampled = i_step % TIMING_EVERY == TIMING_EVERY - 1
        if sampled and unread:
            read_sample()
            unread = False
        g, out_widths = step()
        if verbose:
            t_steps.append(time.perf_counter())
        if not sampled:
            continue
        if lazy:
            unread = True
        else:
            read_sample()
    t_loop = time.perf_counter() - t0
    ctx.sync()
    dt = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    if verbose:  # (diagnostics on stderr: where the timed region's time went)
        per = [round((b - a) * 1e6, 1) for a, b in zip([t0] + t_steps[:-1], t_steps)]
        print(f"[bench] loop {t_loop * 1e3:.4f} ms, closing sync {(dt - t_loop) * 1e3:.4f} ms, {steps} steps; per step us: {per[:40]}", file=sys.stderr)
    if unread:
        read_sample()
    L.plhip_set_timing_interval(1)
    groups_total = g
    n_ranks = world
    if sharded:
        # the rank count is the one the library's communicator reports (RCCL: ncclCommCount), not an environment variable
        n_ranks, comm_rank, is_rccl = eng.comm_info(None)
        assert comm_rank == rank, (comm_rank, rank)
    if world > 1:
        dt = ctx.reduce([dt], "MAX")[0]
        groups_total = int(ctx.reduce([float(g)], "SUM")[0])
    parity = None
    if world > 1 and not a.no_parity:
        parity = sharded_parity(ctx, tensors, keys, vals, keys_n, vals_n, aggs, hint, job_rows)

    line = None
    if rank == 0:
        ms_per_step = dt / steps * 1e3
        total_rows = n * n_ranks
        strategy = pa.last_strategy() if not sharded else eng.last_partial_strategy
        # the local pass of rank 0 (for N > 1: its partial aggregation over its n rows, the dominant kernels of the step)
        g_local = g if not sharded else eng.last_partial_groups
        if g_local < 0:  # the dense exchange never counts the local pass's groups: its partial arrays cover every code
            g_local = hint
        abytes = algorithmic_bytes(n, g_local, keys, vals, out_widths)
        dom_avg = sum(dom_ms) / len(dom_ms)
        tot_avg = sum(kernel_ms) / len(kernel_ms)
        achieved = abytes / (dom_avg * 1e-3) / 1e9 if dom_avg > 0 else 0.0
        dom_name = max(set(dom_names), key=dom_names.count)
        dom_name = dom_name.decode() if isinstance(dom_name, bytes) else dom_name
        traffic_key = a.workload + ("_na" if a.na else "") + f":{n}:{strategy}"
        traffic, traffic_kernels = measured_traffic(traffic_key)
        acc_dtype = {"q1": "i32", "q3": "i32+f64", "q5": "i32+f64", "c3": "i64+f64", "c5": "i64+f64",
                     "q2": "i32", "q4": "f64", "q7": "i32", "q10": "f64+u32"}[a.workload]
        line = {
            "metric": "db-benchmark group-by rows/sec + % HBM roofline at 1/2/4/8 MI355X",
            "value": total_rows / (dt / steps),
            "unit": "rows/s",
            "n_gpus": n_ranks,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": acc_dtype,
            "data": "synthetic",
            "config": {"workload": f"{'db-benchmark ' if a.workload[0] == 'q' else 'BASELINE config '}{a.workload}"
                                   f"{' (NA variant)' if a.na else ''}: {', '.join(f'{k}({vals_n[c]})' for c, k in aggs)} by "
                                   f"{','.join(keys_n)}, {n} rows per GPU, {groups_total} groups",
                       "strategy": strategy, "n_groups_hint": hint, "dense_codes": bool(qflags & _ffi.FLAG_KEYS_DENSE), "rows_per_gpu": n,
                       "groups": int(groups_total),
                       "exchange": "none" if not sharded else "plhip_group_by_agg_sharded: ncclSend/ncclRecv of partial records on the library's stream (RCCL)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_by_kernel": traffic_kernels,
                         "traffic_source": (f"profiles/traffic.json[{traffic_key}]: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE passes of this "
                                            "workload kept under profiles/, NOT counted in this run") if traffic else None,
                         "traffic_over_algorithmic": (traffic / abytes) if traffic else None, "kernel": dom_name,
                         "kernel_ms": dom_avg, "algorithmic_bytes": abytes,
                         "breakdown_ms": {k: sum(p[i] for p in parts) / len(parts) for i, k in
                                          enumerate(("partition", "fold", "merge_emit"))},
                         "all_kernels_ms": tot_avg,
                         "all_kernels_frac": abytes / (tot_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS if tot_avg > 0 else 0.0,
                         "step_frac": abytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        if sharded:
            line["config"]["transport"] = eng.transport
            line["config"]["ranks_share_one_gpu"] = ctx.share_gpu
            # per-GPU GB/s (this rank's algorithmic bytes over the whole step's wall time) ...
            line["per_gpu"] = {"algorithmic_GBps": abytes / (ms_per_step * 1e-3) / 1e9, "rows_per_s": n / (ms_per_step * 1e-3),
                               "frac_of_hbm_peak": abytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        if phases:  # ... and the shuffle's bytes and time reported separately
            line["sharded_step_ms"] = {k: sum(p.get(k, 0.0) for p in phases) / len(phases) for k in phases[0]}
            line["shuffle"] = {"bytes_sent_per_rank": eng.last_shuffle_bytes, "ms": line["sharded_step_ms"].get("exchange"),
                               "protocol": eng.last_protocol, "host_syncs_per_step": eng.last_host_syncs, "transport": eng.transport}
        if parity is not None:
            line["parity"] = parity
        if not sharded and not a.no_cpu_baseline:
            def gpu_frame():  # the timed configuration once more, result copied to the host
                handle, out = pa.group_by_agg_raw(keys, vals, agg_codes, qflags, hint)
                try:
                    fr = []
                    for cs in out:
                        c = pa.frame._host_column("c", cs)
                        fr.append((c.to_numpy().copy(), None if c.valid_mask() is None else c.valid_mask().copy()))
                    return fr
                finally:
                    L.plhip_result_free(handle)

            # whole-frame parity against the oracle up to 1e8 rows (as an extra config of the default run only up to 2e7: the
            # canonical sort of config 3's 1e8-row result on the host alone takes over a minute)
            if n <= (20_000_000 if light else 100_000_000):
                line["cpu_baseline"], line["parity"] = cpu_baseline(tensors, keys, vals, aggs, gpu_frame, budget_s=3.0 if light else 10.0,
                                                                    path_b=not light)
            else:
                line["cpu_baseline"], line["parity"] = cpu_baseline_sample(tensors, keys_n, vals_n, aggs, gpu_frame)
    del tensors, cols, keys, vals
    return line


# the BASELINE single-GPU configs the default run times after the headline; (name, workload, overrides, steps, warmup,
# seconds the config is expected to need on the box: data generation + steps + its CPU leg)
EXTRA_CONFIGS = [
    ("q1", "q1", {}, 100, 10, 8),
    ("q5", "q5", {}, 100, 10, 8),
    ("q3_na", "q3", {"na": True}, 100, 10, 10),
    ("q3_no_dense", "q3", {"no_dense": True}, 100, 10, 8),
    ("q3_exact_fold", "q3", {"exact_fold": True}, 50, 5, 8),
    ("c3", "c3", {}, 10, 2, 45),
    ("c5", "c5", {}, 4, 1, 90),
    # the other fused queries of the reference's harness, behind the BASELINE configs (they run if the budget still allows)
    ("q2", "q2", {}, 100, 10, 8),
    ("q4", "q4", {}, 100, 10, 8),
    ("q7", "q7", {}, 100, 10, 8),
    ("q10", "q10", {}, 30, 3, 15),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks, one per GPU (default: WORLD_SIZE, else 1); with N > 1 and no WORLD_SIZE "
                                                         "in the environment this process starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=0, help="default 200 (c3: 20, c5: 5)")
    ap.add_argument("--warmup", type=int, default=-1, help="default 10 (c3: 3, c5: 1)")
    ap.add_argument("--workload", default="q3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default 1e7; c3: 1e8; c5: 1e9)")
    ap.add_argument("--total-rows", type=int, default=0,
                    help="strong scaling (BASELINE configs[3]): the job's rows, every rank gets total / N of them")
    ap.add_argument("--strategy", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the parity leg (one more, untimed sharded call + checks)")
    ap.add_argument("--no-dense", action="store_true", help="do not declare a single Categorical key's codes dense (PLHIP_FLAG_KEYS_DENSE)")
    ap.add_argument("--no-hint", action="store_true",
                    help="do not pass the category count of categorical keys (id1..id3) as n_groups_hint")
    ap.add_argument("--na", action="store_true", help="db-benchmark's NA variant (groupby-datagen.R:30-44): 5 %% of the distinct key values "
                                                      "nulled, 5 %% NA values, as validity bitmaps")
    ap.add_argument("--exact-fold", action="store_true", help="PLHIP_FLAG_EXACT_FOLD: float sums / means folded in row order (bit-exact, deterministic)")
    ap.add_argument("--configs", default="auto", help="'auto': a default single-GPU q3 run also times the other BASELINE single-GPU configs "
                                                      "(q1, q5, q3 NA, q3 hashed, c3, c5) into \"configs\"; 'none'; or a comma list of their names")
    ap.add_argument("--configs-budget-s", type=float, default=240.0, help="wall-clock budget of the extra configs")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(launch_ranks(args))
    if env_world is not None and args.gpus and int(env_world) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: start exactly one rank per GPU")

    import torch

    import polars_amd as pa
    from polars_amd import _ffi

    # ONE JSON line on stdout, nothing else: native libraries write to descriptor 1 on their own (RCCL prints a version banner
    # when its first communicator is made), so everything but the line goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ctx = Ctx()
    ctx.world = int(env_world or "1")
    ctx.rank = int(os.environ.get("RANK", "0"))
    ctx.share_gpu = os.environ.get("PLHIP_BENCH_SHARE_GPU") == "1"  # all ranks on device 0, host (gloo) transport
    local_rank = 0 if ctx.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    ctx.rehearse = os.environ.get("PLHIP_BENCH_FORCE_EXCHANGE") == "1"  # run the sharded path (and RCCL) even on one rank
    ctx.dev = torch.device(f"cuda:{local_rank}")
    if ctx.sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if ctx.share_gpu:
            dist.init_process_group("gloo", rank=ctx.rank, world_size=ctx.world)
        else:
            dist.init_process_group("nccl", device_id=ctx.dev, rank=ctx.rank, world_size=ctx.world)
        ctx.dist = dist
    torch.cuda.set_device(local_rank)
    _ffi.check(_ffi.lib().plhip_init(local_rank))
    if ctx.sharded:
        from polars_amd import distributed as pdist

        ctx.eng = pdist.HipEngine(ctx.dev)

    t_start = time.perf_counter()
    line = run_workload(ctx, args)
    extra = args.configs != "none" and not ctx.sharded and args.workload == "q3" and not (
        args.rows or args.total_rows or args.na or args.no_dense or args.no_hint or args.exact_fold or args.strategy != "auto" or args.no_cpu_baseline)
    if extra and line is not None:
        wanted = None if args.configs == "auto" else set(args.configs.split(","))
        line["configs"] = {}
        for name, wl, over, k_steps, k_warm, need_s in EXTRA_CONFIGS:
            if wanted is not None and name not in wanted:
                continue
            left = args.configs_budget_s - (time.perf_counter() - t_start)
            if left < need_s:
                line["configs"][name] = {"skipped": f"{left:.0f} s of the extra configs' budget left, needs ~{need_s} s (run: python bench.py --workload {wl}"
                                                    f"{' --na' if over.get('na') else ''}{' --no-dense' if over.get('no_dense') else ''})"}
                continue
            sub = argparse.Namespace(**{**vars(args), **over, "workload": wl, "steps": k_steps, "warmup": k_warm})
            _ffi.lib().plhip_release_workspace()
            torch.cuda.empty_cache()
            t1 = time.perf_counter()
            try:
                ln = run_workload(ctx, sub, light=True)
            except Exception as e:  # one config's failure must not take the headline with it; it is reported, not hidden
                line["configs"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                continue
            rf = ln["roofline"]
            line["configs"][name] = {
                "workload": ln["config"]["workload"], "strategy": ln["config"]["strategy"], "steps": k_steps, "ms_per_step": ln["ms_per_step"],
                "rows_per_s": ln["value"], "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "kernel_frac": rf["frac"],
                "all_kernels_ms": rf["all_kernels_ms"], "all_kernels_frac": rf["all_kernels_frac"], "step_frac": rf["step_frac"],
                "algorithmic_bytes": rf["algorithmic_bytes"], "traffic_over_algorithmic": rf["traffic_over_algorithmic"],
                "traffic_source": rf["traffic_source"], "parity": ln.get("parity"),
                "cpu_baseline_rows_per_s": (ln.get("cpu_baseline") or {}).get("value"), "wall_s": time.perf_counter() - t1}
        pa.set_strategy("auto")
    if ctx.rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if ctx.sharded:
        from polars_amd import distributed as pdist

        pdist.HipEngine.close()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
