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


def cpu_baseline(tensors, keys, vals, aggs, gpu_frame, budget_s=10.0):
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
    if len(hk) == 1:  # path B: the streaming sink's accumulators, timed on the same rows
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


def measured_traffic(workload, n_rows, strategy):
    """HBM bytes of the WHOLE query (summed over its kernels, per launch) from the rocprofv3 --pmc passes kept under
    profiles/ (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; scripts/prof_summary.py
    writes the entry together with the per-kernel figures); None if no such measurement is on file for this
    workload and strategy."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None, None
    try:
        t = json.loads(f.read_text()).get(f"{workload}:{n_rows}:{strategy}")
        if not t:
            return None, None
        return t["hbm_bytes"], {k: int(v["fetch_bytes"] + v["write_bytes"]) for k, v in t.get("kernels", {}).items()}
    except (ValueError, OSError, KeyError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="default 200 (c3: 20, c5: 5)")
    ap.add_argument("--warmup", type=int, default=-1, help="default 10 (c3: 3, c5: 1)")
    ap.add_argument("--workload", default="q3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default 1e7; c3: 1e8; c5: 1e9)")
    ap.add_argument("--total-rows", type=int, default=0,
                    help="strong scaling (BASELINE configs[3]): the job's rows, every rank gets total / N of them")
    ap.add_argument("--strategy", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="do not declare a single Categorical key's codes dense (PLHIP_FLAG_KEYS_DENSE)")
    ap.add_argument("--no-hint", action="store_true",
                    help="do not pass the category count of categorical keys (id1..id3) as n_groups_hint")
    ap.add_argument("--na", action="store_true", help="db-benchmark's NA variant (groupby-datagen.R:30-44): 5 %% of the distinct key values "
                                                      "nulled, 5 %% NA values, as validity bitmaps")
    args = ap.parse_args()

    import torch

    import polars_amd as pa
    from polars_amd import _ffi, h2oai

    # ONE JSON line on stdout, nothing else: native libraries write to descriptor 1 on their own (RCCL prints a version banner
    # when its first communicator is made), so everything but the line goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("PLHIP_BENCH_FORCE_EXCHANGE") == "1"  # run the sharded path (and RCCL) even on one rank
    if world > 1 or rehearse:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    _ffi.check(_ffi.lib().plhip_init(local_rank))

    wl = WORKLOADS[args.workload]
    keys_n, vals_n, aggs, wide = wl
    strong = args.total_rows > 0
    n = args.total_rows // world if strong else (args.rows or DEFAULT_ROWS.get(args.workload, 10_000_000))
    job_rows = n * world  # the generator's N: key cardinalities (N / K ids) follow the JOB's size under strong scaling
    steps = args.steps or {"c3": 20, "c5": 5}.get(args.workload, 200)
    warmup = args.warmup if args.warmup >= 0 else {"c3": 3, "c5": 1}.get(args.workload, 10)
    if args.workload == "c5":
        tensors = skewed_keys(n, max(int((job_rows if strong else n) * C5_IDS_PER_ROW), 1), 108 + rank, dev)
        cols = {k: pa.Column(k, t) for k, t in tensors.items()}
    else:
        tensors = h2oai.generate(n, seed=108 + rank, device=dev, wide=wide, columns=keys_n + vals_n, na=args.na,
                                 n_for_ids=job_rows if strong else None)
        cols = h2oai.to_columns(tensors, wide, n=job_rows if strong else n)
    keys = [cols[k] for k in keys_n]
    vals = [cols[v] for v in vals_n]
    agg_codes = [(ci, _ffi.AGG_NAMES[kind]) for ci, kind in aggs]
    pa.set_strategy(args.strategy)
    L = _ffi.lib()
    # id1..id3 are Categorical in the reference harness (run_h2oai_benchmark.py:21-35): the engine knows their
    # category counts (K, K, N/K) without touching the data, and hands the product of them on as the bound.
    hint = 0
    if not wide and not args.no_hint and all(k in ("id1", "id2", "id3") for k in keys_n):
        hint = 1
        for k in keys:
            hint *= k.n_categories
        hint = min(hint, n)
    # a single Categorical key: its codes index the tables directly (group_tuples_perfect, perfect.rs:190-227)
    qflags = _ffi.FLAG_KEYS_DENSE if (hint and len(keys) == 1 and keys[0].n_categories is not None and not args.no_dense) else 0

    if world > 1 or rehearse:
        from polars_amd import distributed as pdist

        eng = pdist.HipEngine(dev)
        eng.dense_keys = bool(qflags)  # the local pass takes the dense-code path as the single-GPU run does

        group = None  # the default process group

        def step():  # ONE library call per rank; the result frame stays in HBM, its row count is on the host
            handle, cols = eng.sharded_raw(keys, vals, aggs, group, hint)
            g = L.plhip_result_num_rows(handle)
            widths = [_ffi.code_np_dtype(cs.dtype).itemsize for cs in cols]
            L.plhip_result_free(handle)
            return g, widths
    else:
        import ctypes as C

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

    def sync():
        torch.cuda.synchronize()
        if world > 1 or rehearse:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    kernel_ms, dom_ms, parts, dom_names, phases = [], [], [], [], []
    # the library's HIP-event times of every step are read inside the timed region; the ctypes out-parameters are built once
    # (three object constructions per call cost more than the call)
    import ctypes as C

    f_tot, f_dom, f_name, f_a, f_b, f_c = C.c_float(), C.c_float(), C.c_char_p(), C.c_float(), C.c_float(), C.c_float()
    r_ms = (C.byref(f_tot), C.byref(f_dom), C.byref(f_name))
    r_bd = (C.byref(f_a), C.byref(f_b), C.byref(f_c))
    sharded = world > 1 or rehearse
    # the library records its HIP events on every TIMING_EVERY-th query (three event records and their read-back cost ~5 % of a
    # 0.1 ms step); the times below are those of the sampled steps INSIDE the timed region
    TIMING_EVERY = max(1, min(8, steps // 4)) if not sharded else 1
    L.plhip_set_timing_interval(TIMING_EVERY)
    t0 = time.perf_counter()
    for i_step in range(steps):
        g, out_widths = step()
        if i_step % TIMING_EVERY != TIMING_EVERY - 1:
            continue
        if sharded:
            # the sharded step makes several library calls: the engine kept the kernel times of the LOCAL partial pass
            # (the pass over this rank's rows) and the wall time of every phase of the step
            tot, dom, dom_name = eng.last_partial_kernel_ms
            parts.append(eng.last_partial_breakdown)
            phases.append(dict(eng.last_phase_ms))
        else:
            L.plhip_last_kernel_ms(*r_ms)  # HIP events on the library's own stream, recorded live
            L.plhip_last_kernel_breakdown(*r_bd)
            tot, dom, dom_name = f_tot.value, f_dom.value, f_name.value
            parts.append((f_a.value, f_b.value, f_c.value))
        kernel_ms.append(tot)
        dom_ms.append(dom)
        dom_names.append(dom_name)
    sync()
    dt = time.perf_counter() - t0
    L.plhip_set_timing_interval(1)
    groups_total = g
    if world > 1:
        t = torch.tensor([dt, float(g)], device=dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, groups_total = float(tmax[0].item()), int(t[1].item())

    if rank == 0:
        ms_per_step = dt / steps * 1e3
        total_rows = n * world
        strategy = pa.last_strategy() if world == 1 and not rehearse else eng.last_partial_strategy
        # the local pass of rank 0 (for N > 1: its partial aggregation over its n rows, the dominant kernels of the step)
        g_local = g if world == 1 and not rehearse else eng.last_partial_groups
        if g_local < 0:  # the dense exchange never counts the local pass's groups: its partial arrays cover every code
            g_local = hint
        abytes = algorithmic_bytes(n, g_local, keys, vals, out_widths)
        dom_avg = sum(dom_ms) / len(dom_ms)
        tot_avg = sum(kernel_ms) / len(kernel_ms)
        achieved = abytes / (dom_avg * 1e-3) / 1e9 if dom_avg > 0 else 0.0
        dom_name = max(set(dom_names), key=dom_names.count)
        dom_name = dom_name.decode() if isinstance(dom_name, bytes) else dom_name
        traffic, traffic_kernels = measured_traffic(args.workload + ("_na" if args.na else ""), n, strategy)
        acc_dtype = {"q1": "i32", "q3": "i32+f64", "q5": "i32+f64", "c3": "i64+f64", "c5": "i64+f64"}[args.workload]
        line = {
            "metric": "db-benchmark group-by rows/sec + % HBM roofline at 1/2/4/8 MI355X",
            "value": total_rows / (dt / steps),
            "unit": "rows/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": acc_dtype,
            "data": "synthetic",
            "config": {"workload": f"{'db-benchmark ' if args.workload[0] == 'q' else 'BASELINE config '}{args.workload}"
                                   f"{' (NA variant)' if args.na else ''}: {', '.join(f'{k}({vals_n[c]})' for c, k in aggs)} by "
                                   f"{','.join(keys_n)}, {n} rows per GPU, {groups_total} groups",
                       "strategy": strategy, "n_groups_hint": hint, "dense_codes": bool(qflags), "rows_per_gpu": n, "groups": int(groups_total),
                       "exchange": "none" if world == 1 and not rehearse else "plhip_group_by_agg_sharded: ncclSend/ncclRecv of partial records on the library's stream (RCCL)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_by_kernel": traffic_kernels,
                         "traffic_over_algorithmic": (traffic / abytes) if traffic else None, "kernel": dom_name,
                         "kernel_ms": dom_avg, "algorithmic_bytes": abytes,
                         "breakdown_ms": {k: sum(p[i] for p in parts) / len(parts) for i, k in
                                          enumerate(("partition", "fold", "merge_emit"))},
                         "all_kernels_ms": tot_avg,
                         "all_kernels_frac": abytes / (tot_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS if tot_avg > 0 else 0.0},
        }
        if phases:  # shuffle bytes and time reported separately
            line["sharded_step_ms"] = {k: sum(p.get(k, 0.0) for p in phases) / len(phases) for k in phases[0]}
            line["shuffle"] = {"bytes_sent_per_rank": eng.last_shuffle_bytes, "protocol": eng.last_protocol, "host_syncs_per_step": eng.last_host_syncs,
                               "transport": eng.transport}
        if world == 1 and not rehearse and not args.no_cpu_baseline:
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

            if n <= 100_000_000:
                line["cpu_baseline"], line["parity"] = cpu_baseline(tensors, keys, vals, aggs, gpu_frame)
            else:
                line["cpu_baseline"], line["parity"] = cpu_baseline_sample(tensors, keys_n, vals_n, aggs, gpu_frame)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
