"""Per-kernel summary of one scripts/gpu_prof2.sh run: average duration (rocprofv3 --kernel-trace --stats) and HBM
traffic per launch (two --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) of every kernel
of the library.   python scripts/prof_summary.py <dir> [--json key]  (with --json: also updates profiles/traffic.json)"""
import collections, csv, json, re, sys
from pathlib import Path

d = Path(sys.argv[1])


def short(k):
    k = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "")
    return k.split("(")[0].split("<")[0]


def ours(k):
    return k.startswith("k_")


dur = {}
f = d / "stats" / "stats_kernel_stats.csv"
if f.exists():
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if ours(k):
            e = dur.setdefault(k, [0, 0.0])
            e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"])
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, name in (("pmc_fetch", "fetch"), ("pmc_write", "write")):
    f = d / sub / f"{name}_counter_collection.csv"
    if f.exists():
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if ours(k):
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot_ns, tot_b = 0.0, 0.0
rows = {}
for k in sorted(set(dur) | set(pmc)):
    calls, ns = dur.get(k, [0, 0.0])
    avg_us = ns / calls / 1e3 if calls else float("nan")
    fe = pmc[k].get("FETCH_SIZE", [])
    wr = pmc[k].get("WRITE_SIZE", [])
    fb = 2 * 1024 * sum(fe) / len(fe) if fe else 0.0
    wb = 1024 * sum(wr) / len(wr) if wr else 0.0
    rows[k] = {"avg_us": avg_us, "calls": calls, "fetch_bytes": fb, "write_bytes": wb}
    print(f"{k:28s} calls {calls:6d} avg {avg_us:9.2f} us   fetch {fb / 1e6:9.2f} MB  write {wb / 1e6:9.2f} MB  (pmc launches {len(fe)}/{len(wr)})")
if "--json" in sys.argv:
    key = sys.argv[sys.argv.index("--json") + 1]
    out = Path("profiles/traffic.json")
    t = json.loads(out.read_text()) if out.exists() else {}
    # launches per query: a kernel that runs twice per query (the two partition levels' k_tile_sort, k_estimate's two sample
    # sizes) shows twice the calls of the ones that run once
    base = min((r["calls"] for r in rows.values() if r["calls"]), default=1)
    for r in rows.values():
        r["launches_per_query"] = max(1, round(r["calls"] / base)) if r["calls"] else 1
    t[key] = {"kernels": rows, "hbm_bytes": int(sum(r["launches_per_query"] * (r["fetch_bytes"] + r["write_bytes"]) for r in rows.values())),
              "note": "sum over the query's kernels of launches_per_query x (FETCH_SIZE x 2 (gfx950) + WRITE_SIZE per launch)"}
    out.write_text(json.dumps(t, indent=1, sort_keys=True) + "\n")
