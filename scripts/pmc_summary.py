"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python scripts/pmc_summary.py <csv>..."""
import collections, csv, re, sys
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "at::" in k or "rocclr" in k:
            continue
        k = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "").split("(")[0][:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, c in agg.items():
        print(f"{k:48s}", {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}, "calls", len(next(iter(c.values()))))
