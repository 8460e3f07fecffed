"""Prints the heaviest kernels of a rocprofv3 --stats run: python scripts/top_kernels.py <dir> [count]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    n = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "")[:70]
    print(f"{n:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:9.1f} us")
