#!/bin/bash
# kernel stats only: scripts/gpu_stats.sh <tag> <bench args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $R/bench.py --no-cpu-baseline "$@" > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
cd $R
grep -v "^{" $out/stats.log | grep -i "tilesort2\|leaf" | tail -2
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/stats/stats_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    n=re.sub(r"^void ","",r["Name"]).replace("(anonymous namespace)::","")[:70]
    print(f"{n:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:10.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
