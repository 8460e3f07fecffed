#!/bin/bash
# one rocprofv3 --pmc pass: scripts/gpu_pmc.sh <workload> <strategy> <tag> <counters...>
wl=$1; st=$2; tag=$3; shift 3
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --strategy $st --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/$tag/*counter_collection.csv")
if f:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0][:40]
        if k.startswith("void at::") or "rocclr" in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in agg.items():
        print(k, {n: round(sum(v)/len(v)) for n,v in c.items()}, "calls", len(next(iter(c.values()))))
PY
exit $rc
