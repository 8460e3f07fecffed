#!/bin/bash
for rows in 100000000 1000000000; do
tag=part_$rows
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --rows $rows --steps 5 --warmup 2 --no-cpu-baseline --configs none > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python - "$tag" <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/{sys.argv[1]}/*kernel_stats.csv")
print(sys.argv[1])
for r in list(csv.DictReader(open(f[0])))[:9]:
    print("  ", r["Name"][:70].replace("(anonymous namespace)::",""), r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3))
PY
done
