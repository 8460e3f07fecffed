#!/bin/bash
mkdir -p gpurun_out/tuples
timeout -k 10 200 python scripts/tuples_prof.py 1e7 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tuples -o tuples -- python $GRAFT_REPO_ROOT/scripts/tuples_prof.py 1e7 > $GRAFT_REPO_ROOT/gpurun_out/tuples/log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/tuples/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:90].replace("(anonymous namespace)::",""), r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3), "total ms %.2f"%(float(r["TotalDurationNs"])/1e6))
PY
