#!/bin/bash
# rocprofv3 kernel stats of config 3 (1e8 rows, ~1e8 groups): scripts/gpu_prof_c3.sh [strategy]
mkdir -p gpurun_out/prof_c3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --strategy ${1:-auto} --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c3/bench.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_c3/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:70].replace("(anonymous namespace)::",""), r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3))
PY
exit $rc
