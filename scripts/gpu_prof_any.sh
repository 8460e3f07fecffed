#!/bin/bash
# rocprofv3 kernel stats of one bench workload at a given size: scripts/gpu_prof_any.sh <workload> <rows> <strategy> <tag>
wl=$1; rows=$2; st=${3:-auto}; tag=${4:-prof_any}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --rows $rows --strategy $st --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/$tag/bench.log | cut -c1-300
python - "$tag" <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/{sys.argv[1]}/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    print(r["Name"][:64].replace("(anonymous namespace)::",""), r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3))
PY
exit $rc
