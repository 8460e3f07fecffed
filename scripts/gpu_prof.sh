#!/bin/bash
# rocprofv3 kernel-trace stats of one bench workload: scripts/gpu_prof.sh <workload> <strategy> <tag>
wl=${1:-q3}; st=${2:-auto}; tag=${3:-prof}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --strategy $st --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/$tag -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-200 | head -30
exit $rc
