#!/bin/bash
# rocprofv3 kernel stats + HBM traffic counters of one bench command: scripts/gpu_prof2.sh <tag> <bench args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $R/bench.py --no-cpu-baseline --steps 100 "$@" > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o write -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1 || exit 1
cd $R
python scripts/prof_summary.py $out
