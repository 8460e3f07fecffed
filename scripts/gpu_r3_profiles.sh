#!/bin/bash
# round 3 profile set: rocprofv3 --kernel-trace --stats + the two --pmc passes (FETCH_SIZE, WRITE_SIZE) per workload, flattened
# into gpurun_out/r3p/ (copied to profiles/ afterwards), and profiles/traffic.json entries.   scripts/gpu_r3_profiles.sh <tag> [workloads...]
tag=${1:-r3a}; shift
wls=${@:-"q3 q1 q5 q3na c3 q3big c5"}
mkdir -p gpurun_out/r3p
for wl in $wls; do
  case $wl in
    q3) args="--workload q3"; key="q3:10000000:dense";;
    q1) args="--workload q1"; key="q1:10000000:direct";;
    q5) args="--workload q5"; key="q5:10000000:dense";;
    q3na) args="--workload q3 --na"; key="q3_na:10000000:dense";;
    q3hash) args="--workload q3 --no-dense"; key="q3:10000000:tilesort";;
    c3) args="--workload c3"; key="c3:100000000:leafsort";;
    c5) args="--workload c5"; key="c5:1000000000:tilesort2";;
    q3big) args="--workload q3 --rows 1000000000"; key="q3:1000000000:dense_part";;
  esac
  echo "== $wl ($key)"
  bash scripts/gpu_prof2.sh ${tag}_$wl $args --configs none > gpurun_out/r3p/${tag}_${wl}_summary.txt 2>&1 || { tail -5 gpurun_out/r3p/${tag}_${wl}_summary.txt; exit 1; }
  python scripts/prof_summary.py gpurun_out/${tag}_$wl --json "$key" > /dev/null
  cat gpurun_out/r3p/${tag}_${wl}_summary.txt
  cp gpurun_out/${tag}_$wl/stats/stats_kernel_stats.csv gpurun_out/r3p/${tag}_${wl}_kernel_stats.csv
  cp gpurun_out/${tag}_$wl/pmc_fetch/fetch_counter_collection.csv gpurun_out/r3p/${tag}_${wl}_pmc_fetch.csv
  cp gpurun_out/${tag}_$wl/pmc_write/write_counter_collection.csv gpurun_out/r3p/${tag}_${wl}_pmc_write.csv
done
cp profiles/traffic.json gpurun_out/r3p/traffic.json
