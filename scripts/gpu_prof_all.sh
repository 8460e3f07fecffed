#!/bin/bash
# the round's final profile set: scripts/gpu_prof_all.sh <tag>  -> gpurun_out/<tag>_<workload>/
tag=${1:-r2c}
for wl in q3 q3na q1 q5 c3 c5; do
  case $wl in
    q3na) args="--workload q3 --na";;
    c3) args="--workload c3 --steps 20";;
    c5) args="--workload c5 --steps 5";;
    *) args="--workload $wl";;
  esac
  echo "== $wl"
  bash scripts/gpu_prof2.sh ${tag}_$wl $args || exit 1
done
