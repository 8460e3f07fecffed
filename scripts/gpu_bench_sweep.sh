#!/bin/bash
# quick look at every strategy on q1/q3 (development aid)
mkdir -p gpurun_out
for wl in q1 q3; do for st in lds global auto; do
  echo "== $wl $st" >> gpurun_out/sweep.log
  timeout -k 10 120 python bench.py --workload $wl --strategy $st --steps 10 --warmup 2 --no-cpu-baseline >> gpurun_out/sweep.log 2>&1 || exit 1
done; done
cat gpurun_out/sweep.log
