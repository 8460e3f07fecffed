#!/bin/bash
# the db-benchmark queries at config 4's sizes on ONE GPU (the N = 1 point of the strong-scaling curve): scripts/gpu_big.sh <tag>
tag=${1:-big}; out=gpurun_out/$tag; mkdir -p $out
for spec in "q1 100000000" "q1 1000000000" "q3 100000000" "q3 1000000000" "q5 1000000000"; do
  set -- $spec
  timeout -k 10 500 python bench.py --workload $1 --rows $2 --steps 10 --warmup 2 --no-cpu-baseline > $out/$1_$2.json 2> $out/$1_$2.err || { echo "$1 $2 FAILED"; tail -3 $out/$1_$2.err; continue; }
  python - <<PY
import json
d=json.load(open('$out/$1_$2.json')); r=d['roofline']
print('$1 $2', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],3), 'rows/s %.3g'%d['value'], {k: round(v,3) for k,v in r['breakdown_ms'].items()}, 'dom_frac', round(r['frac'],3), 'all_frac', round(r['all_kernels_frac'],3), 'step_frac', round(r['algorithmic_bytes']/(d['ms_per_step']*1e-3)/8e12,3))
PY
done
