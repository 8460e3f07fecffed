#!/bin/bash
# c5 leaf-fold shape experiments: scripts/gpu_c5_var.sh <tag>
tag=${1:-c5var}; out=gpurun_out/$tag; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 3 --warmup 1 > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return 1; }
  python - <<PY
import json
d=json.load(open('$out/$name.json')); r=d['roofline']
print('$name', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in r['breakdown_ms'].items()}, 'parity', d.get('parity',{}).get('ok'))
PY
}
run base PLHIP_X=0 && run uniform PLHIP_BENCH_ZIPF_SHARE=0 && run zipf90 PLHIP_BENCH_ZIPF_SHARE=0.9 && run nohot PLHIP_T2_NOHOT=1
