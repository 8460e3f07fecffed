#!/bin/bash
# compares a variant build of the library with the shipped one on the large workloads: scripts/gpu_lib_var.sh <tag> <path to variant libplhip.so>
tag=${1:-libvar}; lib=$2; out=gpurun_out/$tag; mkdir -p $out
run() { # name, workload args..., then env assignments after --
  name=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout -k 10 300 python bench.py "${args[@]}" --no-cpu-baseline --steps 3 --warmup 1 > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return 1; }
  python - <<PY
import json
d=json.load(open('$out/$name.json')); r=d['roofline']
print('$name', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in r['breakdown_ms'].items()}, 'parity', d.get('parity',{}).get('ok'))
PY
}
run c3_base --workload c3 -- PLHIP_X=0 && run c3_var --workload c3 -- PLHIP_LIBRARY=$lib && \
run c5_base --workload c5 -- PLHIP_X=0 && run c5_var --workload c5 -- PLHIP_LIBRARY=$lib && \
run q5h_base --workload q5 -- PLHIP_NO_RANGE_DENSE=1 && run q5h_var --workload q5 -- PLHIP_NO_RANGE_DENSE=1 PLHIP_LIBRARY=$lib && \
run q3x8_base --workload q3 --rows 100000000 -- PLHIP_X=0 && run q3x8_var --workload q3 --rows 100000000 -- PLHIP_LIBRARY=$lib
