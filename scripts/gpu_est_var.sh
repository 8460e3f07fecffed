#!/bin/bash
# the estimator's second sample: scripts/gpu_est_var.sh <tag>
tag=${1:-estvar}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout -k 10 300 python bench.py "${args[@]}" --no-cpu-baseline --steps 5 --warmup 1 > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return 1; }
  python - <<PY
import json
d=json.load(open('$out/$name.json')); r=d['roofline']
print('$name', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in r['breakdown_ms'].items()}, 'groups', d['config'].get('groups'), 'parity', d.get('parity',{}).get('ok'))
PY
}
for cap in 22 21 20 19; do
  run c3_$cap --workload c3 -- PLHIP_EST_CAP_LOG2=$cap PLHIP_DEBUG_EST=1 || exit 1
  grep "estimate: round 1" $out/c3_$cap.err | tail -1
  run c5_$cap --workload c5 -- PLHIP_EST_CAP_LOG2=$cap PLHIP_DEBUG_EST=1 || exit 1
  grep "estimate: round 1" $out/c5_$cap.err | tail -1
done
