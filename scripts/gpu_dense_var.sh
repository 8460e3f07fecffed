#!/bin/bash
out=gpurun_out/${1:-var}; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline --steps 100 --workload ${WL:-q3} > $out/$name.json 2> $out/$name.err
  python -c "
import json
d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],4), {k: round(v,4) for k,v in r['breakdown_ms'].items()})"
}
for WL in q3 q1; do
export WL
run ${WL}_base PLHIP_DENSE_DBG=0
run ${WL}_noload PLHIP_DENSE_DBG=1
run ${WL}_noload_notree PLHIP_DENSE_DBG=3
done
