#!/bin/bash
out=gpurun_out/${1:-var}; mkdir -p $out
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --workload c3 > $out/$name.json 2> $out/$name.err
  python -c "
import json
d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in r['breakdown_ms'].items()})"
}
run base PLHIP_LEAF_DBG=0
run nosort PLHIP_LEAF_DBG=1
run nofold PLHIP_LEAF_DBG=2
run noemit PLHIP_LEAF_DBG=4
run nosort_nofold PLHIP_LEAF_DBG=3
