#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --configs none 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['config'].get('strategy'), 'step_ms', round(d['ms_per_step'],3), 'breakdown', {k:round(v,2) for k,v in r.get('breakdown_ms').items()}, 'parity', (d.get('parity') or {}).get('ok'))" || { tail -5 gpurun_out/t.err; exit 1; }
}
run base --workload c5 --steps 4 --warmup 1 || exit 1
PLHIP_T2_SWITCH=10 run switch10 --workload c5 --steps 4 --warmup 1 || exit 1
PLHIP_T2_VERBOSE=1 timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --configs none 2>&1 >/dev/null | grep tilesort2 | head -2
PLHIP_T2_SWITCH=10 PLHIP_T2_VERBOSE=1 timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --configs none 2>&1 >/dev/null | grep tilesort2 | head -2
