#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --configs none 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['config'].get('strategy'), 'step_ms', round(d['ms_per_step'],4), 'kernels', round(r['all_kernels_ms'],4), 'breakdown', r.get('breakdown_ms'), 'parity', (d.get('parity') or {}).get('ok'))" || { tail -5 gpurun_out/t.err; exit 1; }
}
run 1e8_auto --workload q3 --rows 100000000 --steps 30 --warmup 3 &&
run 1e8_dense --workload q3 --rows 100000000 --steps 30 --warmup 3 --strategy dense &&
run 5e7_auto --workload q3 --rows 50000000 --steps 30 --warmup 3 &&
run 5e7_dense --workload q3 --rows 50000000 --steps 30 --warmup 3 --strategy dense &&
run 2e8_auto --workload q3 --rows 200000000 --steps 20 --warmup 3 &&
run 2e8_tilesort --workload q3 --rows 200000000 --steps 20 --warmup 3 --strategy tilesort
