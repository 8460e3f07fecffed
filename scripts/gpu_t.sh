#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --configs none 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', d['config'].get('strategy'), 'step_ms', round(d['ms_per_step'],4), 'kernels', round(r['all_kernels_ms'],4), 'dom', r['kernel'], round(r['kernel_ms'],4), 'parity', (d.get('parity') or {}).get('ok'))" || { tail -5 gpurun_out/t.err; exit 1; }
}
# small first: every two-level / partition-mode parity test against the oracle, then the sizes that only time
timeout -k 10 900 python -u -m pytest tests/test_dense_gpu.py tests/test_parity_gpu.py -x -q --timeout 300 --timeout-method thread -k "partition or part or tilesort2 or two_levels or leaf or hot" > gpurun_out/t_small.log 2>&1 || { tail -30 gpurun_out/t_small.log; exit 1; }
tail -2 gpurun_out/t_small.log
run q3_3e8 --workload q3 --rows 300000000 --steps 10 --warmup 2 &&
run q3big --workload q3 --rows 1000000000 --steps 10 --warmup 2 &&
run c3 --workload c3 --steps 10 --warmup 2 &&
run c5 --workload c5 --steps 5 --warmup 1
