#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
PLHIP_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --configs none --no-cpu-baseline 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('default-20', 'step_ms', round(d['ms_per_step'],4), 'kernels', round(r['all_kernels_ms'],4), 'dom', r['kernel'], round(r['kernel_ms'],4), 'parity', (d.get('parity') or {}).get('ok'))" || { tail -5 gpurun_out/t.err; exit 1; }
grep bench gpurun_out/t.err
done
for wl in q3 q1; do
timeout -k 10 300 python bench.py --gpus 1 --workload $wl --steps 300 --warmup 20 --configs none --no-cpu-baseline 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl steps-300', 'step_ms', round(d['ms_per_step'],4), 'kernels', round(r['all_kernels_ms'],4), 'dom', round(r['kernel_ms'],4))"
done
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py tests/test_sharded_gpu.py tests/test_edges_gpu.py -x -q --timeout 300 --timeout-method thread -k "timing or kernel_ms or sharded or threads" > gpurun_out/t_small.log 2>&1 || { tail -30 gpurun_out/t_small.log; exit 1; }
tail -2 gpurun_out/t_small.log
