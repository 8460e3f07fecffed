#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_edges_gpu.py tests/test_dense_gpu.py tests/test_parity_gpu.py -x -q --timeout 300 --timeout-method thread -k "pinned_word or dense or db_benchmark or sorted or edge_keys or wide or multi_key or two_levels" > gpurun_out/t_mm.log 2>&1 || { tail -40 gpurun_out/t_mm.log; exit 1; }
tail -2 gpurun_out/t_mm.log
for i in 1 2; do for wl in q5 q3; do
timeout -k 10 300 python bench.py --gpus 1 --workload $wl --steps 300 --warmup 20 --configs none --no-cpu-baseline 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl', d['config']['strategy'], 'step_ms', round(d['ms_per_step'],4), 'kernels', round(r['all_kernels_ms'],4))"
done; done
PLHIP_MINMAX_COPY=1 timeout -k 10 300 python bench.py --gpus 1 --workload q5 --steps 300 --warmup 20 --configs none --no-cpu-baseline 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('q5 copy path', 'step_ms', round(d['ms_per_step'],4))"
