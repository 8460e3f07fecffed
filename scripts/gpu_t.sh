#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do
PLHIP_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --configs none --no-cpu-baseline 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('default-20', 'step_ms', round(d['ms_per_step'],4), 'kernels', round(r['all_kernels_ms'],4))" || { tail -5 gpurun_out/t.err; exit 1; }
grep bench gpurun_out/t.err
done
