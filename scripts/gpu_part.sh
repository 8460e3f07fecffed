#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_dense_gpu.py tests/test_parity_gpu.py -x -q -k "dense or partition or two_level or leaf or tilesort2 or skew or config3 or strategies" --timeout 600 --timeout-method thread > gpurun_out/part_tests.log 2>&1 || { tail -40 gpurun_out/part_tests.log; exit 1; }
tail -1 gpurun_out/part_tests.log
for args in "--rows 1000000000" "--workload c5" "--workload c3" "--rows 300000000"; do
  timeout -k 10 400 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --configs none 2>gpurun_out/part.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$args]',d['config']['strategy'],'ms',round(d['ms_per_step'],3),'kernels',round(r['all_kernels_ms'],3),{k:round(v,3) for k,v in r['breakdown_ms'].items()},'step_frac',round(r['step_frac'],3))" || { tail -5 gpurun_out/part.err; exit 1; }
done
