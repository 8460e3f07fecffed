#!/bin/bash
timeout -k 10 300 python -u -m pytest tests/test_dense_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_tests.log 2>&1 || { tail -30 gpurun_out/t_tests.log; exit 1; }
tail -1 gpurun_out/t_tests.log
for args in "" "--na" "--workload q5" "--rows 1000000000"; do
  timeout -k 10 300 python bench.py $args --steps 60 --warmup 10 --no-cpu-baseline --configs none 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$args]','fold_ms',round(r['breakdown_ms']['fold'],4),'merge',round(r['breakdown_ms']['merge_emit'],4),'step',round(d['ms_per_step'],4))" || { tail -5 gpurun_out/t.err; exit 1; }
done
