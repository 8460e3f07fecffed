#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/diag.log
for d in 0 1 2 3 4 6 7; do
  for wl in q1 q3; do
  echo "== diag $d $wl" >> gpurun_out/diag.log
  PLHIP_DIAG=$d timeout -k 10 120 python bench.py --workload $wl --strategy auto --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print(d['config']['strategy'], 'dom %s %.3f ms'%(r['kernel'][:24],r['kernel_ms']),'all %.3f ms'%r['all_kernels_ms'])
" >> gpurun_out/diag.log
  done
done
cat gpurun_out/diag.log
