#!/bin/bash
# quick look at strategies on q1/q3/q5 (development aid): prints strategy, kernel ms, roofline fraction
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
for cfg in "q1 lds" "q1 auto" "q3 partitioned" "q3 auto" "q5 auto" ${EXTRA}; do
  set -- $cfg
  echo "== $1 $2" >> gpurun_out/sweep.log
  timeout -k 10 120 python bench.py --workload $1 --strategy $2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print(d['config']['strategy'], 'groups',d['config']['groups'],'ms/step %.3f'%d['ms_per_step'],'dom %s %.3f ms'%(r['kernel'][:24],r['kernel_ms']),'all %.3f ms'%r['all_kernels_ms'],'frac %.3f'%r['frac'],'allfrac %.3f'%r['all_kernels_frac'], 'rows/s %.3g'%d['value'])
" >> gpurun_out/sweep.log || exit 1
done
cat gpurun_out/sweep.log
