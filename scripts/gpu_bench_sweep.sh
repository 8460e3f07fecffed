#!/bin/bash
# quick look at strategies on q1/q3/q5 (development aid): prints strategy, kernel ms, roofline fraction
mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
for cfg in "q1 auto" "q3 auto" "q5 auto" "q1 auto --no-hint" "q3 auto --no-hint" ${EXTRA}; do
  set -- $cfg
  echo "== $*" >> gpurun_out/sweep.log
  timeout -k 10 120 python bench.py --workload $1 --strategy $2 $3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print(d['config']['strategy'], 'groups',d['config']['groups'],'ms/step %.3f'%d['ms_per_step'],'dom %s %.3f ms'%(r['kernel'][:24],r['kernel_ms']),'all %.3f ms'%r['all_kernels_ms'], {k:round(v,4) for k,v in r['breakdown_ms'].items()},'frac %.3f'%r['frac'],'allfrac %.3f'%r['all_kernels_frac'], 'rows/s %.3g'%d['value'])
" >> gpurun_out/sweep.log || exit 1
done
cat gpurun_out/sweep.log
