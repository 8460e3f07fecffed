#!/bin/bash
# q3 shape beyond one partition level: AUTO (two levels) against the sort strategy
for cfg in "q3 300000000 auto" "q3 300000000 sort" "q5 200000000 auto" "q5 200000000 sort"; do
  set -- $cfg
  timeout -k 10 280 python bench.py --workload $1 --rows $2 --strategy $3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('$1 rows $2', d['config']['strategy'], d['config']['groups'], 'ms/step %.3f'%d['ms_per_step'], r['breakdown_ms'], 'allfrac %.3f'%r['all_kernels_frac'])" || exit 1
done
