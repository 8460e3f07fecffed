#!/bin/bash
# config 3 (1e8 rows, three i64 keys, ~1e8 groups) with AUTO (two partition levels) and with the sort strategy
for st in auto sort; do
  timeout -k 10 200 python bench.py --workload c3 --strategy $st --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['strategy'], d['config']['groups'], 'ms/step %.3f'%d['ms_per_step'], r['breakdown_ms'], 'allfrac %.3f'%r['all_kernels_frac'])" || exit 1
done
