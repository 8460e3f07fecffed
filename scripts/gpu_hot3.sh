#!/bin/bash
b() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --workload $W --rows $R --strategy auto --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['strategy'], d['config']['groups'], 'ms/step %.3f'%d['ms_per_step'], r['breakdown_ms'])"; }
W=q3 R=300000000
b PLHIP_T2_KB=32 || exit 1
b PLHIP_T2_KB=76 || exit 1
W=q5 R=200000000
b PLHIP_T2_KB=32 || exit 1
b PLHIP_T2_KB=76 || exit 1
W=q3 R=100000000
b PLHIP_T2_KB=32 || exit 1
b PLHIP_T2_KB=76 || exit 1
