#!/bin/bash
b() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('ms/step %.4f'%d['ms_per_step'], r['breakdown_ms'])"; }
b PLHIP_NOHOT=1 || exit 1
b PLHIP_HOT_EXTRA=0 || exit 1
b PLHIP_HOT_EXTRA=32 || exit 1
b PLHIP_HOT_EXTRA=128 || exit 1
b PLHIP_NOHOT=1 || exit 1
