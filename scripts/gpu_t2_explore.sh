#!/bin/bash
# development aid: the two-level pipeline's leaf-table budget (KB), block size and level-1 bits on one workload
wl=$1; rows=$2; shift 2
for cfg in "$@"; do
  IFS=, read kb blk p1 <<< "$cfg"
  PLHIP_T2_VERBOSE=1 PLHIP_T2_KB=$kb PLHIP_T2_BLOCK=$blk PLHIP_T2_PBITS1=$p1 timeout -k 10 200 python bench.py --workload $wl --rows $rows --strategy auto --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/t2.err | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('$wl $rows kb=$kb blk=$blk p1=$p1', d['config']['strategy'], 'ms/step %.3f'%d['ms_per_step'], {k: round(v,2) for k,v in r['breakdown_ms'].items()})" || exit 1
  grep tilesort2 gpurun_out/t2.err | tail -1
done
