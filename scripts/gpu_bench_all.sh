#!/bin/bash
# every bench workload once on the box: scripts/gpu_bench_all.sh <tag> [workloads...]
tag=${1:-r2}; shift
out=gpurun_out/$tag; mkdir -p $out
wls=${@:-"q3 q1 q5 q3na c3 c5 q3x"}
for wl in $wls; do
  case $wl in
    q3na) args="--workload q3 --na";;
    q3x) export PLHIP_BENCH_FORCE_EXCHANGE=1; args="--workload q3";;
    *) args="--workload $wl";;
  esac
  timeout -k 10 600 python bench.py $args > $out/bench_$wl.json 2> $out/bench_$wl.err || { echo "$wl FAILED"; tail -5 $out/bench_$wl.err; }
  unset PLHIP_BENCH_FORCE_EXCHANGE
  python - <<PY
import json
try:
    d=json.load(open('$out/bench_$wl.json')); r=d['roofline']
    print('$wl', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],4), 'rows/s %.3g'%d['value'], {k: round(v,4) for k,v in r['breakdown_ms'].items()}, 'all_frac', round(r['all_kernels_frac'],3), 'parity', d.get('parity',{}).get('ok'), 'cpu %.3g'%d.get('cpu_baseline',{}).get('value',0), d.get('sharded_step_ms',''))
except Exception as e:
    print('$wl', 'no line', e)
PY
done
