#!/bin/bash
# round 2's closing profile set: scripts/gpu_final_r2e.sh  -> gpurun_out/r2e_*
for wl in q3 q3na q1 q5; do
  case $wl in q3na) args="--workload q3 --na";; *) args="--workload $wl";; esac
  echo "== $wl"
  bash scripts/gpu_prof2.sh r2e_$wl $args || exit 1
  timeout -k 10 300 python bench.py $args > gpurun_out/r2e_${wl}_bench.json 2> gpurun_out/r2e_${wl}_bench.err || exit 1
done
for wl in c3 c5; do
  echo "== $wl"
  timeout -k 10 500 python bench.py --workload $wl > gpurun_out/r2e_${wl}_bench.json 2> gpurun_out/r2e_${wl}_bench.err || exit 1
done
PLHIP_BENCH_FORCE_EXCHANGE=1 timeout -k 10 300 python bench.py > gpurun_out/r2e_q3x_bench.json 2> gpurun_out/r2e_q3x_bench.err || exit 1
for f in gpurun_out/r2e_*_bench.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], d['config']['strategy'], 'ms', round(d['ms_per_step'],4), 'dom', r['kernel'], round(r['frac'],3), 'all', round(r['all_kernels_frac'],3), 'traffic x', r.get('traffic_over_algorithmic'), 'parity', d.get('parity',{}).get('ok'), 'cpu', d.get('cpu_baseline',{}).get('value'))
"; done
