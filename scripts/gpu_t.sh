#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_dense_gpu.py -x -q --timeout 500 --timeout-method thread -k "config4" > gpurun_out/t_c4.log 2>&1 || { tail -30 gpurun_out/t_c4.log; exit 1; }
tail -2 gpurun_out/t_c4.log
export PLHIP_BENCH_SHARE_GPU=1
for nd in 0 1; do
  if [ $nd = 1 ]; then export PLHIP_NO_DENSE_PARTITION=1; else unset PLHIP_NO_DENSE_PARTITION; fi
  timeout -k 10 500 python bench.py --gpus 2 --total-rows 250000000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/t_c4_$nd.json 2>gpurun_out/t.err || { tail -5 gpurun_out/t.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/t_c4_$nd.json').read().strip().splitlines()[-1])
print('no_part=$nd', 'n_gpus', d['n_gpus'], 'rows/gpu', d['config']['rows_per_gpu'], 'groups', d['config']['groups'], d['config']['strategy'], 'ms/step', round(d['ms_per_step'],3), 'shuffle', d['shuffle'], 'parity', d['parity']['ok'])"
done
