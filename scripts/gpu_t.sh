#!/bin/bash
# scratch: the deferred final wait -- its tests, then the step time with and without it
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_deferred_sync_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_defer.log 2>&1 || { tail -30 gpurun_out/t_defer.log; exit 1; }
tail -3 gpurun_out/t_defer.log
for nd in 1 0 1 0; do for wl in q3 q1; do
  if [ $nd = 1 ]; then export PLHIP_NO_DEFERRED_SYNC=1; else unset PLHIP_NO_DEFERRED_SYNC; fi
  timeout -k 10 120 python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --configs none 2>gpurun_out/t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('no_defer',$nd,'$wl','step_ms',round(d['ms_per_step'],4),'kernels',round(r['all_kernels_ms'],4), d.get('parity'))" || { tail -5 gpurun_out/t.err; exit 1; }
done; done
