#!/bin/bash
# dense-path round on the box: tests, bench lines, kernel stats
tag=${1:-r2b}
out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_dense_gpu.py -x -q --timeout 120 --timeout-method thread > $out/dense_tests.log 2>&1; echo rc=$? >> $out/dense_tests.log; tail -6 $out/dense_tests.log
PLHIP_DENSE_VERBOSE=1 timeout -k 10 200 python bench.py --no-cpu-baseline > $out/bench_q3.json 2> $out/bench_q3.err; tail -1 $out/bench_q3.err; python -c "
import json,sys
d=json.load(open('$out/bench_q3.json')); r=d['roofline']; print('q3', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],4), r['breakdown_ms'], 'all', round(r['all_kernels_ms'],4), round(r['all_kernels_frac'],3))"
timeout -k 10 200 python bench.py --no-cpu-baseline --workload q1 > $out/bench_q1.json 2> $out/bench_q1.err; python -c "
import json,sys
d=json.load(open('$out/bench_q1.json')); r=d['roofline']; print('q1', d['config']['strategy'], 'ms/step', round(d['ms_per_step'],4), r['breakdown_ms'], 'all', round(r['all_kernels_ms'],4), round(r['all_kernels_frac'],3))"
