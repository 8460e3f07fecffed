#!/bin/bash
timeout -k 10 500 python bench.py --workload c5 --steps 5 --warmup 1 --configs none > gpurun_out/c5.json 2> gpurun_out/c5.err; rc=$?
[ $rc -eq 0 ] || { tail -5 gpurun_out/c5.err; exit $rc; }
python -c "
import json; d=json.loads(open('gpurun_out/c5.json').read().strip().splitlines()[-1]); print('c5', d['ms_per_step'], d['config']['strategy'], d.get('parity'), d['roofline']['breakdown_ms'])"
bash scripts/gpu_tests.sh 1000
