#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -40 gpurun_out/full_tests.log; exit 1; }
tail -3 gpurun_out/full_tests.log
STRATS=auto timeout -k 10 200 python scripts/skew_profile.py 1e8 1e7 && \
timeout -k 10 200 python bench.py --workload q3 --rows 100000000 --steps 20 --warmup 3 --no-cpu-baseline | cut -c1-200 && \
timeout -k 10 200 python bench.py --workload q5 --no-hint --steps 100 --warmup 10 --no-cpu-baseline | cut -c1-200 && \
timeout -k 10 200 python bench.py --workload q1 --steps 100 --warmup 10 --no-cpu-baseline | cut -c1-200
