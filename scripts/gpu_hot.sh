#!/bin/bash
# hot-leaf splitting: parity tests, then the skewed profile at two sizes and a q3 regression line
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 120 --timeout-method thread -k "two_level or skew or hot" > gpurun_out/hot_tests.log 2>&1 || { tail -30 gpurun_out/hot_tests.log; exit 1; }
tail -3 gpurun_out/hot_tests.log
timeout -k 10 200 python scripts/skew_profile.py 1e8 1e7 && \
timeout -k 10 300 python scripts/skew_profile.py 1e9 1e8 && \
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline && \
timeout -k 10 200 python bench.py --workload q3 --rows 300000000 --strategy auto --steps 5 --warmup 2 --no-cpu-baseline
