#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 120 --timeout-method thread -k "hot or skew or two_level or known_answers" > gpurun_out/hot5_tests.log 2>&1 || { tail -30 gpurun_out/hot5_tests.log; exit 1; }
tail -2 gpurun_out/hot5_tests.log
export STRATS=auto
timeout -k 10 200 python scripts/skew_small.py 1e7 0 0 && timeout -k 10 200 python scripts/skew_small.py 1e7 0.3 0 && timeout -k 10 200 python scripts/skew_small.py 1e7 0 0.05 && timeout -k 10 200 python scripts/skew_small.py 1e8 0.05 0 && \
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline
