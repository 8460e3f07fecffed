#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 120 --timeout-method thread -k "sort or skew or stable or known_answers or wide or hot or long_runs" > gpurun_out/sortlong_tests.log 2>&1 || { tail -30 gpurun_out/sortlong_tests.log; exit 1; }
tail -2 gpurun_out/sortlong_tests.log
export STRATS=sort
timeout -k 10 200 python scripts/skew_small.py 1e7 0.3 0 2>/dev/null | grep "hint       0" && timeout -k 10 200 python scripts/skew_small.py 1e7 0 0.05 2>/dev/null | grep "hint       0" && timeout -k 10 200 python scripts/skew_small.py 1e8 0.05 0 2>/dev/null | grep "hint       0" && \
timeout -k 10 300 python scripts/skew_profile.py 1e9 1e8 2>/dev/null
