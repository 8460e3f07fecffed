#!/bin/bash
mkdir -p gpurun_out
export STRATS=auto,sort
SKEW_FRAC=0 PLHIP_T2_VERBOSE=1 timeout -k 10 300 python scripts/skew_profile.py 1e9 1e8 2>&1 | tail -4 && \
SKEW_FRAC=0 PLHIP_T2_VERBOSE=1 timeout -k 10 300 python scripts/skew_profile.py 2e8 5e7 2>&1 | tail -4 && \
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -40 gpurun_out/full_tests.log; exit 1; }
tail -3 gpurun_out/full_tests.log
