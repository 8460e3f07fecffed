#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_bool_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/t_bool.log 2>&1 || { tail -40 gpurun_out/t_bool.log; exit 1; }
tail -2 gpurun_out/t_bool.log
