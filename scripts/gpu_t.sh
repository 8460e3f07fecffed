#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_io_gpu.py tests/test_chunked_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/t_io.log 2>&1 || { tail -50 gpurun_out/t_io.log; exit 1; }
tail -2 gpurun_out/t_io.log
