#!/bin/bash
# scratch: the sharded dense path in partition mode
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_sharded_gpu.py tests/test_bench_launcher_gpu.py tests/test_dense_gpu.py -x -q --timeout 400 --timeout-method thread > gpurun_out/t_shard.log 2>&1 || { tail -40 gpurun_out/t_shard.log; exit 1; }
tail -3 gpurun_out/t_shard.log
