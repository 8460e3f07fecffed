#!/bin/bash
bash scripts/gpu_tests.sh 1000 || exit 1
bash scripts/gpu_stats.sh c3_now --workload c3 --steps 20 --warmup 3 --configs none --no-parity
