#!/bin/bash
timeout -k 10 400 python -u -m pytest tests/test_agg_over_groups_gpu.py tests/test_utf8_values_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | grep -v amdgpu.ids | tail -40
