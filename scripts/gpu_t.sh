#!/bin/bash
PLHIP_DENSE_VERBOSE=1 timeout -k 10 200 python -u -m pytest tests/test_edges_gpu.py -x -q -s --timeout 100 --timeout-method thread -k "pinned_word" 2>&1 | grep -v amdgpu.ids | tail -30
