#!/bin/bash
timeout -k 10 400 python -u -m pytest tests/test_io_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | grep -v amdgpu.ids | tail -40
