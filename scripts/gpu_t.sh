#!/bin/bash
timeout -k 10 400 python -u -m pytest tests/test_edges_gpu.py -x -q --timeout 300 --timeout-method thread -k "pinned_buffers or release_workspace" 2>&1 | grep -v amdgpu.ids | tail -30
