#!/bin/bash
timeout -k 10 300 python scripts/q6_probe.py 2>&1 | grep -v amdgpu.ids | tail -5
