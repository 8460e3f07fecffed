#!/bin/bash
timeout -k 10 300 python -u scripts/join_profile.py 2>&1 | grep -v amdgpu.ids | tail -6
