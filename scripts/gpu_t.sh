#!/bin/bash
bash scripts/gpu_tests.sh 1000 || exit 1
timeout -k 10 300 python -u scripts/h2oai_harness.py --rows 10000000 > gpurun_out/harness_1e7.json 2> gpurun_out/harness_1e7.err; rc=$?
grep -v amdgpu.ids gpurun_out/harness_1e7.err | tail -12; exit $rc
