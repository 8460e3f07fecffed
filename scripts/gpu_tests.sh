#!/bin/bash
# Runs the GPU parity tests on the box as the driver does at round end; output goes to gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 ${1:-1100} python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread ${@:2} > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu.log
exit $rc
