#!/bin/bash
# GPU test suite with a verbose log kept under gpurun_out/ (a native crash otherwise leaves no trace of the test it hit)
mkdir -p gpurun_out
timeout -k 10 ${1:-700} python -X faulthandler -m pytest tests -m gpu -x -v > gpurun_out/gpu_tests.log 2>&1
rc=$?
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -2
if [ $rc -ne 0 ]; then grep -n "Fatal\|fault\|FAILED\|Error" gpurun_out/gpu_tests.log | head -20; tail -5 gpurun_out/gpu_tests.log | cut -c1-300; fi
exit $rc
