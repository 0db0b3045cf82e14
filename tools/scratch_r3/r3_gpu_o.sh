#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/long.log 2>&1; tail -1 gpurun_out/long.log | cut -c1-600
AMC_OVERLAP=1 timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/long_ov.log 2>&1; tail -1 gpurun_out/long_ov.log | cut -c1-600
AMC_OVERLAP=1 timeout -k 10 300 python tests/soak.py pore_1e6 200 100 > gpurun_out/soak_p6ov.log 2>&1; tail -1 gpurun_out/soak_p6ov.log | cut -c1-330
