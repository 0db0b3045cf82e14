#!/bin/bash
set -u
mkdir -p gpurun_out/r3
timeout -k 10 600 python -X faulthandler -m pytest tests/test_oracle_natural.py tests/test_gpu_parity.py -m gpu -x -q -k "natural or references_own or overlay_history or candidate_overflow" > gpurun_out/r3/new_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3/new_gpu_tests.log
timeout -k 10 300 python tests/mfp_stat.py 100000 > gpurun_out/r3/mfp_stat.json 2> gpurun_out/r3/mfp_stat.err; echo "mfp rc=$?"; cat gpurun_out/r3/mfp_stat.json | head -40
