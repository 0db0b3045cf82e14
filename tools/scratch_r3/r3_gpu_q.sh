#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 800 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q -k "energised or temp or checkpoint or gap_case or facade" > $O/temp_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/temp_tests.log | cut -c1-400
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python bench.py --workload temp_1e6 --steps 200 --warmup 20 --no-cpu-baseline > $O/q_temp.json 2> $O/q_temp.err || { echo "bench failed"; tail -5 $O/q_temp.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/q_temp.json')); print('temp_1e6 us/step', round(d['ms_per_step']*1e3,1), f"{d['value']:.3g}")
PY
timeout -k 10 300 python tests/soak.py temp_1e6 100 50 > gpurun_out/soak_t6.log 2>&1; tail -1 gpurun_out/soak_t6.log | cut -c1-400
