#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "energised or temp or checkpoint" > $O/temp_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/temp_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python bench.py --workload temp_1e6 --steps 200 --warmup 20 --no-cpu-baseline > $O/p_temp.json 2> $O/p_temp.err || { echo "bench failed"; tail -5 $O/p_temp.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/p_temp.json')); print('temp_1e6 us/step', round(d['ms_per_step']*1e3,1), f"{d['value']:.3g}")
PY
timeout -k 10 300 python tests/soak.py temp_1e6 100 50 > gpurun_out/soak_t6.log 2>&1; tail -1 gpurun_out/soak_t6.log | cut -c1-400
