#!/bin/bash
# final build: GPU suite, then the soaks / long runs of record
set -o pipefail
bash tools/run_gpu_tests.sh 700 || exit 1
rm -f gpurun_out/soak_*.json gpurun_out/long_*.json
S="timeout -k 10 300 python tests/soak.py"
$S cube_1e5 3000 500 > gpurun_out/s1.log 2>&1 || { echo s1 failed; tail -3 gpurun_out/s1.log; exit 1; }
$S cube_1e5 1000 500 --cw-blocks 8 > gpurun_out/s2.log 2>&1 || { echo s2 failed; tail -3 gpurun_out/s2.log; exit 1; }
$S pore_1e6 300 100 > gpurun_out/s3.log 2>&1 || { echo s3 failed; tail -3 gpurun_out/s3.log; exit 1; }
$S pore_1e6 100 50 --cw-blocks 8 > gpurun_out/s4.log 2>&1 || { echo s4 failed; tail -3 gpurun_out/s4.log; exit 1; }
$S temp_1e6 100 50 > gpurun_out/s5.log 2>&1 || { echo s5 failed; tail -3 gpurun_out/s5.log; exit 1; }
timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/l1.log 2>&1 || { echo l1 failed; tail -3 gpurun_out/l1.log; exit 1; }
AMC_OVERLAP=1 timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/l2.log 2>&1 || { echo l2 failed; tail -3 gpurun_out/l2.log; exit 1; }
AMC_LIST_KEEP=8 timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/l3.log 2>&1 || { echo l3 failed; tail -3 gpurun_out/l3.log; exit 1; }
cp gpurun_out/l3.log gpurun_out/long_pore_1e6_10000_keep8.json
ls gpurun_out/soak_*.json gpurun_out/long_*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/long_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, {k:d[k] for k in d if 'sha' in k.lower() or k in ('seconds','particle_steps_per_s')})
PY
