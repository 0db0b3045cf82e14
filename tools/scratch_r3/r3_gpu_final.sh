#!/bin/bash
# the round's last GPU call: suite, pore soaks, long run, then the profile collection of record
set -o pipefail
bash tools/run_gpu_tests.sh 700 || exit 1
rm -f gpurun_out/soak_*.json gpurun_out/long_*.json
S="timeout -k 10 300 python tests/soak.py"
$S pore_1e6 300 100 > gpurun_out/s3.log 2>&1 || { echo s3 failed; tail -3 gpurun_out/s3.log; exit 1; }
$S pore_1e6 100 50 --cw-blocks 8 > gpurun_out/s4.log 2>&1 || { echo s4 failed; tail -3 gpurun_out/s4.log; exit 1; }
$S cube_1e5 1000 500 > gpurun_out/s1.log 2>&1 || { echo s1 failed; tail -3 gpurun_out/s1.log; exit 1; }
timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/l1.log 2>&1 || { echo l1 failed; tail -3 gpurun_out/l1.log; exit 1; }
python - <<'PY'
import json,glob
d=json.loads(open('gpurun_out/l1.log').read().strip().splitlines()[-1]); print('long', d['final_state_sha256'][:16], d['histograms_sha256'][:16], '%.3g'%d['particle_steps_per_s'])
for f in sorted(glob.glob('gpurun_out/soak_*.json')):
    d=json.load(open(f)); print(f, d['steps'], d['counters_equal'], d['histograms_equal_np_histogram_of_oracle_paths'])
PY
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; grep -i failed gpurun_out/collect.log; echo collected
