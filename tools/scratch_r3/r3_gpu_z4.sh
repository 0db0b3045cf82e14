#!/bin/bash
# kept lists on by default in the pore (K = 4): whole suite, soaks, long run
set -o pipefail
bash tools/run_gpu_tests.sh 700 || exit 1
rm -f gpurun_out/soak_*.json gpurun_out/long_*.json
S="timeout -k 10 300 python tests/soak.py"
$S pore_1e6 300 100 > gpurun_out/s3.log 2>&1 || { echo s3 failed; tail -3 gpurun_out/s3.log; exit 1; }
$S pore_1e6 100 50 --cw-blocks 8 > gpurun_out/s4.log 2>&1 || { echo s4 failed; tail -3 gpurun_out/s4.log; exit 1; }
$S pore_5e5 300 100 > gpurun_out/s6.log 2>&1 || { echo s6 failed; tail -3 gpurun_out/s6.log; exit 1; }
timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/l1.log 2>&1 || { echo l1 failed; tail -3 gpurun_out/l1.log; exit 1; }
AMC_LIST_KEEP=0 timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/l0.log 2>&1 || { echo l0 failed; tail -3 gpurun_out/l0.log; exit 1; }
python - <<'PY'
import json,glob
for f in ('gpurun_out/l1.log','gpurun_out/l0.log'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['final_state_sha256'][:16], d['histograms_sha256'][:16], '%.3g'%d['particle_steps_per_s'])
for f in sorted(glob.glob('gpurun_out/soak_*.json')):
    d=json.load(open(f)); print(f, d['steps'], d['counters_equal'], d['histograms_equal_np_histogram_of_oracle_paths'])
PY
