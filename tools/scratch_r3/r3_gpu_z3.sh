#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/keep
AMC_LIST_KEEP=8 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "free_run or high_collision or dense_cluster or natural or kept" > gpurun_out/keep/tests4_k8.log 2>&1 || { echo "tests K=8 failed"; tail -15 gpurun_out/keep/tests4_k8.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/keep/tests4_k8.log
run() { # K workload
  AMC_LIST_KEEP=$1 timeout -k 10 100 python bench.py --workload $2 --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/keep/x_$2_K$1_$3.json 2> gpurun_out/keep/err || { echo "bench failed $1 $2"; tail -3 gpurun_out/keep/err; }
  python - <<PY
import json
d=json.load(open('gpurun_out/keep/x_$2_K$1_$3.json')); r=d['roofline']['per_kernel_avg_us']
print('$2 K=$1','%.1f (s %.1f d %.1f c %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0)))
PY
}
for rep in 1 2; do for K in 0 3 4 6 8 12; do run $K pore_1e6 $rep; done; done
for K in 0 4 8; do run $K pore_5e5 1; done; for K in 0 2 3; do run $K cube_1e6 1; done; run 0 cube_1e5 1; run 2 cube_1e5 1
