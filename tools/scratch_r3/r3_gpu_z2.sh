#!/bin/bash
# detect: particles per thread (AMC_DETECT_PPT) and kept lists with the extra nodes walked by the normal blocks
set -o pipefail
mkdir -p gpurun_out/keep
AMC_LIST_KEEP=8 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "free_run or high_collision or dense_cluster or natural or kept" > gpurun_out/keep/tests3_k8.log 2>&1 || { echo "tests K=8 failed"; tail -15 gpurun_out/keep/tests3_k8.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/keep/tests3_k8.log
AMC_DETECT_PPT=2 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "free_run or high_collision or dense_cluster or tiny" > gpurun_out/keep/tests3_ppt2.log 2>&1 || { echo "tests PPT=2 failed"; tail -15 gpurun_out/keep/tests3_ppt2.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/keep/tests3_ppt2.log
run() { # K PPT workload
  AMC_LIST_KEEP=$1 AMC_DETECT_PPT=$2 timeout -k 10 100 python bench.py --workload $3 --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/keep/y_$3_K$1_P$2.json 2> gpurun_out/keep/err || { echo "bench failed $1 $2 $3"; tail -3 gpurun_out/keep/err; }
  python - <<PY
import json
d=json.load(open('gpurun_out/keep/y_$3_K$1_P$2.json')); r=d['roofline']['per_kernel_avg_us']
print('$3 K=$1 PPT=$2','%.1f (s %.1f d %.1f c %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0)))
PY
}
for w in pore_1e6 cube_1e6 cube_1e5; do for P in 1 2 4; do run 0 $P $w; done; done
for K in 2 4 8; do run $K 1 pore_1e6; done
run 8 2 pore_1e6; run 4 1 pore_5e5; run 8 1 pore_5e5; run 2 1 cube_1e6
