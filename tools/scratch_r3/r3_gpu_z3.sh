#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/keep
run() { # K workload
  AMC_LIST_KEEP=$1 timeout -k 10 100 python bench.py --workload $2 --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/keep/x_$2_K$1_$3.json 2> gpurun_out/keep/err || { echo "bench failed $1 $2"; tail -3 gpurun_out/keep/err; }
  python - <<PY
import json
d=json.load(open('gpurun_out/keep/x_$2_K$1_$3.json')); r=d['roofline']['per_kernel_avg_us']
print('$2 K=$1','%.1f (s %.1f d %.1f c %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0)))
PY
}
for rep in 1 2; do for K in 4 5 6 8; do run $K pore_1e6 $rep; done; done
for K in 4 6 8; do run $K pore_5e5 1; done
