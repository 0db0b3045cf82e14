#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for lds in 0 40000 54000 80000; do
for w in pore_1e6 cube_1e5; do
  AMC_DETECT_LDS=$lds timeout -k 10 100 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/n${lds}_$w.json 2> $O/n_$w.err || { echo "bench $w failed"; tail -3 $O/n_$w.err; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/n*_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), 'detect', round(r['per_kernel_avg_us']['detect'],1))
PY
