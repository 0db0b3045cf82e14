#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for w in cube_1e5 pore_1e6 cube_1e6; do
  timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/k_$w.json 2> $O/k_$w.err || { echo "bench $w failed"; exit 1; }
done
AMC_DEBUG_RESOLVE=1 timeout -k 10 100 python bench.py --workload cube_1e5 --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/kdbg_cube_1e5.json 2> $O/kdbg_cube_1e5.err
AMC_DEBUG_RESOLVE=1 timeout -k 10 100 python bench.py --workload pore_1e6 --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/kdbg_pore_1e6.json 2> $O/kdbg_pore_1e6.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/k_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
PY
grep -h "amc k_clusters_wide pair \|3-cluster  \|pair+again" $O/kdbg_cube_1e5.err $O/kdbg_pore_1e6.err | cut -c1-420
