#!/bin/bash
mkdir -p gpurun_out/r3
for w in cube_1e5 pore_1e6 cube_1e6; do
  timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/r3/nt_$w.json 2> gpurun_out/r3/nt.err || { echo "bench $w failed"; tail -3 gpurun_out/r3/nt.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/nt_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
PY
