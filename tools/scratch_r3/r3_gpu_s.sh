#!/bin/bash
mkdir -p gpurun_out/r3
for rep in 1 2 3; do
for v in plain nt; do
  cp libamc_$v.tmp.so argon_monte_carlo_amd/libargonmc.so
  for w in pore_1e6 cube_1e6; do
    timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/r3/ab_${v}_${rep}_$w.json 2> gpurun_out/r3/ab.err || { echo "bench failed"; tail -3 gpurun_out/r3/ab.err; exit 1; }
  done
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/ab_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), 'stream', round(r['per_kernel_avg_us']['drift_walls'],1))
PY
