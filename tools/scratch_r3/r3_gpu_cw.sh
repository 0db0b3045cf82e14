#!/bin/bash
# the wide cluster kernel with more waves at N = 1e6 (2,000+ candidates per sweep: 4-5 per wave at 512 waves)
mkdir -p gpurun_out/cw
for w in pore_1e6 cube_1e6 cube_1e5; do for nb in 512 1024 2048 4096; do
  AMC_CW_BLOCKS=$nb timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/cw/${w}_$nb.json 2> gpurun_out/cw/err || { echo "failed $w $nb"; tail -3 gpurun_out/cw/err; continue; }
  python - <<PY
import json
d=json.load(open('gpurun_out/cw/${w}_$nb.json')); r=d['roofline']['per_kernel_avg_us']
print('$w waves=$nb','%.1f (s %.1f d %.1f c %.1f r %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0), r.get('resolve',0)))
PY
done; done
