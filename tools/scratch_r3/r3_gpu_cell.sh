#!/bin/bash
# cell edge of the detection grid (in mean particle spacings) with kept lists: fewer movers with larger cells
mkdir -p gpurun_out/cell
for w in pore_1e6 pore_5e5; do for sc in 0.5 0.63 0.75 0.9 1.1; do for K in 4 8; do
  AMC_CELL_SCALE=$sc AMC_LIST_KEEP=$K timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/cell/${w}_${sc}_$K.json 2> gpurun_out/cell/err || { echo "failed $w $sc $K"; tail -3 gpurun_out/cell/err; continue; }
  python - <<PY
import json
d=json.load(open('gpurun_out/cell/${w}_${sc}_$K.json')); r=d['roofline']['per_kernel_avg_us']
print('$w scale=$sc K=$K','%.1f (s %.1f d %.1f c %.1f r %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0), r.get('resolve',0)))
PY
done; done; done
