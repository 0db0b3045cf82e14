#!/bin/bash
# kept lists: where does the detect kernel's step of ~21 us come from?  J = blocks per bank for the extra nodes (0: none, timing only)
mkdir -p gpurun_out/keep
cp argon_monte_carlo_amd/libargonmc.so libamc_keepme.tmp.so
run() { # variant K J
  cp libamc_$1.tmp.so argon_monte_carlo_amd/libargonmc.so
  AMC_LIST_KEEP=$2 AMC_KEEP_DET_J=$3 timeout -k 10 100 python bench.py --workload pore_1e6 --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/keep/z_$1_K$2_J$3.json 2> gpurun_out/keep/err || { echo "bench failed $1 $2 $3"; tail -3 gpurun_out/keep/err; }
  python - <<PY
import json
d=json.load(open('gpurun_out/keep/z_$1_K$2_J$3.json')); r=d['roofline']['per_kernel_avg_us']
print('$1 K=$2 J=$3','%.1f (s %.1f d %.1f c %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0)))
PY
}
run base 0 8; run base 8 8; run base 8 1; run base 8 32; run base 8 0; run base 2 0; run fullrec 8 8; run fullrec 2 8
cp libamc_keepme.tmp.so argon_monte_carlo_amd/libargonmc.so
