#!/bin/bash
# round-3 scratch: phase breakdown of the resolve kernels at HEAD
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for w in cube_1e5 pore_1e6 cube_1e6; do
  timeout -k 10 200 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline > $O/base_$w.json 2> $O/base_$w.err || echo "bench $w failed"
  AMC_DEBUG_RESOLVE=1 timeout -k 10 200 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline > $O/dbg_$w.json 2> $O/dbg_$w.err || echo "dbg $w failed"
done
hipcc -O3 --offload-arch=gfx950 tools/ubench_icache.hip -o /tmp/ubench_icache && timeout -k 5 60 /tmp/ubench_icache > $O/icache.txt 2>&1
grep -h "amc k_" $O/dbg_*.err | cut -c1-900
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/base_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['roofline']['per_kernel_avg_us'])
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/icache.txt
