#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pore or temp or energised or dense or free_run or config" > $O/pore_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pore_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for w in pore_1e6 pore_5e5; do
  timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/l_$w.json 2> $O/l_$w.err || { echo "bench $w failed"; exit 1; }
done
AMC_DEBUG_RESOLVE=1 timeout -k 10 100 python bench.py --workload pore_1e6 --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/ldbg_pore_1e6.json 2> $O/ldbg_pore_1e6.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/l_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
PY
grep -h "amc k_clusters_wide pair \|3-cluster  \|pair+again" $O/ldbg_pore_1e6.err | cut -c1-420
