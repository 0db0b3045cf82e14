#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
AMC_OVERLAP_SPLIT=1 timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped" > $O/ovl_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/ovl_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for sp in 0 1; do
for w in pore_1e6 cube_1e6; do
  AMC_OVERLAP=1 AMC_OVERLAP_SPLIT=$sp timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/m${sp}_$w.json 2> $O/m_$w.err || { echo "bench $w failed"; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/m?_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
PY
