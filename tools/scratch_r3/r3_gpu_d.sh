#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/ubench_xstream.hip -o /tmp/ubench_xstream 2>/dev/null && timeout -k 5 100 /tmp/ubench_xstream > $O/xstream.txt 2>&1; cat $O/xstream.txt
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped" > $O/ovl_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/ovl_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for m in 1; do
for w in pore_1e6 cube_1e6; do
  AMC_OVERLAP=$m timeout -k 10 60 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/ovl${m}_$w.json 2> $O/ovl${m}_$w.err || { echo "bench $m $w failed"; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/ovl1_*1e6.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
