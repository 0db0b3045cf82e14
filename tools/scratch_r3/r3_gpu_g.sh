#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for w in cube_1e5 pore_1e6; do
for sb in 64 128 256; do
  AMC_STREAM_BS=$sb timeout -k 10 60 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/bs_s${sb}_$w.json 2> $O/bs.err || { echo "bench failed"; tail -3 $O/bs.err; exit 1; }
  AMC_DETECT_BS=$sb timeout -k 10 60 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/bs_d${sb}_$w.json 2> $O/bs.err || { echo "bench failed"; tail -3 $O/bs.err; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/bs_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
PY
