#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
R=$(pwd)
for m in 2 1; do
for w in cube_1e5 pore_1e6 cube_1e6; do
  AMC_OVERLAP=$m timeout -k 10 60 python bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/ovl${m}_$w.json 2> $O/ovl${m}_$w.err || { echo "bench $m $w failed"; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/ovl?_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
AMC_OVERLAP=1 timeout -k 10 120 rocprofv3 --kernel-trace -d $R/$O/kt_ovl -o out --output-format csv -- python $R/bench.py --workload cube_1e6 --steps 30 --warmup 5 --no-cpu-baseline --extra-workloads none > $R/$O/kt_ovl.log 2>&1 || echo "kt failed"
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3/kt_ovl/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=None
sel=[r for r in rows if 'k_' in r['Kernel_Name']]
sel=sel[-150:-110]
for r in sel:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:40]}")
PY
