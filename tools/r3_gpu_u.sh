#!/bin/bash
# where does the overlapped run (AMC_OVERLAP=1) win?  plain vs overlapped over sizes, both geometries
mkdir -p gpurun_out/ovl
for w in cube pore; do
for n in 300000 500000 1000000 2000000 4000000; do
  base=${w}_1e6
  for m in 0 1; do
    AMC_OVERLAP=$m timeout -k 10 120 python bench.py --workload $base --n $n --steps 500 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/ovl/${w}_${n}_$m.json 2> gpurun_out/ovl/err || { echo "failed $w $n $m"; tail -3 gpurun_out/ovl/err; exit 1; }
  done
done; done
python - <<'PY'
import json,glob
for w in ('cube','pore'):
    for n in (300000,500000,1000000,2000000,4000000):
        a=json.load(open('gpurun_out/ovl/%s_%d_0.json'%(w,n)))['ms_per_step']*1e3
        b=json.load(open('gpurun_out/ovl/%s_%d_1.json'%(w,n)))['ms_per_step']*1e3
        print(w,n,'plain %.1f overlapped %.1f  %+.1f %%'%(a,b,(b/a-1)*100))
PY
