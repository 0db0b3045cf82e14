#!/bin/bash
# kept lists (AMC_LIST_KEEP=K): the whole parity suite with K = 8, soaks, then timing against K = 0
set -o pipefail
mkdir -p gpurun_out/keep
AMC_LIST_KEEP=8 timeout -k 10 700 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/keep/tests_k8.log 2>&1 || { echo "suite K=8 failed"; grep -n "FAILED\|Error\|error" gpurun_out/keep/tests_k8.log | head; tail -15 gpurun_out/keep/tests_k8.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/keep/tests_k8.log
AMC_LIST_KEEP=8 timeout -k 10 400 python tests/soak.py pore_1e6 300 100 > gpurun_out/keep/soak_pore_1e6_k8.json 2> gpurun_out/keep/soak.err || { echo soak failed; tail -5 gpurun_out/keep/soak.err; exit 1; }
AMC_LIST_KEEP=3 timeout -k 10 300 python tests/soak.py cube_1e5 1000 500 > gpurun_out/keep/soak_cube_1e5_k3.json 2> gpurun_out/keep/soak.err || { echo soak2 failed; tail -5 gpurun_out/keep/soak.err; exit 1; }
grep -h -o '"hist[^,]*' gpurun_out/keep/soak_*.json | sort | uniq -c
for rep in 1 2; do for K in 0 4 8 16; do for w in pore_1e6 pore_5e5 cube_1e6 cube_1e5; do
  AMC_LIST_KEEP=$K timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/keep/b_${w}_K${K}_$rep.json 2> gpurun_out/keep/err || { echo "bench failed $w $K"; tail -3 gpurun_out/keep/err; exit 1; }
done; done; done
python - <<'PY'
import json,glob
for w in ('pore_1e6','pore_5e5','cube_1e6','cube_1e5'):
    for K in (0,4,8,16):
        out=[]
        for rep in (1,2):
            d=json.load(open('gpurun_out/keep/b_%s_K%d_%d.json'%(w,K,rep))); r=d['roofline']['per_kernel_avg_us']
            out.append('%.1f (s %.1f d %.1f c %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0)))
        print(w,'K=%d'%K,' | '.join(out))
PY
