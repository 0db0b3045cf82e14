#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/keep
AMC_LIST_KEEP=8 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "free_run or high_collision or dense_cluster or natural" > gpurun_out/keep/tests2_k8.log 2>&1 || { echo "tests K=8 failed"; tail -15 gpurun_out/keep/tests2_k8.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/keep/tests2_k8.log
for rep in 1 2; do for K in 0 2 8; do for w in pore_1e6 pore_5e5; do
  AMC_LIST_KEEP=$K timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/keep/c_${w}_K${K}_$rep.json 2> gpurun_out/keep/err || { echo "bench failed $w $K"; tail -3 gpurun_out/keep/err; exit 1; }
done; done; done
python - <<'PY'
import json,glob
for w in ('pore_1e6','pore_5e5'):
    for K in (0,2,8):
        out=[]
        for rep in (1,2):
            d=json.load(open('gpurun_out/keep/c_%s_K%d_%d.json'%(w,K,rep))); r=d['roofline']['per_kernel_avg_us']
            out.append('%.1f (s %.1f d %.1f c %.1f)'%(d['ms_per_step']*1e3, r.get('drift_walls',0), r.get('detect',0), r.get('clusters_wide',0)))
        print(w,'K=%d'%K,' | '.join(out))
PY
