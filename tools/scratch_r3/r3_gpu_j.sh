#!/bin/bash
set -u
mkdir -p gpurun_out/prof gpurun_out/r3
OUT=gpurun_out/prof
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "sharded_by_index" > gpurun_out/r3/dist_tests2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3/dist_tests2.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for w in cube_1e6 pore_1e6; do
    AMC_OVERLAP=1 timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu-baseline --extra-workloads none > "$OUT/bench_overlap_$w.json" 2> "$OUT/bench_overlap_$w.err" || echo "bench overlap $w failed"
done
timeout -k 10 300 python bench.py --force-sharded --workload cube_1e5 --steps 1000 --warmup 50 --no-cpu-baseline > "$OUT/bench_sharded1_cube_1e5.json" 2> "$OUT/bench_sharded1_cube_1e5.err" || echo "bench sharded failed"
python - <<'PY'
import json,glob
for f in ['gpurun_out/prof/bench_overlap_cube_1e6.json','gpurun_out/prof/bench_overlap_pore_1e6.json','gpurun_out/prof/bench_sharded1_cube_1e5.json']:
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['ms_per_step']*1e3,1), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()})
PY
# soaks of record (GPU and oracle side by side), incl. the many-clusters-per-wave configuration and the overlapped run
timeout -k 10 200 python tests/soak.py cube_1e5 3000 500 > gpurun_out/soak_c5.log 2>&1; tail -1 gpurun_out/soak_c5.log | cut -c1-330
timeout -k 10 200 python tests/soak.py cube_1e5 1000 500 --cw-blocks 8 > gpurun_out/soak_c5cw.log 2>&1; tail -1 gpurun_out/soak_c5cw.log | cut -c1-330
timeout -k 10 300 python tests/soak.py pore_1e6 300 100 > gpurun_out/soak_p6.log 2>&1; tail -1 gpurun_out/soak_p6.log | cut -c1-330
timeout -k 10 200 python tests/soak.py pore_1e6 100 50 --cw-blocks 8 > gpurun_out/soak_p6cw.log 2>&1; tail -1 gpurun_out/soak_p6cw.log | cut -c1-330
AMC_OVERLAP=1 timeout -k 10 200 python tests/soak.py cube_1e6 100 50 > gpurun_out/soak_c6ov.log 2>&1; tail -1 gpurun_out/soak_c6ov.log | cut -c1-330
