#!/bin/bash
# Measurement pass on the GPU box (run from the repo root through gpurun):
#   bench JSON lines for every workload, rocprofv3 kernel-trace statistics and FETCH_SIZE / WRITE_SIZE counter passes
#   (separate runs, as MI355X_MICROARCH.md prescribes).  Raw output under gpurun_out/prof/; tools/summarise_profiles.py
#   turns it into the committed files under profiles/.
set -u
R=$(pwd)
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
# the default workload with every CPU baseline (serial port, all cores, NumPy + multiprocessing structure mirror)
timeout -k 10 900 python bench.py --workload cube_1e5 --steps 1000 --warmup 50 > "$OUT/bench_cube_1e5.json" 2> "$OUT/bench_cube_1e5.err" || echo "bench cube_1e5 failed"
for w in cube_1e6 pore_5e5 pore_1e6; do
    timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-python-mp-baseline --extra-workloads none > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || echo "bench $w failed"
done
# the all-pairs detector against the fp64 vector peak
for w in cube_allpairs_4096 cube_allpairs_1e5; do
    timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || echo "bench $w failed"
done
timeout -k 10 300 python bench.py --workload temp_1e6 --steps 200 --warmup 20 --no-python-mp-baseline > "$OUT/bench_temp_1e6.json" 2> "$OUT/bench_temp_1e6.err" || echo "bench temp failed"
# opt-in: the overlapped run (AMC_OVERLAP=1) on the two large workloads
for w in cube_1e6 pore_1e6; do
    AMC_OVERLAP=1 timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-cpu-baseline --extra-workloads none > "$OUT/bench_overlap_$w.json" 2> "$OUT/bench_overlap_$w.err" || echo "bench overlap $w failed"
done
# the multi-GPU driver with one rank: same step through dist.ShardedSimulation, collective issued over RCCL
timeout -k 10 300 python bench.py --force-sharded --workload cube_1e5 --steps 1000 --warmup 50 --no-cpu-baseline > "$OUT/bench_sharded1_cube_1e5.json" 2> "$OUT/bench_sharded1_cube_1e5.err" || echo "bench sharded failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_sharded1_cube_1e5" -o out --output-format csv -- \
    python "$R/bench.py" --force-sharded --workload cube_1e5 --steps 1000 --warmup 5 --no-cpu-baseline > "$OUT/kt_sharded1_cube_1e5.log" 2>&1 || echo "kernel trace sharded failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_cube_allpairs_1e5" -o out --output-format csv -- \
    python "$R/bench.py" --workload cube_allpairs_1e5 --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/kt_cube_allpairs_1e5.log" 2>&1 || echo "kernel trace allpairs failed"
for w in cube_1e5 pore_5e5 pore_1e6 cube_1e6 temp_1e6; do
    steps=1000; [ $w = temp_1e6 ] && steps=20
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$w" -o out --output-format csv -- \
        python "$R/bench.py" --workload $w --steps $steps --warmup 5 --no-cpu-baseline --extra-workloads none > "$OUT/kt_$w.log" 2>&1 || echo "kernel trace $w failed"
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_$w" -o out --output-format csv -- \
            python "$R/bench.py" --workload $w --steps 20 --warmup 2 --no-cpu-baseline --extra-workloads none > "$OUT/pmc_${c}_$w.log" 2>&1 || echo "pmc $c $w failed"
    done
done
ls "$OUT"
