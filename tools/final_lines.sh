#!/bin/bash
# the bench lines and the kernel statistics of the headline workload for the build as it is (short form of
# collect_profiles.sh: run when only the numbers of the final build are missing)
R=$(pwd); OUT=$R/gpurun_out/prof; mkdir -p "$OUT"
timeout -k 10 400 python bench.py --workload cube_1e5 --steps 1000 --warmup 50 > "$OUT/bench_cube_1e5.json" 2> "$OUT/bench_cube_1e5.err" || echo "bench cube_1e5 failed"
for w in cube_1e6 pore_5e5 pore_1e6; do
    timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 50 --no-python-mp-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || echo "bench $w failed"
done
timeout -k 10 300 python bench.py --workload temp_1e6 --steps 200 --warmup 20 --no-python-mp-baseline > "$OUT/bench_temp_1e6.json" 2> "$OUT/bench_temp_1e6.err" || echo "bench temp failed"
cd /tmp && export TMPDIR=/tmp
for w in cube_1e5 pore_1e6 cube_1e6; do
    rm -rf "$OUT/kt_$w"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$w" -o out --output-format csv -- \
        python "$R/bench.py" --workload $w --steps 1000 --warmup 5 --no-cpu-baseline > "$OUT/kt_$w.log" 2>&1 || echo "kernel trace $w failed"
done
