#!/bin/bash
# parity suite, soaks (also with 8 waves), debug table, then A/B against base
set -o pipefail
mkdir -p gpurun_out/r3
bash tools/run_gpu_tests.sh 600 || exit 1
timeout -k 10 300 python tests/soak.py cube_1e5 3000 500 > gpurun_out/r3/soak_cont_cube_1e5.json 2> gpurun_out/r3/soak.err || { echo soak1 failed; tail -5 gpurun_out/r3/soak.err; exit 1; }
timeout -k 10 300 python tests/soak.py cube_1e5 1000 500 --cw-blocks 8 > gpurun_out/r3/soak_cont_cube_1e5_cw8.json 2> gpurun_out/r3/soak.err || { echo soak2 failed; tail -5 gpurun_out/r3/soak.err; exit 1; }
timeout -k 10 400 python tests/soak.py pore_1e6 300 100 > gpurun_out/r3/soak_cont_pore_1e6.json 2> gpurun_out/r3/soak.err || { echo soak3 failed; tail -5 gpurun_out/r3/soak.err; exit 1; }
grep -h -o '"all_equal[^,]*\|"equal[^,]*\|"hist[^,]*' gpurun_out/r3/soak_cont_*.json | sort | uniq -c
AMC_DEBUG_RESOLVE=1 timeout -k 10 100 python bench.py --workload cube_1e5 --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/r3/tdbg_cube_1e5.json 2> gpurun_out/r3/tdbg_cube_1e5.err || exit 1
grep "k_clusters_wide" gpurun_out/r3/tdbg_cube_1e5.err | cut -c1-400
bash tools/r3_ab.sh "base ab2" "cube_1e5 pore_1e6 cube_1e6" 2
