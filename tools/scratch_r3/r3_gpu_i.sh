#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_dist.py -m gpu -x -q > $O/dist_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/dist_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/rehearse_ranks.py cube 800000 8 100 > $O/rehearsal_cube_8x1e5_sharded.json 2> $O/reh.err || { echo "rehearsal failed"; tail -5 $O/reh.err; exit 1; }
timeout -k 10 200 python tools/rehearse_ranks.py cube 800000 8 100 --replicated > $O/rehearsal_cube_8x1e5_replicated.json 2> $O/reh.err || { echo "rehearsal failed"; tail -5 $O/reh.err; exit 1; }
timeout -k 10 300 python tools/rehearse_ranks.py pore 4000000 8 30 > $O/rehearsal_pore_8x5e5_sharded.json 2> $O/reh.err || { echo "rehearsal failed"; tail -5 $O/reh.err; exit 1; }
timeout -k 10 300 python tools/rehearse_ranks.py pore 4000000 8 30 --replicated > $O/rehearsal_pore_8x5e5_replicated.json 2> $O/reh.err || { echo "rehearsal failed"; tail -5 $O/reh.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/rehearsal_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], d['all_ranks_equal_single_context_bit_for_bit'], round(d['rank0_kernel_us_per_step_sum'],1), {k:round(v,1) for k,v in d['rank0_kernel_us_per_step'].items()}, d['pp_collisions_per_step'])
PY
