#!/bin/bash
set -u
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for n in 1000 4096; do
for mx in 4096 0; do
  AMC_ALLPAIRS_MAX_N=$mx timeout -k 10 60 python bench.py --workload cube_1e5 --n $n --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > $O/small_${n}_${mx}.json 2> $O/small.err || { echo "bench failed"; tail -3 $O/small.err; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/small_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'us/step', round(d['ms_per_step']*1e3,2), {k:round(v,1) for k,v in r['per_kernel_avg_us'].items()}, d['config']['pp_collisions_per_step'])
PY
timeout -k 10 200 python tests/mfp_stat.py 100000 > $O/mfp_stat.json 2> $O/mfp_stat.err; echo "mfp rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/mfp_stat.json'))
for r in d['runs']:
    print(r['dt_is_tau_over'], r['mean_displacement_per_step_in_collision_ranges'], r['pp_collisions'], r['completed_paths_in_histogram_range'], r['hip'], r['chi2_distance_normalised_histograms'], r['chi2_distance_expected_from_sampling_noise'])
PY
