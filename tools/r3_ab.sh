#!/bin/bash
# same-session A/B of library variants: tools/r3_ab.sh "base OWN HEAD" "pore_1e6 cube_1e6" reps
mkdir -p gpurun_out/ab
cp argon_monte_carlo_amd/libargonmc.so libamc_keep.tmp.so
for rep in $(seq 1 ${3:-2}); do
for v in $1; do
  cp libamc_$v.tmp.so argon_monte_carlo_amd/libargonmc.so
  for w in $2; do
    timeout -k 10 100 python bench.py --workload $w --steps 1000 --warmup 20 --no-cpu-baseline --extra-workloads none > gpurun_out/ab/${v}_${rep}_$w.json 2> gpurun_out/ab/err || { echo "bench failed $v $w"; tail -3 gpurun_out/ab/err; exit 1; }
  done
done; done
cp libamc_keep.tmp.so argon_monte_carlo_amd/libargonmc.so
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    d=json.load(open(f)); r=d['roofline']; v,rep,w=f.split('/')[-1][:-5].split('_',2)
    acc[(w,v)].append((d['ms_per_step']*1e3,)+tuple(r['per_kernel_avg_us'].get(k,0) for k in ('drift_walls','detect','clusters_wide','resolve')))
for k in sorted(acc):
    rows=acc[k]; print(k, ' | '.join(' '.join('%.1f'%x for x in row) for row in rows))
PY
