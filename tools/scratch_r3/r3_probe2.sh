#!/bin/bash
# round-3 scratch: dispatch-attached profiling events vs rocprofv3, bench with extra workloads
set -u
R=$(pwd)
mkdir -p gpurun_out/r3
O=$R/gpurun_out/r3
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20.json 2> $O/b20.err || echo "bench 20 failed"
timeout -k 10 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/b1000.json 2> $O/b1000.err || echo "bench 1000 failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o out --output-format csv -- python $R/bench.py --steps 1000 --warmup 5 --no-cpu-baseline --extra-workloads none > $O/kt.log 2>&1 || echo "kt failed"
cd $R
python - <<'PY'
import json,glob
for f in ['gpurun_out/r3/b20.json','gpurun_out/r3/b1000.json']:
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f, 'ms/step', round(d['ms_per_step']*1e3,2), 'prof', round(r['profiled_pass_ms_per_step']*1e3,2))
        print('  per step', {k:round(v,2) for k,v in r['per_kernel_avg_us'].items()}, 'sum', round(sum(r['per_kernel_avg_us'].values()),2))
        print('  per launch', {k:round(v,2) for k,v in r['per_kernel_avg_launch_us'].items()})
        for e in d.get('extra_workloads',[]):
            print('  extra', e['workload'], round(e['ms_per_step']*1e3,2), {k:round(v,2) for k,v in e['per_kernel_avg_us'].items()}, e['whole_step_frac'], e['pair_sweep_frac'])
    except Exception as e: print(f, 'ERR', e)
PY
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
tail -3 $O/b20.err
