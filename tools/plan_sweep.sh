#!/bin/bash
# experiment: step time against the launch-plan threshold (AMC_PLAN_SMALL) at sweeps of ~400 / ~600 candidates
show='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]["per_kernel_avg_us"]
print(sys.argv[1], d["config"]["workload"], d["config"]["n_particles"], d["config"]["pp_collisions_per_step"], round(d["ms_per_step"]*1e3,1), {k:round(v,1) for k,v in r.items()})'
for spec in "cube_1e5 200000" "cube_1e5 300000" "pore_5e5 750000" "pore_5e5 1000000" "pore_5e5 1400000"; do
    set -- $spec
    for t in 640 150; do
        AMC_PLAN_SMALL=$t timeout -k 10 100 python bench.py --workload $1 --n $2 --steps 1000 --warmup 100 --no-cpu-baseline 2>> gpurun_out/pl.err | python -c "$show" $t
    done
done
