#!/bin/bash
# rehearsal of the driver's multi-GPU invocation on the one-GPU box: ranks share GPU 0, gloo in place of RCCL
mkdir -p gpurun_out/mg
for g in 2 4; do
  for w in cube_1e5 pore_1e6; do
    timeout -k 10 200 python bench.py --gpus $g --backend gloo --same-device --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/mg/b_${w}_$g.json 2> gpurun_out/mg/b_${w}_$g.err || { echo "failed $w $g"; tail -5 gpurun_out/mg/b_${w}_$g.err; exit 1; }
    python - <<PY
import json
d=json.load(open('gpurun_out/mg/b_${w}_$g.json'))
print('$w', $g, 'ranks:', d['n_gpus'], d['scaling'], 'us/step %.1f'%(d['ms_per_step']*1e3), 'value %.3g'%d['value'], d['config'].get('parallelism'), {k:round(v,1) for k,v in d['roofline']['per_kernel_avg_us'].items()})
PY
  done
done
