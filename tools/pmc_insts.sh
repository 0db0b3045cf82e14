#!/bin/bash
# instruction counts per kernel (SQ counters), one pass per counter group; summary printed per kernel name
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/insts
mkdir -p $OUT
W=${1:-cube_1e5}
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo $c | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d "$OUT/$tag" -o out --output-format csv -- \
        python "$R/bench.py" --workload $W --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/$tag.log" 2>&1 || echo "pmc $c failed"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[(k, r["Counter_Name"])] += 0
        cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / max(1, cnt[(k, c)]), 1) for c, v in acc[k].items()})
PY
