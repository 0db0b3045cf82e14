"""gpurun_out/prof/ (tools/collect_profiles.sh) -> profiles/<round>_*: bench lines, kernel statistics, HBM traffic."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)

for f in sorted(glob.glob(f"{SRC}/bench_*.json")):
    txt = open(f).read().strip()
    if txt:
        json.loads(txt.splitlines()[-1])
        open(f"profiles/{ROUND}_{os.path.basename(f)}", "w").write(txt.splitlines()[-1] + "\n")

for d in sorted(glob.glob(f"{SRC}/kt_*")):
    if os.path.isdir(d):
        st = glob.glob(f"{d}/**/out_kernel_stats.csv", recursive=True)
        if st:
            shutil.copy(st[0], f"profiles/{ROUND}_{os.path.basename(d)[3:]}_kernel_stats.csv")

for w in sorted({os.path.basename(d).split("_SIZE_")[1] for d in glob.glob(f"{SRC}/pmc_*_SIZE_*") if os.path.isdir(d)}):
    kern = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{SRC}/pmc_{c}_{w}/**/out_counter_collection.csv", recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        two_pass = not w.startswith("cube")       # pore / temp: besides the main streaming pass there are bounds-only passes
        stream_vals = sorted(float(r["Counter_Value"]) for r in rows if "k_stream" in r["Kernel_Name"])
        median = stream_vals[len(stream_vals) // 2] if stream_vals else 0.0
        for r in rows:
            name = r["Kernel_Name"].split("(")[0]
            if "k_stream" in name and two_pass:
                # (inside amc_run the bounds check rides along with the next main pass; the few bounds-only launches
                # are told apart by their much smaller traffic)
                name += " [main pass]" if float(r["Counter_Value"]) >= 0.8 * median else " [bounds-only pass]"
            acc[name].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            kern[k][f"{c}_KiB_avg"] = round(sum(v) / len(v), 1)
            kern[k][f"{c}_launches"] = len(v)
    for k, v in kern.items():
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies a wide coalesced streaming read at half its bytes, WRITE_SIZE
        # is exact for streaming stores; other patterns are uncalibrated.  The main streaming pass is such a read (and its
        # known 81 B per particle calibrate the factor: see DESIGN.md 7); every other kernel is reported as counted.
        if "k_stream" in k and "bounds-only" not in k and "FETCH_SIZE_KiB_avg" in v:
            v["HBM_bytes_corrected"] = round((2.0 * v["FETCH_SIZE_KiB_avg"] + v.get("WRITE_SIZE_KiB_avg", 0.0)) * 1024)
            v["correction"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950 half-count of wide coalesced reads)"
        else:
            v["HBM_bytes_corrected"] = round((v.get("FETCH_SIZE_KiB_avg", 0.0) + v.get("WRITE_SIZE_KiB_avg", 0.0)) * 1024)
            v["correction"] = "none (FETCH_SIZE + WRITE_SIZE as counted: scattered accesses are uncalibrated)"
    out = {"workload": w, "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --steps 20 --warmup 2",
           "units": "KiB per launch as reported by rocprofv3 (uncorrected; MI355X_MICROARCH.md: FETCH_SIZE counts wide "
                    "coalesced reads at half their bytes on gfx950, scattered accesses are uncalibrated)",
           "kernels": kern}
    json.dump(out, open(f"profiles/{ROUND}_pmc_traffic_{w}.json", "w"), indent=1)
# bench lines were produced before the counter files of this pass existed: fill `traffic` the way bench.py does
sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
for f in sorted(glob.glob(f"profiles/{ROUND}_bench_*.json")):
    d = json.loads(open(f).read())
    if d.get("roofline"):
        if d["roofline"].get("bound") != "hbm":
            continue
        t, src = bench.committed_traffic(d["config"]["workload"], d["roofline"]["kernel_class"], ROUND)
        d["roofline"]["traffic"], d["roofline"]["traffic_source"] = t, src
        open(f, "w").write(json.dumps(d) + "\n")
print(sorted(os.listdir("profiles")))
