"""The bench line's contract (task brief, measurement section), checked on the committed lines of this round and on
bench.py's command line — no GPU needed."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r03_bench_*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_keeps_the_contract(path):
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "particle-steps/sec" and d["unit"] == "particle-steps/s"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    # value = whole-job particle-steps per second over exactly `steps` timed steps
    n = d["config"]["n_particles"]
    assert d["value"] == pytest.approx(n / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0 < r["frac"] < 1
    assert (r["unit"], r["peak"]) in (("GB/s", 8000.0), ("TFLOP/s", 78.6))
    if "executed" in r:         # the tiled all-pairs detector: what is executed stays below what is counted
        assert r["executed"]["frac_of_peak"] < r["frac"] and r["executed"]["vector_instruction_issue_frac"] < 1
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


R03 = [p for p in LINES if os.path.basename(p).startswith("r03_")]


@pytest.mark.parametrize("path", R03, ids=[os.path.basename(p) for p in R03])
def test_kernel_classes_add_up_to_no_more_than_the_step(path):
    """Round 3: the per-kernel figures are time per STEP (class total / steps), measured by events the dispatches carry
    themselves (kernel begin -> kernel end).  Kernels of one stream run one after the other, so the classes of a
    single-stream step cannot add up to more than the step; a class launched three times in a thousand steps weighs
    next to nothing; the pair-sweep and streaming-pass lines are sums of these."""
    d = json.load(open(path))
    r = d["roofline"]
    per = r["per_kernel_avg_us"]
    if d["config"]["workload"].startswith("temp"):
        return                                          # (host-bound step: the kernels are a tenth of it)
    overlapped = "overlap" in os.path.basename(path)    # two streams: the pass runs beside the resolve kernels
    # ("allgather" is the collective, bracketed by torch events AROUND the call: dispatch gaps included, not a kernel class)
    total = sum(v for k, v in per.items() if not (overlapped and k == "drift_walls") and k != "allgather")
    assert total <= 1.1 * d["ms_per_step"] * 1e3, (total, d["ms_per_step"])
    assert set(per) == set(r["per_kernel_avg_launch_us"]) == set(r["per_kernel_launches_per_step"])
    for k in per:
        assert per[k] == pytest.approx(r["per_kernel_avg_launch_us"][k] * r["per_kernel_launches_per_step"][k], rel=1e-9)
    if "pair_sweep" in r:
        sweep = sum(v for k, v in per.items() if k in ("bin_count", "detect", "clusters_wide", "resolve", "commit", "fixup"))
        assert r["pair_sweep"]["avg_us"] == pytest.approx(sweep, rel=1e-9) and r["pair_sweep"]["avg_us"] <= 1.1 * d["ms_per_step"] * 1e3
    for e in d.get("extra_workloads", []):
        assert sum(e["per_kernel_avg_us"].values()) <= 1.1 * e["ms_per_step"] * 1e3
        assert e["value"] == pytest.approx(e["n_particles"] / (e["ms_per_step"] * 1e-3), rel=1e-6)


def test_the_default_line_covers_both_sizes_the_metric_names():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_cube_1e5.json")))
    assert [e["workload"] for e in d["extra_workloads"]] == ["cube_1e6", "pore_1e6"]
    assert d["cpu_baseline_all_cores"]["cores_source"] in ("affinity", "cgroup", "assumed", "env")


def test_the_default_workload_is_baseline_configs_1():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_cube_1e5.json")))
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["config"]["workload"] == "cube_1e5" and d["config"]["n_particles"] == 100_000
    assert "cpu_baseline" in d and "cpu_baseline_all_cores" in d and "cpu_baseline_python_mp" in d
    assert "particle-steps" in json.dumps(b)      # the metric BASELINE.json names


def test_bench_command_line_parses_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--strong", "--force-sharded", "--extra-workloads"):
        assert flag in out, flag
