"""Test infrastructure (run by hand / through gpurun, not collected by pytest).  The north-star run shape — pore geometry,
N = 10^6, 10^4 steps — on ONE GPU, checked through size-independent properties (the oracle would need ~1.6 hours for it;
the first 1,000 steps of this very workload are compared bit for bit in tests/soak.py): kinetic energy conserved by the
specular walls and the elastic collisions, every particle inside the geometry, histogram totals equal the number of
completed paths, no capacity / grid flags.  Writes a JSON summary (committed under profiles/).

    python tests/long_run.py pore_1e6 10000
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import make_workload
from argon_monte_carlo_amd.engine import Engine

workload = sys.argv[1] if len(sys.argv) > 1 else "pore_1e6"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
chunk = 1000
p, c, init = make_workload(workload)
eng = Engine(p)
eng.upload(*init)
v2_0 = float(np.sum(np.asarray(init[3]) ** 2 + np.asarray(init[4]) ** 2 + np.asarray(init[5]) ** 2))
tot = {}
t_gpu = 0.0
for done in range(0, steps, chunk):
    t0 = time.perf_counter()
    st = eng.run(c["dt"], min(chunk, steps - done))
    t_gpu += time.perf_counter() - t0
    for k, v in st.items():
        tot[k] = tot.get(k, 0) + v if k != "flags" else (tot.get(k, 0) | v)
s = eng.download()
counts, npaths = eng.histograms()
v2 = float(np.sum(s["vx"] ** 2 + s["vy"] ** 2 + s["vz"] ** 2))
outside = int(eng.stage_bounds())
import hashlib
state_sha = hashlib.sha256(b"".join(np.ascontiguousarray(s[k]).tobytes() for k in ("x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz"))).hexdigest()
out = {"workload": workload, "n": int(p.n), "steps": steps, "gpu_seconds": round(t_gpu, 3),
       "overlapped_run": eng.overlap_stats(), "final_state_sha256": state_sha,
       "histograms_sha256": hashlib.sha256(np.ascontiguousarray(counts).tobytes()).hexdigest(),
       "particle_steps_per_s": p.n * steps / t_gpu,
       "counters": {k: int(tot[k]) for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors", "flags")},
       "kinetic_energy_relative_drift": abs(v2 / v2_0 - 1.0),
       "particles_outside_after_run": outside,
       "histogram_total_paths": int(npaths), "histogram_rows_le_total": bool(counts.sum(axis=1).max() <= npaths),
       "all_finite": bool(all(np.isfinite(s[k]).all() for k in ("x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz")))}
out["ok"] = bool(out["kinetic_energy_relative_drift"] < 1e-9 and outside == 0 and npaths == tot["n_paths"] and
                 tot["flags"] == 0 and out["all_finite"] and out["histogram_rows_le_total"])
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/long_{workload}_{steps}" + ("_overlap" if out["overlapped_run"]["steps"] else "") + (("_keep" + os.environ["AMC_LIST_KEEP"]) if os.environ.get("AMC_LIST_KEEP", "0") not in ("", "0", "1") else "") + ".json", "w") as f:
    f.write(json.dumps(out) + "\n")
print(json.dumps(out))
sys.exit(0 if out["ok"] else 1)
