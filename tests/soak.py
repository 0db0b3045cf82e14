"""Test infrastructure (run by hand / through gpurun, not collected by pytest).  Long parity run on the GPU box: the HIP path and the CPU oracle advance the same workload side by side; the state
is compared bit for bit every `chunk` steps and the device histograms with np.histogram of the oracle's completed
paths at the end.  Writes a JSON summary (committed under profiles/ as evidence).

    python tests/soak.py pore_1e6 1000 100 [--cw-blocks 8]

--cw-blocks K runs the wide cluster kernel with K waves instead of 512 (AMC_CW_BLOCKS): many clusters per wave, emulated
in lockstep and sharing one work-item list — the configuration, not the step count, that exposed the one protocol bug of
round 2; every soak of record runs once in it.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import make_workload
from argon_monte_carlo_amd.engine import Engine
from oracle import oracle as O

cw_blocks = 0
if "--cw-blocks" in sys.argv:
    k = sys.argv.index("--cw-blocks")
    cw_blocks = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
    os.environ["AMC_CW_BLOCKS"] = str(cw_blocks)        # (read when the context is created)
workload = sys.argv[1] if len(sys.argv) > 1 else "pore_1e6"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 100
p, c, init = make_workload(workload)
temp = workload.startswith("temp")
if temp:
    # energised walls: both sides draw from identically seeded np.random / random streams (parity mode) and the per-step
    # z-momentum / energy sums (the momentum_energy.csv columns) are compared as well
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
    from argon_monte_carlo_amd.engine import EnergisedEngine
    energies = SurfaceEnergies(c, start_workers=True)      # (forked before the engine's GPU context exists)
    p.reserved0 |= 1
    p.E_cold, p.E_hot = energies.cold, energies.hot
    eng = EnergisedEngine(p)
    s_gpu = DirectionSampler(np.random.RandomState(17), random.Random(17))
    s_cpu = DirectionSampler(np.random.RandomState(17), random.Random(17))
else:
    eng = Engine(p)
orc = O.Oracle(p, mode="mul", path_capacity=1 << 24)
eng.upload(*init)
orc.upload(*init)
csv_equal = True
keys = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz", "flag"]
tot = {}
otot = {}
t_gpu = t_cpu = 0.0
done = 0
ok = True
first_bad = None
while done < steps and ok:
    k = min(chunk, steps - done)
    if temp:
        for _ in range(k):
            t0 = time.perf_counter()
            g = eng.temp_timestep(c["dt"], s_gpu, energies)
            t_gpu += time.perf_counter() - t0
            t0 = time.perf_counter()
            o = orc.temp_timestep(c["dt"], s_cpu, energies)
            t_cpu += time.perf_counter() - t0
            if o[0] != 0:
                raise SystemExit("oracle reported an error")
            csv_equal = csv_equal and tuple(float(v) for v in g[1:4]) == tuple(float(v) for v in o[2:5]) and g[4:] == o[5:]
            for kk in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"):
                otot[kk] = otot.get(kk, 0) + o[1][kk]
                tot[kk] = tot.get(kk, 0) + g[0][kk]
    else:
        t0 = time.perf_counter()
        st = eng.run(c["dt"], k)
        t_gpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(k):
            rc, so = orc.timestep(c["dt"])
            if rc != 0:
                raise SystemExit(f"oracle aborted the step (rc={rc}): the reference would have raised here")
            for kk in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"):
                otot[kk] = otot.get(kk, 0) + so[kk]
        t_cpu += time.perf_counter() - t0
        for kk in otot:
            tot[kk] = tot.get(kk, 0) + st[kk]
    done += k
    g, o = eng.download(), orc.state()
    for kk in keys:
        if not np.array_equal(g[kk], o[kk]):
            ok = False
            first_bad = (done, kk, int(np.flatnonzero(g[kk] != o[kk])[0]))
            break
    print(f"step {done}: {'equal' if ok else 'DIFFERENT ' + str(first_bad)}  counters gpu={tot} cpu={otot}", flush=True)
counts, npaths = eng.histograms()
paths = orc.paths()
hist_equal = bool(npaths == len(paths))
for row, key in enumerate(("total", "px", "py", "pz")):
    ref, _ = np.histogram(paths[key], bins=p.hist_bins, range=(p.hist_lo, p.hist_hi))
    hist_equal = hist_equal and bool(np.array_equal(counts[row], ref.astype(np.uint64)))
out = {"workload": workload, "n": int(p.n), "steps": done, "wide_kernel_waves": cw_blocks or 512, "compared_every": chunk, "state_bit_identical": ok,
       "first_difference": first_bad, "counters_equal": tot == otot, "counters": tot, "completed_paths": int(npaths),
       "histograms_equal_np_histogram_of_oracle_paths": hist_equal,
       "per_step_momentum_energy_sums_equal": (csv_equal if temp else None), "gpu_seconds": round(t_gpu, 3),
       "oracle_seconds_1_core": round(t_cpu, 1)}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/soak_{workload}_{done}" + (f"_cw{cw_blocks}" if cw_blocks else "") + ".json", "w"), indent=1)
