"""Test infrastructure (run by hand / through gpurun, not collected by pytest).  The largest sizes: does the path hold
beyond BASELINE's (indices above 2^24, tens of thousands of candidates per sweep, the large-sweep plan, gigabytes of
state)?  A few steps of the cube / pore at N particles on the GPU and in the oracle, compared bit for bit, plus the
step time.

    python tests/big_n.py cube 16000000 3
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from argon_monte_carlo_amd import ic as IC
from argon_monte_carlo_amd import params as PR
from argon_monte_carlo_amd.engine import Engine
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "cube"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 16_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t0 = time.time()
if kind == "cube":
    p, c = PR.cube_params_for_n(n)
    init = IC.cube_ic(p, c, seed=127)
else:
    p, c = PR.pore_params(n=n)
    init = IC.pore_ic(p, c, seed=17)
p.reserved1 = 1
p.max_paths = -1
print(f"{kind} N = {n}: initial conditions in {time.time() - t0:.1f} s", flush=True)
eng = Engine(p)
orc = O.Oracle(p, mode="mul", path_capacity=1 << 24)
eng.upload(*init)
orc.upload(*init)
keys = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz", "flag"]
ok, npp = True, 0
for s in range(steps):
    st = eng.timestep(c["dt"])
    t1 = time.time()
    rc, so = orc.timestep(c["dt"])
    assert rc == 0
    npp += st["n_pp"]
    g, o = eng.download(), orc.state()
    same = all(np.array_equal(np.asarray(g[k]).view(np.uint8), np.asarray(o[k]).view(np.uint8)) for k in keys)
    cnt = all(st[k] == so[k] for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"))
    ok = ok and same and cnt
    print(f"step {s}: collisions {st['n_pp']}, candidates {st.get('n_candidates')}, state identical {same}, counters equal {cnt} (oracle step {time.time() - t1:.1f} s)", flush=True)
# step time without the host in the loop
eng.run(c["dt"], 5)
t0 = time.perf_counter()
k = 50
eng.run(c["dt"], k)
dt_step = (time.perf_counter() - t0) / k
out = {"geometry": kind, "n": n, "steps_compared": steps, "state_bit_identical": bool(ok), "collisions": int(npp),
       "ms_per_step": dt_step * 1e3, "particle_steps_per_s": n / dt_step,
       "fraction_of_hbm_peak_at_137_B": 137.0 * n / dt_step / 8.0e12}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/big_{kind}_{n}.json", "w"), indent=1)
