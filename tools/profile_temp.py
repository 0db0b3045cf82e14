"""Where the energised-wall step spends its host time (per case: hits, RNG draws, mpmath, API calls)."""
import os
import sys
import time
import random

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import make_workload
from argon_monte_carlo_amd.engine import EnergisedEngine
from argon_monte_carlo_amd.energised import CASES, GAP_CASE, COLD_CASES, DirectionSampler, SurfaceEnergies

p, c, init = make_workload(sys.argv[1] if len(sys.argv) > 1 else "temp_1e6")
p.reserved0 |= 1
e = EnergisedEngine(p)
e.upload(*init)
sampler = DirectionSampler(np.random.RandomState(17), random.Random(17))
energies = SurfaceEnergies(c)
T = {}
hits = {k: 0 for k in CASES}


def tm(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0


steps, warm = 20, 5
for s in range(steps + warm):
    if s == warm:
        T.clear(); hits = {k: 0 for k in CASES}; t_all = time.perf_counter()
    t0 = time.perf_counter(); e.temp_begin(c["dt"]); tm("temp_begin", t0)
    t0 = time.perf_counter(); sess = sampler.session(); sess.__enter__(); tm("rng states", t0)
    for case in CASES:
        t0 = time.perf_counter(); idx, normals, cz, ok = e.wall_hits(case); tm("wall_hits", t0)
        n = len(idx); hits[case] += n
        if n == 0:
            continue
        t0 = time.perf_counter(); dirs = sampler.sample_case(normals, ok); tm("rng", t0)
        t0 = time.perf_counter()
        good = [k for k in range(n) if ok[k]]
        Es = np.zeros(n)
        if case == GAP_CASE:
            Es[good] = energies.gap_many([cz[k] for k in good])
        else:
            Es[good] = energies.cold if case in COLD_CASES else energies.hot
        tm("energies", t0)
        t0 = time.perf_counter(); e.wall_apply(case, dirs, Es); tm("wall_apply", t0)
    t0 = time.perf_counter(); sess.__exit__(None, None, None); tm("rng states", t0)
    t0 = time.perf_counter(); e.temp_end(); tm("temp_end", t0)
tot = time.perf_counter() - t_all
print("total %.2f ms/step; hits per step by case: %s" % (tot / steps * 1e3, {k: v / steps for k, v in hits.items()}))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-12s %8.2f ms/step" % (k, v / steps * 1e3))
