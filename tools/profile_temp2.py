"""Where the energised-wall step spends its host time with the round-3 driver (gap case started early and parked): the
hooks, the sampler and the energies are wrapped with timers, the step itself is energised.drive_energised_cases."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import make_workload
from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
from argon_monte_carlo_amd.engine import EnergisedEngine

p, c, init = make_workload(sys.argv[1] if len(sys.argv) > 1 else "temp_1e6")
p.reserved0 |= 1
energies = SurfaceEnergies(c, start_workers=True)
p.E_cold, p.E_hot = energies.cold, energies.hot
e = EnergisedEngine(p)
e.upload(*init)
sampler = DirectionSampler(np.random.RandomState(17), random.Random(17))
T, N = {}, {}


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        N[label] = N.get(label, 0) + 1
        return r
    setattr(obj, name, g)


for m in ("temp_begin", "wall_hits", "wall_apply", "wall_park", "wall_finish", "temp_end"):
    wrap(e, m)
for m in ("gap_start", "gap_finish", "gap_many"):
    wrap(energies, m)
wrap(sampler, "sample_case")
steps, warm = 50, 10
for s in range(steps + warm):
    if s == warm:
        T.clear(); N.clear(); t_all = time.perf_counter()
    e.temp_timestep(c["dt"], sampler, energies)
tot = time.perf_counter() - t_all
print("total %.3f ms/step" % (tot / steps * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-12s %7.3f ms/step  (%.1f calls/step)" % (k, v / steps * 1e3, N[k] / steps))
print("  %-12s %7.3f ms/step" % ("(rest: python glue, rng state hand-over, sums)", (tot - sum(T.values())) / steps * 1e3))
