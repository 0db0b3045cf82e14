"""How long the gap energies of one case take: serial, in the forked workers, and what the pool itself costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from argon_monte_carlo_amd import energised as E, params as PR

p, c = PR.pore_params(n=1000, energised=True)
en = E.SurfaceEnergies(c)
z0 = c["open_air_height"] + c["hot_coating_height"]
rng = np.random.default_rng(1)
def zs(k):
    return list(z0 + c["gap_height"] * rng.random(k))
for k in (1, 2, 5, 8, 16):
    ts, tp = [], []
    for rep in range(30):
        z = zs(k)
        t = time.perf_counter(); a = [en.gap(v) for v in z]; ts.append(time.perf_counter() - t)
        time.sleep(0.002)                     # the rest of a step
        t = time.perf_counter(); b = en.gap_many(z); tp.append(time.perf_counter() - t)
        assert a == b
    print("%2d hits: serial %.2f ms, gap_many %.2f ms (median of 30)" % (k, np.median(ts) * 1e3, np.median(tp) * 1e3))
# (measured with this tool before the workers became plain forked processes with a pipe each: a multiprocessing.Pool took
# 1.38 ms for five hits — its handler threads compete with the caller for the interpreter — and waking the workers
# 0.2-0.6 ms beforehand changed nothing)
