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
import multiprocessing as mp
pool = E.SurfaceEnergies._pool
if pool is not None:
    t = time.perf_counter()
    for _ in range(50): pool.map(abs, [1.0] * 5, chunksize=1)
    print("pool.map of 5 trivial tasks: %.3f ms" % ((time.perf_counter() - t) / 50 * 1e3))

# do the workers run faster when they were woken shortly before?  (a step has ~0.6 ms of other work before the gap case)
def _spin(ms):
    t = time.perf_counter()
    while time.perf_counter() - t < ms * 1e-3:
        pass
    return 0
if pool is not None:
    for warm_ms in (0.0, 0.2, 0.6):
        tp = []
        for rep in range(30):
            z = zs(5)
            time.sleep(0.002)
            if warm_ms > 0:
                pool.map_async(_spin, [warm_ms] * 8, chunksize=1)
            time.sleep(0.0006)
            t = time.perf_counter(); b = en.gap_many(z); tp.append(time.perf_counter() - t)
        print("5 hits, workers spun %.1f ms beforehand: gap_many %.2f ms" % (warm_ms, np.median(tp) * 1e3))
