"""Test infrastructure: the initial state of Open_Air_Pore_MC.py at its OWN parameters (N = 557,649), regenerated.

tests/golden/step_pore_natural.npz (oracle/gen_golden.py --only pore_natural) holds what the unmodified reference did in
its first steps — counters, completed paths, event sets and the SHA-256 of every state array — but not the 27 MB of
random doubles it started from.  That state is a pure function of the two seeds (Pore:89-90) and of the library calls the
reference makes while it initialises (Pore:106-158: NumPy's legacy global generator, CPython's `random`, SciPy's
`maxwell.rvs`, libm through `math`), so it is made again here by issuing the same calls in the same order, and checked
against the stored hashes before anything is compared with it (a different NumPy / SciPy build skips the test instead of
failing it).  The reference itself is not needed and not read.

Order of the draws (what the recipe has to reproduce):
  1. theta = np.random.uniform(0, 2 pi, N); u = np.random.uniform(0, 1, N)                       (Pore:115-116)
  2. per region, in the order bottom cap, hot, gap, cold, top cap: z = np.random.uniform(lo, hi, count)   (Pore:122-139)
     with x = (R * sqrt(u)) * cos(theta), y = (R * sqrt(u)) * sin(theta) through math.cos / math.sin     (Pore:107-111)
  3. speeds = scipy.stats.maxwell.rvs(loc=0, scale=a_shape, size=N)                              (Pore:144)
  4. per particle: cos_t = np.random.uniform(-1, 1); phi = random.uniform(0, pi); sign = np.random.choice([-1, 1])
     -> (v cos(phi) sin(t), v sin(phi) sin(t) sign, v cos(t)), t = acos(cos_t)                   (Pore:97-104, 149-153)
"""
import hashlib
import math
import random

import numpy as np


def sha(a):
    a = np.asarray(a)
    a = a.astype(np.uint8) if a.dtype == bool else np.ascontiguousarray(a, dtype=np.float64)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)


def reference_pore_initial_state(consts, seed=17):
    """(x, y, z, vx, vy, vz) as Open_Air_Pore_MC.py builds them from np.random.seed(seed) / random.seed(seed)."""
    from scipy.stats import maxwell
    c = consts
    n = int(c["num_molecules"])
    np.random.seed(seed)
    random.seed(seed)
    theta = np.random.uniform(0, 2 * np.pi, n)
    u = np.random.uniform(0, 1, n)
    ar = c["argon_radius"]
    oa, hot, gap, cold = (int(c[k]) for k in ("open_air_particles", "hot_pore_particles", "gap_particles", "cold_pore_particles"))
    h_oa, h_hot, h_gap, h_cold, H = (c[k] for k in ("open_air_height", "hot_coating_height", "gap_height", "cold_coating_height",
                                                   "total_height"))
    R_oa, R_p, R_g = c["open_air_radius"], c["pore_coated_radius"], c["gap_radius"]
    regions = [(oa, R_oa - ar, 0 + ar, h_oa - ar),
               (hot, R_p - ar, h_oa, h_oa + h_hot),
               (gap, R_g - ar, h_oa + h_hot + ar, h_oa + h_hot + h_gap - ar),
               (cold, R_p - ar, h_oa + h_hot + h_gap, h_oa + h_hot + h_gap + h_cold),
               (n - oa - hot - gap - cold, R_oa - ar, h_oa + h_hot + h_gap + h_cold + ar, H - ar)]
    x = np.zeros(n); y = np.zeros(n); z = np.zeros(n)
    root = np.sqrt(u)
    o = 0
    for cnt, radius, z_lo, z_hi in regions:
        for k in range(o, o + cnt):
            rr = radius * root[k]
            x[k] = rr * math.cos(theta[k])
            y[k] = rr * math.sin(theta[k])
        z[o:o + cnt] = np.random.uniform(z_lo, z_hi, cnt)
        o += cnt
    speeds = maxwell.rvs(loc=0, scale=c["a_shape"], size=n)
    vx = np.empty(n); vy = np.empty(n); vz = np.empty(n)
    uni, pyuni, choice = np.random.uniform, random.uniform, np.random.choice
    signs = [-1, 1]
    for k in range(n):
        v = speeds[k]
        cos_t = uni(low=-1.0, high=1.0)
        phi = pyuni(0, math.pi)
        t = math.acos(cos_t)
        vx[k] = v * math.cos(phi) * math.sin(t)
        vy[k] = v * math.sin(phi) * math.sin(t) * choice(signs)
        vz[k] = v * math.cos(t)
    return x, y, z, vx, vy, vz
