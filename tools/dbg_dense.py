import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from argon_monte_carlo_amd import params as PR
from argon_monte_carlo_amd.engine import Engine
from oracle import oracle as O
n = 400
rng = np.random.default_rng(99)
p, c = PR.cube_params(n=n)
p.detect_mode = 1
cr = p.collision_range
side = (n * (4.0 / 3.0) * np.pi * (cr / 2) ** 3 / 0.2) ** (1.0 / 3.0)
pos = rng.random((3, n)) * side + 40e-9
vel = rng.normal(size=(3, n)) * 250.0
eng = Engine(p)
orc = O.Oracle(p, mode="mul")
eng.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
orc.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
dt = 2.0e-14
for s in range(6):
    try:
        st = eng.timestep(dt)
    except Exception as e:
        print("step", s, "FAILED", e); break
    rc, so = orc.timestep(dt)
    g, o = eng.download(), orc.state()
    bad = [k for k in ("x","y","z","vx","vy","vz","d","dx","dy","dz","flag") if not np.array_equal(g[k], o[k])]
    print("step", s, st, "oracle npp", so["n_pp"], "diff", bad)
eng.kernel_times()
