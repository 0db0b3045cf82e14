"""The NumPy + multiprocessing mirror of the reference's parallel sweep structure (oracle/pymp_structure.py — the
`cpu_baseline_python_mp` leg of bench.py) against the reference's own step dumps: bit for bit."""
import os

import numpy as np
import pytest

from oracle import pymp_structure as PM
from tests.test_oracle_steps import STATE_KEYS, load, make_params


def test_python_mp_structure_reproduces_the_reference_dump(golden_dir):
    G = load(golden_dir, "step_pore_a.npz")
    p, dt = make_params(G, "pore")
    s = PM.PyMpStepper(p, workers=4)
    init = [G[f"s-001_{k}"] for k in STATE_KEYS]
    s.upload(*init[:10], flag=init[10])
    per = G["per_step"]
    snaps = sorted({int(k[1:5]) for k in G.files if k.startswith("s0")})
    nsteps = 6                                       # (every step forks 8 pools: keep the CPU suite short)
    ncoll = 0
    for step in range(nsteps):
        st = s.timestep(dt)
        assert st["n_pp"] + st["n_wall"] == int(per[step, 1]), (step, st, per[step, 1])
        ncoll += st["n_pp"]
        if step in snaps:
            cur = s.state()
            for k, f in zip(STATE_KEYS[:10], ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz"]):
                assert np.array_equal(cur[f], G[f"s{step:04d}_{k}"]), (step, k)
            assert np.array_equal(cur["flag"].astype(bool), G[f"s{step:04d}_full_path_traveled"].astype(bool))
    assert ncoll > 10


def test_pruned_and_full_mask_building_agree():
    """The two ways of building the per-cell masks (every cell's full expression, as the reference does, or hoisted per
    x- and (x, y)-layer) select the same cells."""
    rng = np.random.default_rng(1)
    from argon_monte_carlo_amd import params as PR
    p, c = PR.pore_params(n=400)
    g = PM.geometry_of(p)
    X = rng.uniform(-1.5e-7, 1.5e-7, 400); Y = rng.uniform(-1.5e-7, 1.5e-7, 400); Z = rng.uniform(0, 3.2e-6, 400)
    a = PM.cell_masks(X, Y, Z, g, 1, 0, 1, full=True)
    b = PM.cell_masks(X, Y, Z, g, 1, 0, 1, full=False)
    assert len(a) == len(b) and all(np.array_equal(u, v) for u, v in zip(a, b))
