"""The all-cores variant of the oracle's sweep (OpenMP over the disjoint cells of a colour group = the reference's
Pool.starmap structure) against the serial, pinned oracle: same state, same counters."""
import numpy as np

from argon_monte_carlo_amd import ic as IC, params as PR
from oracle import oracle as O

KEYS = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz", "flag"]


def test_parallel_pore_sweep_equals_the_serial_oracle():
    p, c = PR.pore_params(n=60_000)
    p.reserved1 = 1
    init = IC.pore_ic(p, c, seed=5)
    a, b = O.Oracle(p, mode="mul"), O.Oracle(p, mode="mul")
    a.upload(*init)
    b.upload(*init)
    npp = 0
    for s in range(6):
        rc1, s1 = a.timestep(c["dt"])
        rc2, s2 = b.timestep_par(c["dt"])
        assert rc1 == 0 and rc2 == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
            assert s1[k] == s2[k], (s, k, s1, s2)
        npp += s1["n_pp"]
        sa, sb = a.state(), b.state()
        for k in KEYS:
            assert np.array_equal(sa[k], sb[k]), (s, k)
    assert npp > 20


def test_parallel_cube_colouring_is_a_valid_sweep():
    """Cube geometry with the Pore script's colouring (a timing baseline, not the reference's serial order): elastic
    collisions conserve momentum and kinetic energy, and about as many happen as in the serial order."""
    p, c = PR.cube_params_for_n(30_000)
    init = IC.cube_ic(p, c, seed=9)
    a, b = O.Oracle(p, mode="mul"), O.Oracle(p, mode="mul")
    a.upload(*init)
    b.upload(*init)
    n1 = n2 = 0
    for s in range(5):
        rc1, s1 = a.timestep(c["dt"])
        rc2, s2 = b.timestep_par(c["dt"])
        assert rc1 == 0 and rc2 == 0
        n1 += s1["n_pp"]; n2 += s2["n_pp"]
    assert n1 > 100 and abs(n1 - n2) <= 0.05 * n1
    sb = b.state()
    v2 = sb["vx"] ** 2 + sb["vy"] ** 2 + sb["vz"] ** 2
    v20 = np.asarray(init[3]) ** 2 + np.asarray(init[4]) ** 2 + np.asarray(init[5]) ** 2
    np.testing.assert_allclose(v2.sum(), v20.sum(), rtol=1e-12)
