"""The oracle pinned at the reference's OWN parameters: tests/golden/step_{cube,pore}_natural.npz are
dumps of Open_Air_Cube_MC.py (N = 24,627, sigma x 1, dt = tau / 25) and Open_Air_Pore_MC.py (N = 557,649, sigma = 3.6e-19)
run AS WRITTEN except for the loop bound and the dump hook (oracle/gen_golden.py --only cube_natural / pore_natural).  Per
step they hold the collision counter, the number of completed paths, the SHA-256 of every state array and the event set
(particles whose velocity changed); the oracle's `pow` variant free-runs from the same initial state and has to reproduce
all of it bit for bit — at these sizes its counting sort by cell and its 32-bit member indices are exercised as they are
at BASELINE sizes."""
import os

import numpy as np
import pytest

from argon_monte_carlo_amd import params as PR
from oracle import oracle as O
from tests.natural_ic import reference_pore_initial_state, sha

STATE_KEYS = ["x_vals", "y_vals", "z_vals", "x_velocities", "y_velocities", "z_velocities", "dist_since_collision",
              "dist_x_since_collision", "dist_y_since_collision", "dist_z_since_collision", "full_path_traveled"]


def load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def natural_params(G, kind):
    p, c = (PR.cube_params() if kind == "cube" else PR.pore_params())
    assert int(p.n) == int(G["num_molecules"]) and p.collision_range == float(G["collision_range"]) and c["dt"] == float(G["dt"])
    return p, c


def initial_state(G, kind, c):
    """(10 float arrays, flag) the reference started its time loop from; for the pore regenerated from the seeds and
    checked against the stored hashes."""
    n = int(G["num_molecules"])
    if kind == "cube":
        init = [G[f"s-001_{k}"] for k in STATE_KEYS]
        return init[:10], init[10]
    x, y, z, vx, vy, vz = reference_pore_initial_state(c)
    for k, a in zip(STATE_KEYS[:6], (x, y, z, vx, vy, vz)):
        if not np.array_equal(sha(a), G[f"h-001_{k}"]):
            pytest.skip(f"this NumPy / SciPy build does not regenerate the reference's initial {k} (hash differs)")
    zeros = [np.zeros(n) for _ in range(4)]
    return [x, y, z, vx, vy, vz] + zeros, np.zeros(n, dtype=np.uint8)


def check_against_reference(G, step_fn, state_fn, nsteps, exact_hashes, v0=None):
    """Per step: counters and event set equal the reference's; with exact_hashes every state array's SHA-256 too."""
    per = G["per_step"]
    total_paths = 0
    prev_v = None if v0 is None else np.stack(v0)
    for s in range(nsteps):
        st = step_fn()
        assert st["n_pp"] + st["n_wall"] == int(per[s, 1]), (s, st, per[s, 1])
        total_paths += st["n_paths"]
        assert total_paths == int(per[s, 2]), (s, total_paths, per[s, 2])
        cur = state_fn()
        if exact_hashes:
            for k, f in zip(STATE_KEYS[:10], O.STATE_FIELDS):
                assert np.array_equal(sha(cur[f]), G[f"h{s:04d}_{k}"]), (s, k)
            assert np.array_equal(sha(np.asarray(cur["flag"]).astype(bool)), G[f"h{s:04d}_full_path_traveled"]), s
        v = np.stack([cur["vx"], cur["vy"], cur["vz"]])
        if prev_v is not None:
            ev = np.nonzero((v != prev_v).any(axis=0))[0]
            assert np.array_equal(ev.astype(np.int32), G[f"ev{s:04d}"]), (s, len(ev), len(G[f"ev{s:04d}"]))
        prev_v = v
    return total_paths


def run_oracle(G, kind, mode):
    p, c = natural_params(G, kind)
    init, flag = initial_state(G, kind, c)
    o = O.Oracle(p, mode=mode)
    o.upload(*init, flag=flag)

    def step():
        rc, st = o.timestep(c["dt"])
        assert rc == 0
        return st
    return o, p, c, init, flag, step


def test_cube_natural_oracle_equals_the_reference(golden_dir):
    G = load(golden_dir, "step_cube_natural.npz")
    o, p, c, init, flag, step = run_oracle(G, "cube", "pow")
    n = check_against_reference(G, step, o.state, G["per_step"].shape[0], exact_hashes=True, v0=init[3:6])
    r = o.paths()
    assert n == len(G["completed_paths"]) > 100
    for k, g in (("total", "completed_paths"), ("px", "completed_x_paths"), ("py", "completed_y_paths"), ("pz", "completed_z_paths")):
        assert np.array_equal(r[k], G[g]), k         # (the cube is serial: list order is deterministic)


def test_pore_natural_oracle_equals_the_reference(golden_dir):
    G = load(golden_dir, "step_pore_natural.npz")
    o, p, c, init, flag, step = run_oracle(G, "pore", "pow")
    n = check_against_reference(G, step, o.state, G["per_step"].shape[0], exact_hashes=True, v0=init[3:6])
    r = o.paths()
    got = np.stack([r["total"], r["px"], r["py"], r["pz"]], axis=1)
    exp = np.stack([G["completed_paths"], G["completed_x_paths"], G["completed_y_paths"], G["completed_z_paths"]], axis=1)
    assert got.shape == exp.shape and n == len(exp)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], exp[np.lexsort(exp.T[::-1])])     # (Manager().list order is scheduling dependent)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cube", "pore"])
def test_hip_at_the_references_own_parameters(golden_dir, kind):
    """HIP path at N = 24,627 / 557,649, the reference's sigma and dt, from the reference's own initial state: bit for
    bit equal to orc_mul after every step, and per step the same collision counter, completed-path count and event set
    as the reference itself (its `pow` arithmetic differs from x*x in the last bit of some values, which cannot change
    who collides within these few steps)."""
    from argon_monte_carlo_amd.engine import Engine
    G = load(golden_dir, f"step_{kind}_natural.npz")
    p, c = natural_params(G, kind)
    init, flag = initial_state(G, kind, c)
    nsteps = G["per_step"].shape[0]
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(*init, flag)
    orc.upload(*init, flag=flag)
    state = {}

    def step():
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_paths"):
            assert st[k] == so[k], (k, st, so)
        dev, ref = eng.download(), orc.state()
        for k in O.STATE_FIELDS:
            assert np.array_equal(dev[k], ref[k]), k
        assert np.array_equal(np.asarray(dev["flag"]).astype(bool), np.asarray(ref["flag"]).astype(bool))
        state["cur"] = dev
        return st
    total = check_against_reference(G, step, lambda: state["cur"], nsteps, exact_hashes=False, v0=init[3:6])
    assert total == len(G["completed_paths"])
