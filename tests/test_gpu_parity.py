"""GPU parity (-m gpu): libargonmc.so (HIP, through the C ABI) against the oracle and the reference-generated goldens.

Bars
  * vs oracle `mul` (same algorithm, exact squares — the arithmetic the kernels implement): BIT-EXACT, including the
    order of the completed-path lists.
  * vs the reference goldens (NumPy scalar `**2` = libm pow, 1-ulp different from x*x for ~0.08 % of inputs):
    identical event sets / counters; state rtol 1e-9 for one function call, 1e-6 on free-running trajectories.
"""
import os

import numpy as np
import pytest

from argon_monte_carlo_amd import ic as IC
from argon_monte_carlo_amd import params as PR

pytestmark = pytest.mark.gpu

FIELDS = ["cont", "cx", "cy", "cz", "flag", "x", "y", "z", "vx", "vy", "vz"]
STATE_KEYS = ["x_vals", "y_vals", "z_vals", "x_velocities", "y_velocities", "z_velocities", "dist_since_collision",
              "dist_x_since_collision", "dist_y_since_collision", "dist_z_since_collision", "full_path_traveled"]
SF = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz"]


@pytest.fixture(scope="module")
def Engine():
    from argon_monte_carlo_amd.engine import Engine as E
    return E


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "func_pore.npz"))


def assert_state_equal(dev, orc, ctx=""):
    for k in SF:
        assert np.array_equal(dev[k], orc[k]), (ctx, k, np.flatnonzero(dev[k] != orc[k])[:5])
    assert np.array_equal(dev["flag"].astype(bool), orc["flag"].astype(bool)), (ctx, "flag")


def paths_of(rec):
    return np.stack([rec["total"], rec["px"], rec["py"], rec["pz"]], axis=1) if len(rec) else np.zeros((0, 4))


# ---------------------------------------------------------------------------------------------- function level
def test_pairwise_cell_single_pairs(Engine, O, G):
    p, _ = PR.cell_params(n=2)
    eng = Engine(p)
    n = G["pair_in_x"].shape[0]
    for k in range(0, n, 3):
        args = [np.ascontiguousarray(G[f"pair_in_{f}"][k], dtype=np.float64) for f in FIELDS]
        args[4] = np.ascontiguousarray(G["pair_in_flag"][k].astype(np.uint8))
        ref, rpaths, rnc, rc = O.pair_cell(p, *[G[f"pair_in_{f}"][k] for f in FIELDS], mode="mul")
        paths, nc = eng.pairwise_cell(*args)
        assert nc == rnc == G["pair_ncoll"][k]
        for f, a in zip(FIELDS, args):
            assert np.array_equal(a.astype(np.float64), ref[f].astype(np.float64)), (k, f)
            np.testing.assert_allclose(a.astype(np.float64), G[f"pair_out_{f}"][k], rtol=1e-9, atol=0)
        assert np.array_equal(paths, rpaths)
    eng.close()


def test_pairwise_cell_whole_cells_with_chains(Engine, O, G):
    off = G["cell_off"]
    poff = G["cell_path_off"]
    p, _ = PR.cell_params(n=int(np.max(np.diff(off))))
    eng = Engine(p)
    for c in range(len(off) - 1):
        sl = slice(off[c], off[c + 1])
        args = [np.ascontiguousarray(G[f"cell_in_{f}"][sl], dtype=np.float64) for f in FIELDS]
        args[4] = np.ascontiguousarray(G["cell_in_flag"][sl].astype(np.uint8))
        ref, rpaths, rnc, rc = O.pair_cell(p, *[G[f"cell_in_{f}"][sl] for f in FIELDS], mode="mul")
        paths, nc = eng.pairwise_cell(*args)
        assert nc == rnc == G["cell_ncoll"][c], c
        for f, a in zip(FIELDS, args):
            assert np.array_equal(a.astype(np.float64), ref[f].astype(np.float64)), (c, f)
            np.testing.assert_allclose(a.astype(np.float64), G[f"cell_out_{f}"][sl], rtol=1e-9, atol=1e-300)
        assert np.array_equal(paths, rpaths), c          # same values in the same (reference) order
        np.testing.assert_allclose(paths, G["cell_paths"][poff[c]:poff[c + 1]], rtol=1e-9)
    eng.close()


def test_empty_and_tiny_cells(Engine):
    p, _ = PR.cell_params(n=4)
    eng = Engine(p)
    for n in (0, 1):
        args = [np.zeros(n) for _ in range(4)] + [np.zeros(n, dtype=np.uint8)] + [np.zeros(n) for _ in range(6)]
        paths, nc = eng.pairwise_cell(*args)
        assert nc == 0 and len(paths) == 0
    eng.close()


def test_zero_relative_velocity_is_reported_like_the_reference(Engine):
    """a == 0 in the contact solve: the reference raises FloatingPointError (np.seterr(all='raise'), Pore:11)."""
    from argon_monte_carlo_amd._lib import ArgonMCError
    p, _ = PR.cell_params(n=2)
    eng = Engine(p)
    cr = p.collision_range
    args = [np.zeros(2) for _ in range(4)] + [np.zeros(2, dtype=np.uint8)] + \
           [np.array([0.0, 0.5 * cr]), np.zeros(2), np.zeros(2), np.array([10.0, 10.0]), np.zeros(2), np.zeros(2)]
    with pytest.raises(ArgonMCError) as ei:
        eng.pairwise_cell(*args)
    assert ei.value.code == -5
    eng.close()


# ---------------------------------------------------------------------------------------------- step level
def load_step(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def golden_params(Gs, kind, **kw):
    K = int(Gs["meta_K"])
    sigma = 3.6 * 10**(-19) * float(Gs["meta_sigma_mult"])
    p, c = (PR.cube_params(n=K, sigma=sigma) if kind == "cube" else PR.pore_params(n=K, sigma=sigma))
    for k, v in kw.items():
        setattr(p, k, v)
    return p, float(Gs["dt"])


@pytest.mark.parametrize("name,kind", [("step_cube_a.npz", "cube"), ("step_cube_dense.npz", "cube"),
                                       ("step_pore_a.npz", "pore")])
@pytest.mark.parametrize("detect_mode", [1, 2])
def test_free_run_matches_oracle_and_reference(Engine, O, golden_dir, name, kind, detect_mode):
    Gs = load_step(golden_dir, name)
    p, dt = golden_params(Gs, kind, detect_mode=detect_mode)
    init = [Gs[f"s-001_{k}"] for k in STATE_KEYS]
    eng = Engine(p)
    eng.upload(*init[:10], flag=init[10])
    orc = O.Oracle(p, mode="mul")
    orc.upload(*init[:10], flag=init[10])
    per = Gs["per_step"]
    snaps = sorted({int(k[1:5]) for k in Gs.files if k.startswith("s0")})
    for s in range(per.shape[0]):
        st = eng.timestep(dt)
        rc, so = orc.timestep(dt)
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
            assert st[k] == so[k], (s, k, st, so)
        assert st["n_pp"] + st["n_wall"] == int(per[s, 1])              # the reference's own counter
        assert_state_equal(eng.download(), orc.state(), ctx=(name, s))   # bit-exact vs oracle, every step
        if s in snaps:                                                    # and against the reference's dump
            dev = eng.download()
            for k, f in zip(STATE_KEYS[:10], SF):
                np.testing.assert_allclose(dev[f], Gs[f"s{s:04d}_{k}"], rtol=1e-6, atol=1e-18, err_msg=f"{name} {s} {k}")
    # completed paths: same values, same (reference) order as the oracle
    rec = eng.drain_paths(sort=True)
    ro = orc.paths()
    assert len(rec) == len(ro) == Gs["completed_paths"].shape[0]
    if kind == "cube":
        assert np.array_equal(paths_of(rec), paths_of(ro))
        np.testing.assert_allclose(rec["total"], Gs["completed_paths"], rtol=1e-6)
    else:
        a, b = paths_of(rec), paths_of(ro)
        assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    # on-device histograms == np.histogram of the drained paths (Pore:575)
    counts, tot = eng.histograms()
    assert tot == len(rec)
    for row, key in enumerate(["total", "px", "py", "pz"]):
        ref_counts, _ = np.histogram(rec[key], bins=int(p.hist_bins), range=(p.hist_lo, p.hist_hi))
        assert np.array_equal(counts[row].astype(np.int64), ref_counts)
    eng.close()


def test_stages_match_oracle(Engine, O, golden_dir):
    """drift / walls / bounds / sweep as separate calls (function-level view of one Pore step)."""
    Gs = load_step(golden_dir, "step_pore_a.npz")
    p, dt = golden_params(Gs, "pore")
    init = [Gs[f"s-001_{k}"] for k in STATE_KEYS]
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(*init[:10], flag=init[10])
    orc.upload(*init[:10], flag=init[10])
    for s in range(6):
        eng.stage_drift(dt); orc.drift(dt)
        assert_state_equal(eng.download(), orc.state(), ("drift", s))
        st = eng.stage_walls(); rc, nw = orc.pore_walls()
        assert st["n_wall"] == nw
        assert_state_equal(eng.download(), orc.state(), ("walls", s))
        assert eng.stage_bounds() == orc.bounds(False)
        assert_state_equal(eng.download(), orc.state(), ("bounds", s))
        st = eng.stage_sweep(); rc, npp, _ = orc.sweep()
        assert st["n_pp"] == npp
        assert_state_equal(eng.download(), orc.state(), ("sweep", s))
        assert eng.stage_bounds() == orc.bounds(False)
        orc.step += 1
    eng.close()


# ---------------------------------------------------------------------------------------------- BASELINE-size runs
def conserved(st):
    v2 = st["vx"] ** 2 + st["vy"] ** 2 + st["vz"] ** 2
    return np.array([st["vx"].sum(), st["vy"].sum(), st["vz"].sum(), v2.sum()])


def test_cube_1e5_vs_oracle_and_invariants(Engine, O):
    """BASELINE config 2 (cube geometry, N = 100,000): GPU == oracle bit for bit over several steps; the sweep
    conserves momentum and kinetic energy; no pair is left overlapping by the sweep's own collisions."""
    p, c = PR.cube_params_for_n(100_000)
    x, y, z, vx, vy, vz = IC.cube_ic(p, c, seed=127)
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(x, y, z, vx, vy, vz)
    orc.upload(x, y, z, vx, vy, vz)
    npp = 0
    for s in range(5):
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0 and st["n_pp"] == so["n_pp"] and st["n_paths"] == so["n_paths"], (s, st, so)
        npp += st["n_pp"]
        assert_state_equal(eng.download(), orc.state(), ("cube1e5", s))
    assert npp > 100
    # invariants of the p-p sweep alone
    before = eng.download()
    st = eng.stage_sweep()
    after = eng.download()
    np.testing.assert_allclose(conserved(after), conserved(before), rtol=1e-12, atol=1e-6)
    eng.close()


def test_pore_natural_density_vs_oracle(Engine, O):
    """Pore geometry at N = 200,000 (reference density x0.36): walls + bounds + sweep, GPU == oracle bit for bit."""
    p, c = PR.pore_params(n=200_000)
    x, y, z, vx, vy, vz = IC.pore_ic(p, c, seed=17)
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(x, y, z, vx, vy, vz)
    orc.upload(x, y, z, vx, vy, vz)
    nw = 0
    for s in range(4):
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
            assert st[k] == so[k], (s, k, st, so)
        nw += st["n_wall"]
        assert_state_equal(eng.download(), orc.state(), ("pore2e5", s))
    assert nw > 50
    eng.close()


def test_run_many_steps_equals_stepwise(Engine):
    """amc_run (no host sync between steps) == the same number of amc_timestep calls."""
    p, c = PR.cube_params_for_n(20_000)
    ic = IC.cube_ic(p, c, seed=3)
    a, b = Engine(p), Engine(p)
    a.upload(*ic); b.upload(*ic)
    tot = a.run(c["dt"], 20)
    acc = 0
    for _ in range(20):
        acc += b.timestep(c["dt"])["n_pp"]
    assert tot["n_pp"] == acc
    assert_state_equal(a.download(), b.download(), "run")
    a.close(); b.close()


# ---------------------------------------------------------------------------------------------- Temperature_Pore_MC
def test_energised_walls_match_oracle_and_reference(O, golden_dir):
    """Temp:662-853 on the GPU (kernels + host RNG/mpmath sliver) == oracle `mul` bit for bit, every step, including
    the per-step z-momentum / energy transfer; and == the reference's own per-step values within 1e-6 relative."""
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
    from argon_monte_carlo_amd.engine import EnergisedEngine
    from tests.test_oracle_steps import restore_rngs, temp_setup
    Gs = load_step(golden_dir, "step_temp_a.npz")
    p, c, dt, _, energies = temp_setup(Gs)
    init = [Gs[f"s-001_{k}"] for k in STATE_KEYS]
    p.reserved0 |= 1
    eng = EnergisedEngine(p)
    eng.upload(*init[:10], flag=init[10])
    orc = O.Oracle(p, mode="mul")
    orc.upload(*init[:10], flag=init[10])
    # two independent copies of the reference's RNG streams, one per implementation
    restore_rngs(Gs)
    st_np, st_py = np.random.get_state(), random.getstate()
    rs_dev, rs_orc = np.random.RandomState(), np.random.RandomState()
    rs_dev.set_state(st_np); rs_orc.set_state(st_np)
    py_dev, py_orc = random.Random(), random.Random()
    py_dev.setstate(st_py); py_orc.setstate(st_py)
    s_dev, s_orc = DirectionSampler(rs_dev, py_dev), DirectionSampler(rs_orc, py_orc)
    per = Gs["per_step"]
    nwall = 0
    for s in range(per.shape[0]):
        st, m, ec, eh, hm, hc, hh = eng.temp_timestep(dt, s_dev, energies)
        rc, so, m2, ec2, eh2, hm2, hc2, hh2 = orc.temp_timestep(dt, s_orc, energies)
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
            assert st[k] == so[k], (s, k, st, so)
        assert (m, ec, eh, hm, hc, hh) == (m2, ec2, eh2, hm2, hc2, hh2), s
        assert st["n_pp"] + st["n_wall"] == int(per[s, 1])
        nwall += st["n_wall"]
        assert_state_equal(eng.download(), orc.state(), ctx=("temp", s))
        np.testing.assert_allclose(m, Gs["momentum"][s], rtol=1e-6, atol=0)
        np.testing.assert_allclose(ec, Gs["energy_cold"][s], rtol=1e-6, atol=0)
        np.testing.assert_allclose(eh, Gs["energy_hot"][s], rtol=1e-6, atol=0)
    assert nwall > 20
    rec, ro = eng.drain_paths(sort=True), orc.paths()
    a, b = paths_of(rec), paths_of(ro)
    assert a.shape == b.shape and np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    eng.close()


# ---------------------------------------------------------------------------------------------- the reference-named facade
def test_facade_pairwise_particles_in_cell_is_a_drop_in(G):
    """argon_monte_carlo_amd.sim.pairwise_particles_in_cell: the reference's signature, return tuple and side effects."""
    import multiprocessing
    from argon_monte_carlo_amd import sim as S
    off, poff = G["cell_off"], G["cell_path_off"]
    with pytest.raises(NameError):
        S.num_collisions_per_step = None
        S.pairwise_particles_in_cell([], [], [], [], np.ones(2, bool), *[np.zeros(2)] * 4, np.zeros(2, bool), *[np.zeros(2)] * 6)
    for c in (0, 2, 5, 9):
        sl = slice(off[c], off[c + 1])
        counter = multiprocessing.Value('i', 0)
        S.init_globals(counter)
        lists = [[], [], [], []]
        args = [G[f"cell_in_{f}"][sl].copy() for f in FIELDS]
        args[4] = args[4].astype(bool)
        res = S.pairwise_particles_in_cell(*lists, np.ones(off[c + 1] - off[c], dtype=bool), *args)
        assert len(res) == 12 and counter.value == G["cell_ncoll"][c]
        for f, a in zip(FIELDS, res[1:]):
            np.testing.assert_allclose(np.asarray(a, dtype=np.float64), G[f"cell_out_{f}"][sl], rtol=1e-9, atol=1e-300)
        got = np.array(list(zip(*lists)), dtype=np.float64).reshape(-1, 4)
        np.testing.assert_allclose(got, G["cell_paths"][poff[c]:poff[c + 1]], rtol=1e-9)


def test_facade_simulation_reproduces_the_reference_run_and_its_files(golden_dir, tmp_path):
    """Simulation('pore').timestep() from the reference's initial state: same per-step collision counter, same
    completed-path multiset, and the 8 histogram text files byte-identical to the ones the reference wrote."""
    from argon_monte_carlo_amd import outputs as OUT
    from argon_monte_carlo_amd.sim import Simulation
    Gs = load_step(golden_dir, "step_pore_a.npz")
    p, dt = golden_params(Gs, "pore")
    from argon_monte_carlo_amd import params as PR2
    _, consts = PR2.pore_params(n=int(Gs["meta_K"]), sigma=3.6 * 10**(-19) * float(Gs["meta_sigma_mult"]))
    sim = Simulation("pore", params=p, consts=consts)
    sim.set_state(*[Gs[f"s-001_{k}"] for k in STATE_KEYS])
    per = Gs["per_step"]
    for s in range(per.shape[0]):
        sim.timestep(dt)
        assert sim.num_collisions_per_step == int(per[s, 1])
    assert sim.total_cols == int(Gs["total_cols"])
    assert sorted(sim.completed_paths) == sorted(Gs["completed_paths"].tolist())
    np.testing.assert_array_equal(sim.x_vals, Gs[f"s{per.shape[0] - 1:04d}_x_vals"])
    sim.write_outputs(str(tmp_path))
    for _, fx, fy in OUT.HIST_FILES:
        for fn in (fx, fy):
            assert open(tmp_path / fn, "rb").read() == bytes(Gs["file_" + fn]), fn
    sim.close()


def test_facade_temperature_simulation_writes_the_reference_csv(golden_dir, tmp_path):
    from argon_monte_carlo_amd import params as PR2
    from argon_monte_carlo_amd.sim import TemperatureSimulation
    from tests.test_oracle_steps import restore_rngs
    Gs = load_step(golden_dir, "step_temp_a.npz")
    p, consts = PR2.pore_params(n=int(Gs["meta_K"]), sigma=3.6 * 10**(-19) * float(Gs["meta_sigma_mult"]), energised=True)
    sim = TemperatureSimulation(params=p, consts=consts)
    sim.set_state(*[Gs[f"s-001_{k}"] for k in STATE_KEYS])
    restore_rngs(Gs)                       # the module-level np.random / random streams, as the reference had them
    dt = float(Gs["dt"])
    for s in range(Gs["per_step"].shape[0]):
        sim.timestep(dt)
    np.testing.assert_allclose(np.array(sim.momentum_z_change_per_step, dtype=float), Gs["momentum"], rtol=1e-6)
    sim.write_outputs(str(tmp_path))
    assert open(tmp_path / "momentum_energy.csv", "rb").read() == bytes(Gs["file_momentum_energy.csv"])
    sim.close()


# ---------------------------------------------------------------------------------------------- BASELINE configs 3-4 at full size
def test_pore_5e5_properties(Engine):
    """BASELINE config 3 (pore geometry, N = 500,000): size-independent properties over 20 steps — the p-p sweep
    conserves momentum and kinetic energy, every particle ends inside the geometry, collisions separate the pair,
    counters are consistent, and the device histograms hold exactly the emitted paths."""
    p, c = PR.pore_params(n=500_000)
    p.reserved1 = 1
    init = IC.pore_ic(p, c, seed=17)
    eng = Engine(p)
    eng.upload(*init)
    tot = eng.run(c["dt"], 19)
    before = eng.download()
    # the sweep alone (stage call) on the current state
    eng.stage_drift(c["dt"]); sw = eng.stage_walls(); eng.stage_bounds()
    pre = eng.download()
    st = eng.stage_sweep()
    post = eng.download()
    assert st["n_pp"] > 50
    np.testing.assert_allclose(conserved(post), conserved(pre), rtol=1e-12)
    moved = np.flatnonzero((post["x"] != pre["x"]) | (post["y"] != pre["y"]) | (post["z"] != pre["z"]))
    assert st["n_pp"] < len(moved) <= 2 * st["n_pp"]                                # (a chain touches a particle twice)
    eng.stage_bounds()
    fin = eng.download()
    r2 = fin["x"] ** 2 + fin["y"] ** 2
    assert np.all(fin["z"] >= 0) and np.all(fin["z"] <= p.H) and np.all(r2 <= p.R_oa_sq * (1 + 1e-12))
    counts, npaths = eng.histograms()
    assert npaths == tot["n_paths"] + sw["n_paths"] + st["n_paths"]                  # run + the wall stage + the sweep stage
    assert counts.sum(axis=1).max() <= npaths
    eng.close()


@pytest.mark.parametrize("n", [1_000_000, 4_000_000])
def test_temp_two_steps_vs_oracle(O, n):
    """BASELINE configs[3] / [4] (energised pore, N = 1,000,000 / 4,000,000, here on one GPU): GPU == oracle bit for bit
    incl. the step's momentum/energy."""
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
    from argon_monte_carlo_amd.engine import EnergisedEngine
    p, c = PR.pore_params(n=n, energised=True)
    p.reserved0 |= 1
    init = IC.pore_ic(p, c, seed=17)
    energies = SurfaceEnergies(c)
    eng = EnergisedEngine(p)
    orc = O.Oracle(p, mode="mul", path_capacity=1 << 16)
    eng.upload(*init); orc.upload(*init)
    s_dev = DirectionSampler(np.random.RandomState(17), random.Random(17))
    s_orc = DirectionSampler(np.random.RandomState(17), random.Random(17))
    for s in range(2):
        st, m, ec, eh, *_ = eng.temp_timestep(c["dt"], s_dev, energies)
        rc, so, m2, ec2, eh2, *_ = orc.temp_timestep(c["dt"], s_orc, energies)
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
            assert st[k] == so[k], (s, k, st, so)
        assert (m, ec, eh) == (m2, ec2, eh2) and st["n_wall"] > 100
        assert_state_equal(eng.download(), orc.state(), ("temp", n, s))
    eng.close()


@pytest.mark.parametrize("mode", ["park", "force_redo", "no_park", "no_workers"])
def test_gap_case_parked_and_finished_later_equals_oracle(O, monkeypatch, mode):
    """The gap case's integrals start ahead of case 3 and the case is PARKED at its turn and finished after case 9
    (energised.drive_energised_cases).  N = 1,000,000 has about five gap hits per step: three steps == oracle bit for bit,
    incl. the step's momentum / energy sums, (a) as is, (b) with the finish-first path forced (what happens when a later
    case hits a parked particle), (c) without parking, (d) without worker processes."""
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
    from argon_monte_carlo_amd.engine import EnergisedEngine
    if mode == "force_redo":
        monkeypatch.setenv("AMC_TEMP_FORCE_GAP_REDO", "1")
    if mode == "no_park":
        monkeypatch.setenv("AMC_TEMP_NO_PARK", "1")
    if mode == "no_workers":
        monkeypatch.setenv("AMC_GAP_WORKERS", "0")
    p, c = PR.pore_params(n=1_000_000, energised=True)
    p.reserved0 |= 1
    init = IC.pore_ic(p, c, seed=17)
    energies = SurfaceEnergies(c)
    eng = EnergisedEngine(p)
    calls = {"park": 0, "finish": 0, "again": 0}
    for name in ("park", "finish"):
        orig = getattr(eng, "wall_" + name)
        setattr(eng, "wall_" + name, (lambda f, k: (lambda *a: (calls.__setitem__(k, calls[k] + 1), f(*a))[1]))(orig, name))
    orig_again = eng.wall_hits_again
    eng.wall_hits_again = lambda: (calls.__setitem__("again", calls["again"] + 1), orig_again())[1]
    orc = O.Oracle(p, mode="mul", path_capacity=1 << 16)
    eng.upload(*init); orc.upload(*init)
    s_dev = DirectionSampler(np.random.RandomState(17), random.Random(17))
    s_orc = DirectionSampler(np.random.RandomState(17), random.Random(17))
    try:
        for s in range(3):
            st, m, ec, eh, hm, hc, hh = eng.temp_timestep(c["dt"], s_dev, energies)
            rc, so, m2, ec2, eh2, hm2, hc2, hh2 = orc.temp_timestep(c["dt"], s_orc, energies)
            assert rc == 0
            for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"):
                assert st[k] == so[k], (mode, s, k, st, so)
            assert (m, ec, eh, hm, hc, hh) == (m2, ec2, eh2, hm2, hc2, hh2), (mode, s)
            assert_state_equal(eng.download(), orc.state(), ("gap", mode, s))
    finally:
        SurfaceEnergies._shutdown_pool()
    if mode == "no_park":
        assert calls["park"] == 0
    else:
        assert calls["park"] == calls["finish"] >= 2, calls         # (a step without a gap hit parks nothing)
    assert (calls["again"] >= 1) == (mode == "force_redo"), calls
    eng.close()


# ---------------------------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17])
@pytest.mark.parametrize("kind", ["cube", "pore"])
def test_tiny_and_empty_systems(Engine, O, kind, n):
    """Empty / tiny inputs through the full step (both detectors): same result as the oracle, no launch errors."""
    rng = np.random.default_rng(n + 5)
    for mode in (1, 2):
        p, c = (PR.cube_params(n=n) if kind == "cube" else PR.pore_params(n=n))
        p.detect_mode = mode
        eng = Engine(p)
        orc = O.Oracle(p, mode="mul")
        if kind == "cube":
            pos = rng.random((3, n)) * 2e-9 + 49e-9            # packed tightly so that the few particles interact
        else:
            pos = np.stack([rng.random(n) * 2e-9, rng.random(n) * 2e-9, 1.5e-6 + rng.random(n) * 2e-9])
        vel = rng.normal(size=(3, n)) * 300.0
        eng.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        orc.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        for s in range(4):
            st = eng.timestep(c["dt"])
            rc, so = orc.timestep(c["dt"])
            assert rc == 0 and st["n_pp"] == so["n_pp"] and st["n_wall"] == so["n_wall"], (kind, n, mode, s, st, so)
            assert_state_equal(eng.download(), orc.state(), (kind, n, mode, s))
        assert eng.run(c["dt"], 0) is not None
        eng.close()


@pytest.mark.parametrize("wide_waves", [0, 4, 16])
def test_dense_cluster_chains_match_oracle(Engine, O, monkeypatch, wide_waves):
    """A dense blob (volume fraction ~20 %): long collision chains and many validation rounds; both detectors agree
    with the oracle bit for bit.  With 4 or 16 waves in the wide cluster kernel instead of 512 (AMC_CW_BLOCKS, read at
    context creation) up to 64 clusters share a wave: multi-hit clusters emulated in lockstep, their work items
    interleaved — the configuration a sweep with tens of thousands of candidates runs in."""
    if wide_waves:
        monkeypatch.setenv("AMC_CW_BLOCKS", str(wide_waves))
    n = 400
    rng = np.random.default_rng(99)
    for mode in (1, 2):
        p, c = PR.cube_params(n=n)
        p.detect_mode = mode
        cr = p.collision_range
        side = (n * (4.0 / 3.0) * np.pi * (cr / 2) ** 3 / 0.2) ** (1.0 / 3.0)
        pos = rng.random((3, n)) * side + 40e-9
        vel = rng.normal(size=(3, n)) * 250.0
        eng = Engine(p)
        orc = O.Oracle(p, mode="mul")
        eng.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        orc.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        dt = 2.0e-14
        tot = 0
        for s in range(6):
            st = eng.timestep(dt)
            rc, so = orc.timestep(dt)
            assert rc == 0 and st["n_pp"] == so["n_pp"], (mode, s, st, so)
            tot += st["n_pp"]
            assert_state_equal(eng.download(), orc.state(), ("dense", mode, s))
        assert tot > 100
        eng.close()


@pytest.mark.parametrize("kind,n,sigma_mult,steps,wide_waves", [("cube", 30_000, 16.0, 25, 0), ("pore", 60_000, 30.0, 25, 0),
                                                              ("cube", 30_000, 16.0, 10, 8), ("pore", 60_000, 30.0, 10, 8)])
def test_high_collision_rate_stresses_the_wide_cluster_kernel(Engine, O, monkeypatch, kind, n, sigma_mult, steps, wide_waves):
    """A cross-section 16-30 times the reference's: several per cent of the particles collide in every step, so that
    three- to ten-particle clusters, pulled-in particles, re-emulations and cluster-cluster conflicts (the concurrent
    publish-then-probe protocol of k_clusters_wide) all happen in every sweep.  State and counters equal the oracle's bit
    for bit at every step."""
    if wide_waves:              # (AMC_CW_BLOCKS: 8 waves of 64 clusters each, several passes — see the dense-blob test)
        monkeypatch.setenv("AMC_CW_BLOCKS", str(wide_waves))
    sigma = 3.6e-19 * sigma_mult
    if kind == "cube":
        p, c = PR.cube_params_for_n(n, sigma=sigma)
        init = IC.cube_ic(p, c, seed=41)
    else:
        p, c = PR.pore_params(n=n, sigma=sigma)
        init = IC.pore_ic(p, c, seed=41)
    p.detect_mode = 1
    p.reserved1 = 1
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul", path_capacity=1 << 22)
    eng.upload(*init)
    orc.upload(*init)
    npp = rounds = 0
    for s in range(steps):
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"):
            assert st[k] == so[k], (kind, s, k, st, so)
        npp += st["n_pp"]
        rounds += st["n_rounds"]
        assert_state_equal(eng.download(), orc.state(), ("stress", kind, s))
    assert npp > (0.01 * n * steps / 2 if kind == "cube" else 60 * steps), npp      # really a high collision rate (the pore's dt shrinks with the cross-section)
    eng.close()


def test_allpairs_tiled_detector_in_the_pore_vs_oracle(Engine, O):
    """The register-tiled all-pairs kernel tests d^2 in the expanded form |ri|^2 + |rj|^2 - 2 ri.rj, which cancels; its
    threshold is raised by the cancellation bound of the grid's extent.  The pore is where that matters — 3.2 um long,
    coordinates 10^4 collision ranges from the origin — so: detect_mode 2 at N = 20,000 with a raised cross-section
    (two hundred collisions per step) against the oracle bit for bit, and against the binned detector."""
    sigma = 3.6e-19 * 80.0
    p, c = PR.pore_params(n=20_000, sigma=sigma)
    init = IC.pore_ic(p, c, seed=43)
    runs = {}
    for mode in (2, 1):
        p.detect_mode = mode
        eng = Engine(p)
        eng.upload(*init)
        orc = O.Oracle(p, mode="mul", path_capacity=1 << 22)
        orc.upload(*init)
        npp = 0
        for s in range(6):
            st = eng.timestep(c["dt"])
            rc, so = orc.timestep(c["dt"])
            assert rc == 0
            for k in ("n_pp", "n_wall", "n_paths", "n_fp_errors"):
                assert st[k] == so[k], (mode, s, k, st, so)
            npp += st["n_pp"]
            assert_state_equal(eng.download(), orc.state(), ("tiled allpairs, pore", mode, s))
        assert npp > 1000, npp
        runs[mode] = eng.download()
        eng.close()
    for k in runs[1]:
        assert np.array_equal(runs[1][k], runs[2][k]), k


def test_candidate_overflow_is_reported_not_silent(Engine):
    from argon_monte_carlo_amd._lib import ArgonMCError
    n = 3000
    rng = np.random.default_rng(1)
    p, c = PR.cube_params(n=n)
    p.detect_mode = 2
    p.max_candidates = 16
    cr = p.collision_range
    pos = rng.random((3, n)) * 6 * cr + 40e-9                   # everything overlaps everything
    vel = rng.normal(size=(3, n)) * 250.0
    eng = Engine(p)
    eng.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
    with pytest.raises(ArgonMCError) as ei:
        eng.timestep(1e-14)
    assert ei.value.code == -4
    eng.close()


# ---------------------------------------------------------------------------------------------- checkpoint / resume
def test_checkpoint_resume_continues_bit_identically(golden_dir, tmp_path):
    """save_checkpoint / load_checkpoint: a run interrupted and resumed in a new context ends in the same
    state, completed-path lists, histograms — and, for Temp, the same momentum_energy.csv — as the uninterrupted one."""
    import random
    from argon_monte_carlo_amd import params as PR2
    from argon_monte_carlo_amd.sim import Simulation, TemperatureSimulation
    Gs = load_step(golden_dir, "step_temp_a.npz")
    sigma = 3.6 * 10**(-19) * float(Gs["meta_sigma_mult"])
    init = [Gs[f"s-001_{k}"] for k in STATE_KEYS]
    dt = float(Gs["dt"])
    nsteps, cut = Gs["per_step"].shape[0], Gs["per_step"].shape[0] // 2

    def make(kind):
        if kind == "temp":
            p, consts = PR2.pore_params(n=int(Gs["meta_K"]), sigma=sigma, energised=True)
            return TemperatureSimulation(params=p, consts=consts, np_rng=np.random.RandomState(3), py_rng=random.Random(3))
        p, consts = PR2.pore_params(n=int(Gs["meta_K"]), sigma=sigma)
        return Simulation("pore", params=p, consts=consts)

    for kind in ("pore", "temp"):
        ref = make(kind)
        ref.set_state(*init)
        for s in range(nsteps):
            ref.timestep(dt)
        a = make(kind)
        a.set_state(*init)
        for s in range(cut):
            a.timestep(dt, collect_paths=(s % 2 == 0))          # some records are still on the device at the checkpoint
        ck = str(tmp_path / f"ck_{kind}.npz")
        a.save_checkpoint(ck)
        a.close()
        b = make(kind)                                           # fresh context, fresh RNG objects
        b.load_checkpoint(ck)
        for s in range(cut, nsteps):
            b.timestep(dt)
        for name in ("x_vals", "y_vals", "z_vals", "x_velocities", "y_velocities", "z_velocities", "dist_since_collision",
                     "dist_x_since_collision", "dist_y_since_collision", "dist_z_since_collision", "full_path_traveled"):
            assert np.array_equal(getattr(b, name), getattr(ref, name)), (kind, name)
        assert len(ref.completed_paths) > 0
        for name in ("completed_paths", "completed_x_paths", "completed_y_paths", "completed_z_paths"):
            assert getattr(b, name) == getattr(ref, name), (kind, name)
        assert b.total_cols == ref.total_cols and b.steps_done == ref.steps_done
        hb, hr = b.histograms()[0], ref.histograms()[0]
        for k in hr:
            assert np.array_equal(hb[k], hr[k]), (kind, k)
        if kind == "temp":
            os.makedirs(tmp_path / "b", exist_ok=True)
            os.makedirs(tmp_path / "r", exist_ok=True)
            b.write_outputs(str(tmp_path / "b"))
            ref.write_outputs(str(tmp_path / "r"))
            assert open(tmp_path / "b" / "momentum_energy.csv", "rb").read() == open(tmp_path / "r" / "momentum_energy.csv", "rb").read()
            assert any(float(v) != 0.0 for v in ref.momentum_z_change_per_step)
        b.close()
        ref.close()


# ---------------------------------------------------------------------------------------------- opt-in device-side energised sampling
def test_device_rng_energised_walls_match_oracle_on_the_same_draws(O):
    """SURVEY 8f-4 (non-parity mode): with directions / energies drawn on the GPU, everything ELSE must still be the
    reference's arithmetic — the oracle replays each case with the draws the device used and has to land on the same
    state bit for bit; the draws themselves are checked against the Philox reference (pinned to the published
    known-answer vectors), the re-emission recipe (Temp:119-141) and the mpmath gap energy (Temp:143-152)."""
    import math
    from argon_monte_carlo_amd.energised import CASES, GAP_CASE, SurfaceEnergies, device_rng_config
    from argon_monte_carlo_amd.engine import EnergisedEngine
    from tests.philox_ref import direction_draw
    p, c = PR.pore_params(n=1_000_000, energised=True)
    p.reserved0 |= 1
    init = IC.pore_ic(p, c, seed=29)
    seed = 0x1234ABCD5678
    cfg = device_rng_config(c, seed)
    energies = SurfaceEnergies(c)
    p.E_cold, p.E_hot = energies.cold, energies.hot          # Temp:83-84 (the constants the device uses for the coated walls)
    eng = EnergisedEngine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(*init)
    orc.upload(*init)
    dt = c["dt"]
    cos85 = math.cos(85 * math.pi / 180)
    n_hits = n_gap = 0
    for s in range(3):
        st, mom, cold, hot, hm, hc, hh = eng.temp_timestep_device(dt, cfg)
        # the library's per-step sums == the reference-order accumulation of the per-hit results
        from argon_monte_carlo_amd.energised import sum_device_cases
        assert (mom, cold, hot, hm, hc, hh) == sum_device_cases({case: eng.device_results(case)[1:] for case in CASES})
        orc._temp_wall_count = 0
        orc._temp_errs = 0
        orc.drift(dt, True)
        orc.temp_specular()
        omom = 0.0
        for case in CASES:
            idx, nm, cz, ok = orc.wall_hits(case)
            didx, dn, dcz, ddir, dEs = eng.device_draws(case)
            assert np.array_equal(idx, didx), (s, case)
            assert np.array_equal(nm, dn) and np.array_equal(cz, dcz), (s, case)
            dpz, dE = orc.wall_apply(case, ddir, dEs)
            ridx, rdpz, rdE, rok = eng.device_results(case)
            assert np.array_equal(ridx, idx) and np.array_equal(rok, ok)
            assert np.array_equal(rdpz, dpz) and np.array_equal(rdE, dE), (s, case)
            for k in range(len(idx)):
                if not ok[k]:
                    continue
                n_hits += 1
                omom = omom + float(dpz[k])
                d = ddir[k]
                assert abs(np.dot(d, d) - 1.0) < 1e-14
                assert np.dot(d, nm[k]) >= cos85 * (1 - 1e-12)                      # inbound, outside the grazing band
                if k < 40:                                                           # the generator itself: first accepted attempt
                    for attempt in range(64):
                        ct, phi, sg = direction_draw(seed, int(idx[k]), s, case, attempt)
                        th = math.acos(ct)
                        f = np.array([math.cos(phi) * math.sin(th), math.sin(phi) * math.sin(th) * sg, math.cos(th)])
                        dot = float(np.dot(f, nm[k]))
                        if abs(dot) < cos85:
                            continue
                        if dot < cos85:
                            f = -f
                        break
                    np.testing.assert_allclose(d, f, rtol=0, atol=1e-13)
                if case == GAP_CASE:
                    n_gap += 1
                    assert abs(dEs[k] / energies.gap(cz[k]) - 1.0) < 1e-12
                else:
                    assert dEs[k] == (energies.cold if case in (3, 7, 9) else energies.hot)
        orc.bounds(True)
        rc, npp, _ = orc.sweep()
        orc.bounds(True)
        orc.step += 1
        assert rc == 0 and st["n_pp"] == npp and st["n_wall"] == orc._temp_wall_count
        assert_state_equal(eng.download(), orc.state(), ("device rng", s))
        assert hm and float(mom) != 0.0
    assert n_hits > 500 and n_gap >= 3
    eng.close()


# ---------------------------------------------------------------------------------------------- BASELINE config 1 and 2 at full length
def test_baseline_config1_cube_1000_particles_1000_steps(Engine, O):
    """BASELINE configs[0]: Open_Air_Cube_MC geometry, 1,000 particles, 1,000 steps — the whole run on the GPU
    (`amc_run`, both detectors) against the oracle stepping the same run: state bit for bit at every 100th step and at the
    end, collision totals, and the free-path histograms from the device equal np.histogram of the oracle's path list."""
    for mode in (1, 2):
        p, c = PR.cube_params_for_n(1000)
        p.detect_mode = mode
        init = IC.cube_ic(p, c, seed=127)
        eng = Engine(p)
        orc = O.Oracle(p, mode="mul", path_capacity=1 << 18)
        eng.upload(*init)
        orc.upload(*init)
        tot = dict(n_pp=0, n_paths=0)
        otot = dict(n_pp=0, n_paths=0)
        for block in range(10):
            st = eng.run(c["dt"], 100)
            for s in range(100):
                rc, so = orc.timestep(c["dt"])
                assert rc == 0
                for k in otot:
                    otot[k] += so[k]
            for k in tot:
                tot[k] += st[k]
            assert_state_equal(eng.download(), orc.state(), ("config1", mode, block))
        assert tot == otot and tot["n_pp"] > 1000
        counts, npaths = eng.histograms()
        paths = orc.paths()
        assert npaths == len(paths) == tot["n_paths"]
        for row, key in enumerate(("total", "px", "py", "pz")):
            ref, _ = np.histogram(paths[key], bins=p.hist_bins, range=(p.hist_lo, p.hist_hi))
            assert np.array_equal(counts[row], ref.astype(np.uint64)), (mode, key)
        eng.close()


def test_baseline_config2_cube_1e5_histograms_vs_cpu(Engine, O):
    """BASELINE configs[1]: cube geometry, N = 100,000 — free-path histograms of a 60-step run (`amc_run`, no host
    synchronisation inside) equal the CPU oracle's, and so does the final state."""
    p, c = PR.cube_params_for_n(100_000)
    init = IC.cube_ic(p, c, seed=127)
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul", path_capacity=1 << 20)
    eng.upload(*init)
    orc.upload(*init)
    st = eng.run(c["dt"], 60)
    npp = 0
    for s in range(60):
        rc, so = orc.timestep(c["dt"])
        assert rc == 0
        npp += so["n_pp"]
    assert st["n_pp"] == npp and npp > 5000
    assert_state_equal(eng.download(), orc.state(), "config2")
    counts, npaths = eng.histograms()
    paths = orc.paths()
    assert npaths == len(paths) and npaths > 1000
    for row, key in enumerate(("total", "px", "py", "pz")):
        ref, _ = np.histogram(paths[key], bins=p.hist_bins, range=(p.hist_lo, p.hist_hi))
        assert np.array_equal(counts[row], ref.astype(np.uint64)), key
    eng.close()


def test_baseline_config3_pore_5e5_vs_oracle(Engine, O):
    """BASELINE configs[2]: pore geometry, N = 500,000, specular walls — GPU == oracle bit for bit over several steps
    (state, every counter), completed-path multiset included."""
    p, c = PR.pore_params(n=500_000)
    init = IC.pore_ic(p, c, seed=17)
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(*init)
    orc.upload(*init)
    tot = 0
    for s in range(4):
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
            assert st[k] == so[k], (s, k, st, so)
        tot += st["n_pp"] + st["n_wall"]
        assert_state_equal(eng.download(), orc.state(), ("config3", s))
    assert tot > 500
    rec = eng.drain_paths(sort=True)
    ref = orc.paths()
    assert len(rec) == len(ref)
    for key in ("total", "px", "py", "pz"):
        assert np.array_equal(np.sort(rec[key]), np.sort(ref[key])), key
    eng.close()


# ---------------------------------------------------------------------------------------------- count-and-continue option
def test_tolerated_fp_events_match_oracle(Engine, O):
    """amc_params.reserved1 bit0 (count and continue, Temp:340-342 semantics for every geometry): a side-wall solve
    without a real root and a pair with zero relative velocity are counted, leave the particles untouched, and the run
    goes on — identically on the GPU and in the oracle; without the option both report AMC_ERR_FP."""
    n = 64
    rng = np.random.default_rng(4)
    for mode in (1, 2):
        p, c = PR.pore_params(n=n)
        p.detect_mode = mode
        p.reserved1 = 1
        pos = np.stack([rng.random(n) * 2e-8 - 1e-8, rng.random(n) * 2e-8 - 1e-8, 1.0e-6 + rng.random(n) * 5e-7])
        vel = rng.normal(size=(3, n)) * 300.0
        # particle 0: outside the open-air radius, flying tangentially -> the backward ray never meets the wall (Pore:336)
        pos[:, 0] = [1.02 * p.R_oa, 0.0, 5.0e-8]
        vel[:, 0] = [0.0, 300.0, 10.0]
        # particles 1, 2: overlapping with identical velocities -> a == 0 (Pore:182,185)
        pos[:, 1] = [0.0, 0.0, 2.0e-6]
        pos[:, 2] = [0.3 * p.collision_range, 0.0, 2.0e-6]
        vel[:, 1] = vel[:, 2] = [50.0, -20.0, 10.0]
        eng = Engine(p)
        orc = O.Oracle(p, mode="mul")
        eng.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        orc.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        nfp = 0
        for s in range(3):
            st = eng.timestep(c["dt"])
            rc, so = orc.timestep(c["dt"])
            assert rc == 0
            for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"):
                assert st[k] == so[k], (mode, s, k, st, so)
            nfp += st["n_fp_errors"]
            assert_state_equal(eng.download(), orc.state(), ("tolerant", mode, s))
        assert nfp >= 2
        eng.close()
        # strict mode: both sides refuse
        p.reserved1 = 0
        eng = Engine(p)
        orc = O.Oracle(p, mode="mul")
        eng.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        orc.upload(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2])
        rc, so = orc.timestep(c["dt"])
        assert rc != 0
        with pytest.raises(Exception):
            eng.timestep(c["dt"])
        eng.close()


# ---------------------------------------------------------------------------------------------- large-sweep launch plan
@pytest.mark.parametrize("kind,n", [("cube", 400_000), ("pore", 1_000_000)])
def test_large_sweep_plan_with_wide_pair_kernel_vs_oracle(Engine, O, kind, n):
    """Every sweep emulates its small clusters in the wide kernel (k_clusters_wide, validation included) and leaves the
    entangled rest to the ordered workgroup; the commit rides along with the next streaming pass (pore: the bounds check
    after the sweep) or runs as k_commit when the counters are read first (cube).  State and counters equal the oracle's
    bit for bit at every step, and the profile shows that the wide kernel ran in every sweep."""
    if kind == "cube":
        p, c = PR.cube_params_for_n(n)
        init = IC.cube_ic(p, c, seed=127)
    else:
        p, c = PR.pore_params(n=n)
        init = IC.pore_ic(p, c, seed=17)
    p.reserved1 = 1
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul", path_capacity=1 << 20)
    eng.upload(*init)
    orc.upload(*init)
    eng.profile(True)
    ncand = 0
    for s in range(6):
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0
        for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"):
            assert st[k] == so[k], (kind, s, k, st, so)
        ncand = max(ncand, st["n_candidates"])
        assert_state_equal(eng.download(), orc.state(), ("large plan", kind, s))
    kt = eng.kernel_times()
    eng.profile(False)
    assert ncand > 640
    assert kt["clusters_wide"][1] >= 6 and kt["resolve"][1] >= 6, kt
    assert kt["commit"][1] >= (4 if kind == "cube" else 0), kt
    eng.close()


def test_allpairs_detector_in_front_of_the_grid_resolve_vs_oracle(Engine, O):
    """detect_mode 2 above 4096 particles: candidates from the LDS-tiled all-pairs kernel (what the reference's pairwise
    loop, Pore:168-174, maps to directly), resolved by the same wide-cluster / ordered-workgroup kernels as the binned
    detector's — state and counters equal the oracle's bit for bit, and equal the binned detector's."""
    p, c = PR.cube_params_for_n(20_000)
    init = IC.cube_ic(p, c, seed=3)
    runs = {}
    for mode in (2, 1):
        p.detect_mode = mode
        eng = Engine(p)
        eng.upload(*init)
        orc = O.Oracle(p, mode="mul")
        orc.upload(*init)
        npp = 0
        for s in range(5):
            st = eng.timestep(c["dt"])
            rc, so = orc.timestep(c["dt"])
            assert rc == 0
            for k in ("n_pp", "n_paths", "n_fp_errors"):
                assert st[k] == so[k], (mode, s, k, st, so)
            npp += st["n_pp"]
            assert_state_equal(eng.download(), orc.state(), ("allpairs detector", mode, s))
        assert npp > 100
        runs[mode] = eng.download()
        eng.close()
    for k in runs[1]:
        assert np.array_equal(runs[1][k], runs[2][k]), k


def test_candidate_overflow_in_the_large_sweep_plan_is_reported(Engine):
    """Too small a candidate buffer with the detection grid and the large-sweep plan (wide pair kernel): every step
    reports AMC_ERR_CAPACITY, nothing crashes, and the context stays usable for a following call."""
    from argon_monte_carlo_amd._lib import ArgonMCError
    p, c = PR.cube_params_for_n(400_000)
    p.detect_mode = 1
    p.max_candidates = 700                      # ~800 candidates per step at this size
    init = IC.cube_ic(p, c, seed=127)
    eng = Engine(p)
    eng.upload(*init)
    for s in range(3):                          # the first call runs the small plan (no history yet), the next ones the large one
        with pytest.raises(ArgonMCError) as ei:
            eng.timestep(c["dt"])
        assert ei.value.code == -4, (s, ei.value)
    st = eng.download()
    assert np.all(np.isfinite(st["x"]))
    eng.close()


def test_overlay_history_overflow_in_the_wide_kernel_is_reported(Engine, O, monkeypatch):
    """The host-visible contract of the wide kernel's publish-then-probe protocol at its capacity limit: with fewer
    history / overlay entries than the sweep's clusters need (AMC_MAX_HIST caps what a sweep may use), every step reports
    AMC_ERR_CAPACITY — no hang, no crash, no silent loss of a collision — and the context stays usable: the state is
    intact, and a context with the full work space resolves the same state and equals the oracle."""
    from argon_monte_carlo_amd._lib import ArgonMCError
    p, c = PR.cube_params_for_n(400_000)
    init = IC.cube_ic(p, c, seed=127)
    monkeypatch.setenv("AMC_MAX_HIST", "600")           # ~800 candidates per step own the entries 4k .. 4k + 3
    eng = Engine(p)
    monkeypatch.delenv("AMC_MAX_HIST")
    eng.upload(*init)
    for s in range(3):
        with pytest.raises(ArgonMCError) as ei:
            eng.timestep(c["dt"])
        assert ei.value.code == -4, (s, ei.value)
    st = eng.download()
    assert all(np.all(np.isfinite(st[k])) for k in ("x", "y", "z", "vx", "vy", "vz"))
    eng.close()
    # the same workload with the full work space: equal to the oracle
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(*init)
    orc.upload(*init)
    for s in range(2):
        st = eng.timestep(c["dt"])
        rc, so = orc.timestep(c["dt"])
        assert rc == 0 and st["n_pp"] == so["n_pp"] > 0
    dev, ref = eng.download(), orc.state()
    for k in ("x", "y", "z", "vx", "vy", "vz", "d"):
        assert np.array_equal(dev[k], ref[k]), k
    eng.close()


# ---------------------------------------------------------------------------------------------- device-side initial conditions
def test_device_initial_conditions_cube(Engine):
    """amc_init_synthetic, cube: the documented mapping of Philox numbers to positions / velocities
    (positions bit for bit, velocities to the libm/ocml difference), bounds, moments, determinism."""
    import math
    from tests.philox_ref import ic_uniforms
    n, seed = 200_000, 12345
    p, c = PR.cube_params_for_n(n)
    eng = Engine(p)
    eng.init_synthetic(IC.device_ic_config(p, c, seed, "cube"))
    s = eng.download()
    a = c["a_shape"]
    for q in (0, 1, 77_777, n - 1):
        u = ic_uniforms(seed, q)
        assert s["x"][q] == u[0] * p.cube_x and s["y"][q] == u[1] * p.cube_y and s["z"][q] == u[2] * p.cube_z
        m0, m1 = math.sqrt(-2.0 * math.log(1.0 - u[3])), math.sqrt(-2.0 * math.log(1.0 - u[5]))
        want = (a * m0 * math.cos(2 * math.pi * u[4]), a * m0 * math.sin(2 * math.pi * u[4]), a * m1 * math.cos(2 * math.pi * u[6]))
        np.testing.assert_allclose([s["vx"][q], s["vy"][q], s["vz"][q]], want, rtol=1e-12)
    for k, L in (("x", p.cube_x), ("y", p.cube_y), ("z", p.cube_z)):
        assert s[k].min() >= 0.0 and s[k].max() < L
        assert abs(s[k].mean() / L - 0.5) < 5 * (1 / 12) ** 0.5 / n ** 0.5
    for k in ("vx", "vy", "vz"):
        assert abs(s[k].mean()) < 5 * a / n ** 0.5                       # N(0, a^2) components
        assert abs(s[k].var() / a ** 2 - 1.0) < 5 * (2.0 / n) ** 0.5
    for k in ("d", "dx", "dy", "dz", "flag"):
        assert not s[k].any()
    eng2 = Engine(p)
    eng2.init_synthetic(IC.device_ic_config(p, c, seed, "cube"))
    assert_state_equal(eng2.download(), s, "device ic: same seed")
    eng2.init_synthetic(IC.device_ic_config(p, c, seed + 1, "cube"))
    assert not np.array_equal(eng2.download()["x"], s["x"])
    st = eng.run(c["dt"], 5)                                             # and the state is usable: a few steps run
    assert st["n_pp"] > 0 and st["flags"] == 0
    eng.close(); eng2.close()


def test_device_initial_conditions_pore(Engine):
    """amc_init_synthetic, pore: every particle inside its region with the reference's insets (Pore:120-139), region
    populations as ic.pore_regions says, radial density uniform (mean r^2 = R^2 / 2), and the bounds check finds nobody outside."""
    n, seed = 300_000, 17
    p, c = PR.pore_params(n=n)
    eng = Engine(p)
    eng.init_synthetic(IC.device_ic_config(p, c, seed, "pore"))
    s = eng.download()
    counts, regions = IC.pore_regions(p, c)
    o = 0
    for cnt, (R, zlo, zhi) in zip(counts, regions):
        sl = slice(o, o + cnt)
        r2 = s["x"][sl] ** 2 + s["y"][sl] ** 2
        assert r2.max() <= R * R * (1 + 1e-12) and s["z"][sl].min() >= zlo and s["z"][sl].max() <= zhi
        assert abs(r2.mean() / (R * R) - 0.5) < 5 * (1 / 12) ** 0.5 / cnt ** 0.5
        assert abs((s["z"][sl].mean() - zlo) / (zhi - zlo) - 0.5) < 5 * (1 / 12) ** 0.5 / cnt ** 0.5
        o += cnt
    assert o == n
    assert eng.stage_bounds() == 0                                         # nobody starts outside (Pore:354-375 finds nothing)
    st = eng.run(c["dt"], 3)
    assert st["flags"] == 0 and st["n_pp"] > 0 and st["n_wall"] > 0
    eng.close()


# ---------------------------------------------------------------------------------------------- the overlapped run
@pytest.mark.parametrize("mode", ["2", "1"])
@pytest.mark.parametrize("name,kind", [("step_cube_dense.npz", "cube"), ("step_cube_a.npz", "cube"), ("step_pore_a.npz", "pore")])
def test_overlapped_run_equals_the_oracle_on_the_reference_runs(Engine, O, golden_dir, monkeypatch, name, kind, mode):
    """amc_run overlaps the streaming pass of step s + 1 with the resolve of sweep s.  From the reference's
    initial states — the dense cube among them, where sweeps pull uninvolved particles into clusters in most steps, i.e.
    particles the early pass has already advanced are advanced again from the sweep's result and filed under extra list
    nodes — the state after run(k) equals the oracle's bit for bit, so do counters, histograms and the completed paths.
    mode 2 = the same kernels in order on one stream, mode 1 = on two streams."""
    Gs = load_step(golden_dir, name)
    sigma = 3.6 * 10**(-19) * float(Gs["meta_sigma_mult"])
    K = int(Gs["meta_K"])
    p, c = (PR.cube_params(n=K, sigma=sigma) if kind == "cube" else PR.pore_params(n=K, sigma=sigma))
    dt = float(Gs["dt"])
    p.detect_mode = 1                           # (the binned detector also below 4,096 particles: the overlapped run needs the grid)
    init = [Gs[f"s-001_{k}"] for k in STATE_KEYS]
    monkeypatch.setenv("AMC_OVERLAP", mode)
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul")
    eng.upload(*init[:10], init[10])
    orc.upload(*init[:10], flag=init[10])
    tot = {}
    refiled = 0
    for chunk in (7, 2, 16):                    # odd and even numbers of steps: either state buffer ends up current
        st = eng.run(dt, chunk)
        so = {}
        for _ in range(chunk):
            rc, s1 = orc.timestep(dt)
            assert rc == 0
            for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
                so[k] = so.get(k, 0) + s1[k]
        for k in so:
            assert st[k] == so[k], (chunk, k, st, so)
        assert_state_equal(eng.download(), orc.state(), (name, mode, chunk))
    ov = eng.overlap_stats()
    assert ov["steps"] == 25 and ov["mode"] == int(mode)
    if name == "step_cube_dense.npz":
        assert ov["refiled"] > 0, ov          # the case the extra list nodes exist for did occur
    rec = eng.drain_paths()
    ref = orc.paths()
    got = paths_of(rec)
    exp = np.stack([ref["total"], ref["px"], ref["py"], ref["pz"]], axis=1) if len(ref["total"]) else np.zeros((0, 4))
    assert got.shape == exp.shape
    assert np.array_equal(got[np.lexsort(got.T[::-1])], exp[np.lexsort(exp.T[::-1])])
    eng.close()


@pytest.mark.parametrize("workload,n,steps", [("cube", 100_000, 120), ("pore", 300_000, 60)])
def test_overlapped_run_equals_the_plain_sequence_at_size(Engine, O, monkeypatch, workload, n, steps):
    """The same at BASELINE-like sizes against BOTH the plain sequence (AMC_OVERLAP=0) and the oracle: state, counters and
    device histograms."""
    if workload == "cube":
        p, c = PR.cube_params_for_n(n)
        init = IC.cube_ic(p, c, seed=127)
    else:
        p, c = PR.pore_params(n=n)
        init = IC.pore_ic(p, c, seed=17)
    p.reserved1 = 1
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("AMC_OVERLAP", mode)
        eng = Engine(p)
        eng.upload(*init)
        st = eng.run(c["dt"], steps)
        res[mode] = (st, eng.download(), eng.histograms(), eng.overlap_stats())
        eng.close()
    assert res["0"][3]["steps"] == 0 and res["1"][3]["steps"] == steps
    for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_candidates"):
        assert res["0"][0][k] == res["1"][0][k], (k, res["0"][0], res["1"][0])
    assert_state_equal(res["1"][1], res["0"][1], workload)
    assert np.array_equal(res["0"][2][0], res["1"][2][0]) and res["0"][2][1] == res["1"][2][1]
    orc = O.Oracle(p, mode="mul")
    orc.upload(*init)
    for _ in range(steps):
        rc, _ = orc.timestep(c["dt"])
        assert rc == 0
    assert_state_equal(res["1"][1], orc.state(), workload + " vs oracle")


@pytest.mark.parametrize("kind,n,sigma_mult,keep,steps", [("pore", 60_000, 30.0, 3, 14), ("pore", 200_000, 1.0, 8, 20),
                                                         ("cube", 30_000, 16.0, 4, 13)])
def test_kept_lists_equal_the_oracle(Engine, O, monkeypatch, kind, n, sigma_mult, keep, steps):
    """AMC_LIST_KEEP=K (the pore's default is K = 4, here other K and the cube): the per-cell lists of a full build are kept for K - 1 more steps — a particle
    still in its cell refreshes the position in its node, one that left poisons its node and files a new one from its
    wave's pool.  Several cycles, run() and timestep() mixed (both go through the same lists), high collision rates so
    that clusters, pulled-in particles and validation walk lists full of dead nodes: state, counters and completed paths
    equal the oracle's bit for bit at every step."""
    monkeypatch.setenv("AMC_LIST_KEEP", str(keep))
    sigma = 3.6e-19 * sigma_mult
    if kind == "cube":
        p, c = PR.cube_params_for_n(n, sigma=sigma)
        init = IC.cube_ic(p, c, seed=43)
    else:
        p, c = PR.pore_params(n=n, sigma=sigma)
        init = IC.pore_ic(p, c, seed=43)
    p.detect_mode = 1
    p.reserved1 = 1
    eng = Engine(p)
    orc = O.Oracle(p, mode="mul", path_capacity=1 << 22)
    eng.upload(*init)
    orc.upload(*init)
    npp = 0
    s = 0
    while s < steps:
        k = 3 if s % 5 == 2 else 1                  # (a run of three steps now and then: the cycle goes on across calls)
        st = eng.run(c["dt"], k) if k > 1 else eng.timestep(c["dt"])
        so = None
        tot = dict.fromkeys(("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors"), 0)
        for _ in range(k):
            rc, so = orc.timestep(c["dt"])
            assert rc == 0
            for key in tot:
                tot[key] += so[key]
        for key in tot:
            assert st[key] == tot[key], (kind, s, key, st, tot)
        npp += st["n_pp"]
        s += k
        assert_state_equal(eng.download(), orc.state(), ("kept lists", kind, keep, s))
    assert npp > 0
    # an upload starts a new cycle (full build), and the context goes on
    eng.upload(*init)
    orc.upload(*init)
    eng.timestep(c["dt"])
    orc.timestep(c["dt"])
    assert_state_equal(eng.download(), orc.state(), ("kept lists after upload", kind))
    eng.close()
