"""CPU-only checks of the host layer: output writers against the files the reference itself wrote (tests/golden),
the C-ABI library's exports, parameter evaluation, and the no-fallback rule."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from argon_monte_carlo_amd import _lib, outputs as OUT, params as PR
from argon_monte_carlo_amd._abi import AMC_ABI_VERSION, AmcParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "argonmc.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*\*?\s*(amc_[a-z_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 30
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.amc_abi_version() == AMC_ABI_VERSION == 2


def test_a_library_that_does_not_match_the_sources_is_refused(monkeypatch):
    """libargonmc.so travels with the working tree, not with git: the loader compares the digest build() left next to
    it with the sources in the tree and refuses a stale library instead of silently running old kernels."""
    assert _lib.stale_sources() == []
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "source_digest", lambda: "0" * 64)
    with pytest.raises(_lib.ArgonMCError) as ei:
        _lib.load()
    assert ei.value.code == -3 and "rebuild" in str(ei.value)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from argon_monte_carlo_amd.engine import Engine
    with pytest.raises(_lib.ArgonMCError) as ei:
        Engine(PR.cube_params(n=100)[0])
    assert ei.value.code == -2


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under argon_monte_carlo_amd/ may import, link or dlopen it."""
    pkg = os.path.join(ROOT, "argon_monte_carlo_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|liboracle|oracle/|orc_(pow|mul)_", re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, fn)).read()
                m = bad.search(src)
                assert m is None or fn == "amc_device.h" and "orc_" in m.group(0), (fn, m.group(0) if m else None)
    src = open(os.path.join(pkg, "csrc", "Makefile")).read()
    assert "oracle" not in src


def test_params_match_reference_constants(golden_dir):
    G = np.load(os.path.join(golden_dir, "consts.npz"))
    p, c = PR.pore_params()
    for k in ["dx", "dz", "dt", "argon_radius", "collision_range", "a_shape", "num_molecules", "lambda_mfp", "tau",
              "open_air_particles", "cold_pore_particles", "hot_pore_particles", "gap_particles", "remaining_particles"]:
        assert float(G["pore_" + k]) == float(c[k]), k
    assert p.z_gap_top == float(G["pore_z_gap_top_expr"]) and p.z_gap_bottom == float(G["pore_z_gap_bottom_expr"])
    assert p.R_oa_sq == float(G["pore_R_oa_sq"]) and p.R_g_sq == float(G["pore_R_g_sq"]) and p.R_p_sq == float(G["pore_R_p_sq"])
    assert p.R_oa_c == float(G["pore_open_air_collision_radius"]) and p.R_g_c == float(G["pore_gap_collision_radius"])
    assert p.R_p_c == float(G["pore_pore_collision_radius"])
    pt, ct = PR.pore_params(energised=True)
    for k in ["dt", "a_shape", "num_molecules", "lambda_mfp"]:
        assert float(G["temp_" + k]) == float(ct[k]), k
    assert pt.R_g_c_sq == float(G["temp_R_g_c_sq"]) and pt.R_p_c_sq == float(G["temp_R_p_c_sq"])
    assert pt.z_gap_top == float(G["temp_gap_top_height"]) and pt.z_gap_bottom == float(G["temp_gap_bottom_height"])
    pc, cc = PR.cube_params()
    for k in ["dt", "num_molecules", "a_shape", "tau"]:
        assert float(G["cube_" + k]) == float(cc[k]), k
    assert pc.overlap_x == float(G["cube_collision_x_overlap"]) and pc.dx == float(G["cube_dx"])
    assert C.sizeof(AmcParams) == pc.struct_size


def test_histogram_files_byte_identical_to_the_reference(golden_dir, tmp_path):
    """The 8 text files the (patched) reference run wrote are reproduced from its completed-path lists."""
    G = np.load(os.path.join(golden_dir, "step_pore_a.npz"))
    dens = {}
    edges = None
    for key, name in (("total", "completed_paths"), ("x", "completed_x_paths"), ("y", "completed_y_paths"),
                      ("z", "completed_z_paths")):
        dens[key], edges = OUT.density_from_paths(G[name])
        # the device path produces integer counts; density from counts must be the same floats
        counts, _ = np.histogram(G[name], bins=200, range=(0, 10**-6))
        d2, e2 = OUT.density_from_counts(counts)
        assert np.array_equal(d2, dens[key]) and np.array_equal(e2, edges)
    OUT.write_histograms(str(tmp_path), dens, edges)
    for _, fx, fy in OUT.HIST_FILES:
        for fn in (fx, fy):
            assert open(tmp_path / fn, "rb").read() == bytes(G["file_" + fn]), fn


def test_direction_sampler_sign_draw_equals_numpy_choice():
    """energised.DirectionSampler replaces np.random.choice([-1, 1]) (Temp:124) by randint(0, 2): same values, same
    position of the legacy Mersenne Twister stream afterwards."""
    a, b = np.random.RandomState(5), np.random.RandomState(5)
    for i in range(20000):
        assert a.choice([-1, 1]) == (-1, 1)[b.randint(0, 2)]
        if i % 7 == 0:
            assert a.uniform(low=-1.0, high=1.0) == b.uniform(low=-1.0, high=1.0)
    assert a.get_state()[2] == b.get_state()[2] and np.array_equal(a.get_state()[1], b.get_state()[1])


def _random_normals(n, seed):
    rs = np.random.RandomState(seed)
    nm = rs.standard_normal((n, 3))
    nm /= np.linalg.norm(nm, axis=1)[:, None]
    nm[::5] = (0.0, 0.0, 1.0)                   # the planes of cases 3, 4, 6
    nm[1::5] = (0.0, 0.0, -1.0)
    nm[2::5, 2] = 0.0                           # the gap's side wall: radial normals
    nm[2::5] /= np.linalg.norm(nm[2::5], axis=1)[:, None]
    return nm


def test_host_direction_helper_equals_the_library_calls():
    """amc_host_directions (the C helper behind DirectionSampler.sample_case) against the per-hit calls of np.random /
    random / math / np.dot it replaces (Temp:119-141): same directions bit for bit, and both Mersenne Twisters left
    exactly where the library calls leave them — with RandomState / random.Random instances and with the module-level
    generators the reference uses, inside a session and outside."""
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler
    assert DirectionSampler.fast_path(), "the helper did not reproduce the library calls on this interpreter"
    nm = _random_normals(700, 1)
    ok = np.ones(len(nm), dtype=np.uint8)
    ok[3::13] = 0
    slow = DirectionSampler(np.random.RandomState(17), random.Random(17), fast=False)
    fast = DirectionSampler(np.random.RandomState(17), random.Random(17))
    a1, b1 = slow.sample_case(nm, ok), fast.sample_case(nm, ok)
    assert np.array_equal(a1.view(np.uint64), b1.view(np.uint64))
    assert not a1[ok == 0].any() and (np.abs(np.linalg.norm(a1[ok == 1], axis=1) - 1) < 1e-12).all()
    with fast.session():                        # a step: several cases on one borrowed state
        b2 = fast.sample_case(nm[:40], ok[:40])
        b3 = fast.sample_case(nm[40:45], ok[40:45])
        assert len(fast.sample_case(nm[:0], ok[:0])) == 0
    a2, a3 = slow.sample_case(nm[:40], ok[:40]), slow.sample_case(nm[40:45], ok[40:45])
    assert np.array_equal(a2, b2) and np.array_equal(a3, b3)
    sa, sb = slow.np_rng.get_state(), fast.np_rng.get_state()
    assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    assert slow.py_rng.getstate() == fast.py_rng.getstate()
    # rejections happened (|d.n| < cos 85 deg redraws) and flips happened
    assert (np.einsum("ij,ij->i", a1, nm)[ok == 1] >= slow.cos85).all()
    # the module-level streams, seeded like the reference seeds them (Temp:108-109)
    np.random.seed(17); random.seed(17)
    c1 = DirectionSampler(fast=False).sample_case(nm[:90], ok[:90])
    tail1 = (np.random.uniform(), random.random(), np.random.randint(0, 2))
    np.random.seed(17); random.seed(17)
    d1 = DirectionSampler().sample_case(nm[:90], ok[:90])
    tail2 = (np.random.uniform(), random.random(), np.random.randint(0, 2))
    assert np.array_equal(c1, d1) and tail1 == tail2


def test_host_direction_helper_crosses_a_twister_refill():
    """624 words per refill, 3 + 2 words per attempt: long cases cross many refills of both generators; positions 0 and
    624 at entry are the edge cases of the state layout."""
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler
    nm = _random_normals(4000, 2)
    ok = np.ones(len(nm), dtype=np.uint8)
    for burn in (0, 1, 623, 624, 625):
        slow = DirectionSampler(np.random.RandomState(99), random.Random(99), fast=False)
        fast = DirectionSampler(np.random.RandomState(99), random.Random(99))
        for s in (slow, fast):
            for _ in range(burn):
                s.np_rng.randint(0, 2)          # one 32-bit word each
                s.py_rng.getrandbits(32)
        assert np.array_equal(slow.sample_case(nm, ok), fast.sample_case(nm, ok))
        assert slow.py_rng.getstate() == fast.py_rng.getstate()
        assert np.array_equal(slow.np_rng.get_state()[1], fast.np_rng.get_state()[1])
        assert slow.np_rng.get_state()[2] == fast.np_rng.get_state()[2]


def test_host_direction_helper_rejects_broken_states():
    """amc_host_directions validates what it is handed (positions outside 0..624, missing buffers, a BLAS form without a
    function) instead of reading past the key arrays."""
    lib = _lib.load()
    key = np.zeros(624, dtype=np.uint32)
    nm = np.array([[0.0, 0.0, 1.0]])
    out = np.zeros((1, 3))
    u32p, dp = C.POINTER(C.c_uint32), C.POINTER(C.c_double)

    def call(np_pos, py_pos, kind=0, fn=None, normals=nm, dirs=out, n=1):
        a, b = C.c_int32(np_pos), C.c_int32(py_pos)
        return lib.amc_host_directions(key.ctypes.data_as(u32p), C.byref(a), key.ctypes.data_as(u32p), C.byref(b),
                                       normals.ctypes.data_as(dp) if normals is not None else None, None, n, 0.0871557, 3.141592653589793,
                                       kind, fn, dirs.ctypes.data_as(dp) if dirs is not None else None)

    assert call(625, 0) != 0 and call(0, -1) != 0
    assert call(0, 0, kind=2, fn=None) != 0
    assert call(0, 0, normals=None) != 0 and call(0, 0, dirs=None) != 0
    assert call(0, 0, n=0) == 0


def test_direction_sampler_falls_back_for_other_generators():
    """Anything that is not one of the two Mersenne Twisters keeps the per-hit loop (no state to borrow)."""
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler

    class Mine(random.Random):
        pass

    nm = _random_normals(20, 3)
    ok = np.ones(20, dtype=np.uint8)
    s = DirectionSampler(np.random.RandomState(1), Mine(1))
    assert not s._states_borrowable()
    ref = DirectionSampler(np.random.RandomState(1), random.Random(1), fast=False)
    assert np.array_equal(s.sample_case(nm, ok), ref.sample_case(nm, ok))


def test_gap_energies_in_worker_processes_equal_the_serial_ones(monkeypatch):
    """SurfaceEnergies.gap_many spreads the mpmath integrals of a case over forked workers: same values, same order."""
    from argon_monte_carlo_amd.energised import SurfaceEnergies
    _, c = PR.pore_params(n=100, energised=True)
    en = SurfaceEnergies(c)
    z0 = c["open_air_height"] + c["hot_coating_height"]
    zs = [z0 + f * c["gap_height"] for f in (0.0, 0.123, 0.5, 0.77, 1.0)]
    want = [en.gap(z) for z in zs]
    monkeypatch.setenv("AMC_GAP_WORKERS", "3")
    try:
        assert en.gap_many(zs) == want
        assert SurfaceEnergies._pool is not None and SurfaceEnergies._pool.alive()
        pids = [w[0] for w in SurfaceEnergies._pool.workers]
        assert len(pids) == 3
        # a worker holds nothing of the parent's but its two pipe ends (no inherited stdout pipe, socket or device file)
        for pid in pids:
            links = [os.readlink(f"/proc/{pid}/fd/{f}") for f in os.listdir(f"/proc/{pid}/fd")]
            assert sum(l.startswith("pipe:") for l in links) == 2 and all(l.startswith("pipe:") or l == "/dev/null" for l in links), links
        assert en.gap_many(zs[:1]) == want[:1] and en.gap_many([]) == []
    finally:
        SurfaceEnergies._shutdown_pool()
    for pid in pids:
        assert not os.path.exists(f"/proc/{pid}") or open(f"/proc/{pid}/stat").read().split()[2] == "Z"
    monkeypatch.setenv("AMC_GAP_WORKERS", "0")
    assert en.gap_many(zs) == want and SurfaceEnergies._pool is None


def test_gap_workers_can_be_started_ahead_of_the_gpu_context(monkeypatch):
    """SurfaceEnergies(consts, start_workers=True) forks the workers in the constructor — the facade and bench.py build it
    before amc_create, so no worker is ever forked from a process with an initialised HIP runtime — and gap_many then
    uses exactly those processes."""
    import inspect
    from argon_monte_carlo_amd import sim as SIM
    from argon_monte_carlo_amd.energised import SurfaceEnergies
    _, c = PR.pore_params(n=100, energised=True)
    monkeypatch.setenv("AMC_GAP_WORKERS", "2")
    try:
        en = SurfaceEnergies(c, start_workers=True)
        assert SurfaceEnergies._pool is not None and SurfaceEnergies._pool.alive()
        pids = [w[0] for w in SurfaceEnergies._pool.workers]
        z0 = c["open_air_height"] + c["hot_coating_height"]
        zs = [z0 + f * c["gap_height"] for f in (0.2, 0.6)]
        assert en.gap_many(zs) == [en.gap(z) for z in zs]
        assert [w[0] for w in SurfaceEnergies._pool.workers] == pids        # the early workers served the case
    finally:
        SurfaceEnergies._shutdown_pool()
    # the facade creates the energies (and their workers) before the engine
    src = inspect.getsource(SIM.TemperatureSimulation.__init__)
    assert 0 < src.index("SurfaceEnergies(consts, start_workers=True)") < src.index("EnergisedEngine(params)")


def test_gap_energies_started_early_equal_the_serial_ones(monkeypatch):
    """gap_start / gap_finish (the integrals run in the workers while the host handles other cases): same values, same
    order, several batches in a row; without workers gap_start declines."""
    from argon_monte_carlo_amd.energised import SurfaceEnergies
    _, c = PR.pore_params(n=100, energised=True)
    en = SurfaceEnergies(c)
    z0 = c["open_air_height"] + c["hot_coating_height"]
    zs = [z0 + f * c["gap_height"] for f in (0.05, 0.3, 0.31, 0.62, 0.9, 0.97, 0.5)]
    want = [en.gap(z) for z in zs]
    monkeypatch.setenv("AMC_GAP_WORKERS", "3")
    try:
        for batch in (zs, zs[:1], zs[2:5]):
            h = en.gap_start(batch)
            assert h is not None
            assert en.gap_finish(h) == [want[zs.index(z)] for z in batch]
        assert en.gap_start([]) is None
    finally:
        SurfaceEnergies._shutdown_pool()
    monkeypatch.setenv("AMC_GAP_WORKERS", "0")
    assert en.gap_start(zs) is None


def test_gap_energies_survive_a_dead_worker(monkeypatch):
    """A worker that went away (killed, out of memory) must not cost the step: gap_many notices, does the integrals in
    this process and stops using workers."""
    import signal
    from argon_monte_carlo_amd.energised import SurfaceEnergies
    _, c = PR.pore_params(n=100, energised=True)
    en = SurfaceEnergies(c)
    z0 = c["open_air_height"] + c["hot_coating_height"]
    zs = [z0 + f * c["gap_height"] for f in (0.1, 0.4, 0.9)]
    want = [en.gap(z) for z in zs]
    monkeypatch.setenv("AMC_GAP_WORKERS", "2")
    try:
        assert en.gap_many(zs) == want
        os.kill(SurfaceEnergies._pool.workers[0][0], signal.SIGKILL)
        assert en.gap_many(zs) == want                  # (either path: restarted workers or this process)
        assert en.gap_many(zs) == want
    finally:
        SurfaceEnergies._shutdown_pool()


def test_direct_debye_integrand_equals_the_operator_form():
    """SurfaceEnergies integrates x**3 / (exp(x) - 1) through libmp's functions directly (no mpf operator wrappers): the
    gap energies and the two plate energies equal the ones the operator form gives — the reference's lambda, Temp:80 —
    over the whole gap."""
    from mpmath import exp
    from argon_monte_carlo_amd.energised import SurfaceEnergies, _direct_debye_integrand
    assert _direct_debye_integrand() is not None
    _, c = PR.pore_params(n=100, energised=True)
    fast = SurfaceEnergies(c)
    assert fast._integrand.__name__ == "direct"
    slow = SurfaceEnergies(c)
    slow._integrand = lambda x: (x ** 3) / (exp(x) - 1)
    z0 = c["open_air_height"] + c["hot_coating_height"]
    rng = np.random.default_rng(5)
    for z in np.concatenate([[z0, z0 + c["gap_height"]], z0 + c["gap_height"] * rng.random(25)]):
        assert fast.gap(z) == slow.gap(z)
    # the plate energies were integrated with the direct form too: compare with the operator form from scratch
    from mpmath import quad
    f = lambda x: (x ** 3) / (exp(x) - 1)                                    # noqa: E731
    q = quad(f, [0, c["t_debye_graphene"] / c["t_cold"]])
    want = 9 * c["t_cold"] * c["num_atoms_unitcell_graphene"] * c["boltzman"] * (c["t_cold"] / c["t_debye_graphene"]) ** 3 * q
    assert float(want) == fast.cold


def test_philox_reference_known_answers():
    """tests/philox_ref.py against the known-answer vectors of the Random123 distribution (kat_vectors, philox4x32 10)."""
    from tests.philox_ref import philox4x32_10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
