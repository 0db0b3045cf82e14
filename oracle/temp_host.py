"""The oracle's OWN host loop of the energised walls of Temperature_Pore_MC.py.

TEST INFRASTRUCTURE ONLY.  The product has the same job in argon_monte_carlo_amd/energised.py; this file is a separate
restatement (written from SURVEY.md App. B.2 and from reading Temp:119-152, 705-758 — no code shared with, or imported
from, the product), so that "HIP path == oracle" on the energised geometry compares two implementations of the host
sliver as well.  It is pinned by tests/golden/step_temp_a.npz (the reference's own dump incl. its RNG states) through
tests/test_oracle_steps.py::test_temp_free_run_bit_exact.

What it restates:
  * random_components(r) (Temp:119-126): cos(theta) from np.random.uniform(-1, 1), phi from random.uniform(0, pi), the
    sign of the y component from np.random.choice([-1, 1]) — called exactly like that, on the two Mersenne Twisters;
  * random_inbounds_direction(normal) (Temp:132-141): redraw while |d.n| < cos 85 deg, flip when d.n < cos 85 deg;
  * surface_energy_cold / _hot (Temp:80-84) and surface_energy_gap(z) (Temp:143-152): Debye integrals by mpmath.quad;
  * the per-step loop over the seven case ids in evaluation order (Temp:705-758): hits in ascending particle index, one
    direction (and, in the gap, one energy) per hit that has a contact point, sums accumulated left to right from the
    integer 0 — momentum over all cases, energy into the cold or the hot total by surface.
"""
from __future__ import annotations

import math

import numpy as np

ORDER = (3, 4, 5, 6, 7, 8, 9)      # case-3 cold plane, case-3 hot plane, gap side wall, case-5 bottom, case-5 top, case-6 hot, case-6 cold
COLD_SURFACES = {3, 7, 9}
HOT_SURFACES = {4, 6, 8}
GAP = 5


class Directions:
    """Re-emission directions drawn from a NumPy legacy generator and a Python `random` generator."""

    def __init__(self, np_gen, py_gen):
        self.np_gen, self.py_gen = np_gen, py_gen
        self.limit = math.cos(math.radians(85))

    def components(self, r=1):
        c = self.np_gen.uniform(-1.0, 1.0)
        phi = self.py_gen.uniform(0, math.pi)
        th = math.acos(c)
        sign = self.np_gen.choice([-1, 1])
        return np.array([r * math.cos(phi) * math.sin(th), r * math.sin(phi) * math.sin(th) * sign, r * math.cos(th)])

    def inbounds(self, normal):
        normal = np.asarray(normal, dtype=np.float64)
        while True:
            d = self.components(1)
            s = np.dot(d, normal)
            if abs(s) < self.limit:
                continue
            return -d if s < self.limit else d


class Energies:
    """Surface energies from the constants of the run (a dict or any object with the same attributes)."""

    def __init__(self, k):
        import mpmath
        get = (lambda name: k[name]) if isinstance(k, dict) else (lambda name: getattr(k, name))
        self.mp = mpmath
        self.kB = get("boltzman")
        self.Tc, self.Th = get("t_cold"), get("t_hot")
        self.theta_g, self.theta_a = get("t_debye_graphene"), get("t_debye_alumina")
        try:
            self.n_g, self.n_a = get("num_atoms_unitcell_graphene"), get("num_atoms_unitcell_alumina")
        except (KeyError, AttributeError):
            self.n_g, self.n_a = get("n_graphene"), get("n_alumina")
        self.h_gap = get("gap_height")
        try:
            self.z_gap0 = get("open_air_height") + get("hot_coating_height")
        except (KeyError, AttributeError):
            self.z_gap0 = get("gap_bottom_height")
        self.cold = float(self._debye(self.Tc, self.theta_g, self.n_g))
        self.hot = float(self._debye(self.Th, self.theta_g, self.n_g))

    def _debye(self, T, theta, natoms):
        f = lambda x: (x ** 3) / (self.mp.exp(x) - 1)
        return 9 * T * natoms * self.kB * (T / theta) ** 3 * self.mp.quad(f, [0, theta / T])

    def gap(self, z):
        z = float(z)
        slope = (self.Tc - self.Th) / self.h_gap
        T = slope * (z - self.z_gap0) + self.Th
        return float(self._debye(T, self.theta_a, self.n_a))


def run_cases(walls, directions, energies):
    """One step's energised cases.  `walls.wall_hits(case)` -> (particle indices, inward normals, contact z, has-contact
    flags) in ascending particle index; `walls.wall_apply(case, directions, surface energies)` -> (dp_z, dE) per hit.
    Returns (momentum, energy_cold, energy_hot, any_momentum, any_cold, any_hot)."""
    tot_p, tot_c, tot_h = 0, 0, 0
    seen_p = seen_c = seen_h = False
    for case in ORDER:
        idx, normals, cz, has = walls.wall_hits(case)
        if len(idx) == 0:
            continue
        d = np.zeros((len(idx), 3))
        e = np.zeros(len(idx))
        for q in range(len(idx)):
            if has[q]:
                d[q] = directions.inbounds(normals[q])
                e[q] = energies.gap(cz[q]) if case == GAP else (energies.cold if case in COLD_SURFACES else energies.hot)
        dp, dE = walls.wall_apply(case, d, e)
        p_case, e_case, any_hit = 0, 0, False
        for q in range(len(idx)):
            if has[q]:
                p_case = p_case + float(dp[q])
                e_case = e_case + float(dE[q])
                any_hit = True
        tot_p = tot_p + p_case
        seen_p = seen_p or any_hit
        if case in COLD_SURFACES:
            tot_c = tot_c + e_case
            seen_c = seen_c or any_hit
        elif case in HOT_SURFACES:
            tot_h = tot_h + e_case
            seen_h = seen_h or any_hit
    return tot_p, tot_c, tot_h, seen_p, seen_c, seen_h
