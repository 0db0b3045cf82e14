"""Host-side sliver of the energised-wall path of Temperature_Pore_MC.py (hard part 3).

Two things cannot move to the GPU without changing results: the re-emission direction is drawn from NumPy's legacy
global Mersenne Twister AND Python's ``random`` module in strict particle order with a data-dependent rejection loop
(Temp:119-141), and the gap wall's surface energy is an ``mpmath.quad`` per hit (Temp:143-152).  They are restated here
with the same library calls, so a caller that seeds ``np.random.seed(17); random.seed(17)`` like the reference
(Temp:108-109) consumes the identical streams.  Everything else of the wall handlers (masks, contact points, energy
accommodation, bookkeeping) runs in the HIP kernels; ``amc_wall_hits`` / ``amc_wall_apply`` are the hand-over.
"""
from __future__ import annotations

import random as _py_random
from math import acos, cos, pi, sin

import numpy as np

CASES = (3, 4, 5, 6, 7, 8, 9)          # evaluation order, Temp:708-751 (see include/argonmc.h, amc_wall_hits)
COLD_CASES = (3, 7, 9)                  # contribute to energy_change_cold_in_step (Temp:711,738,753)
HOT_CASES = (4, 6, 8)                   # contribute to energy_change_hot_in_step (Temp:716,732,747)
GAP_CASE = 5                            # momentum only (Temp:722-723)


class DirectionSampler:
    """random_components / random_inbounds_direction (Temp:119-141) on the given generators.
    Defaults: the module-level ``np.random`` and ``random`` streams, exactly what the reference uses."""

    def __init__(self, np_rng=None, py_rng=None):
        self.np_rng = np.random if np_rng is None else np_rng
        self.py_rng = _py_random if py_rng is None else py_rng
        self.cos85 = cos(85 * pi / 180)

    def random_components(self, r):
        costheta = self.np_rng.uniform(low=-1.0, high=1.0)                   # Temp:120
        phi = self.py_rng.uniform(0, pi)                                     # Temp:121
        theta = acos(costheta)
        Fx = r * cos(phi) * sin(theta)
        # Temp:124 is np.random.choice([-1, 1]): for a two-element list the legacy generator draws randint(0, 2) — same
        # value, same stream consumption (tests/test_host.py checks the equivalence), a quarter of the call overhead
        Fy = r * sin(phi) * sin(theta) * (-1, 1)[self.np_rng.randint(0, 2)]
        Fz = r * cos(theta)
        return Fx, Fy, Fz

    def random_inbounds_direction(self, norm):
        while True:                                                          # Temp:133-141
            nx, ny, nz = self.random_components(1)
            new_direction = np.array([nx, ny, nz])
            if abs(np.dot(new_direction, norm)) < self.cos85:
                continue
            if np.dot(new_direction, norm) < self.cos85:
                new_direction = -new_direction
            break
        return new_direction


class SurfaceEnergies:
    """surface_energy_cold / surface_energy_hot (Temp:80-84) and surface_energy_gap(z) (Temp:143-152) via mpmath,
    rounded to double (mpmath works at 53 bits there, so the mpf values ARE doubles)."""

    def __init__(self, consts):
        from mpmath import exp, quad
        self._quad = quad
        self._integrand = lambda x: (x ** 3) / (exp(x) - 1)                  # Temp:80
        c = consts
        self.boltzman = c["boltzman"]
        self.t_cold, self.t_hot = c["t_cold"], c["t_hot"]
        self.t_debye_graphene, self.t_debye_alumina = c["t_debye_graphene"], c["t_debye_alumina"]
        self.n_graphene, self.n_alumina = c["num_atoms_unitcell_graphene"], c["num_atoms_unitcell_alumina"]
        self.gap_height = c["gap_height"]
        self.gap_bottom_height = c["open_air_height"] + c["hot_coating_height"]            # Temp:45
        q_cold = quad(self._integrand, [0, self.t_debye_graphene / self.t_cold])           # Temp:81
        q_hot = quad(self._integrand, [0, self.t_debye_graphene / self.t_hot])             # Temp:82
        self.cold_mpf = 9 * self.t_cold * self.n_graphene * self.boltzman * (self.t_cold / self.t_debye_graphene) ** 3 * q_cold
        self.hot_mpf = 9 * self.t_hot * self.n_graphene * self.boltzman * (self.t_hot / self.t_debye_graphene) ** 3 * q_hot
        self.cold, self.hot = float(self.cold_mpf), float(self.hot_mpf)

    def gap(self, z_value):
        z_value = float(z_value)
        m = (self.t_cold - self.t_hot) / self.gap_height                                   # Temp:144
        t_gap = m * (z_value - self.gap_bottom_height) + self.t_hot                        # Temp:145
        q = self._quad(self._integrand, [0, self.t_debye_alumina / t_gap])                 # Temp:148
        return float(9 * t_gap * self.n_alumina * self.boltzman * (t_gap / self.t_debye_alumina) ** 3 * q)   # Temp:152


def sequential_sum(values):
    """momentem_z_change_in_case += ... (Temp:389): a left-to-right sum starting from the int 0."""
    s = 0
    for v in values:
        s = s + float(v)
    return s


def format_mpf(value, is_zero_int):
    """What pandas writes for one momentum_energy.csv cell: str(mpf) (15 significant digits) or the int 0 of a step
    without a contributing hit (Temp:366-367, 685-687)."""
    if is_zero_int:
        return "0"
    from mpmath import mpf
    return str(mpf(float(value)))


def drive_energised_cases(hooks, sampler, energies):
    """The host loop over the seven energised cases of one step (Temp:705-758).

    ``hooks.wall_hits(case)`` -> (idx, normals[n,3], contact_z[n], ok[n]) in ascending particle index;
    ``hooks.wall_apply(case, dirs[n,3], Es[n])`` -> (dpz[n], dE[n]).  Returns (momentum, energy_cold, energy_hot,
    had_momentum, had_cold, had_hot) for the step, accumulated in the reference's order."""
    mom = cold = hot = 0
    had_m = had_c = had_h = False
    for case in CASES:
        idx, normals, contact_z, ok = hooks.wall_hits(case)
        n = len(idx)
        if n == 0:
            continue
        dirs = np.zeros((n, 3))
        Es = np.zeros(n)
        for k in range(n):
            if not ok[k]:
                continue                                   # the reference's try-block fails before any RNG draw
            dirs[k] = sampler.random_inbounds_direction(np.array(normals[k]))
            Es[k] = (energies.gap(contact_z[k]) if case == GAP_CASE else
                     energies.cold if case in COLD_CASES else energies.hot)
        dpz, dE = hooks.wall_apply(case, dirs, Es)
        good = [k for k in range(n) if ok[k]]
        m_case = sequential_sum(dpz[k] for k in good)
        mom = mom + m_case
        had_m = had_m or len(good) > 0
        if case in COLD_CASES:
            cold = cold + sequential_sum(dE[k] for k in good)
            had_c = had_c or len(good) > 0
        elif case in HOT_CASES:
            hot = hot + sequential_sum(dE[k] for k in good)
            had_h = had_h or len(good) > 0
    return mom, cold, hot, had_m, had_c, had_h


def device_rng_config(consts, seed, n_gl=32):
    """amc_temp_rng for the opt-in, NON-PARITY device-side sampling (include/argonmc.h): Philox seed, the constants of
    surface_energy_gap (Temp:143-152) and a Gauss-Legendre rule for its Debye integral."""
    from ._abi import AmcTempRng
    import ctypes as C
    x, w = np.polynomial.legendre.leggauss(int(n_gl))
    g = AmcTempRng()
    g.struct_size, g.n_gl, g.seed = C.sizeof(AmcTempRng), int(n_gl), int(seed) & 0xFFFFFFFFFFFFFFFF
    g.t_cold, g.t_hot = consts["t_cold"], consts["t_hot"]
    g.gap_height = consts["gap_height"]
    g.gap_bottom_height = consts["open_air_height"] + consts["hot_coating_height"]          # Temp:45
    g.t_debye_alumina, g.n_alumina = consts["t_debye_alumina"], consts["num_atoms_unitcell_alumina"]
    g.boltzman = consts["boltzman"]
    for i in range(int(n_gl)):
        g.gl_x[i], g.gl_w[i] = float(x[i]), float(w[i])
    return g


def sum_device_cases(results):
    """Per-step sums from the per-hit results of the seven cases, accumulated like drive_energised_cases does
    (left-to-right in ascending particle index within a case, cases in Temp:708-751 order).  ``results[case]`` =
    (dpz, dE, ok)."""
    mom = cold = hot = 0
    had_m = had_c = had_h = False
    for case in CASES:
        dpz, dE, ok = results[case]
        good = [k for k in range(len(ok)) if ok[k]]
        if len(ok) == 0:
            continue
        mom = mom + sequential_sum(dpz[k] for k in good)
        had_m = had_m or len(good) > 0
        if case in COLD_CASES:
            cold = cold + sequential_sum(dE[k] for k in good)
            had_c = had_c or len(good) > 0
        elif case in HOT_CASES:
            hot = hot + sequential_sum(dE[k] for k in good)
            had_h = had_h or len(good) > 0
    return mom, cold, hot, had_m, had_c, had_h
