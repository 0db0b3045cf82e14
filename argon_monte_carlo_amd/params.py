"""Constants of the three reference geometries, evaluated with the reference's own expressions.

Every value below is computed in the same order of operations as the line it cites (Cube = Open_Air_Cube_MC.py,
Pore = Open_Air_Pore_MC.py, Temp = Temperature_Pore_MC.py under /root/reference) so that the resulting doubles are
bit-identical to the reference's module constants (checked in tests/test_host.py against values captured from the
imported reference).  The result is an ``AmcParams`` (ctypes mirror of ``amc_params``, include/argonmc.h).
"""
from __future__ import annotations

import math

import numpy as np

from ._abi import (AMC_GEOM_CELL, AMC_GEOM_CUBE, AMC_GEOM_PORE, AMC_GEOM_PORE_ENERGISED, AmcParams)

# ---- physics shared by the three scripts (Cube:42-57, Pore:49-64, Temp:56-71) --------------------------------
ARGON_MASS = 6.63 * 10**-26
AR_MOLAR_MASS = 0.039948
MOLECULES_PER_MOLE = 6.02214179 * 10**23
IDEAL_GAS_CONST = 8.3145
BOLTZMAN = 1.38 * 10**(-23)            # Cube:46, Pore:53
BOLTZMAN_TEMP = 1.38064852 * 10**(-23)  # Temp:60
TEMP_AMBIENT = 298
SIGMA = 3.6 * 10**(-19)
PRESSURE = 101325
NUM_BINS = 200                          # Pore:93
HIST_RANGE = (0.0, 10**-6)              # Pore:575


def _physics(sigma=SIGMA, boltzman=BOLTZMAN, temp=TEMP_AMBIENT, pressure=PRESSURE):
    argon_radius = float(np.sqrt(sigma / (4 * np.pi)))                       # Pore:56
    collision_radius = argon_radius * 1                                      # Pore:57
    collision_range = collision_radius * 2                                   # Pore:58
    lambda_mfp = float(boltzman * temp / (np.sqrt(2) * sigma * pressure))    # Pore:60
    v_mean = float(np.sqrt(3 * IDEAL_GAS_CONST * temp / AR_MOLAR_MASS))      # Pore:61
    a_shape = float(np.sqrt(boltzman * temp / ARGON_MASS))                   # Pore:63
    tau = lambda_mfp / v_mean                                                # Pore:72
    return dict(argon_radius=argon_radius, collision_range=collision_range, lambda_mfp=lambda_mfp, v_mean=v_mean,
                a_shape=a_shape, tau=tau)


def _common(p: AmcParams, ph, n, device=0):
    p.n = int(n)
    p.collision_range = ph["collision_range"]
    p.argon_mass = ARGON_MASS
    p.argon_radius = ph["argon_radius"]
    p.hist_bins = NUM_BINS
    p.hist_lo, p.hist_hi = HIST_RANGE
    p.device = device
    p.struct_size = __import__("ctypes").sizeof(AmcParams)


class SimConstants(dict):
    """dict of derived host-side constants (dt, a_shape, region populations ...) next to the C struct."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def cell_params(sigma=SIGMA, n=0, device=0):
    """A single cell: only pairwise_particles_in_cell (Pore:160-255) runs."""
    ph = _physics(sigma)
    p = AmcParams()
    _common(p, ph, n, device)
    p.geometry = AMC_GEOM_CELL
    p.nx = p.ny = p.nz = 1
    return p, SimConstants(ph)


def cube_params(n=None, cube_side=100 * 10**-9, n_sub=15, sigma=SIGMA, pressure=PRESSURE, steps_per_mft=25,
                device=0):
    """Open_Air_Cube_MC.py constants (Cube:26-74).  ``n=None`` derives num_molecules from the ideal-gas law as
    Cube:55-57 does; passing ``n`` re-parameterises the particle count and leaves everything else."""
    ph = _physics(sigma, BOLTZMAN, TEMP_AMBIENT, pressure)
    cube_x = cube_y = cube_z = cube_side                                     # Cube:26-28
    cube_volume = cube_x * cube_y * cube_z
    dx, dy, dz = cube_x / n_sub, cube_y / n_sub, cube_z / n_sub              # Cube:33-35
    num_moles = cube_volume * pressure / (IDEAL_GAS_CONST * TEMP_AMBIENT)    # Cube:55
    num_molecules = int(np.round(num_moles * MOLECULES_PER_MOLE).astype(int))  # Cube:57
    if n is None:
        n = num_molecules
    Nmft = 20
    num_timesteps = Nmft * steps_per_mft                                     # Cube:63
    dt = Nmft * ph["tau"] / num_timesteps                                    # Cube:64
    p = AmcParams()
    _common(p, ph, n, device)
    p.geometry = AMC_GEOM_CUBE
    p.nx = p.ny = p.nz = n_sub
    p.dx, p.dy, p.dz = dx, dy, dz
    p.overlap_x, p.overlap_y, p.overlap_z = dx / 10, dy / 10, dz / 10        # Cube:36-38
    p.cube_x, p.cube_y, p.cube_z = cube_x, cube_y, cube_z
    c = SimConstants(ph)
    c.update(dt=dt, num_timesteps=num_timesteps, num_molecules=num_molecules, cube_side=cube_side, seed=127)
    return p, c


def cube_params_for_n(n, sigma=SIGMA, device=0, cell_target=6.65e-9):
    """Cube geometry holding ``n`` particles at the reference number density:
    side L = (n/n0)^(1/3), cells of about 6.65 nm with overlap cell/10, dt = tau/25 as Cube:61-64."""
    ph = _physics(sigma)
    n0 = PRESSURE / (BOLTZMAN * TEMP_AMBIENT)  # molecules per m^3 used for the synthetic box size only
    side = (n / n0) ** (1.0 / 3.0)
    n_sub = max(1, int(round(side / cell_target)))
    p, c = cube_params(n=n, cube_side=side, n_sub=n_sub, sigma=sigma, device=device)
    return p, c


def _pore_geometry():
    g = SimConstants()
    g.pore_coated_radius = 30 * 10 ** -9                                     # Pore:25
    g.gap_radius = g.pore_coated_radius + 4 * 10 ** -9                       # Pore:26
    g.pore_height = 3000 * 10 ** -9                                          # Pore:27
    g.hot_coating_height = 30 * 10 ** -9                                     # Pore:28
    g.gap_height = g.hot_coating_height                                      # Pore:29
    g.cold_coating_height = g.pore_height - g.hot_coating_height - g.gap_height  # Pore:30
    g.open_air_radius = 5 * g.pore_coated_radius                             # Pore:35
    g.open_air_height = 100 * 10 ** -9                                       # Pore:36
    cyl = lambda r, h: np.pi * r ** 2 * h                                    # utils.py:3-4
    g.hot_volume = cyl(g.pore_coated_radius, g.hot_coating_height)
    g.gap_volume = cyl(g.gap_radius, g.gap_height)
    g.cold_volume = cyl(g.pore_coated_radius, g.cold_coating_height)
    g.open_air_volume = cyl(g.open_air_radius, g.open_air_height)
    g.total_volume = g.hot_volume + g.gap_volume + g.cold_volume + g.open_air_volume * 2   # Pore:38
    g.total_height = g.pore_height + g.open_air_height * 2                   # Pore:39
    g.num_x_subdivions, g.num_y_subdivions, g.num_z_subdivions = 7, 7, 148   # Pore:41-43
    g.dx = g.open_air_radius / g.num_x_subdivions                            # Pore:44
    g.dy = g.open_air_radius / g.num_y_subdivions
    g.dz = g.total_height / g.num_z_subdivions                               # Pore:46
    return g


def _region_counts(g, num_molecules):
    """Pore:79-83 region populations (open air bottom, hot, gap, cold, open air top + remainder)."""
    fl = lambda v: int(np.floor(v).astype(int))
    open_air = fl(num_molecules * (g.open_air_volume / g.total_volume))
    cold = fl(num_molecules * (g.cold_volume / g.total_volume))
    hot = fl(num_molecules * (g.hot_volume / g.total_volume))
    gap = fl(num_molecules * (g.gap_volume / g.total_volume))
    remaining = num_molecules - gap - hot - cold - open_air * 2
    return dict(open_air_particles=open_air, cold_pore_particles=cold, hot_pore_particles=hot, gap_particles=gap,
                remaining_particles=remaining)


def pore_params(n=None, sigma=SIGMA, energised=False, device=0):
    """Open_Air_Pore_MC.py (Pore:25-86) or, with energised=True, Temperature_Pore_MC.py (Temp:30-105) constants.
    ``n`` re-parameterises the particle count (a pressure scale); geometry, cells and dt stay."""
    boltz = BOLTZMAN_TEMP if energised else BOLTZMAN
    temp = 298.0 if energised else TEMP_AMBIENT
    ph = _physics(sigma, boltz, temp, PRESSURE)
    g = _pore_geometry()
    ar = ph["argon_radius"]
    num_moles = g.total_volume * PRESSURE / (IDEAL_GAS_CONST * temp)         # Pore:62
    num_molecules = int(np.round(num_moles * MOLECULES_PER_MOLE).astype(int))  # Pore:64
    if n is None:
        n = num_molecules
    Nmft, NMFT_slice = 20, 1000                                              # Pore:73-74
    num_timesteps = Nmft * NMFT_slice
    dt = Nmft * ph["tau"] / num_timesteps                                    # Pore:76
    p = AmcParams()
    _common(p, ph, n, device)
    p.geometry = AMC_GEOM_PORE_ENERGISED if energised else AMC_GEOM_PORE
    p.nx, p.ny, p.nz = g.num_x_subdivions, g.num_y_subdivions, g.num_z_subdivions
    p.dx, p.dy, p.dz = g.dx, g.dy, g.dz
    p.overlap_x = p.overlap_y = p.overlap_z = ph["collision_range"]          # Pore:527-529
    p.R_oa, p.R_p, p.R_g = g.open_air_radius, g.pore_coated_radius, g.gap_radius
    p.R_oa_c = g.open_air_radius - ar                                        # Pore:67
    p.R_g_c = g.gap_radius - ar                                              # Pore:68
    p.R_p_c = g.pore_coated_radius - ar                                      # Pore:69
    p.H, p.h_oa = g.total_height, g.open_air_height
    p.z_cold = g.total_height - g.open_air_height                            # Pore:457
    p.z_gap_bottom = g.open_air_height + g.hot_coating_height                # Pore:465 / Temp:45
    if energised:
        p.z_gap_top = g.open_air_height + g.hot_coating_height + g.gap_height          # Temp:46
    else:
        p.z_gap_top = g.total_height - g.open_air_height - g.cold_coating_height        # Pore:465
    if energised:
        p.oob_z_lo_fix = 50 * 10 ** -9                                       # Temp:599
        p.oob_z_hi_fix = g.total_height - (50 * 10 ** -9)                    # Temp:602
    else:
        p.oob_z_lo_fix = 10 * ar                                             # Pore:358
        p.oob_z_hi_fix = 10 * ar                                             # Pore:361
    p.R_oa_sq = g.open_air_radius ** 2                                       # Pore:363
    p.R_g_sq = g.gap_radius ** 2                                             # Pore:367
    p.R_p_sq = g.pore_coated_radius ** 2                                     # Pore:371
    p.z_oob_hot_top = g.open_air_height + g.hot_coating_height               # Pore:371
    p.z_oob_gap_top = g.open_air_height + g.hot_coating_height + g.gap_height  # Pore:371
    c = SimConstants(ph)
    c.update(g)
    c.update(_region_counts(g, n))
    c.update(dt=dt, num_timesteps=num_timesteps, num_molecules=num_molecules, seed=17)
    if energised:
        gap_bottom, gap_top = p.z_gap_bottom, p.z_gap_top
        p.t_z3_cold = g.total_height - g.open_air_height + ar                # Temp:708
        p.t_z3_hot = g.open_air_height - ar                                  # Temp:713
        p.t_zgap_lo = gap_bottom + ar                                        # Temp:720
        p.t_zgap_hi = gap_top - ar                                           # Temp:720
        p.R_g_c_sq = float(np.float64(p.R_g_c) ** 2)                         # Temp:721
        p.R_p_c_sq = float(np.float64(p.R_p_c) ** 2)                         # Temp:728
        p.alpha_coated, p.alpha_gap = 0.95, 0.8                              # Temp:76-77
        p.cos85 = math.cos(85 * math.pi / 180)                               # Temp:136
        c.update(t_cold=293.0, t_hot=353.0, t_debye_graphene=1813.0, t_debye_alumina=980.0,
                 num_atoms_unitcell_graphene=2, num_atoms_unitcell_alumina=10, boltzman=boltz)
    return p, c


def params_to_dict(p: AmcParams):
    return p.as_dict()
