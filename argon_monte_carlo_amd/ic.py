"""Synthetic initial conditions.  The reference's own generators (Pore:106-158, Cube:144-172) depend on
scipy/NumPy legacy RNG streams and are one-off host work; benchmarks and large parity tests use these seeded
generators instead.  Positions are uniform per region, velocities Maxwell-Boltzmann with scale ``a_shape``."""
from __future__ import annotations

import numpy as np


def maxwell_velocities(n, a_shape, rng):
    """|v| ~ Maxwell(a_shape), isotropic direction  ==  each component ~ N(0, a_shape^2)."""
    v = rng.normal(size=(3, n)) * a_shape
    return v[0].copy(), v[1].copy(), v[2].copy()


def cube_ic(p, consts, seed=127):
    """Uniform in [0, L)^3 (the reference stratifies by cell then tops up uniformly, Cube:145-156)."""
    rng = np.random.default_rng(seed)
    n = int(p.n)
    x = rng.random(n) * p.cube_x
    y = rng.random(n) * p.cube_y
    z = rng.random(n) * p.cube_z
    vx, vy, vz = maxwell_velocities(n, consts["a_shape"], rng)
    return x, y, z, vx, vy, vz


def pore_regions(p, consts):
    """(counts, [(radius, z_lo, z_hi)]) of the five stacked cylinders, populated in proportion to their volume, with the
    reference's argon_radius insets (Pore:115-139)."""
    c = consts
    n = int(p.n)
    ar = p.argon_radius
    oa, hot, gap, cold = (c["open_air_particles"], c["hot_pore_particles"], c["gap_particles"],
                          c["cold_pore_particles"])
    counts = [oa, hot, gap, cold, n - oa - hot - gap - cold]
    h_oa, h_hot, h_gap, h_cold, H = (c["open_air_height"], c["hot_coating_height"], c["gap_height"],
                                    c["cold_coating_height"], c["total_height"])
    regions = [
        (c["open_air_radius"] - ar, 0 + ar, h_oa - ar),
        (c["pore_coated_radius"] - ar, h_oa, h_oa + h_hot),
        (c["gap_radius"] - ar, h_oa + h_hot + ar, h_oa + h_hot + h_gap - ar),
        (c["pore_coated_radius"] - ar, h_oa + h_hot + h_gap, h_oa + h_hot + h_gap + h_cold),
        (c["open_air_radius"] - ar, h_oa + h_hot + h_gap + h_cold + ar, H - ar),
    ]
    return counts, regions


def pore_ic(p, consts, seed=17):
    """r = (R - r_ar) * sqrt(u), theta ~ U(0, 2pi), z ~ U(z_lo, z_hi) per region (Pore:120-139)."""
    rng = np.random.default_rng(seed)
    n = int(p.n)
    counts, regions = pore_regions(p, consts)
    x = np.empty(n); y = np.empty(n); z = np.empty(n)
    o = 0
    for cnt, (R, zlo, zhi) in zip(counts, regions):
        th = rng.uniform(0, 2 * np.pi, cnt)
        r = R * np.sqrt(rng.uniform(0, 1, cnt))
        x[o:o + cnt] = r * np.cos(th)
        y[o:o + cnt] = r * np.sin(th)
        z[o:o + cnt] = rng.uniform(zlo, zhi, cnt)
        o += cnt
    vx, vy, vz = maxwell_velocities(n, consts["a_shape"], rng)
    return x, y, z, vx, vy, vz


def device_ic_config(p, consts, seed, kind):
    """amc_ic_config for ``Engine.init_synthetic``: the same regions and Maxwell scale as cube_ic / pore_ic, generated on
    the GPU from a counter-based generator (include/argonmc.h) — a different random stream than the host generators."""
    import ctypes as C
    from ._abi import AmcIcConfig
    g = AmcIcConfig()
    g.struct_size, g.seed, g.a_shape = C.sizeof(AmcIcConfig), int(seed) & 0xFFFFFFFFFFFFFFFF, float(consts["a_shape"])
    if kind == "cube":
        g.n_regions = 0
        return g
    counts, regions = pore_regions(p, consts)
    g.n_regions = len(regions)
    o = 0
    for r, (cnt, (R, zlo, zhi)) in enumerate(zip(counts, regions)):
        g.first[r] = o
        g.radius[r], g.z_lo[r], g.z_hi[r] = float(R), float(zlo), float(zhi)
        o += int(cnt)
    g.first[len(regions)] = o
    return g
