"""The reference's parallel STRUCTURE for the p-p sweep, restated in NumPy + multiprocessing.

TEST INFRASTRUCTURE / CPU BASELINE ONLY (bench.py's `cpu_baseline_python_mp` leg, tests/test_pymp_structure.py): the
product package never imports this module.  Nothing here is copied from the reference; it is written from SURVEY.md
App. A (per-pair arithmetic A.1, cell membership A.2, processing order A.3) and mirrors HOW the reference organises the
work (Open_Air_Pore_MC.py:520-549), because that organisation — not the arithmetic — is what its run time consists of:

  * 8 colour groups (gx, gy, gz) in sequence (Pore:522-524);
  * per group one boolean mask over ALL N particles for every cell (lx, ly, lz) of the group (Pore:527-530), 7 x 7 x 74
    of them, only the non-empty ones kept;
  * per non-empty cell a gather by boolean indexing of the eleven per-particle arrays (Pore:533-543);
  * one task per cell on a `multiprocessing.Pool(cpu_count() + 1)` — a FRESH pool per colour group — through `starmap`
    (Pore:545-546), arguments and results pickled through pipes;
  * scatter of the returned per-cell arrays (Pore:547).

The per-pair loop is NumPy-SCALAR Python like the reference's (`x ** 2` on a NumPy float64 is libm pow), so the results
are bit-identical to the reference's: tests/test_pymp_structure.py checks that against the reference's own step dumps
(tests/golden/step_pore_a.npz), with the C oracle supplying drift / walls / bounds check.
"""
from __future__ import annotations

import multiprocessing as mp
import os

import numpy as np

FIELDS = ("d", "dx", "dy", "dz", "x", "y", "z", "vx", "vy", "vz")


def pair_cell(cr, mass, cont, cx, cy, cz, flag, x, y, z, vx, vy, vz):
    """One cell's O(n^2) pair loop (i = 1..n-1, j = 0..i-1; subscript 1 = j, 2 = i), in place on the
    cell's own copies.  Returns (the eleven arrays, completed paths [[total, x, y, z], ...], collisions)."""
    n = len(x)
    paths = []
    ncoll = 0
    for i in range(n):
        for j in range(i):
            ex, ey, ez = x[i] - x[j], y[i] - y[j], z[i] - z[j]
            if np.sqrt(ex ** 2 + ey ** 2 + ez ** 2) < cr:
                ux, uy, uz = -vx[i] + vx[j], -vy[i] + vy[j], -vz[i] + vz[j]
                a = ux ** 2 + uy ** 2 + uz ** 2
                b = 2 * (ex * ux + ey * uy + ez * uz)
                c = ex ** 2 + ey ** 2 + ez ** 2 - cr ** 2
                root = np.sqrt(b ** 2 - 4 * a * c)
                t = max((-b + root) / (2 * a), (-b - root) / (2 * a))
                for p in (j, i):                                        # j first, then i
                    if flag[p]:
                        speed = np.sqrt(vx[p] ** 2 + vy[p] ** 2 + vz[p] ** 2)
                        paths.append([abs(cont[p] - abs(speed * t)), abs(cx[p] - abs(vx[p] * t)),
                                      abs(cy[p] - abs(vy[p] * t)), abs(cz[p] - abs(vz[p] * t))])
                    else:
                        flag[p] = True
                # back to the contact configuration
                r1 = np.array([x[j] - vx[j] * t, y[j] - vy[j] * t, z[j] - vz[j] * t])
                r2 = np.array([x[i] - vx[i] * t, y[i] - vy[i] * t, z[i] - vz[i] * t])
                nrm = (r2 - r1) / cr
                v1 = np.array([vx[j], vy[j], vz[j]])
                v2 = np.array([vx[i], vy[i], vz[i]])
                p_ = (np.dot(v1, nrm) - np.dot(v2, nrm)) / mass
                w1 = np.array([v1[0] - p_ * mass * nrm[0], v1[1] - p_ * mass * nrm[1], v1[2] - p_ * mass * nrm[2]])
                w2 = np.array([v2[0] + p_ * mass * nrm[0], v2[1] + p_ * mass * nrm[1], v2[2] + p_ * mass * nrm[2]])
                x[j], y[j], z[j] = r1[0] + w1[0] * t, r1[1] + w1[1] * t, r1[2] + w1[2] * t
                x[i], y[i], z[i] = r2[0] + w2[0] * t, r2[1] + w2[1] * t, r2[2] + w2[2] * t
                vx[j], vy[j], vz[j] = w1[0], w1[1], w1[2]
                vx[i], vy[i], vz[i] = w2[0], w2[1], w2[2]
                cont[i] = abs(np.sqrt(w2[0] ** 2 + w2[1] ** 2 + w2[2] ** 2) * t)
                cont[j] = abs(np.sqrt(w1[0] ** 2 + w1[1] ** 2 + w1[2] ** 2) * t)
                cx[i], cz[i], cy[i] = abs(w2[0] * t), abs(w2[2] * t), abs(w2[1] * t)
                cx[j], cy[j], cz[j] = abs(w1[0] * t), abs(w1[1] * t), abs(w1[2] * t)
                ncoll += 1
    return cont, cx, cy, cz, flag, x, y, z, vx, vy, vz, paths, ncoll


def _task(cr, mass, *arrays):
    return pair_cell(cr, mass, *arrays)


def cell_masks(X, Y, Z, geom, gx, gy, gz, full=True):
    """Boolean membership masks of the non-empty cells of colour group (gx, gy, gz), in (lx, ly, lz) list order
    (Pore:527-530).  full=True evaluates every cell's complete expression over all N particles, which is what the
    reference does and what half of its sweep time is; full=False hoists the x- and (x, y)-layer tests (same masks)."""
    cr, dx, dy, dz, nx, ny, nz = geom["cr"], geom["dx"], geom["dy"], geom["dz"], geom["nx"], geom["ny"], geom["nz"]
    masks = []
    if full:
        for lx in range(nx):
            for ly in range(ny):
                for lz in range(nz // 2):
                    m = (((2 * lx + gx - nx) * dx - cr < X) & (X < (2 * lx + gx - nx + 1) * dx) &
                         ((2 * ly + gy - ny) * dy - cr < Y) & (Y < (2 * ly + gy - ny + 1) * dy) &
                         ((2 * lz + gz) * dz - cr < Z) & (Z < (2 * lz + gz + 1) * dz))
                    if m.any():
                        masks.append(m)
        return masks
    for lx in range(nx):
        mx = ((2 * lx + gx - nx) * dx - cr < X) & (X < (2 * lx + gx - nx + 1) * dx)
        if not mx.any():
            continue
        for ly in range(ny):
            mxy = mx & ((2 * ly + gy - ny) * dy - cr < Y) & (Y < (2 * ly + gy - ny + 1) * dy)
            if not mxy.any():
                continue
            for lz in range(nz // 2):
                m = mxy & ((2 * lz + gz) * dz - cr < Z) & (Z < (2 * lz + gz + 1) * dz)
                if m.any():
                    masks.append(m)
    return masks


def sweep(state, geom, workers=None, full_masks=True):
    """One p-p sweep over the module-global-like arrays in `state` (dict with FIELDS + "flag"), in place.
    geom: dict(cr, mass, dx, dy, dz, nx, ny, nz).  Returns (collisions, completed paths as an array [k, 4])."""
    cr, mass = geom["cr"], geom["mass"]
    workers = workers or (os.cpu_count() or 1) + 1
    X, Y, Z = state["x"], state["y"], state["z"]
    ncoll, paths = 0, []
    for gx in (0, 1):
        for gy in (0, 1):
            for gz in (0, 1):
                masks = cell_masks(X, Y, Z, geom, gx, gy, gz, full=full_masks)
                if not masks:
                    continue
                args = [(cr, mass, state["d"][m], state["dx"][m], state["dy"][m], state["dz"][m], state["flag"][m],
                         X[m], Y[m], Z[m], state["vx"][m], state["vy"][m], state["vz"][m]) for m in masks]
                with mp.get_context("fork").Pool(workers) as pool:      # a fresh pool per colour group (Pore:545)
                    results = pool.starmap(_task, args)
                for m, r in zip(masks, results):
                    (state["d"][m], state["dx"][m], state["dy"][m], state["dz"][m], state["flag"][m], X[m], Y[m], Z[m],
                     state["vx"][m], state["vy"][m], state["vz"][m]) = r[:11]
                    paths.extend(r[11])
                    ncoll += r[12]
    return ncoll, (np.array(paths, dtype=np.float64).reshape(-1, 4) if paths else np.zeros((0, 4)))


def geometry_of(params):
    """The sweep's geometry constants from an amc_params structure (pore geometries)."""
    return dict(cr=float(params.collision_range), mass=float(params.argon_mass), dx=float(params.dx), dy=float(params.dy),
                dz=float(params.dz), nx=int(params.nx), ny=int(params.ny), nz=int(params.nz))


class PyMpStepper:
    """Whole reference step for the specular pore (Pore:416-557) with the sweep done by `sweep` above and the per-particle
    stages (drift, walls, bounds checks — vectorised in the reference, negligible there) by the C oracle in `pow` mode."""

    def __init__(self, params, workers=None, full_masks=True):
        from oracle import oracle as O
        self.o = O.Oracle(params, mode="pow")
        self.geom = geometry_of(params)
        self.workers = workers
        self.full_masks = full_masks

    def upload(self, *a, **kw):
        self.o.upload(*a, **kw)

    def timestep(self, dt):
        o = self.o
        o.drift(dt, True)
        rc, nwall = o.pore_walls()
        assert rc == 0
        oob1 = o.bounds(False)
        st = {k: o.arr[k] for k in FIELDS}
        flag = o.flag.astype(bool)
        st["flag"] = flag
        npp, paths = sweep(st, self.geom, self.workers, self.full_masks)
        o.flag[:] = flag.astype(np.uint8)
        oob2 = o.bounds(False)
        o.step += 1
        return dict(n_pp=npp, n_wall=nwall, n_oob_walls=oob1, n_oob_pp=oob2, paths=paths)

    def state(self):
        return self.o.state()
