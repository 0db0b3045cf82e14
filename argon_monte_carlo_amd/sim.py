"""Host-side mirror of the reference's interface for the hot path.

The reference has no functions around its loop body; its "API" is a set of module-global ndarrays
(``x_vals``, ``x_velocities``, ``dist_since_collision`` ... Pore:385-400), four completed-path lists (Pore:408-413),
the per-step counter ``num_collisions_per_step`` (Pore:424) and ONE real function on the p-p path,
``pairwise_particles_in_cell`` (Pore:160-255).  ``Simulation`` exposes exactly those names; ``timestep(dt)`` is one
iteration of ``for i in range(num_timesteps)`` (Pore:416-557 / Cube:175-338) executed by libargonmc.so on the GPU.
"""
from __future__ import annotations

import numpy as np

from . import ic as IC
from . import outputs as OUT
from . import params as PR
from .engine import Engine

_NAMES = {"x_vals": "x", "y_vals": "y", "z_vals": "z", "x_velocities": "vx", "y_velocities": "vy",
          "z_velocities": "vz", "dist_since_collision": "d", "dist_x_since_collision": "dx",
          "dist_y_since_collision": "dy", "dist_z_since_collision": "dz", "full_path_traveled": "flag"}


class Simulation:
    """``kind``: "cube" (Open_Air_Cube_MC.py), "pore" (Open_Air_Pore_MC.py) — see params.py for the constants."""

    def __init__(self, kind="pore", n=None, sigma=PR.SIGMA, device=0, keep_prior=False, tolerate_fp_errors=False,
                 params=None, consts=None):
        if params is None:
            if kind == "cube":
                params, consts = PR.cube_params(n=n, sigma=sigma, device=device) if n is None else \
                    PR.cube_params_for_n(n, sigma=sigma, device=device)
            elif kind == "pore":
                params, consts = PR.pore_params(n=n, sigma=sigma, device=device)
            else:
                raise ValueError(kind)
        if keep_prior:
            params.reserved0 |= 1
        if tolerate_fp_errors:
            params.reserved1 |= 1
        self.kind = kind
        self.params, self.consts = params, consts
        self.dt = consts["dt"]
        self.engine = Engine(params)
        self.completed_paths, self.completed_x_paths = [], []
        self.completed_y_paths, self.completed_z_paths = [], []
        self.num_collisions_per_step = 0          # Pore:424 (value of the last step)
        self.total_cols = 0                        # Pore:556
        self.steps_done = 0
        self._cache = None

    # ---- state, under the reference's names ----------------------------------------------------------------------
    def set_state(self, x_vals, y_vals, z_vals, x_velocities, y_velocities, z_velocities, dist_since_collision=None,
                  dist_x_since_collision=None, dist_y_since_collision=None, dist_z_since_collision=None,
                  full_path_traveled=None):
        self.engine.upload(x_vals, y_vals, z_vals, x_velocities, y_velocities, z_velocities, dist_since_collision,
                           dist_x_since_collision, dist_y_since_collision, dist_z_since_collision, full_path_traveled)
        self._cache = None

    def init_synthetic(self, seed=None, device=False):
        """Seeded synthetic initial conditions — the reference's own generators are host-side, one-off.
        ``device=True`` generates them on the GPU (counter-based generator: another random stream, no PCIe transfer)."""
        seed = seed if seed is not None else self.consts["seed"]
        if device:
            self.engine.init_synthetic(IC.device_ic_config(self.params, self.consts, seed, "cube" if self.kind == "cube" else "pore"))
            self._cache = None
            return
        gen = IC.cube_ic if self.kind == "cube" else IC.pore_ic
        self.set_state(*gen(self.params, self.consts, seed))

    def _state(self):
        if self._cache is None:
            self._cache = self.engine.download()
        return self._cache

    def __getattr__(self, name):
        if name in _NAMES:
            v = self._state()[_NAMES[name]]
            return v.astype(bool) if name == "full_path_traveled" else v
        if name in ("prior_x_vals", "prior_y_vals", "prior_z_vals"):
            return self.engine.download_prior()["xyz".index(name[6])]
        raise AttributeError(name)

    # ---- the loop body -----------------------------------------------------------------------------------------------
    def timestep(self, dt=None, collect_paths=True):
        """One iteration of the reference's time loop.  Returns the step's counters."""
        st = self.engine.timestep(self.dt if dt is None else dt)
        self._cache = None
        self.num_collisions_per_step = st["n_pp"] + st["n_wall"]
        self.total_cols += self.num_collisions_per_step
        self.steps_done += 1
        if collect_paths:
            self._collect()
        return st

    def run(self, nsteps, dt=None):
        """``nsteps`` iterations without host synchronisation in between (histograms accumulate on the device)."""
        st = self.engine.run(self.dt if dt is None else dt, nsteps)
        self._cache = None
        self.num_collisions_per_step = None
        self.total_cols += st["n_pp"] + st["n_wall"]
        self.steps_done += nsteps
        return st

    def _collect(self):
        rec = self.engine.drain_paths(sort=True)
        if len(rec):
            self.completed_paths.extend(rec["total"].tolist())
            self.completed_x_paths.extend(rec["px"].tolist())
            self.completed_y_paths.extend(rec["py"].tolist())
            self.completed_z_paths.extend(rec["pz"].tolist())

    # ---- end of run (Pore:559-630) -------------------------------------------------------------------------------------
    def histograms(self):
        """(densities dict, bin edges) from the histograms accumulated on the device."""
        counts, _ = self.engine.histograms()
        if getattr(self, "_hist_base", None) is not None:
            counts = counts + self._hist_base          # paths completed before the checkpoint this run resumed from
        dens = {}
        edges = None
        for row, key in enumerate(["total", "x", "y", "z"]):
            dens[key], edges = OUT.density_from_counts(counts[row], self.params.hist_lo, self.params.hist_hi)
        return dens, edges

    def write_outputs(self, directory="."):
        dens, edges = self.histograms()
        OUT.write_histograms(directory, dens, edges)

    # ---- checkpoint / resume (the reference has none: a 10^4-step run is one process lifetime there) ---------------
    def _checkpoint_extra(self):
        return {}

    def _restore_extra(self, z):
        pass

    def save_checkpoint(self, path):
        """Everything a later ``load_checkpoint`` needs to continue bit-identically: the particle arrays, the
        completed-path lists, the device histograms and the counters (+ RNG streams and per-step lists for Temp)."""
        st = self.engine.download()
        self._collect()
        counts, _ = self.engine.histograms()
        base = getattr(self, "_hist_base", None)
        if base is not None:
            counts = counts + base
        np.savez(path, kind=self.kind, n=int(self.params.n), collision_range=float(self.params.collision_range),
                 hist_counts=counts.astype(np.uint64), completed_paths=np.array(self.completed_paths, dtype=np.float64),
                 completed_x_paths=np.array(self.completed_x_paths, dtype=np.float64),
                 completed_y_paths=np.array(self.completed_y_paths, dtype=np.float64),
                 completed_z_paths=np.array(self.completed_z_paths, dtype=np.float64),
                 total_cols=int(self.total_cols), steps_done=int(self.steps_done),
                 **{f"state_{k}": v for k, v in st.items()}, **self._checkpoint_extra())

    def load_checkpoint(self, path):
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
        if str(z["kind"]) != self.kind or int(z["n"]) != int(self.params.n) or \
                float(z["collision_range"]) != float(self.params.collision_range):
            raise ValueError("checkpoint was written by a different configuration")
        self.engine.upload(*[z[f"state_{k}"] for k in ("x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz")],
                           flag=z["state_flag"])
        self.engine.reset_outputs()
        self._hist_base = z["hist_counts"].astype(np.uint64)
        self.completed_paths = z["completed_paths"].tolist()
        self.completed_x_paths = z["completed_x_paths"].tolist()
        self.completed_y_paths = z["completed_y_paths"].tolist()
        self.completed_z_paths = z["completed_z_paths"].tolist()
        self.total_cols, self.steps_done = int(z["total_cols"]), int(z["steps_done"])
        self._cache = None
        self._restore_extra(z)

    def close(self):
        self.engine.close()


class TemperatureSimulation(Simulation):
    """Temperature_Pore_MC.py: energised walls, per-step z-momentum / energy transfer and momentum_energy.csv
    (Temp:634-638, 756-758, 929-933).  The random directions come from ``np.random`` / ``random`` (module-level streams
    by default, as in the reference) in strict particle order; see energised.py."""

    def __init__(self, n=None, sigma=PR.SIGMA, device=0, np_rng=None, py_rng=None, params=None, consts=None,
                 device_rng_seed=None):
        """``device_rng_seed``: opt-in, NON-PARITY mode — re-emission directions and gap energies are drawn on the GPU
        (Philox4x32-10 keyed by the seed, Gauss-Legendre Debye integral) instead of from ``np.random`` / ``random`` /
        ``mpmath``; same physics and recipe, different random numbers, no per-case host hand-over."""
        from .energised import DirectionSampler, SurfaceEnergies
        from .engine import EnergisedEngine
        if params is None:
            params, consts = PR.pore_params(n=n, sigma=sigma, energised=True, device=device)
        params.reserved0 |= 1
        self.kind = "temp"
        self.params, self.consts = params, consts
        self.dt = consts["dt"]
        # (the gap-energy workers are forked here, ahead of amc_create: no child of this process inherits a HIP context)
        self.energies = SurfaceEnergies(consts, start_workers=True)
        params.E_cold, params.E_hot = self.energies.cold, self.energies.hot
        self.engine = EnergisedEngine(params)
        self.sampler = DirectionSampler(np_rng, py_rng)
        self._device_rng = None
        if device_rng_seed is not None:
            from .energised import device_rng_config
            self._device_rng = device_rng_config(consts, device_rng_seed)
        self.completed_paths, self.completed_x_paths = [], []
        self.completed_y_paths, self.completed_z_paths = [], []
        self.num_collisions_per_step = 0
        self.total_cols = 0
        self.total_errs = 0
        self.steps_done = 0
        self._cache = None
        self.momentum_z_change_per_step = []          # Temp:634
        self.energy_transfer_hot_per_step = []        # Temp:637
        self.energy_transfer_cold_per_step = []       # Temp:638
        self._zero_flags = []

    def init_synthetic(self, seed=None, device=False):
        seed = seed if seed is not None else self.consts["seed"]
        if device:
            self.engine.init_synthetic(IC.device_ic_config(self.params, self.consts, seed, "pore"))
            self._cache = None
            return
        self.set_state(*IC.pore_ic(self.params, self.consts, seed))

    def timestep(self, dt=None, collect_paths=True):
        if self._device_rng is not None:
            st, mom, cold, hot, had_m, had_c, had_h = self.engine.temp_timestep_device(self.dt if dt is None else dt,
                                                                                       self._device_rng)
        else:
            st, mom, cold, hot, had_m, had_c, had_h = self.engine.temp_timestep(self.dt if dt is None else dt,
                                                                                self.sampler, self.energies)
        self._cache = None
        self.momentum_z_change_per_step.append(mom)
        self.energy_transfer_cold_per_step.append(cold)
        self.energy_transfer_hot_per_step.append(hot)
        self._zero_flags.append((not had_m, not had_c, not had_h))
        self.num_collisions_per_step = st["n_pp"] + st["n_wall"]
        self.total_cols += self.num_collisions_per_step
        self.total_errs += st["n_fp_errors"]
        self.steps_done += 1
        if collect_paths:
            self._collect()
        return st

    def run(self, nsteps, dt=None):
        for _ in range(int(nsteps)):
            self.timestep(dt, collect_paths=False)

    def _checkpoint_extra(self):
        np_state = self.sampler.np_rng.get_state()
        py_state = self.sampler.py_rng.getstate()
        return dict(momentum=np.array([float(v) for v in self.momentum_z_change_per_step]),
                    energy_cold=np.array([float(v) for v in self.energy_transfer_cold_per_step]),
                    energy_hot=np.array([float(v) for v in self.energy_transfer_hot_per_step]),
                    zero_flags=np.array(self._zero_flags, dtype=bool).reshape(-1, 3), total_errs=int(self.total_errs),
                    np_rng_keys=np.asarray(np_state[1], dtype=np.uint32),
                    np_rng_rest=np.array([np_state[2], np_state[3]], dtype=np.int64), np_rng_gauss=float(np_state[4]),
                    py_rng_state=np.array(py_state[1], dtype=np.uint64), py_rng_version=int(py_state[0]))

    def _restore_extra(self, z):
        self.momentum_z_change_per_step = z["momentum"].tolist()
        self.energy_transfer_cold_per_step = z["energy_cold"].tolist()
        self.energy_transfer_hot_per_step = z["energy_hot"].tolist()
        self._zero_flags = [tuple(bool(b) for b in row) for row in z["zero_flags"]]
        self.total_errs = int(z["total_errs"])
        self.sampler.np_rng.set_state(("MT19937", z["np_rng_keys"], int(z["np_rng_rest"][0]), int(z["np_rng_rest"][1]),
                                       float(z["np_rng_gauss"])))
        self.sampler.py_rng.setstate((int(z["py_rng_version"]), tuple(int(v) for v in z["py_rng_state"]), None))

    def write_outputs(self, directory="."):
        import os
        from .energised import format_mpf
        super().write_outputs(directory)
        m = [format_mpf(v, z[0]) for v, z in zip(self.momentum_z_change_per_step, self._zero_flags)]
        c = [format_mpf(v, z[1]) for v, z in zip(self.energy_transfer_cold_per_step, self._zero_flags)]
        h = [format_mpf(v, z[2]) for v, z in zip(self.energy_transfer_hot_per_step, self._zero_flags)]
        OUT.write_momentum_energy_csv(os.path.join(directory, "momentum_energy.csv"), m, c, h)


# ---- the one real function boundary of the reference: pairwise_particles_in_cell (Pore:160-255) ---------------------
num_collisions_per_step = None       # injected by init_globals(counter), exactly like Pore:350-352
_cell_engines = {}


def init_globals(counter):
    global num_collisions_per_step
    num_collisions_per_step = counter


def pairwise_particles_in_cell(completed_paths, completed_x_paths, completed_y_paths, completed_z_paths, in_cell,
                               continue_path, continue_x_path, continue_y_path, continue_z_path, has_collided,
                               x_positions_in_cell, y_positions_in_cell, z_positions_in_cell, x_velocities_in_cell,
                               y_velocities_in_cell, z_velocities_in_cell, sigma=PR.SIGMA, device=0):
    """Same signature, argument meaning, return tuple and side effects as the reference function: the per-cell arrays
    are updated as the sequential i>j loop would, completed free paths are appended to the four lists in loop order,
    and the shared counter gets the number of collisions (Pore:244-251).  The pair loop runs on the GPU."""
    if num_collisions_per_step is None:
        raise NameError("name 'num_collisions_per_step' is not defined")     # what the reference raises (Pore:244)
    n = int(np.sum(in_cell))
    cap = 1
    while cap < max(n, 2):
        cap <<= 1
    key = (cap, float(sigma), device)
    eng = _cell_engines.get(key)
    if eng is None:
        p, _ = PR.cell_params(sigma=sigma, n=cap, device=device)
        eng = _cell_engines[key] = Engine(p)
    arrs = [np.ascontiguousarray(a, dtype=np.float64).copy() for a in
            (continue_path, continue_x_path, continue_y_path, continue_z_path)]
    flag = np.ascontiguousarray(np.asarray(has_collided).astype(np.uint8))
    pos = [np.ascontiguousarray(a, dtype=np.float64).copy() for a in
           (x_positions_in_cell, y_positions_in_cell, z_positions_in_cell, x_velocities_in_cell,
            y_velocities_in_cell, z_velocities_in_cell)]
    paths, ncoll = eng.pairwise_cell(arrs[0], arrs[1], arrs[2], arrs[3], flag, *pos)
    with num_collisions_per_step.get_lock():
        num_collisions_per_step.value += ncoll
    completed_paths.extend(paths[:, 0].tolist())
    completed_x_paths.extend(paths[:, 1].tolist())
    completed_y_paths.extend(paths[:, 2].tolist())
    completed_z_paths.extend(paths[:, 3].tolist())
    return (in_cell, arrs[0], arrs[1], arrs[2], arrs[3], flag.astype(bool), pos[0], pos[1], pos[2], pos[3], pos[4],
            pos[5])
