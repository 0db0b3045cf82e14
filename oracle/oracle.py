"""ctypes wrapper of liboracle.so — the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (argon_monte_carlo_amd) never imports this module.

``Oracle(params, mode)`` with mode "pow" (bit-for-bit the reference: NumPy scalar ``**2`` = libm pow) or "mul"
(exact squares — the arithmetic the HIP kernels implement).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from argon_monte_carlo_amd._abi import (AmcParams, AmcPathRecord, AmcStepStats, path_record_dtype)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATE_FIELDS = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz"]


class OrcState(C.Structure):
    _fields_ = [("n", C.c_int64)] + [(k, C.POINTER(C.c_double)) for k in ["x", "y", "z", "vx", "vy", "vz"]] + \
               [(k, C.POINTER(C.c_double)) for k in ["d", "dx", "dy", "dz"]] + [("flag", C.POINTER(C.c_uint8))] + \
               [(k, C.POINTER(C.c_double)) for k in ["px", "py", "pz"]]


class OrcSink(C.Structure):
    _fields_ = [("rec", C.POINTER(AmcPathRecord)), ("cap", C.c_int64), ("n", C.c_int64), ("overflow", C.c_int64)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("amc_oracle.c", "amc_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        assert _LIB.orc_sizeof_params() == C.sizeof(AmcParams), "amc_params layout mismatch"
        assert _LIB.orc_sizeof_path_record() == C.sizeof(AmcPathRecord)
        assert _LIB.orc_sizeof_step_stats() == C.sizeof(AmcStepStats)
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """Holds the reference's module-global arrays (Pore:385-400) and advances them like the reference does."""

    def __init__(self, params: AmcParams, mode="mul", path_capacity=1 << 20):
        assert mode in ("pow", "mul")
        self.p = params
        self.mode = mode
        self.L = lib()
        n = int(params.n)
        self.n = n
        self.arr = {k: np.zeros(n) for k in STATE_FIELDS + ["px", "py", "pz"]}
        self.flag = np.zeros(n, dtype=np.uint8)
        self._cap = path_capacity
        self._rec = np.zeros(path_capacity, dtype=path_record_dtype())
        self._sink = OrcSink(self._rec.ctypes.data_as(C.POINTER(AmcPathRecord)), path_capacity, 0, 0)
        self.step = 0
        self._mk_state()

    def _mk_state(self):
        s = OrcState()
        s.n = self.n
        for k in STATE_FIELDS + ["px", "py", "pz"]:
            setattr(s, k, _dp(self.arr[k]))
        s.flag = self.flag.ctypes.data_as(C.POINTER(C.c_uint8))
        self._state = s

    def _fn(self, name):
        return getattr(self.L, f"orc_{self.mode}_{name}")

    # -- state --------------------------------------------------------------------------------------------
    def upload(self, x, y, z, vx, vy, vz, d=None, dx=None, dy=None, dz=None, flag=None):
        for k, v in zip(STATE_FIELDS, [x, y, z, vx, vy, vz, d, dx, dy, dz]):
            if v is not None:
                self.arr[k][:] = np.asarray(v, dtype=np.float64)
        if flag is not None:
            self.flag[:] = np.asarray(flag).astype(np.uint8)

    def state(self):
        d = {k: self.arr[k].copy() for k in STATE_FIELDS}
        d["flag"] = self.flag.copy()
        return d

    # -- stages -------------------------------------------------------------------------------------------
    def timestep(self, dt):
        st = AmcStepStats()
        rc = self._fn("timestep")(C.byref(self.p), C.byref(self._state), C.c_double(dt), C.byref(self._sink),
                                  C.c_int32(self.step), C.byref(st))
        self.step += 1
        self.last_rc = rc
        return rc, st.as_dict()

    def timestep_par(self, dt, threads=None):
        """One step with the p-p sweep spread over the host's cores (OpenMP threads over the disjoint cells of a colour
        group — the reference's Pool.starmap structure, Pore:545-549).  Specular geometries; completed paths are
        counted, not recorded.  For the pore the state equals timestep()'s; for the cube the colouring is NOT the
        reference's serial order (see amc_oracle_impl.h)."""
        if threads:
            self.L.orc_set_threads(C.c_int(int(threads)))
        st = AmcStepStats()
        rc = self._fn("timestep_par")(C.byref(self.p), C.byref(self._state), C.c_double(dt), C.byref(st))
        self.step += 1
        return rc, st.as_dict()

    def drift(self, dt, save_prior=True):
        self._fn("drift")(C.byref(self.p), C.byref(self._state), C.c_double(dt), C.c_int(int(save_prior)))

    def cube_walls(self):
        self._fn("cube_walls")(C.byref(self.p), C.byref(self._state))

    def pore_walls(self):
        n = C.c_int64(0)
        rc = self._fn("pore_walls")(C.byref(self.p), C.byref(self._state), C.byref(self._sink), C.c_int32(self.step),
                                    C.byref(n))
        return rc, n.value

    def bounds(self, energised=False):
        f = self._fn("bounds")
        f.restype = C.c_int64
        return f(C.byref(self.p), C.byref(self._state), C.c_int(int(energised)))

    def sweep(self):
        n, t = C.c_int64(0), C.c_int64(0)
        name = "cube_sweep" if self.p.geometry == 1 else "pore_sweep"
        rc = self._fn(name)(C.byref(self.p), C.byref(self._state), C.byref(self._sink), C.c_int32(self.step),
                            C.byref(n), C.byref(t))
        return rc, n.value, t.value

    def vertical_wall(self, hits, z_plane, phase=2):
        n = C.c_int64(0)
        h = np.ascontiguousarray(hits, dtype=np.uint8)
        self._fn("pore_vertical_wall")(C.byref(self.p), C.byref(self._state), h.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       C.c_double(z_plane), C.byref(self._sink), C.c_int32(self.step),
                                       C.c_int32(phase), C.byref(n))
        return n.value

    def side_wall(self, hits, Rc, bookkeeping=True, phase=1):
        n, e = C.c_int64(0), C.c_int64(0)
        h = np.ascontiguousarray(hits, dtype=np.uint8)
        rc = self._fn("side_wall")(C.byref(self.p), C.byref(self._state), h.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   C.c_double(Rc), C.c_int(int(bookkeeping)), C.byref(self._sink), C.c_int32(self.step),
                                   C.c_int32(phase), C.byref(n), C.byref(e))
        return rc, n.value, e.value

    # -- Temperature_Pore_MC.py (energised walls): deterministic parts in C, RNG / mpmath on the host --------
    def temp_specular(self):
        e = C.c_int64(0)
        self._fn("temp_specular")(C.byref(self.p), C.byref(self._state), C.byref(e))
        return e.value

    def wall_hits(self, case):
        n = self.n
        hits = np.zeros(n + 1, dtype=np.uint8)
        self._fn("temp_mask")(C.byref(self.p), C.byref(self._state), C.c_int(case), hits.ctypes.data_as(C.POINTER(C.c_uint8)))
        nh = int(hits[:n].sum())
        idx = np.zeros(max(1, nh), dtype=np.int32)
        t = np.zeros(max(1, nh)); contact = np.zeros((max(1, nh), 3)); normal = np.zeros((max(1, nh), 3))
        ok = np.zeros(max(1, nh), dtype=np.uint8)
        f = self._fn("temp_geometry")
        f.restype = C.c_int64
        got = f(C.byref(self.p), C.byref(self._state), C.c_int(case), hits.ctypes.data_as(C.POINTER(C.c_uint8)),
                idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(t), _dp(contact), _dp(normal),
                ok.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert got == nh
        self._pending = (case, nh, idx, t, contact, ok)
        return idx[:nh], normal[:nh], contact[:nh, 2].copy(), ok[:nh].astype(bool)

    def wall_apply(self, case, dirs, Es):
        pc, nh, idx, t, contact, ok = self._pending
        assert pc == case
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        Es = np.ascontiguousarray(Es, dtype=np.float64)
        dpz = np.zeros(max(1, nh)); dE = np.zeros(max(1, nh))
        nerr = C.c_int64(0)
        f = self._fn("temp_apply")
        f.restype = C.c_int64
        cnt = f(C.byref(self.p), C.byref(self._state), C.c_int(case), C.c_int64(nh), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                _dp(t), _dp(contact), ok.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(dirs), _dp(Es), _dp(dpz), _dp(dE),
                C.byref(self._sink), C.c_int32(self.step), C.byref(nerr))
        self._temp_wall_count += cnt
        self._temp_errs += nerr.value
        return dpz[:nh], dE[:nh]

    def _own_host_objects(self, sampler, energies):
        """The oracle's own direction sampler / surface energies (oracle/temp_host.py) on the SAME two random generators and
        constants the caller's objects carry — the product's classes are only looked at for those, never called."""
        from oracle import temp_host as TH
        key = (id(sampler), id(energies))
        if getattr(self, "_host_key", None) != key:
            d = sampler if isinstance(sampler, TH.Directions) else TH.Directions(sampler.np_rng, sampler.py_rng)
            e = energies if isinstance(energies, TH.Energies) else TH.Energies(energies)
            self._host_key, self._host_objs = key, (d, e, sampler, energies)     # (keeps the ids alive)
        return self._host_objs[0], self._host_objs[1]

    def temp_timestep(self, dt, sampler, energies):
        """One iteration of Temp:662-853.  Returns (rc, stats dict, momentum, energy_cold, energy_hot, had flags).
        `sampler` / `energies` may be the product's objects: only their random generators and constants are used, the
        host loop itself is the oracle's own (oracle/temp_host.py)."""
        from oracle import temp_host as TH
        directions, own_energies = self._own_host_objects(sampler, energies)
        before = self._sink.n
        self._temp_wall_count = 0
        self._temp_errs = 0
        self.drift(dt, True)                                        # Temp:672-683
        self._temp_errs += self.temp_specular()                     # Temp:693-703
        res = TH.run_cases(self, directions, own_energies)          # Temp:705-758
        oob1 = self.bounds(True)                                    # Temp:804
        rc, npp, _ = self.sweep()                                   # Temp:813-842
        oob2 = self.bounds(True)                                    # Temp:844
        st = dict(n_pp=npp, n_wall=self._temp_wall_count, n_oob_walls=oob1, n_oob_pp=oob2,
                  n_paths=self._sink.n - before, n_fp_errors=self._temp_errs)
        self.step += 1
        return (rc, st) + res

    # -- outputs ------------------------------------------------------------------------------------------
    def paths(self):
        """completed-path records so far, in the reference's append order."""
        return self._rec[: self._sink.n].copy()

    def drain_paths(self):
        r = self.paths()
        self._sink.n = 0
        return r


def pair_cell(params, cont, cx, cy, cz, flag, x, y, z, vx, vy, vz, mode="pow"):
    """pairwise_particles_in_cell (Pore:160-255) on one cell; returns (outputs dict, paths[k,4], ncoll, rc)."""
    L = lib()
    n = len(x)
    arrs = [np.array(a, dtype=np.float64, copy=True) for a in (cont, cx, cy, cz)]
    f = np.array(flag, copy=True).astype(np.uint8)
    parr = [np.array(a, dtype=np.float64, copy=True) for a in (x, y, z, vx, vy, vz)]
    rec = np.zeros(max(16, 4 * n * n), dtype=path_record_dtype())
    sink = OrcSink(rec.ctypes.data_as(C.POINTER(AmcPathRecord)), len(rec), 0, 0)
    nc = C.c_int64(0)
    fn = getattr(L, f"orc_{mode}_pair_cell")
    rc = fn(C.byref(params), C.c_int64(n), *[_dp(a) for a in arrs], f.ctypes.data_as(C.POINTER(C.c_uint8)),
            *[_dp(a) for a in parr], None, C.byref(sink), C.c_int32(0), C.c_int32(16), C.c_int64(0), C.byref(nc))
    r = rec[: sink.n]
    paths = np.stack([r["total"], r["px"], r["py"], r["pz"]], axis=1) if sink.n else np.zeros((0, 4))
    out = dict(cont=arrs[0], cx=arrs[1], cy=arrs[2], cz=arrs[3], flag=f.astype(bool), x=parr[0], y=parr[1], z=parr[2],
               vx=parr[3], vy=parr[4], vz=parr[5])
    return out, paths, nc.value, rc
