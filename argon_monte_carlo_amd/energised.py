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


def _numpy_ddot():
    """cblas_ddot of the BLAS this NumPy loaded, as (dot_kind, address) for amc_host_directions — np.dot of two
    float64[3] IS that function, so handing it over keeps the dot product NumPy's own.  None if it cannot be found."""
    import ctypes as C
    try:
        with open("/proc/self/maps") as fh:
            paths = sorted({ln.split()[-1] for ln in fh if "blas" in ln.lower() and ".so" in ln})
    except OSError:
        return None
    for path in paths:
        try:
            lib = C.CDLL(path)
        except OSError:
            continue
        for name, kind in (("scipy_cblas_ddot64_", 2), ("cblas_ddot64_", 2), ("scipy_cblas_ddot", 3), ("cblas_ddot", 3)):
            try:
                fn = getattr(lib, name)
            except AttributeError:
                continue
            return kind, C.cast(fn, C.c_void_p).value, lib         # (the library object keeps the mapping alive)
    return None


class DirectionSampler:
    """random_components / random_inbounds_direction (Temp:119-141) on the given generators.
    Defaults: the module-level ``np.random`` and ``random`` streams, exactly what the reference uses.

    ``sample_case`` draws the directions of all hits of one case.  When both generators are Mersenne Twisters whose state
    can be taken and put back (``np.random`` / ``RandomState``, ``random`` / ``random.Random``) it does so in the library's
    host helper ``amc_host_directions`` — the same draws from the same streams, 40 times faster than three library calls
    and a small ndarray per attempt — after that helper has reproduced the per-hit calls bit for bit, generator states
    included, on this interpreter's NumPy / libm / BLAS (``fast_path()``); otherwise it is the per-hit loop."""

    _fast = None            # (dot_kind, dot_fn, keep-alive) once the self-test has passed, False when it has failed

    def __init__(self, np_rng=None, py_rng=None, fast=True):
        self.np_rng = np.random if np_rng is None else np_rng
        self.py_rng = _py_random if py_rng is None else py_rng
        self.cos85 = cos(85 * pi / 180)
        self.want_fast = fast
        self._session = None

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

    # ---- all hits of a case at once -------------------------------------------------------------------------------------
    def _states_borrowable(self):
        npr, pyr = self.np_rng, self.py_rng
        ok_np = npr is np.random or isinstance(npr, np.random.RandomState)
        ok_py = pyr is _py_random or type(pyr) is _py_random.Random
        if not (ok_np and ok_py):
            return False
        try:
            return npr.get_state()[0] == "MT19937"
        except Exception:
            return False

    @classmethod
    def fast_path(cls):
        """(dot_kind, dot_fn) of a helper configuration that reproduces the per-hit library calls, or False.  Decided once
        per process by running both on equally seeded generators over hits with all sorts of normals."""
        if cls._fast is not None:
            return cls._fast
        cls._fast = False
        try:
            from . import _lib
            lib = _lib.load()
        except Exception:
            return False
        rs = np.random.RandomState(20240611)
        normals = rs.standard_normal((1500, 3))
        normals /= np.linalg.norm(normals, axis=1)[:, None]
        normals[::7] = (0.0, 0.0, 1.0)
        normals[1::7] = (0.0, 0.0, -1.0)
        normals[2::7, 2] = 0.0                                  # side walls: radial normals
        normals[2::7] /= np.linalg.norm(normals[2::7], axis=1)[:, None]
        ok = np.ones(len(normals), dtype=np.uint8)
        ok[5::11] = 0
        ref = cls(np.random.RandomState(4711), _py_random.Random(4711), fast=False)
        want = ref._sample_loop(normals, ok)
        want_np, want_py = ref.np_rng.get_state(), ref.py_rng.getstate()
        blas = _numpy_ddot()
        for cand in ([blas] if blas else []) + [(0, None, None), (1, None, None)]:
            s = cls(np.random.RandomState(4711), _py_random.Random(4711))
            try:
                got = s._sample_helper(lib, cand[0], cand[1], normals, ok)
            except Exception:
                continue
            got_np, got_py = s.np_rng.get_state(), s.py_rng.getstate()
            if (np.array_equal(got.view(np.uint64), want.view(np.uint64)) and np.array_equal(got_np[1], want_np[1])
                    and got_np[2] == want_np[2] and got_py == want_py):
                cls._fast = cand
                break
        return cls._fast

    def _sample_loop(self, normals, ok):
        n = len(normals)
        dirs = np.zeros((n, 3))
        for k in range(n):
            if ok[k]:                                       # (else: the reference's try-block fails before any draw)
                dirs[k] = self.random_inbounds_direction(np.array(normals[k]))
        return dirs

    def _take_states(self):
        st = self.np_rng.get_state()
        ver, internal, gauss = self.py_rng.getstate()
        import ctypes as C
        return {"np_key": np.array(st[1], dtype=np.uint32), "np_pos": C.c_int32(int(st[2])), "np_rest": (st[3], st[4]),
                "py_key": np.array(internal[:-1], dtype=np.uint32), "py_pos": C.c_int32(int(internal[-1])), "py_rest": (ver, gauss)}

    def _put_states(self, S):
        self.np_rng.set_state(("MT19937", S["np_key"], int(S["np_pos"].value)) + tuple(S["np_rest"]))
        self.py_rng.setstate((S["py_rest"][0], tuple(S["py_key"].tolist()) + (int(S["py_pos"].value),), S["py_rest"][1]))

    def _sample_helper(self, lib, dot_kind, dot_fn, normals, ok):
        import ctypes as C
        normals = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        ok8 = np.ascontiguousarray(ok, dtype=np.uint8)
        n = len(normals)
        dirs = np.zeros((n, 3))
        S = self._session if self._session is not None else self._take_states()
        try:
            rc = lib.amc_host_directions(S["np_key"].ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(S["np_pos"]),
                                         S["py_key"].ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(S["py_pos"]),
                                         normals.ctypes.data_as(C.POINTER(C.c_double)), ok8.ctypes.data_as(C.POINTER(C.c_uint8)),
                                         n, self.cos85, pi, int(dot_kind), C.c_void_p(dot_fn), dirs.ctypes.data_as(C.POINTER(C.c_double)))
        finally:
            if self._session is None:
                self._put_states(S)
        if rc != 0:
            raise RuntimeError(f"amc_host_directions failed ({rc})")
        return dirs

    def session(self):
        """Context manager around a step's cases: the generator states are taken once and put back at the end (taking and
        restoring CPython's 625-word tuple costs as much as a hundred hits).  Nothing else may draw from the two
        generators inside."""
        sampler = self

        class _Session:
            def __enter__(self_inner):
                if sampler.want_fast and sampler._session is None and sampler._states_borrowable() and sampler.fast_path():
                    sampler._session = sampler._take_states()
                    self_inner.mine = True
                else:
                    self_inner.mine = False
                return sampler

            def __exit__(self_inner, *exc):
                if self_inner.mine:
                    S, sampler._session = sampler._session, None
                    sampler._put_states(S)
                return False

        return _Session()

    def sample_case(self, normals, ok):
        """Directions [n, 3] for the hits of one case (zero rows where ``ok`` is false), in hit order."""
        if len(normals) == 0:
            return np.zeros((0, 3))
        if self.want_fast and self._states_borrowable():
            cfg = self.fast_path()
            if cfg:
                from . import _lib
                return self._sample_helper(_lib.load(), cfg[0], cfg[1], normals, ok)
        return self._sample_loop(normals, ok)


def _direct_debye_integrand():
    """x**3 / (exp(x) - 1) on mpmath numbers through libmp's functions — what the operators of `mpf` call, without
    their wrappers (a quarter of the integrand's time).  Returned only after it has reproduced the operator form bit for
    bit at the precisions mpmath.quad works at; None otherwise (the operator form stays)."""
    try:
        from mpmath import mp, mpf, exp
        from mpmath.libmp import fone, mpf_div, mpf_exp, mpf_pow_int, mpf_sub

        def direct(x):
            prec, rnd = mp._prec_rounding
            v = x._mpf_
            return mp.make_mpf(mpf_div(mpf_pow_int(v, 3, prec, rnd), mpf_sub(mpf_exp(v, prec, rnd), fone, prec, rnd), prec, rnd))

        plain = lambda x: (x ** 3) / (exp(x) - 1)                            # noqa: E731
        saved = mp.prec
        try:
            for prec in (53, 73, 93):
                mp.prec = prec
                for k in range(1, 240):
                    x = mpf(k) / 59 + mpf(1) / (3 + k)                       # 0.02 .. 4: the Debye integrals' range and beyond
                    if direct(x) != plain(x) or direct(x / 1024) != plain(x / 1024):
                        return None
        finally:
            mp.prec = saved
        return direct
    except Exception:
        return None


class SurfaceEnergies:
    """surface_energy_cold / surface_energy_hot (Temp:80-84) and surface_energy_gap(z) (Temp:143-152) via mpmath,
    rounded to double (mpmath works at 53 bits there, so the mpf values ARE doubles)."""

    def __init__(self, consts, start_workers=False):
        """``start_workers``: fork the gap-energy worker processes NOW instead of at the first case with several gap hits —
        what a driver does that builds this object BEFORE its GPU context (sim.TemperatureSimulation, bench.py,
        tests/soak.py): the workers are then forked from a process that has not initialised HIP, and nothing of a GPU
        runtime (threads' locks, mapped device memory) is inherited by them."""
        from mpmath import exp, quad
        self._quad = quad
        self._integrand = lambda x: (x ** 3) / (exp(x) - 1)                  # Temp:80
        import os
        fast = _direct_debye_integrand() if os.environ.get("AMC_DEBYE_DIRECT", "1") != "0" else None
        if fast is not None:
            self._integrand = fast
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
        if start_workers:
            self._get_pool()

    # ---- several gap energies at once ---------------------------------------------------------------------------------
    # One mpmath.quad costs 0.7-1.5 ms of pure-Python multiprecision arithmetic and a step at N = 1e6 has about five gap
    # hits: half of the energised step's host time.  The integrals are independent and consume no random numbers, so the
    # hits of a case are spread over forked worker processes running this very method — same code, same mpmath, same
    # bits (_GapWorkers below; AMC_GAP_WORKERS=0 turns it off, =k sets the number of workers).
    _pool = None
    _pool_owner = None

    def _workers_wanted(self):
        import os
        v = os.environ.get("AMC_GAP_WORKERS")
        if v is not None:
            try:
                return max(0, int(v))
            except ValueError:
                return 0
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        try:                                        # (a container's CPU quota: a GPU box shows all of its cores to every job)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                cores = min(cores, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
        # (one integral per worker at a time: a step at N = 4e6 has about twenty gap hits, a 16-core share serves them in two rounds)
        return min(16, cores - 1) if cores > 2 else 0

    def _get_pool(self):
        cls = SurfaceEnergies
        if cls._pool is not None and cls._pool_owner is self and cls._pool.alive():
            return cls._pool
        cls._shutdown_pool()
        nw = self._workers_wanted()
        if nw < 2:
            return None
        try:
            cls._pool = _GapWorkers(self, nw)
        except OSError:
            cls._pool = None
            return None
        cls._pool_owner = self
        return cls._pool

    @classmethod
    def _shutdown_pool(cls):
        if cls._pool is not None:
            try:
                cls._pool.close()
            except Exception:
                pass
            cls._pool = None
            cls._pool_owner = None

    def gap_many(self, z_values):
        """surface_energy_gap for every contact height of a case, in order (Temp:143-152 per hit)."""
        zs = [float(z) for z in z_values]
        if len(zs) >= 2:
            pool = self._get_pool()
            if pool is not None:
                try:
                    return pool.map(zs)
                except Exception:
                    SurfaceEnergies._shutdown_pool()            # (a dead worker: this process does it, now and from now on)
                    import os
                    os.environ["AMC_GAP_WORKERS"] = "0"
        return [self.gap(z) for z in zs]

    def gap_start(self, z_values):
        """Hand the contact heights of a case to the worker processes and return at once; ``gap_finish`` collects the
        energies.  None when there are no workers (or nothing to do): the caller evaluates them when it needs them."""
        zs = [float(z) for z in z_values]
        if not zs:
            return None
        pool = self._get_pool()
        if pool is None:
            return None
        try:
            return (pool, pool.start(zs), zs)
        except Exception:
            SurfaceEnergies._shutdown_pool()
            return None

    def gap_finish(self, handle):
        pool, count, zs = handle
        try:
            return pool.finish(count)
        except Exception:
            SurfaceEnergies._shutdown_pool()            # (a dead worker: this process does it, now and from now on)
            import os
            os.environ["AMC_GAP_WORKERS"] = "0"
            return [self.gap(z) for z in zs]

    def gap(self, z_value):
        z_value = float(z_value)
        m = (self.t_cold - self.t_hot) / self.gap_height                                   # Temp:144
        t_gap = m * (z_value - self.gap_bottom_height) + self.t_hot                        # Temp:145
        q = self._quad(self._integrand, [0, self.t_debye_alumina / t_gap])                 # Temp:148
        return float(9 * t_gap * self.n_alumina * self.boltzman * (t_gap / self.t_debye_alumina) ** 3 * q)   # Temp:152


class _GapWorkers:
    """A few forked copies of this process that evaluate SurfaceEnergies.gap on request.

    fork without exec: a child keeps the parent's memory (mpmath, the constants, the quadrature nodes already cached) and
    nothing else — right after the fork it closes every inherited descriptor but its own two pipe ends (the parent's
    stdout / stderr pipes, sockets of a process group, the GPU's device files: a child that held them would keep a pipe
    reader waiting after the parent is gone, and count as a process with the GPU open), points 0-2 at /dev/null, asks
    the kernel to kill it when the parent dies, never touches the GPU runtime and leaves through os._exit.  A request is
    the 8 bytes of a double; the answer likewise."""

    def __init__(self, energies, n):
        import os
        import struct
        self._struct = struct.Struct("<d")
        self.workers = []
        parent = os.getpid()
        try:
            for _ in range(n):
                req_r, req_w = os.pipe()
                ans_r, ans_w = os.pipe()
                pid = os.fork()
                if pid == 0:
                    self._child(energies, req_r, ans_w, parent)          # never returns
                os.close(req_r)
                os.close(ans_w)
                self.workers.append((pid, req_w, ans_r))
        except OSError:
            self.close()
            raise
        import atexit
        atexit.register(self.close)

    def _child(self, energies, req_r, ans_w, parent):
        import os
        code = 0
        try:
            try:
                import ctypes
                import signal
                ctypes.CDLL(None).prctl(1, int(signal.SIGKILL), 0, 0, 0)     # PR_SET_PDEATHSIG
                if os.getppid() != parent:
                    os._exit(0)
            except Exception:
                pass
            keep = {req_r, ans_w}
            try:
                fds = [int(f) for f in os.listdir("/proc/self/fd")]
            except OSError:
                fds = list(range(3, 1024))
            null = os.open(os.devnull, os.O_RDWR)
            keep.add(null)
            for fd in fds:
                if fd > 2 and fd not in keep:
                    try:
                        os.close(fd)
                    except OSError:
                        pass
            for fd in (0, 1, 2):
                os.dup2(null, fd)
            st = self._struct
            while True:
                buf = b""
                while len(buf) < 8:
                    chunk = os.read(req_r, 8 - len(buf))
                    if not chunk:
                        os._exit(0)                                         # the parent closed the pipe
                    buf += chunk
                os.write(ans_w, st.pack(energies.gap(st.unpack(buf)[0])))
        except BaseException:
            code = 1
        finally:
            os._exit(code)

    def alive(self):
        import os
        for pid, _, _ in self.workers:
            try:
                if os.waitpid(pid, os.WNOHANG)[0] != 0:
                    return False
            except ChildProcessError:
                return False
        return bool(self.workers)

    def map(self, zs):
        import os
        st, nw = self._struct, len(self.workers)
        out = [0.0] * len(zs)
        for base in range(0, len(zs), nw):                       # one request in flight per worker
            batch = zs[base:base + nw]
            for k, z in enumerate(batch):
                os.write(self.workers[k][1], st.pack(z))
            for k in range(len(batch)):
                buf = b""
                while len(buf) < 8:
                    chunk = os.read(self.workers[k][2], 8 - len(buf))
                    if not chunk:
                        raise RuntimeError("a gap-energy worker went away")
                    buf += chunk
                out[base + k] = st.unpack(buf)[0]
        return out

    # the two halves of map(), for callers that have something else to do while the workers integrate: every request is
    # written at once (worker k gets requests k, k + nw, ...: a pipe holds thousands of them and a worker answers in order),
    # the answers are read later
    def start(self, zs):
        import os
        st, nw = self._struct, len(self.workers)
        for k, z in enumerate(zs):
            os.write(self.workers[k % nw][1], st.pack(z))
        return len(zs)

    def finish(self, count):
        import os
        st, nw = self._struct, len(self.workers)
        out = [0.0] * count
        for k in range(count):
            buf = b""
            while len(buf) < 8:
                chunk = os.read(self.workers[k % nw][2], 8 - len(buf))
                if not chunk:
                    raise RuntimeError("a gap-energy worker went away")
                buf += chunk
            out[k] = st.unpack(buf)[0]
        return out

    def close(self):
        import os
        import signal
        workers, self.workers = self.workers, []
        for pid, req_w, ans_r in workers:
            for fd in (req_w, ans_r):
                try:
                    os.close(fd)
                except OSError:
                    pass
        for pid, _, _ in workers:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
            try:
                os.waitpid(pid, 0)
            except (ChildProcessError, OSError):
                pass


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
    had_momentum, had_cold, had_hot) for the step, accumulated in the reference's order.

    Per hit the reference draws the direction first and evaluates the surface energy second (Temp:367-368 and
    alike); the energy consumes no random numbers, so all directions of a case are drawn first (``sample_case``) and the
    gap energies of the case are evaluated together afterwards (``gap_many``: worker processes when there are several)."""
    # The gap case's integrals (mpmath.quad, ~1 ms each, five per step at N = 1e6) are half of the hand-over's host time.  Two
    # things take them off the critical path; neither assumes anything that is not checked:
    # * EARLY START.  The gap mask reads positions and prior positions only (Temp:720-721), and what the two cases before it
    #   change — particles parked on the planes z = h_oa -+ r_ar outside the gap zone — cannot enter or leave it: the gap
    #   case's hits are looked at once more ahead of case 3 (wall_hits changes nothing) and their contact heights go to the
    #   worker processes.  When the gap case's turn comes its hits are taken again and the early energies are used only if
    #   particle indices, contact heights and solve flags are identical.
    # * PARKING.  At its turn the gap case is parked (hooks.wall_park: completed paths, counters, particles at their contact
    #   points — all the following masks read) and finished (hooks.wall_finish: new velocities, dp_z) after case 9, when the
    #   energies have had the whole hand-over to arrive.  A parked particle keeps its old velocity meanwhile; should a later
    #   case hit one, the gap case is finished first and that case's hits are evaluated anew.
    # The per-case sums are folded in case order at the end: the same additions in the same order as the reference's loop.
    import os
    part = {}                                   # case -> (m_case, e_case, any good hit)
    early = None
    can_park = hasattr(hooks, "wall_park") and os.environ.get("AMC_TEMP_NO_PARK") != "1"
    force_redo = os.environ.get("AMC_TEMP_FORCE_GAP_REDO") == "1"       # (tests: take the finish-first path at the next case with hits)
    if getattr(hooks, "early_gap", False) and hasattr(energies, "gap_start"):
        e_idx, _, e_cz, e_ok = hooks.wall_hits(GAP_CASE)
        if len(e_idx):
            h = energies.gap_start(np.asarray(e_cz)[np.flatnonzero(np.asarray(e_ok))].tolist())
            if h is not None:
                early = (np.array(e_idx), np.array(e_cz), np.array(e_ok), h)
    parked = None                               # (idx, good, energy handle or contact heights)

    def finish_parked():
        nonlocal parked
        p_idx, p_good, p_src = parked
        Es = np.zeros(len(p_idx))
        Es[p_good] = energies.gap_finish(p_src) if isinstance(p_src, tuple) else energies.gap_many(p_src)
        dpz, _ = hooks.wall_finish(GAP_CASE, Es)
        part[GAP_CASE] = (sequential_sum(np.asarray(dpz, dtype=np.float64)[p_good].tolist()), None, len(p_good) > 0)
        parked = None

    with (sampler.session() if hasattr(sampler, "session") else _NoSession(sampler)):
        for case in CASES:
            idx, normals, contact_z, ok = hooks.wall_hits(case)
            n = len(idx)
            if parked is not None and n and (force_redo or np.intersect1d(parked[0], idx).size):
                # a later case hits a parked particle: its contact solve needs the velocity the gap case gives it
                force_redo = False
                finish_parked()
                hooks.wall_hits_again()
                idx, normals, contact_z, ok = hooks.wall_hits(case)
                n = len(idx)
            if n == 0:
                if case == GAP_CASE and early is not None:
                    energies.gap_finish(early[3])           # (drain the workers' answers: nothing hit after all)
                    early = None
                continue
            if hasattr(sampler, "sample_case"):
                dirs = sampler.sample_case(normals, ok)
            else:
                dirs = np.zeros((n, 3))
                for k in range(n):
                    if ok[k]:                               # (else: the reference's try-block fails before any RNG draw)
                        dirs[k] = sampler.random_inbounds_direction(np.array(normals[k]))
            good = np.flatnonzero(np.asarray(ok))
            Es = np.zeros(n)
            if case == GAP_CASE:
                src = None                                  # the energies: an early handle that matches, or still to be computed
                if early is not None:
                    if np.array_equal(early[0], idx) and np.array_equal(early[1], contact_z) and np.array_equal(early[2], ok):
                        src = early[3]
                    else:
                        energies.gap_finish(early[3])       # (discarded)
                    early = None
                zs = np.asarray(contact_z)[good].tolist()
                if can_park:
                    if src is None and hasattr(energies, "gap_start"):
                        src = energies.gap_start(zs)
                    hooks.wall_park(case, dirs)
                    parked = (np.array(idx), good, src if src is not None else zs)
                    continue
                if src is not None:
                    Es[good] = energies.gap_finish(src)
                elif hasattr(energies, "gap_many"):
                    Es[good] = energies.gap_many(zs)
                else:
                    for k in good:
                        Es[k] = energies.gap(contact_z[k])
            else:
                Es[good] = energies.cold if case in COLD_CASES else energies.hot
            dpz, dE = hooks.wall_apply(case, dirs, Es)
            # (left-to-right sums over plain Python floats: the same additions in the same order as the reference's loop)
            m_case = sequential_sum(np.asarray(dpz, dtype=np.float64)[good].tolist())
            e_case = None if case == GAP_CASE else sequential_sum(np.asarray(dE, dtype=np.float64)[good].tolist())
            part[case] = (m_case, e_case, len(good) > 0)
        if parked is not None:
            finish_parked()
    mom = cold = hot = 0
    had_m = had_c = had_h = False
    for case in CASES:
        if case not in part:
            continue
        m_case, e_case, any_good = part[case]
        mom = mom + m_case
        had_m = had_m or any_good
        if case in COLD_CASES:
            cold = cold + e_case
            had_c = had_c or any_good
        elif case in HOT_CASES:
            hot = hot + e_case
            had_h = had_h or any_good
    return mom, cold, hot, had_m, had_c, had_h


class _NoSession:
    def __init__(self, sampler):
        self.sampler = sampler

    def __enter__(self):
        return self.sampler

    def __exit__(self, *exc):
        return False


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
