"""Thin object wrapper over the C ABI: one ``Engine`` = one ``amc_ctx`` = the particle state resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._abi import (AMC_K_NAMES, AmcParams, AmcPathRecord, AmcStepStats, path_record_dtype)

_dp = C.POINTER(C.c_double)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a, n):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != (n,):
        raise ValueError(f"expected float64[{n}], got {a.shape}")
    return a


class Engine:
    def __init__(self, params: AmcParams):
        self.lib = _lib.load()
        self.params = params
        self.n = int(params.n)
        self._ctx = C.c_void_p()
        rc = self.lib.amc_create(C.byref(self._ctx), C.byref(params))
        if rc != 0:
            msg = self.lib.amc_last_error(None)
            raise _lib.ArgonMCError(rc, msg.decode() if msg else "amc_create failed")

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.amc_last_error(self._ctx)
            raise _lib.ArgonMCError(rc, msg.decode() if msg else "")

    def close(self):
        if self._ctx:
            self.lib.amc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state -----------------------------------------------------------------------------------------------
    def upload(self, x=None, y=None, z=None, vx=None, vy=None, vz=None, d=None, dx=None, dy=None, dz=None, flag=None):
        arrs = [_f64(a, self.n) for a in (x, y, z, vx, vy, vz, d, dx, dy, dz)]
        f = None
        if flag is not None:
            f = np.ascontiguousarray(np.asarray(flag).astype(np.uint8))
        self._ck(self.lib.amc_upload(self._ctx, *[_d(a) for a in arrs],
                                     None if f is None else f.ctypes.data_as(C.POINTER(C.c_uint8))))

    def init_synthetic(self, cfg):
        """Synthetic initial conditions generated on the device (``ic.device_ic_config``); replaces ``upload``."""
        self._ck(self.lib.amc_init_synthetic(self._ctx, C.byref(cfg)))

    def download(self):
        n = self.n
        arrs = [np.empty(n) for _ in range(10)]
        f = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.amc_download(self._ctx, *[_d(a) for a in arrs], f.ctypes.data_as(C.POINTER(C.c_uint8))))
        keys = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz"]
        out = dict(zip(keys, arrs))
        out["flag"] = f
        return out

    def download_prior(self):
        a = [np.empty(self.n) for _ in range(3)]
        self._ck(self.lib.amc_download_prior(self._ctx, *[_d(v) for v in a]))
        return a

    # -- stepping ----------------------------------------------------------------------------------------------
    def timestep(self, dt):
        st = AmcStepStats()
        self._ck(self.lib.amc_timestep(self._ctx, float(dt), C.byref(st)))
        return st.as_dict()

    def run(self, dt, nsteps):
        st = AmcStepStats()
        self._ck(self.lib.amc_run(self._ctx, float(dt), int(nsteps), C.byref(st)))
        return st.as_dict()

    def stage_drift(self, dt):
        self._ck(self.lib.amc_stage_drift(self._ctx, float(dt)))

    def stage_walls(self):
        st = AmcStepStats()
        self._ck(self.lib.amc_stage_walls(self._ctx, C.byref(st)))
        return st.as_dict()

    def stage_bounds(self):
        n = C.c_int64(0)
        self._ck(self.lib.amc_stage_bounds(self._ctx, C.byref(n)))
        return n.value

    def stage_sweep(self):
        st = AmcStepStats()
        self._ck(self.lib.amc_stage_sweep(self._ctx, C.byref(st)))
        return st.as_dict()

    def pairwise_cell(self, cont, cx, cy, cz, flag, x, y, z, vx, vy, vz):
        """One cell through the device pair kernel; arrays are updated IN PLACE (float64 / uint8, contiguous)."""
        n = len(x)
        cap = max(16, 8 * n)
        paths = np.zeros((4, cap))
        npaths = C.c_size_t(0)
        ncoll = C.c_int64(0)
        self._ck(self.lib.amc_pairwise_cell(self._ctx, n, _d(cont), _d(cx), _d(cy), _d(cz),
                                            flag.ctypes.data_as(C.POINTER(C.c_uint8)), _d(x), _d(y), _d(z), _d(vx),
                                            _d(vy), _d(vz), _d(paths), cap, C.byref(npaths), C.byref(ncoll)))
        return paths[:, :npaths.value].T.copy(), ncoll.value

    # -- outputs -------------------------------------------------------------------------------------------------
    def drain_paths(self, sort=True):
        pend = C.c_size_t(0)
        self._ck(self.lib.amc_paths_pending(self._ctx, C.byref(pend)))
        rec = np.zeros(max(1, pend.value), dtype=path_record_dtype())
        got = C.c_size_t(0)
        self._ck(self.lib.amc_drain_paths(self._ctx, rec.ctypes.data_as(C.POINTER(AmcPathRecord)), len(rec),
                                          C.byref(got)))
        rec = rec[:got.value]
        if sort and len(rec):
            # the reference's append order: step, wall cases in evaluation order, then colour groups / cells,
            # inside a cell i ascending, j ascending, particle j before particle i (Pore:168-199)
            rec = rec[np.lexsort((rec["which"], rec["j"], rec["i"], rec["cell"], rec["phase"], rec["step"]))]
        return rec

    def histograms(self):
        nb = int(self.params.hist_bins)
        counts = np.zeros((4, nb), dtype=np.uint64)
        tot = C.c_uint64(0)
        self._ck(self.lib.amc_histograms(self._ctx, counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(tot)))
        return counts, tot.value

    def reset_outputs(self):
        self._ck(self.lib.amc_reset_outputs(self._ctx))

    # -- measurement -----------------------------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        """Run on the given hipStream_t; 0 = HIP's NULL stream (torch's default current stream)."""
        if not stream_ptr:
            self._ck(self.lib.amc_use_null_stream(self._ctx))
        else:
            self._ck(self.lib.amc_set_stream(self._ctx, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._ck(self.lib.amc_synchronize(self._ctx))

    def overlap_stats(self):
        """(steps amc_run has overlapped, particles advanced again after a sweep pulled them in, mode, extra list nodes)."""
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.lib.amc_overlap_stats(self._ctx, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return dict(steps=int(out[0]), refiled=int(out[1]), mode=int(out[2]), extra_nodes=int(out[3]))

    def profile(self, on=True):
        self._ck(self.lib.amc_profile(self._ctx, int(on)))

    def kernel_times(self):
        nk = len(AMC_K_NAMES)
        ms = np.zeros(nk)
        cnt = np.zeros(nk, dtype=np.int64)
        self._ck(self.lib.amc_kernel_times(self._ctx, _d(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return {AMC_K_NAMES[k]: (float(ms[k]), int(cnt[k])) for k in range(nk)}


class ShardEngine(Engine):
    """Engine + the multi-GPU entry points (include/argonmc.h, "multi-GPU"): the object dist.ShardedSimulation drives.
    The buffers of the per-step all-gather are exposed as torch tensors that alias the library's device memory."""

    def __init__(self, params, lo, hi):
        super().__init__(params)
        self.lo, self.hi = int(lo), int(hi)
        self._ck(self.lib.amc_set_shard(self._ctx, self.lo, self.hi))

    @staticmethod
    def _wrap(ptr, count, typestr):
        import torch

        class _Dev:
            pass
        d = _Dev()
        d.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(d, device="cuda")

    def exchange_buffers(self, world):
        """(send, recv) tensors of the all-gather: this rank's block (float64[block]: positions of the shard + the list of
        velocity changes) and the blocks of all ranks (float64[world * block])."""
        if getattr(self, "_xb_world", None) != world:
            send, recv, blk = C.c_void_p(), C.c_void_p(), C.c_int64(0)
            self._ck(self.lib.amc_mg_exchange_view(self._ctx, int(world), C.byref(send), C.byref(recv), C.byref(blk)))
            self._xb = (self._wrap(send.value, blk.value, "<f8"), self._wrap(recv.value, blk.value * world, "<f8"))
            self._xb_world = world
        return self._xb

    def mg_local(self, dt):
        self._ck(self.lib.amc_mg_local(self._ctx, float(dt)))

    def mg_pack(self, world):
        self._ck(self.lib.amc_mg_pack(self._ctx, int(world)))

    def mg_sweep(self, world, rank):
        self._ck(self.lib.amc_mg_sweep(self._ctx, int(world), int(rank)))

    def candidate_buffers(self, world):
        """(send, recv) tensors of the step's SECOND all-gather (detection sharded by index): this rank's candidate pairs
        (int32[block]: count, -, then (i, j) pairs) and the blocks of all ranks."""
        if getattr(self, "_cb_world", None) != world:
            send, recv, blk = C.c_void_p(), C.c_void_p(), C.c_int64(0)
            self._ck(self.lib.amc_mg_candidates_view(self._ctx, int(world), C.byref(send), C.byref(recv), C.byref(blk)))
            self._cb = (self._wrap(send.value, blk.value, "<i4"), self._wrap(recv.value, blk.value * world, "<i4"))
            self._cb_world = world
        return self._cb

    def mg_detect(self, world, rank):
        self._ck(self.lib.amc_mg_detect(self._ctx, int(world), int(rank)))

    def mg_resolve(self, world):
        self._ck(self.lib.amc_mg_resolve(self._ctx, int(world)))

    def mg_finish(self, want_stats=True):
        if not want_stats:
            self._ck(self.lib.amc_mg_finish(self._ctx, None))
            return None
        st = AmcStepStats()
        self._ck(self.lib.amc_mg_finish(self._ctx, C.byref(st)))
        return st.as_dict()


class EnergisedEngine(Engine):
    """Engine + the energised-wall hand-over (Temp:705-758): ``wall_hits`` / ``wall_apply`` are the hooks that
    ``energised.drive_energised_cases`` drives once per case."""
    early_gap = True            # wall_hits(case) may be called ahead of its turn (it changes nothing): the gap case's integrals start early

    def temp_begin(self, dt):
        self._ck(self.lib.amc_temp_begin(self._ctx, float(dt)))

    def wall_hits(self, case):
        cap = max(4096, self.n // 8 + 1024)
        buf = getattr(self, "_wall_hit_buffers", None)
        if buf is None or len(buf[0]) != cap:       # (kept across calls: seven allocations of megabytes per step otherwise)
            buf = self._wall_hit_buffers = (np.empty(cap, dtype=np.int32), np.empty((cap, 3)), np.empty(cap))
        idx, normal, cz = buf
        n = C.c_size_t(0)
        self._ck(self.lib.amc_wall_hits(self._ctx, int(case), idx.ctypes.data_as(C.POINTER(C.c_int32)), _d(normal), _d(cz),
                                        cap, C.byref(n)))
        k = n.value
        nm = normal[:k].copy()
        ok = np.any(nm != 0.0, axis=1)          # a zero normal marks a failed contact solve (Temp:472-474)
        return idx[:k].copy(), nm, cz[:k].copy(), ok

    def wall_apply(self, case, dirs, Es):
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        Es = np.ascontiguousarray(Es, dtype=np.float64)
        n = len(Es)
        dpz = np.zeros(max(1, n))
        dE = np.zeros(max(1, n))
        self._ck(self.lib.amc_wall_apply(self._ctx, int(case), _d(dirs), _d(Es), n, _d(dpz), _d(dE)))
        return dpz[:n], dE[:n]

    # a case parked while its surface energies are still being integrated (energised.drive_energised_cases)
    def wall_park(self, case, dirs):
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        self._ck(self.lib.amc_wall_park(self._ctx, int(case), _d(dirs), len(dirs)))

    def wall_finish(self, case, Es):
        Es = np.ascontiguousarray(Es, dtype=np.float64)
        n = len(Es)
        dpz = np.zeros(max(1, n))
        dE = np.zeros(max(1, n))
        self._ck(self.lib.amc_wall_finish(self._ctx, int(case), _d(Es), n, _d(dpz), _d(dE)))
        return dpz[:n], dE[:n]

    def wall_hits_again(self):
        self._ck(self.lib.amc_wall_hits_again(self._ctx))

    def temp_end(self):
        st = AmcStepStats()
        self._ck(self.lib.amc_temp_end(self._ctx, C.byref(st)))
        return st.as_dict()

    # ---- opt-in, non-parity: directions / energies drawn on the device (no host hand-over) ---------------------------
    def temp_cases_device(self, cfg):
        self._ck(self.lib.amc_temp_cases_device(self._ctx, C.byref(cfg)))

    def device_results(self, case):
        cap = max(4096, self.n // 64 + 1024)
        idx = np.empty(cap, dtype=np.int32)
        dpz, dE = np.empty(cap), np.empty(cap)
        ok = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._ck(self.lib.amc_temp_device_results(self._ctx, int(case), idx.ctypes.data_as(C.POINTER(C.c_int32)), _d(dpz), _d(dE),
                                                  ok.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
        k = n.value
        return idx[:k].copy(), dpz[:k].copy(), dE[:k].copy(), ok[:k].astype(bool)

    def device_draws(self, case):
        cap = max(4096, self.n // 64 + 1024)
        idx = np.empty(cap, dtype=np.int32)
        normal, dirs = np.empty((cap, 3)), np.empty((cap, 3))
        cz, Es = np.empty(cap), np.empty(cap)
        n = C.c_size_t(0)
        self._ck(self.lib.amc_temp_device_draws(self._ctx, int(case), idx.ctypes.data_as(C.POINTER(C.c_int32)), _d(normal), _d(cz),
                                                _d(dirs), _d(Es), cap, C.byref(n)))
        k = n.value
        return idx[:k].copy(), normal[:k].copy(), cz[:k].copy(), dirs[:k].copy(), Es[:k].copy()

    def temp_timestep_device(self, dt, cfg):
        """temp_timestep with the random draws on the device: same return tuple, no per-case host round trip."""
        self.temp_begin(dt)
        self.temp_cases_device(cfg)
        st = self.temp_end()
        sums = (C.c_double * 3)()
        had = (C.c_int32 * 3)()
        self._ck(self.lib.amc_temp_device_sums(self._ctx, sums, had))
        return (st, sums[0], sums[1], sums[2], bool(had[0]), bool(had[1]), bool(had[2]))

    def temp_timestep(self, dt, sampler, energies):
        """One iteration of Temperature_Pore_MC.py's loop (Temp:662-853)."""
        from .energised import drive_energised_cases
        self.temp_begin(dt)
        res = drive_energised_cases(self, sampler, energies)
        st = self.temp_end()
        return (st,) + res


class ShardEnergisedEngine(ShardEngine, EnergisedEngine):
    """Energised walls on a shard: ``temp_begin`` / ``wall_hits`` / ``wall_apply`` act on the owned index range,
    ``mg_bounds`` is the bounds check between the walls and the sweep (Temp:804); the sweep is ShardEngine's."""

    def mg_bounds(self):
        self._ck(self.lib.amc_mg_bounds(self._ctx))
