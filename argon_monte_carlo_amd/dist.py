"""Multi-GPU driver: one process per GPU, particles sharded by contiguous index range.

Per step every rank advances only its shard (drift + walls + bounds), then ONE all-gather over RCCL/xGMI hands the
positions of every shard and the velocities that changed to everybody (~28 B per particle).  Detection is sharded by
index: a rank examines its own particles against everybody and keeps the pairs whose partner has the lower index (every
close pair is found once, by the owner of its higher index); a second, small all-gather hands the pairs of all ranks to
everybody, and every rank then resolves the whole system's candidates as a single GPU would (ordered resolve, commit).
(``replicated_detect=True`` or AMC_MG_REPLICATED=1: the round-2 form, detection of the whole system on every rank, one
collective per step — the default below 1.5 million particles in all, where the second collective costs more than the
detection it saves: measured)  All ranks compute every collision from
identical inputs, so cross-shard pairs and chains need no locking, no ownership logic inside the kernels and no further
exchange: the path accumulators and the flag of a particle only feed its own bookkeeping and are meaningful on its owner
alone, which is also the rank that emits the particle's completed paths.  Nothing in the step waits for the host.

``ShardedSimulation`` only needs an *engine* with the ``mg_*`` methods of ``engine.ShardEngine`` and a communicator;
tests/test_dist_gloo.py drives it with a NumPy engine over gloo on CPU.
"""
from __future__ import annotations

import numpy as np


def shard_range(n, rank, world):
    """Contiguous index range [lo, hi) of ``rank``; equal shards when world divides n."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TorchComm:
    """torch.distributed collectives on tensors that alias engine memory.  backend "nccl" (= RCCL on ROCm) works on the
    device tensors directly; "gloo" stages GPU tensors through the host (used for rehearsals on one GPU and on CPU)."""

    def __init__(self, rank, world, group=None):
        import torch.distributed as dist
        import os
        self.dist, self.rank, self.world, self.group = dist, rank, world, group
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # world == 1 normally skips the collectives; AMC_DIST_NOSHORTCUT=1 issues them anyway (single-GPU rehearsal
        # of the RCCL code path: tensors aliasing library memory, all-gather)
        self.shortcut = not (os.environ.get("AMC_DIST_NOSHORTCUT") == "1" and dist.is_initialized())

    def _stage(self, t):
        return t.is_cuda and self.backend != "nccl"

    def allgather_packed(self, send, recv):
        """recv = concatenation of every rank's `send` (equal sizes), in rank order."""
        if self._stage(send):
            import torch
            h = send.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(parts, h, group=self.group)
            recv.copy_(torch.cat(parts))
        else:
            self.dist.all_gather_into_tensor(recv, send, group=self.group)

    def gather_shards(self, mine, ranges):
        """all-gather of (possibly unequal) host shards: padded to the longest one."""
        import torch
        m = max(b - a for a, b in ranges)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        pad = torch.zeros(m, dtype=mine.dtype, device=dev)
        pad[:mine.numel()] = mine
        parts = [torch.empty(m, dtype=mine.dtype, device=dev) for _ in ranges]
        self.dist.all_gather(parts, pad, group=self.group)
        return [p[:b - a].cpu() for (a, b), p in zip(ranges, parts)]

    def allreduce_sum_ints(self, values):
        import torch
        if self.world == 1 and self.shortcut:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.int64)
        if self.backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, group=self.group)
        return [int(v) for v in t.cpu().tolist()]

    def allgather_var(self, rows):
        """all-gather of float64 row blocks of different lengths: list of the blocks of all ranks, in rank order."""
        import torch
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if self.world == 1 and self.shortcut:
            return [rows]
        counts = [0] * self.world
        counts[self.rank] = rows.shape[0]
        counts = self.allreduce_sum_ints(counts)
        width = rows.shape[1]
        if sum(counts) == 0:
            return [rows[:0].copy() for _ in counts]
        ranges = [(0, c * width) for c in counts]
        parts = self.gather_shards(torch.from_numpy(rows.reshape(-1)), ranges)
        return [p.numpy().reshape(-1, width) for p in parts]


class ShardedSimulation:
    SHARDED_DETECT_MIN_N = 1_500_000
    SUM_KEYS = ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors")

    def __init__(self, params, rank, world, backend="nccl", stream_ptr=None, engine=None, comm=None, replicated_detect=None):
        import os
        self.params, self.rank, self.world = params, int(rank), int(world)
        self.n = int(params.n)
        if replicated_detect is None:
            env = os.environ.get("AMC_MG_REPLICATED")
            # sharded detection saves ~0.7 x 54 ns x N of detect time per rank (eight ranks) and costs a second collective
            # (~30 us over RCCL): worth it from ~1.5 million particles in all
            replicated_detect = (env == "1") if env is not None else (self.n < self.SHARDED_DETECT_MIN_N)
        self.replicated_detect = bool(replicated_detect)
        self.lo, self.hi = shard_range(self.n, rank, world)
        if engine is None:
            from .engine import ShardEngine
            engine = ShardEngine(params, self.lo, self.hi)
            if stream_ptr is None:
                # the collectives (and their host staging under gloo) are torch operations: the library's kernels must
                # run on the same stream, or a table could be read before the kernel that fills it has finished
                import torch
                stream_ptr = torch.cuda.current_stream().cuda_stream
            engine.set_stream(stream_ptr)
        self.engine = engine
        self.comm = comm if comm is not None else TorchComm(rank, world)
        self._comm_events = None

    def upload(self, *arrays, **kw):
        """Every rank uploads the full initial state (only its shard of the non-position arrays is ever used)."""
        self.engine.upload(*arrays, **kw)

    def init_synthetic(self, cfg):
        """Initial conditions generated on the device (``ic.device_ic_config``): a particle's numbers depend on the seed and
        its index only, so every rank builds the identical system."""
        self.engine.init_synthetic(cfg)

    # ---- one step -------------------------------------------------------------------------------------------------------
    def timestep(self, dt, reduce_stats=True, want_stats=True):
        self.engine.mg_local(dt)
        return self._sweep(reduce_stats, want_stats)

    def _sweep(self, reduce_stats=True, want_stats=True):
        """positions (+ changed velocities) of all shards -> everybody, then the sweep of the whole system on every rank"""
        e = self.engine
        if not (self.world == 1 and self.comm.shortcut):
            send, recv = e.exchange_buffers(self.world)
            e.mg_pack(self.world)
            if self._comm_events is None:
                self.comm.allgather_packed(send, recv)
            else:                                   # measurement: the collective bracketed by events on the launch stream
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.comm.allgather_packed(send, recv)
                b.record()
                self._comm_events.append((a, b))
        if self.replicated_detect or (self.world == 1 and self.comm.shortcut) or not hasattr(e, "mg_detect"):
            e.mg_sweep(self.world, self.rank)
        else:
            csend, crecv = e.candidate_buffers(self.world)
            e.mg_detect(self.world, self.rank)              # the other shards in, then my particles against everybody
            if self._comm_events is None:
                self.comm.allgather_packed(csend, crecv)    # everybody's candidate pairs to everybody
            else:
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.comm.allgather_packed(csend, crecv)
                b.record()
                self._comm_events.append((a, b))
            e.mg_resolve(self.world)                        # the same candidate graph and ordered resolve on every rank
        st = e.mg_finish(want_stats)
        if st is None:
            return None
        if reduce_stats and (self.world > 1 or not self.comm.shortcut):
            tot = self.comm.allreduce_sum_ints([st[k] for k in self.SUM_KEYS])
            st.update(dict(zip(self.SUM_KEYS, tot)))
        return st

    def profile_collective(self, enable):
        """Bracket every all-gather with events on the launch stream (bench.py's per-kernel pass, GPU engines only)."""
        self._comm_events = [] if enable else None

    def collective_times(self):
        """(total milliseconds, count) of the bracketed all-gathers since profile_collective(True); synchronises."""
        ev = self._comm_events or []
        if not ev:
            return 0.0, 0
        ev[-1][1].synchronize()
        tot = sum(a.elapsed_time(b) for a, b in ev)
        n = len(ev)
        self._comm_events = []
        return tot, n

    def run(self, dt, nsteps):
        acc = None
        nsteps = int(nsteps)
        for k in range(nsteps):
            # counters are cumulative on the device: only the last step reads them back (deltas since the last read)
            st = self.timestep(dt, reduce_stats=False, want_stats=(k == nsteps - 1))
            if st is not None:
                acc = st
        if acc is not None and self.world > 1:
            tot = self.comm.allreduce_sum_ints([acc[k] for k in self.SUM_KEYS])
            acc.update(dict(zip(self.SUM_KEYS, tot)))
        return acc

    # ---- results ---------------------------------------------------------------------------------------------------------
    def download(self):
        """Full state assembled from the owners (every rank returns the same arrays)."""
        st = self.engine.download()
        if self.world == 1:
            return st
        import torch
        out = {}
        ranges = [shard_range(self.n, r, self.world) for r in range(self.world)]
        for k, v in st.items():
            t = torch.from_numpy(np.ascontiguousarray(v))
            parts = self.comm.gather_shards(t[self.lo:self.hi].contiguous(), ranges)
            out[k] = torch.cat(parts).numpy()
        return out

    def histograms(self):
        """Global free-path histograms = sum over ranks (every completed path is emitted by its particle's owner)."""
        counts, tot = self.engine.histograms()
        if self.world == 1:
            return counts, tot
        flat = self.comm.allreduce_sum_ints(list(counts.astype(np.int64).ravel()) + [int(tot)])
        return np.array(flat[:-1], dtype=np.uint64).reshape(counts.shape), flat[-1]


class _GlobalWallHooks:
    """``drive_energised_cases`` hooks over all shards: the hits of a case are concatenated in rank order — ascending
    particle index, the order of the reference's ``np.where(hits)`` loop (Temp:132-152, 311-553) — on every rank, so
    every rank draws the same random directions from identically seeded streams and applies its own slice."""

    def __init__(self, engine, comm, rank):
        self.e, self.comm, self.rank = engine, comm, rank
        # the gap case parked at its turn and finished after the last case (energised.drive_energised_cases), when the engine can
        if hasattr(engine, "wall_park"):
            self.early_gap = True
            self.wall_park, self.wall_finish, self.wall_hits_again = self._wall_park, self._wall_finish, engine.wall_hits_again

    def _wall_park(self, case, dirs):
        self._parked = (self._off, self._cnt)
        a, b = self._off, self._off + self._cnt
        self.e.wall_park(case, np.asarray(dirs)[a:b])

    def _wall_finish(self, case, Es):
        a, b = self._parked[0], self._parked[0] + self._parked[1]
        dpz, dE = self.e.wall_finish(case, np.asarray(Es)[a:b])
        parts = self.comm.allgather_var(np.column_stack([dpz, dE]).reshape(-1, 2))
        allr = np.concatenate(parts)
        return allr[:, 0].copy(), allr[:, 1].copy()

    def wall_hits(self, case):
        idx, normals, contact_z, ok = self.e.wall_hits(case)
        mine = np.column_stack([idx.astype(np.float64), normals.reshape(-1, 3), contact_z, ok.astype(np.float64)])
        parts = self.comm.allgather_var(mine.reshape(-1, 6))
        self._off = sum(len(p) for p in parts[:self.rank])
        self._cnt = len(parts[self.rank])
        allr = np.concatenate(parts) if parts else mine
        return allr[:, 0].astype(np.int32), allr[:, 1:4].copy(), allr[:, 4].copy(), allr[:, 5] != 0.0

    def wall_apply(self, case, dirs, Es):
        a, b = self._off, self._off + self._cnt
        dpz, dE = self.e.wall_apply(case, np.asarray(dirs)[a:b], np.asarray(Es)[a:b])
        parts = self.comm.allgather_var(np.column_stack([dpz, dE]).reshape(-1, 2))
        allr = np.concatenate(parts)
        return allr[:, 0].copy(), allr[:, 1].copy()


class ShardedTemperatureSimulation(ShardedSimulation):
    """Temperature_Pore_MC.py's step (Temp:662-853) over index-range shards.  ``sampler`` / ``energies`` are the host
    objects of argon_monte_carlo_amd.energised; every rank must construct them with the same seeds."""

    def __init__(self, params, rank, world, backend="nccl", stream_ptr=None, engine=None, comm=None):
        if engine is None:
            from .engine import ShardEnergisedEngine
            lo, hi = shard_range(int(params.n), rank, world)
            params.reserved0 |= 1                   # the energised masks read prior_*_vals
            engine = ShardEnergisedEngine(params, lo, hi)
            if stream_ptr is None:
                import torch
                stream_ptr = torch.cuda.current_stream().cuda_stream
            engine.set_stream(stream_ptr)
        super().__init__(params, rank, world, backend=backend, stream_ptr=stream_ptr, engine=engine, comm=comm)
        self._hooks = _GlobalWallHooks(self.engine, self.comm, self.rank)

    def temp_timestep(self, dt, sampler, energies, reduce_stats=True):
        from .energised import drive_energised_cases
        e = self.engine
        e.temp_begin(dt)                                                # drift + specular cases on the shard
        res = drive_energised_cases(self._hooks, sampler, energies)     # the seven energised cases, global RNG order
        e.mg_bounds()                                                   # Temp:804
        st = self._sweep(reduce_stats, True)                            # Temp:813-844
        return (st,) + res
