"""world_size-2 gloo test (CPU) of the multi-GPU driver argon_monte_carlo_amd.dist.ShardedSimulation.

The driver is exercised with a NumPy engine that implements the mg_* protocol with simple deterministic rules (not the
physics — that is tested on the GPU against the oracle): state owned by another rank is poisoned with NaN, so any use
of non-exchanged state, a wrong shard range, a non-canonical exchange order or a lost -0.0 in the bit-exact exchange
shows up as a difference from the single-process run of the same engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from argon_monte_carlo_amd import params as PR
from argon_monte_carlo_amd.dist import ShardedSimulation, TorchComm, shard_range

KEYS = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz"]


class NumpyShardEngine:
    def __init__(self, n, lo, hi, cr):
        self.n, self.lo, self.hi, self.cr = n, lo, hi, cr
        self.a = {k: np.zeros(n) for k in KEYS}
        self.flag = np.zeros(n, dtype=np.uint8)
        self.table = np.zeros(11 * n)
        self.known = set()
        self.cand = None
        self.round = 0
        self.pending = {}

    # -- Engine surface used by the driver
    def upload(self, x, y, z, vx, vy, vz, d=None, dx=None, dy=None, dz=None, flag=None):
        for k, v in zip(KEYS, [x, y, z, vx, vy, vz, d, dx, dy, dz]):
            if v is not None:
                self.a[k][:] = v
        own = np.zeros(self.n, dtype=bool)
        own[self.lo:self.hi] = True
        for k in KEYS[3:]:
            self.a[k][~own] = np.nan              # state of other ranks' particles is NOT available locally

    def download(self):
        out = {k: v.copy() for k, v in self.a.items()}
        out["flag"] = self.flag.copy()
        return out

    def position_tensors(self):
        return [torch.from_numpy(self.a[k]) for k in "xyz"]

    def exchange_tensor(self, m):
        return torch.from_numpy(self.table[:11 * m].view(np.int64))

    def mg_local(self, dt):
        s = slice(self.lo, self.hi)
        for p, v in zip("xyz", ("vx", "vy", "vz")):
            self.a[p][s] += dt * self.a[v][s]
        self.a["d"][s] += 1.0

    def mg_detect(self):
        P = np.stack([self.a[k] for k in "xyz"], 1)
        assert not np.isnan(P).any()
        d2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
        i, j = np.nonzero(np.tril(d2 < self.cr ** 2, -1))
        order = np.lexsort((j, i))
        self.cand = (i[order].astype(np.int32), j[order].astype(np.int32))
        # hand the pairs over in a rank-dependent order: the driver must canonicalise
        perm = np.random.default_rng(self.lo + 1).permutation(len(order))
        self.cand_out = (self.cand[0][perm], self.cand[1][perm])
        self.known, self.round, self.pending = set(), 0, {}
        return len(order)

    def mg_candidates(self, ncand):
        return self.cand_out

    def mg_pack(self, particles):
        m = len(particles)
        t = self.table[:11 * m].reshape(11, m)
        t[:] = 0.0
        for u, p in enumerate(particles):
            if self.lo <= p < self.hi:
                t[:10, u] = [self.a[k][p] for k in KEYS]
                t[10, u] = float(self.flag[p])

    def mg_unpack(self, particles):
        m = len(particles)
        t = self.table[:11 * m].reshape(11, m)
        for u, p in enumerate(particles):
            self.known.add(int(p))
            if not (self.lo <= p < self.hi):
                for e, k in enumerate(KEYS):
                    self.a[k][p] = t[e, u]
                self.flag[p] = t[10, u] != 0

    def mg_resolve_round(self, first):
        self.round += 1
        ci, cj = self.cand
        self.pending = {}
        new = set()
        v = {k: self.a[k].copy() for k in ("vx", "vy", "vz")}
        for i, j in zip(ci, cj):
            assert int(i) in self.known and int(j) in self.known
            for k in v:                               # "collision": exchange velocities (needs both particles' state)
                v[k][i], v[k][j] = v[k][j], v[k][i]
                assert not np.isnan(v[k][i]) and not np.isnan(v[k][j])
            m = int((7 * int(i) + 3 * int(j)) % self.n)   # rule that pulls a third particle into the "cluster"
            if m not in self.known:
                new.add(m)
            elif self.round > 1 and m != i and m != j:
                v["vx"][i] = v["vx"][i] + self.a["vx"][m] * 0.5
                assert not np.isnan(v["vx"][i])
            self.pending[int(i)] = None
            self.pending[int(j)] = None
        self.vnew = v
        return (len(new) > 0), np.array(sorted(new), dtype=np.int32)

    def mg_commit(self):
        for p in self.pending:
            for k in ("vx", "vy", "vz"):
                self.a[k][p] = self.vnew[k][p]

    # -- energised-wall hooks (fake rules; the point is the global ordering of hits and of the RNG draws)
    def temp_begin(self, dt):
        self.mg_local(dt)
        self.step = getattr(self, "step", 0) + 1

    def wall_hits(self, case):
        own = np.arange(self.lo, self.hi)
        hit = own[(own * 7 + case * 3 + self.step) % 11 == 0]
        nm = np.zeros((len(hit), 3))
        nm[:, case % 3] = 1.0 if case % 2 else -1.0
        ok = (hit % 5) != 0                                        # some contact solves "fail": no RNG draw for them
        self._hit = hit
        return hit.astype(np.int32), nm, self.a["z"][hit].copy(), ok

    def wall_apply(self, case, dirs, Es):
        hit = self._hit
        assert len(dirs) == len(hit) == len(Es)
        sp = np.sqrt(self.a["vx"][hit] ** 2 + self.a["vy"][hit] ** 2 + self.a["vz"][hit] ** 2)
        assert not np.isnan(sp).any()
        old_vz = self.a["vz"][hit].copy()
        for k, v in enumerate(("vx", "vy", "vz")):
            self.a[v][hit] = np.asarray(dirs)[:, k] * sp
        return old_vz - self.a["vz"][hit], np.asarray(Es) * sp

    def mg_bounds(self):
        pass

    def mg_finish(self, want_stats=True):
        return dict(n_pp=len(self.cand[0]) if self.lo == 0 else 0, n_wall=self.hi - self.lo, n_oob_walls=0, n_oob_pp=0,
                    n_paths=0, n_candidates=len(self.cand[0]), n_clusters=0, n_rounds=self.round, n_fp_errors=0, flags=0)


def make_state(n, seed=5):
    rng = np.random.default_rng(seed)
    pos = rng.random((3, n)) * 4e-9
    vel = rng.normal(size=(3, n)) * 300.0
    vel[0, ::7] = -0.0                                  # the int64-sum exchange must keep the sign of zero
    return [pos[0], pos[1], pos[2], vel[0], vel[1], vel[2]]


def run_sim(n, rank, world, steps, comm):
    cr = 3.385137501286538e-10
    p, _ = PR.cube_params(n=n)
    lo, hi = shard_range(n, rank, world)
    eng = NumpyShardEngine(n, lo, hi, cr)
    sim = ShardedSimulation(p, rank, world, engine=eng, comm=comm)
    sim.upload(*make_state(n))
    tot = None
    for s in range(steps):
        st = sim.timestep(2e-13)
        tot = st if tot is None else {k: tot[k] + st[k] for k in st}
    return sim, tot


def _worker(rank, world, port, n, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sim, tot = run_sim(n, rank, world, steps, TorchComm(rank, world))
        full = sim.download()
        if rank == 0:
            q.put(({k: v for k, v in full.items()}, tot))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _NoComm:
    world = 1
    backend = "none"

    def allgather_inplace(self, *a): pass
    def allreduce_bits(self, *a): pass
    def allreduce_sum_ints(self, v): return list(v)


@pytest.mark.parametrize("n", [240, 251])       # equal and unequal shards
def test_two_ranks_equal_one_rank(n):
    steps = 6
    ref_sim, ref_tot = run_sim(n, 0, 1, steps, _NoComm())
    ref = ref_sim.engine.download()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, tot = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ref_tot["n_candidates"] > 0 and ref_tot["n_rounds"] > steps      # multi-round exchange was exercised
    for k in KEYS:
        a, b = got[k], ref[k]
        assert np.array_equal(a.view(np.int64), b.view(np.int64)), k          # bitwise, incl. -0.0
    for k in ("n_pp", "n_wall"):
        assert tot[k] == ref_tot[k], k


def test_shard_ranges_cover_everything():
    for n in (1, 7, 100, 1000003):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))


# ---- energised walls over shards: global hit order and RNG consumption ---------------------------------------------------------
class _Energies:
    cold, hot = 1.0e-21, 2.0e-21

    def gap(self, z):
        return 1.5e-21 + z


def run_temp(n, rank, world, steps, comm):
    import random
    from argon_monte_carlo_amd.dist import ShardedTemperatureSimulation
    from argon_monte_carlo_amd.energised import DirectionSampler
    p, _ = PR.cube_params(n=n)
    lo, hi = shard_range(n, rank, world)
    sim = ShardedTemperatureSimulation(p, rank, world, engine=NumpyShardEngine(n, lo, hi, 3.385137501286538e-10), comm=comm)
    sim.upload(*make_state(n, seed=9))
    sampler = DirectionSampler(np.random.RandomState(3), random.Random(3))
    out = []
    for s in range(steps):
        st, mom, cold, hot, hm, hc, hh = sim.temp_timestep(2e-13, sampler, _Energies())
        out.append((float(mom), float(cold), float(hot), hm, hc, hh, st["n_pp"]))
    return sim, out


def _temp_worker(rank, world, port, n, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sim, out = run_temp(n, rank, world, steps, TorchComm(rank, world))
        full = sim.download()
        if rank == 0:
            q.put((full, out))
    finally:
        dist.destroy_process_group()


class _NoCommVar(_NoComm):
    def allgather_var(self, rows):
        return [np.ascontiguousarray(rows, dtype=np.float64)]


def test_energised_walls_two_ranks_equal_one_rank():
    n, steps = 251, 5
    ref_sim, ref_out = run_temp(n, 0, 1, steps, _NoCommVar())
    ref = ref_sim.engine.download()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_temp_worker, args=(r, 2, port, n, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert any(o[3] for o in ref_out)                                          # wall hits happened
    assert out == ref_out                                                       # per-step momentum / energy sums, bitwise
    for k in KEYS:
        assert np.array_equal(got[k].view(np.int64), ref[k].view(np.int64)), k
