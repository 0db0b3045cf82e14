"""GPU (-m gpu): the sharded path with 2 (and 3) ranks on ONE MI355X (gloo, collectives staged through the host — RCCL refuses
two ranks on one device) must equal the single-context engine bit for bit: state, counters and histograms.  On a box with
two or more devices the same comparison runs over backend "nccl" (= RCCL), one rank per device."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KEYS = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz", "flag"]


def _case(kind, n):
    from argon_monte_carlo_amd import ic as IC, params as PR
    if kind == "cube":
        p, c = PR.cube_params_for_n(n)
        init = IC.cube_ic(p, c, seed=11)
    else:
        p, c = PR.pore_params(n=n)
        init = IC.pore_ic(p, c, seed=11)
    p.detect_mode = 1
    return p, c, init


def _worker(rank, world, port, kind, n, steps, q, device_ic_seed=None, backend="gloo"):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if backend == "nccl":                   # one rank per device, collectives on the device tensors
            import torch
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            from argon_monte_carlo_amd.dist import ShardedSimulation
            p, c, init = _case(kind, n)
            if backend == "nccl":
                p.device = rank
            sim = ShardedSimulation(p, rank, world, backend=backend)
            if device_ic_seed is None:
                sim.upload(*init)
            else:
                from argon_monte_carlo_amd import ic as IC
                sim.init_synthetic(IC.device_ic_config(p, c, device_ic_seed, kind))
            tot = sim.run(c["dt"], steps)
            full = sim.download()
            counts, npaths = sim.histograms()
            if rank == 0:
                q.put(("ok", (full, tot, counts, npaths)))
        finally:
            dist.destroy_process_group()
    except BaseException as e:                  # the parent shows the rank's own error, not a queue timeout
        import traceback
        q.put(("error", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))
        raise


def _run_ranks(world, args, timeout=300.0, target=None):
    """Start `world` ranks of _worker, return rank 0's result.  A rank that dies ends the test at once with its own
    error; whatever happens, no child process (they hold GPU contexts) outlives the call."""
    import queue
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    if target is None:
        procs = [ctx.Process(target=_worker, args=(r, world, port) + tuple(args[:3]) + (q,) + tuple(args[3:])) for r in range(world)]
    else:
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for pr in procs:
        pr.start()
    try:
        t0 = time.time()
        while True:
            try:
                kind, payload = q.get(timeout=1.0)
                break
            except queue.Empty:
                dead = [pr for pr in procs if pr.exitcode not in (None, 0)]
                if dead:
                    try:
                        kind, payload = q.get(timeout=2.0)      # its error message, if it got that far
                    except queue.Empty:
                        kind, payload = "error", f"a rank exited with code {dead[0].exitcode} without a message"
                    break
                if time.time() - t0 > timeout:
                    kind, payload = "error", f"no result after {timeout:.0f} s"
                    break
        if kind != "ok":
            pytest.fail(payload)
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
        return payload
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
        for pr in procs:
            pr.join(timeout=10)
            if pr.is_alive():
                pr.kill()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("kind,n,steps,world", [("cube", 30000, 8, 2), ("pore", 60001, 6, 2), ("cube", 400000, 5, 2),   # equal / unequal shards; large-sweep plan
                                                ("cube", 200000, 40, 2), ("pore", 500001, 30, 2),   # longer runs: deferred commits, slot release
                                                ("cube", 100003, 10, 3)])                           # three ranks, unequal shards
def test_ranks_on_one_gpu_equal_single_engine(kind, n, steps, world):
    from argon_monte_carlo_amd.engine import Engine
    p, c, init = _case(kind, n)
    eng = Engine(p)
    eng.upload(*init)
    ref_tot = eng.run(c["dt"], steps)
    ref = eng.download()
    ref_counts, ref_npaths = eng.histograms()
    eng.close()
    assert ref_tot["n_pp"] > 0
    full, tot, counts, npaths = _run_ranks(world, (kind, n, steps))
    for k in KEYS:
        assert np.array_equal(full[k], ref[k]), (kind, k, np.flatnonzero(full[k] != ref[k])[:5])
    for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
        assert tot[k] == ref_tot[k], (k, tot, ref_tot)
    assert npaths == ref_npaths and np.array_equal(counts, ref_counts)


@pytest.mark.parametrize("kind,n,steps,world", [("cube", 30000, 8, 2), ("pore", 60001, 6, 2), ("cube", 100003, 10, 3), ("pore", 500001, 30, 2)])
def test_detection_sharded_by_index_equals_single_engine(kind, n, steps, world, monkeypatch):
    """The two-collective form (a rank examines its own particles against everybody, candidate pairs all-gathered)
    forced at sizes where the driver would pick the replicated one: state, counters, histograms == the single engine."""
    monkeypatch.setenv("AMC_MG_REPLICATED", "0")
    test_ranks_on_one_gpu_equal_single_engine(kind, n, steps, world)


@pytest.mark.parametrize("kind,n,steps", [("cube", 200000, 20), ("pore", 500001, 12)])
def test_two_devices_over_rccl_equal_single_engine(kind, n, steps):
    """One rank per DEVICE, collectives on the device tensors over backend "nccl" (= RCCL over xGMI): the replacement of
    the reference's Pool.starmap over colour groups (Pore:545-549).  Needs two devices; a one-GPU box skips it (the same
    protocol runs there over gloo, above)."""
    import torch
    if torch.cuda.device_count() < 2:           # (counting devices does not initialise the GPU)
        pytest.skip("needs two GPUs")
    from argon_monte_carlo_amd.engine import Engine
    p, c, init = _case(kind, n)
    eng = Engine(p)
    eng.upload(*init)
    ref_tot = eng.run(c["dt"], steps)
    ref = eng.download()
    ref_counts, ref_npaths = eng.histograms()
    eng.close()
    full, tot, counts, npaths = _run_ranks(2, (kind, n, steps, None, "nccl"))
    for k in KEYS:
        assert np.array_equal(full[k], ref[k]), (kind, k, np.flatnonzero(full[k] != ref[k])[:5])
    for k in ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths"):
        assert tot[k] == ref_tot[k], (k, tot, ref_tot)
    assert npaths == ref_npaths and np.array_equal(counts, ref_counts)


def test_device_initial_conditions_do_not_depend_on_the_shard_layout():
    """amc_init_synthetic on every rank of a 2-rank run == on a single context (then 6 steps, bit for bit)."""
    from argon_monte_carlo_amd import ic as IC
    from argon_monte_carlo_amd.engine import Engine
    kind, n, steps, seed = "pore", 120001, 6, 5
    p, c, _ = _case(kind, n)
    eng = Engine(p)
    eng.init_synthetic(IC.device_ic_config(p, c, seed, kind))
    ref_tot = eng.run(c["dt"], steps)
    ref = eng.download()
    eng.close()
    full, tot, counts, npaths = _run_ranks(2, (kind, n, steps, seed))
    for k in KEYS:
        assert np.array_equal(full[k], ref[k]), (k, np.flatnonzero(full[k] != ref[k])[:5])
    assert tot["n_pp"] == ref_tot["n_pp"] > 0 and tot["n_wall"] == ref_tot["n_wall"]


def test_velocity_change_list_overflow_is_reported(monkeypatch):
    """More velocity changes in one step than the exchange block has room for (forced: room for two) must surface as
    AMC_ERR_CAPACITY, not as silently stale velocities on the other ranks."""
    from argon_monte_carlo_amd._lib import ArgonMCError
    from argon_monte_carlo_amd.engine import ShardEngine
    monkeypatch.setenv("AMC_MG_VELOCITY_LIST", "2")
    p, c, init = _case("cube", 50000)
    e = ShardEngine(p, 0, 50000)
    e.upload(*init)
    e.exchange_buffers(1)
    with pytest.raises(ArgonMCError, match="velocity changes"):
        for _ in range(3):                      # wall hits of step 1 and collisions of the sweeps change > 2 velocities
            e.mg_local(c["dt"])
            e.mg_pack(1)
            e.mg_sweep(1, 0)
            e.mg_finish(True)
    e.close()


# ---- energised walls (Temperature_Pore_MC.py) over two shards ---------------------------------------------------------------------
def _temp_case(n):
    from argon_monte_carlo_amd import ic as IC, params as PR
    p, c = PR.pore_params(n=n, energised=True)
    init = IC.pore_ic(p, c, seed=23)
    p.detect_mode = 1
    p.reserved0 |= 1
    return p, c, init


def _temp_objects(c):
    import random
    from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
    return DirectionSampler(np.random.RandomState(5), random.Random(5)), SurfaceEnergies(c)


def _temp_worker(rank, world, port, n, steps, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            from argon_monte_carlo_amd.dist import ShardedTemperatureSimulation
            p, c, init = _temp_case(n)
            sim = ShardedTemperatureSimulation(p, rank, world, backend="gloo")
            sim.upload(*init)
            sampler, energies = _temp_objects(c)
            rows = []
            for s in range(steps):
                st, mom, cold, hot, hm, hc, hh = sim.temp_timestep(c["dt"], sampler, energies)
                rows.append((float(mom), float(cold), float(hot), hm, hc, hh, st["n_pp"], st["n_wall"], st["n_oob_walls"], st["n_oob_pp"]))
            full = sim.download()
            if rank == 0:
                q.put(("ok", (full, rows)))
        finally:
            dist.destroy_process_group()
    except BaseException as e:
        import traceback
        q.put(("error", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))
        raise


@pytest.mark.parametrize("n,steps", [(200000, 5), (4000000, 2)])      # the second: BASELINE configs[4] at its full size
def test_energised_walls_two_ranks_one_gpu_equal_single_engine(n, steps):
    from argon_monte_carlo_amd.engine import EnergisedEngine
    p, c, init = _temp_case(n)
    eng = EnergisedEngine(p)
    eng.upload(*init)
    sampler, energies = _temp_objects(c)
    ref_rows = []
    for s in range(steps):
        st, mom, cold, hot, hm, hc, hh = eng.temp_timestep(c["dt"], sampler, energies)
        ref_rows.append((float(mom), float(cold), float(hot), hm, hc, hh, st["n_pp"], st["n_wall"], st["n_oob_walls"], st["n_oob_pp"]))
    ref = eng.download()
    eng.close()
    assert sum(r[7] for r in ref_rows) > 0 and sum(r[6] for r in ref_rows) > 0      # wall hits and p-p collisions happened
    full, rows = _run_ranks(2, (n, steps), timeout=600.0, target=_temp_worker)
    assert rows == ref_rows, (rows, ref_rows)
    for k in KEYS:
        assert np.array_equal(full[k], ref[k]), (k, np.flatnonzero(full[k] != ref[k])[:5])
