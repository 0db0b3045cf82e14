"""Test infrastructure (run by hand / through gpurun, not collected by pytest).  Long parity run of the multi-GPU driver on ONE GPU: `world` ranks (gloo, collectives staged through the host — RCCL
refuses several ranks on one device) against the single-context engine on the same workload; state, counters and
histograms must be identical.  Writes a JSON summary (committed under profiles/ as evidence).

    python tests/soak_sharded.py pore_1e6 200 2
"""
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

KEYS = ["x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz", "flag"]


def worker(rank, world, port, workload, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bench import make_workload
        from argon_monte_carlo_amd.dist import ShardedSimulation
        p, c, init = make_workload(workload)
        sim = ShardedSimulation(p, rank, world, backend="gloo")
        sim.upload(*init)
        t0 = time.perf_counter()
        tot = sim.run(c["dt"], steps)
        el = time.perf_counter() - t0
        full = sim.download()
        counts, npaths = sim.histograms()
        if rank == 0:
            q.put((full, tot, counts, npaths, el))
    finally:
        dist.destroy_process_group()


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "pore_1e6"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    from bench import make_workload
    from argon_monte_carlo_amd.engine import Engine
    p, c, init = make_workload(workload)
    eng = Engine(p)
    eng.upload(*init)
    ref_tot = eng.run(c["dt"], steps)
    ref = eng.download()
    ref_counts, ref_npaths = eng.histograms()
    eng.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, workload, steps, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    import queue
    try:
        t0 = time.time()
        while True:                                 # a rank that dies ends the run at once (its peers hold GPU contexts)
            try:
                full, tot, counts, npaths, el = q.get(timeout=1.0)
                break
            except queue.Empty:
                if any(pr.exitcode not in (None, 0) for pr in procs) or time.time() - t0 > 900:
                    raise SystemExit("a rank died or the run timed out")
        for pr in procs:
            pr.join(timeout=120)
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
        for pr in procs:
            pr.join(timeout=10)
            if pr.is_alive():
                pr.kill()
    first_bad = next((k for k in KEYS if not np.array_equal(full[k], ref[k])), None)
    ckeys = ("n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_fp_errors")
    out = {"workload": workload, "n": int(p.n), "steps": steps, "ranks_on_one_gpu": world, "collectives": "gloo, staged through the host",
           "state_bit_identical_to_single_engine": first_bad is None, "first_difference": first_bad,
           "counters_equal": all(tot[k] == ref_tot[k] for k in ckeys), "counters": {k: tot[k] for k in ckeys},
           "histograms_equal": bool(npaths == ref_npaths and np.array_equal(counts, ref_counts)),
           "sharded_seconds_incl_host_staging": round(el, 3)}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/soak_sharded{world}_{workload}_{steps}.json", "w") as f:
        f.write(json.dumps(out) + "\n")
    print(json.dumps(out))
    sys.exit(0 if out["state_bit_identical_to_single_engine"] and out["counters_equal"] and out["histograms_equal"] else 1)


if __name__ == "__main__":
    main()
