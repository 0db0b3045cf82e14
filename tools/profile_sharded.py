"""Host-side phase timing of the sharded step (world = 1 rehearsal over RCCL): where the per-step overhead goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ["AMC_DIST_NOSHORTCUT"] = "1"
import numpy as np
import torch
import torch.distributed as dist

from bench import make_workload
from argon_monte_carlo_amd.dist import ShardedSimulation

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p, c, init = make_workload(sys.argv[1] if len(sys.argv) > 1 else "cube_1e5")
sim = ShardedSimulation(p, 0, 1, backend="nccl")
sim.upload(*init)
sim.run(c["dt"], 20)
torch.cuda.synchronize()
T = {}
SYNC = len(sys.argv) > 2


def timed(name, fn, *a, **k):
    t0 = time.perf_counter()
    r = fn(*a, **k)
    if SYNC:
        torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return r


e = sim.engine
steps = 200
t_all = time.perf_counter()
for s in range(steps):
    timed("mg_local", e.mg_local, c["dt"])
    send, recv = e.exchange_buffers(sim.world)
    timed("pack", e.mg_pack, sim.world)
    timed("allgather", sim.comm.allgather_packed, send, recv)
    timed("sweep", e.mg_sweep, sim.world, sim.rank)
    timed("finish", e.mg_finish, False)
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print("total %.1f us/step" % (tot / steps * 1e6))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-14s %8.1f us/step" % (k, v / steps * 1e6))
dist.destroy_process_group()
