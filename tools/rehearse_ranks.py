"""Measurement tool (GPU box): what ONE rank of a G-rank job does per step, measured on one GPU.

    python tools/rehearse_ranks.py cube 800000 8 [steps] [--replicated]     > gpurun_out/rehearsal_cube_8x1e5.json

No multi-GPU node is available to the builder, so the per-rank cost of the sharded step cannot be read off a real run.
This tool runs all G ranks of the job in ONE process on ONE GPU — G ShardEngine contexts, each owning its index range of
the same system — and moves the blocks of the two per-step all-gathers between them with device-to-device copies (what the
collectives deliver, without their time).  Every rank's kernels are the real ones on the real data, so rank 0's per-kernel
times are what a rank of the G-rank job spends computing per step; the collectives' time comes on top (SURVEY 5's link
budget).  The final state is compared with a single context stepping the same system (bit for bit).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from argon_monte_carlo_amd import ic as IC
from argon_monte_carlo_amd import params as PR
from argon_monte_carlo_amd.dist import shard_range
from argon_monte_carlo_amd.engine import Engine, ShardEngine


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    replicated = "--replicated" in sys.argv
    kind = args[0] if args else "cube"
    n = int(args[1]) if len(args) > 1 else 800_000
    world = int(args[2]) if len(args) > 2 else 8
    steps = int(args[3]) if len(args) > 3 else 50
    if kind == "cube":
        p, c = PR.cube_params_for_n(n)
        init = IC.cube_ic(p, c, seed=127)
    else:
        p, c = PR.pore_params(n=n)
        init = IC.pore_ic(p, c, seed=17)
    p.reserved1 = 1
    p.max_paths = -1
    dt = c["dt"]
    stream = torch.cuda.current_stream().cuda_stream
    ranks = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        e = ShardEngine(p, lo, hi)
        e.set_stream(stream)
        e.upload(*init)
        ranks.append(e)
    xb = [e.exchange_buffers(world) for e in ranks]
    cb = [e.candidate_buffers(world) for e in ranks] if not replicated else None

    def gather(bufs):
        blk = bufs[0][0].numel()
        for r, (_, recv) in enumerate(bufs):
            for q, (send, _) in enumerate(bufs):
                recv[q * blk:(q + 1) * blk].copy_(send)

    def step(want=False):
        for e in ranks:
            e.mg_local(dt)
            e.mg_pack(world)
        gather(xb)
        if replicated:
            for r, e in enumerate(ranks):
                e.mg_sweep(world, r)
        else:
            for r, e in enumerate(ranks):
                e.mg_detect(world, r)
            gather(cb)
            for e in ranks:
                e.mg_resolve(world)
        return [e.mg_finish(want) for e in ranks]

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ranks[0].profile(True)
    t0 = time.perf_counter()
    for s in range(steps):
        st = step(want=(s == steps - 1))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kt = ranks[0].kernel_times()
    ranks[0].profile(False)
    # parity: the assembled state equals a single context's
    single = Engine(p)
    single.set_stream(stream)
    single.upload(*init)
    single.run(dt, 5 + steps)
    ref = single.download()
    equal = True
    for r, e in enumerate(ranks):
        lo, hi = shard_range(n, r, world)
        got = e.download()
        for k in ("x", "y", "z", "vx", "vy", "vz", "d", "dx", "dy", "dz", "flag"):
            equal = equal and bool(np.array_equal(got[k][lo:hi], ref[k][lo:hi]))
    per_step = {k: v[0] * 1e3 / steps for k, v in kt.items() if v[1]}
    out = {"what": "one rank's kernels per step in a %d-rank job, all ranks run in one process on one GPU (collectives replaced by device copies)" % world,
           "geometry": kind, "n_total": n, "world": world, "n_per_rank": n // world, "steps": steps,
           "detection": "replicated on every rank (round 2)" if replicated else "sharded by index + candidate all-gather",
           "rank0_kernel_us_per_step": per_step, "rank0_kernel_us_per_step_sum": sum(per_step.values()),
           "pp_collisions_per_step": (sum(s["n_pp"] for s in st if s) / (steps + 5)) if st[0] else None,
           "all_ranks_equal_single_context_bit_for_bit": equal,
           "bytes_gathered_per_step": {"positions_and_velocity_changes": int(xb[0][1].numel() * 8),
                                       "candidate_pairs": int(cb[0][1].numel() * 4) if cb else 0}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
