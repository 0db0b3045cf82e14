#!/usr/bin/env python3
"""bench.py — particle-steps/s of the hot path (drift -> walls -> bounds -> p-p sweep) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cube_1e5|pore_5e5|pore_1e6|cube_1e6]

A "step" is one full timestep(dt) over all particles (BASELINE.json metric: particle-steps/sec + achieved HBM GB/s).
At N=1 the default workload is BASELINE configs[1]: Open_Air_Cube_MC geometry, N = 100,000 synthetic uniform-cube
argon.  State is resident in HBM before the timed region starts.  For N>1 one rank runs per GPU
over RCCL: either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`:
RANK / WORLD_SIZE are in the environment) or `python bench.py --gpus N` starts them itself — as child processes of a
parent that has not touched the GPU — and relays rank 0's JSON line.  Particles are sharded by index range with a
per-step all-gather of positions (argon_monte_carlo_amd/dist.py) and per-GPU work is fixed (weak scaling).

Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (dominant kernel, HIP-event timed on the
launch stream) and `cpu_baseline` (the C oracle = single-thread port of the reference algorithm, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
BYTES_PER_PARTICLE_STEP = 137    # 81 B read + 56 B written by a complete timestep(dt)
# algorithmic bytes per particle for each kernel class (DESIGN.md "kernels")
ALGO_BYTES = {"drift_walls": 137, "detect": 24, "bin_count": 24, "bounds": 24}
KERNEL_OF_CLASS = {"drift_walls": "k_stream", "bin_count": "k_bin_lists / k_kin_pack / k_kin_unpack (list build)",
                   "detect": "k_detect_lists", "resolve": "k_resolve<GEOM,0> (ordered workgroup)",
                   "bounds": "k_stream (bounds-only pass)", "clusters_wide": "k_clusters_wide",
                   "commit": "k_commit", "fixup": "k_fixup (overlapped run: sweep results -> next state, sweep commit)",
                   "allgather": "all-gather (RCCL)"}
SWEEP_CLASSES = ("bin_count", "detect", "clusters_wide", "resolve", "commit", "fixup")     # the p-p pair sweep (24 B per particle)

WORKLOADS = {
    "cube_1e5": ("cube", 100_000),
    "cube_1e6": ("cube", 1_000_000),
    "pore_5e5": ("pore", 500_000),
    "pore_1e6": ("pore", 1_000_000),
    # BASELINE configs[3]: Temperature_Pore_MC energised walls; every step hands the wall hits to the host, which draws
    # the re-emission directions from the two Mersenne Twisters in particle order and evaluates mpmath.quad per gap hit
    "temp_1e6": ("temp", 1_000_000),
    # the LDS-tiled ALL-PAIRS detector (k_detect_allpairs: what the reference's O(n^2) pairwise loop, Pore:168-174, maps to
    # directly) in front of the same resolve: reported against the fp64 vector peak, 9 flop per unordered pair
    "cube_allpairs_4096": ("cube", 4096),
    "cube_allpairs_1e5": ("cube", 100_000),
}
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (MI355X_MICROARCH.md / SURVEY 8d)


def make_workload(name, n_override=None, device=0):
    from argon_monte_carlo_amd import ic as IC
    from argon_monte_carlo_amd import params as PR
    kind, n = WORKLOADS[name]
    if n_override:
        n = n_override
    if kind == "cube":
        p, c = PR.cube_params_for_n(n, device=device)
        init = IC.cube_ic(p, c, seed=127)
        if "allpairs" in name:
            p.detect_mode = 2
    else:
        p, c = PR.pore_params(n=n, device=device, energised=(kind == "temp"))
        init = IC.pore_ic(p, c, seed=17)
    # count-and-continue on a degenerate wall/contact solve (Temp:340-342 semantics) instead of aborting like Pore:336-338
    p.reserved1 = 1
    p.max_paths = -1                # completed paths go into the device histograms only (no per-path records to drain)
    if os.environ.get("AMC_BENCH_NOHIST"):
        p.hist_bins = 0             # experiment: no histogram atomics at all
    return p, c, init


def host_cores():
    """(cores, source): host cores this process may really use — the affinity mask, cut by a cgroup CPU quota if there is
    one ("affinity" / "cgroup").  A GPU box shows all of its (hundreds of) cores to a job that owns a share of them:
    without a visible limit the count is capped at 16, the share that goes with one GPU on the measurement boxes, and the
    source says "assumed" (AMC_BENCH_CORES overrides: "env")."""
    if os.environ.get("AMC_BENCH_CORES"):
        return max(1, int(os.environ["AMC_BENCH_CORES"])), "env"
    src = "affinity"
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max" and int(int(q) / int(per)) < n:
            n, src = max(1, int(int(q) / int(per))), "cgroup"
    except (OSError, ValueError):
        pass
    return (n, src) if n <= 64 else (16, "assumed")


def cpu_baseline(workload, budget_s=12.0):
    """The oracle (oracle/amc_oracle.c, `mul` variant) on the host, on a bounded number of steps of the SAME workload.
    Reported next to the GPU number; it is a baseline, not the target.  Three figures:
      * `cpu_baseline`           the faithful port on the cores the reference's algorithm can use for this geometry: ONE for
                                 the cube (Open_Air_Cube_MC.py's cell loop is serial, Cube:231-336), ALL for the pore (OpenMP
                                 threads over the disjoint cells of a colour group = the Pool.starmap structure, Pore:545-549);
      * `cpu_baseline_1core`     the same port on one core (always);
      * `cpu_baseline_all_cores` all host cores (cube: the Pore script's colouring applied to the cube's cells — a
                                 different processing order than the reference's serial loop, timing only)."""
    from oracle import oracle as O
    p, c, init = make_workload(workload)
    kind = WORKLOADS[workload][0]
    ncores, cores_src = host_cores()

    def timed(one, budget):
        one()                                       # warm-up (page faults)
        t0 = time.perf_counter()
        steps = 0
        while True:
            one()
            steps += 1
            el = time.perf_counter() - t0
            if el > budget or steps >= 2000:
                return steps, el

    orc = O.Oracle(p, mode="mul")
    orc.upload(*init)
    if kind == "temp":
        import random
        from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
        sampler, energies = DirectionSampler(np.random.RandomState(17), random.Random(17)), SurfaceEnergies(c)
        one = lambda: orc.temp_timestep(c["dt"], sampler, energies)     # noqa: E731
    else:
        one = lambda: orc.timestep(c["dt"])                             # noqa: E731
    steps, el = timed(one, budget_s)
    serial = {"value": p.n * steps / el, "unit": "particle-steps/s", "cores": 1, "cores_source": "single thread", "kind": "port",
              "sample": f"{steps} steps of {workload} (N={p.n}) in {el:.1f} s, oracle/amc_oracle.c single thread"}
    out = {"cpu_baseline_1core": serial}
    if kind == "temp":
        out["cpu_baseline"] = serial                # (the energised step's host part is sequential by construction)
        return out
    par = O.Oracle(p, mode="mul")
    par.upload(*init)
    steps, el = timed(lambda: par.timestep_par(c["dt"], threads=ncores), budget_s)
    allc = {"value": p.n * steps / el, "unit": "particle-steps/s", "cores": ncores, "cores_source": cores_src, "kind": "port",
            "sample": f"{steps} steps of {workload} (N={p.n}) in {el:.1f} s, oracle sweep with OpenMP threads over the "
                      "disjoint cells of a colour group" + (" (the Pore script's colouring on the cube's cells: NOT the "
                      "reference's serial cube order)" if kind == "cube" else " (= the reference's Pool.starmap structure)")}
    out["cpu_baseline_all_cores"] = allc
    out["cpu_baseline"] = serial if kind == "cube" else allc
    return out


def cpu_baseline_python_mp(n=100_000, steps=3):
    """The reference's own parallel STRUCTURE in NumPy + multiprocessing (oracle/pymp_structure.py: 8 colour groups, one
    boolean mask per cell over all N particles, one Pool task per cell, Pool(cpu_count() + 1), a fresh pool per group —
    Pore:520-549), validated against the reference's step dump, timed on this host for `steps` steps of the specular pore
    at N = n.  The like-for-like "Python/multiprocessing" number of the north star."""
    from argon_monte_carlo_amd import ic as IC
    from argon_monte_carlo_amd import params as PR
    from oracle import pymp_structure as PM
    p, c = PR.pore_params(n=n)
    init = IC.pore_ic(p, c, seed=17)
    ncores, cores_src = host_cores()
    s = PM.PyMpStepper(p, workers=ncores + 1)           # the reference's Pool(cpu_count() + 1) on the cores this job has
    s.upload(*init)
    t0 = time.perf_counter()
    npp = 0
    for _ in range(steps):
        npp += s.timestep(c["dt"])["n_pp"]
    el = time.perf_counter() - t0
    return {"value": n * steps / el, "unit": "particle-steps/s", "cores": ncores, "cores_source": cores_src, "kind": "port",
            "workers": ncores + 1,
            "sample": f"{steps} steps (no warm-up) of the specular pore at N={n} in {el:.1f} s, {npp} p-p collisions; "
                      "oracle/pymp_structure.py (NumPy-scalar pair loop, full per-cell boolean masks, fresh Pool per colour group)"}


def class_report(kt, steps, n_total, n_local, value_per_gpu, profiled_ms_per_step):
    """Kernel-class times of the profiled repeat of the timed steps.  `per_kernel_avg_us` is microseconds PER STEP (total
    time of the class / steps: a class launched three times in 2,000 steps weighs next to nothing), the per-launch averages
    and launch counts are beside it; the durations come from events carried by the dispatches themselves (kernel begin ->
    kernel end, what rocprofv3 --kernel-trace reports), so the classes add up to no more than the step."""
    per_step = {k: v[0] * 1e3 / steps for k, v in kt.items() if v[1]}
    out = {"per_kernel_avg_us": per_step,
           "per_kernel_avg_launch_us": {k: v[0] * 1e3 / v[1] for k, v in kt.items() if v[1]},
           "per_kernel_launches_per_step": {k: v[1] / steps for k, v in kt.items() if v[1]},
           "per_kernel_timing": "dispatch-attached HIP events (hipExtLaunchKernelGGL start/stop) on the launch stream, profiled repeat of the timed steps",
           "profiled_pass_ms_per_step": profiled_ms_per_step,
           "whole_step_frac_of_hbm_peak": BYTES_PER_PARTICLE_STEP * value_per_gpu / 1e9 / HBM_PEAK_GBS}
    # the whole pair sweep (list build outside the streaming pass, detect, wide clusters, ordered workgroup, commit)
    # against its 24 B per particle, and the streaming pass against its 137 B: the two numbers the north star names
    sweep_us = sum(v for kk, v in per_step.items() if kk in SWEEP_CLASSES)
    if sweep_us > 0:
        g = 24.0 * n_total / (sweep_us * 1e-6) / 1e9
        out["pair_sweep"] = {"avg_us": sweep_us, "achieved_GBps": g, "frac": g / HBM_PEAK_GBS, "bytes_per_particle": 24}
    if per_step.get("drift_walls"):
        g = 137.0 * n_local / (per_step["drift_walls"] * 1e-6) / 1e9
        out["streaming_pass"] = {"avg_us": per_step["drift_walls"], "achieved_GBps": g, "frac": g / HBM_PEAK_GBS, "bytes_per_particle": 137}
    return out


def extra_workload(name, steps, warmup, device, stream_ptr, sync):
    """One more single-GPU workload timed in the same invocation (GPU legs only): the driver's record then covers both
    sizes BASELINE.json's metric names.  Same protocol as the headline: warm-up, K timed steps between synchronisations,
    then the K steps again with the per-kernel events."""
    from argon_monte_carlo_amd.engine import Engine
    p, c, init = make_workload(name, device=device)
    eng = Engine(p)
    eng.set_stream(stream_ptr)
    eng.upload(*init)
    eng.run(c["dt"], warmup)
    sync()
    t0 = time.perf_counter()
    stats = eng.run(c["dt"], steps)
    sync()
    el = time.perf_counter() - t0
    eng.profile(True)
    t0 = time.perf_counter()
    eng.run(c["dt"], steps)
    sync()
    el_prof = time.perf_counter() - t0
    kt = eng.kernel_times()
    eng.profile(False)
    n = int(p.n)
    value = n * steps / el
    rep = class_report(kt, steps, n, n, value, el_prof / steps * 1e3)
    out = {"workload": name, "geometry": WORKLOADS[name][0], "n_particles": n, "steps": steps, "warmup": warmup,
           "ms_per_step": el / steps * 1e3, "value": value, "unit": "particle-steps/s",
           "pp_collisions_per_step": stats["n_pp"] / steps if stats else None,
           "whole_step_frac": rep["whole_step_frac_of_hbm_peak"],
           "pair_sweep_frac": rep.get("pair_sweep", {}).get("frac"),
           "streaming_pass_frac": rep.get("streaming_pass", {}).get("frac"),
           "per_kernel_avg_us": rep["per_kernel_avg_us"], "per_kernel_avg_launch_us": rep["per_kernel_avg_launch_us"]}
    eng.close()
    return out


def committed_traffic(workload, kclass, tag):
    """HBM bytes per launch of the dominant kernel from the committed counter passes (profiles/<round>_pmc_traffic_*.json,
    produced by tools/collect_profiles.sh + tools/summarise_profiles.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs).  bench.py itself cannot read PMC counters; None when no pass exists for this workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{tag}_pmc_traffic_{workload}.json")
    if not os.path.exists(path):
        return None, None
    needle = {"resolve": "k_resolve<", "detect": "k_detect_lists", "drift_walls": "k_stream", "clusters_wide": "k_clusters_wide",
              "commit": "k_commit"}.get(kclass)
    kern = json.load(open(path)).get("kernels", {})
    for name, v in kern.items():
        if needle and needle in name and (kclass != "resolve" or name.rstrip().endswith("0>")) and "bounds-only" not in name:
            if "HBM_bytes_corrected" in v:
                return float(v["HBM_bytes_corrected"]), f"profiles/{os.path.basename(path)}: {name} ({v.get('correction', '')})"
            kib = v.get("FETCH_SIZE_KiB_avg", 0.0) + v.get("WRITE_SIZE_KiB_avg", 0.0)
            return kib * 1024.0, f"profiles/{os.path.basename(path)}: {name} (FETCH_SIZE + WRITE_SIZE, uncorrected)"
    return None, None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run as a CHILD
    process — this parent has made no HIP / torch.cuda call, and it never replaces itself with another program — and pass
    rank 0's JSON line through.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round-tag", default="r03", help="prefix of the committed profile files to take `traffic` from")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cube_1e5", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override the particle count per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-workloads", default="auto",
                    help="comma-separated single-GPU workloads timed after the headline one (GPU legs only) and attached as "
                         "`extra_workloads`; auto = cube_1e6,pore_1e6 behind the default headline on one GPU; none = skip")
    ap.add_argument("--no-python-mp-baseline", action="store_true", help="skip the NumPy + multiprocessing leg (~30 s)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: all ranks share GPU 0")
    ap.add_argument("--sorted-ic", action="store_true", help="experiment: upload the particles in spatial (cell) order")
    ap.add_argument("--device-rng", action="store_true",
                    help="temp workloads: opt-in NON-PARITY mode, re-emission directions / gap energies drawn on the GPU (Philox)")
    ap.add_argument("--force-sharded", action="store_true", help="rehearsal: run the sharded driver (and its collectives) even with one rank")
    ap.add_argument("--strong", action="store_true", help="N>1: keep the TOTAL particle count at the workload's size (strong scaling) instead of the per-GPU count")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    # stdout carries exactly one JSON line: native libraries that print there (RCCL's version banner at communicator
    # creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    energies_early = None
    if WORKLOADS[args.workload][0] == "temp":
        # the gap-energy worker processes (energised.py) are forked before this process initialises the GPU
        from argon_monte_carlo_amd import params as PR_
        from argon_monte_carlo_amd.energised import SurfaceEnergies
        energies_early = SurfaceEnergies(PR_.pore_params(n=args.n or WORKLOADS[args.workload][1], energised=True)[1], start_workers=True)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.force_sharded:
            os.environ["AMC_DIST_NOSHORTCUT"] = "1"
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from argon_monte_carlo_amd.engine import Engine

    kind, n_per_gpu = WORKLOADS[args.workload]
    if args.n:
        n_per_gpu = args.n
    stream_ptr = torch.cuda.current_stream().cuda_stream
    driver = None                                       # the multi-GPU driver object, when one is in use

    if kind == "temp":
        import random
        from argon_monte_carlo_amd.energised import DirectionSampler, SurfaceEnergies
        from argon_monte_carlo_amd.engine import EnergisedEngine
        sharded = world > 1 or args.force_sharded
        n_total = n_per_gpu if args.strong else n_per_gpu * world       # weak scaling (default): per-GPU particles fixed
        p, c, init = make_workload(args.workload, n_total, device=local_rank)
        p.reserved0 |= 1
        energies = energies_early
        p.E_cold, p.E_hot = energies.cold, energies.hot          # Temp:83-84
        if sharded:
            from argon_monte_carlo_amd.dist import ShardedTemperatureSimulation
            eng = ShardedTemperatureSimulation(p, rank, world, backend=args.backend, stream_ptr=stream_ptr)
            driver = eng
        else:
            eng = EnergisedEngine(p)
            eng.set_stream(stream_ptr)
        eng.upload(*init)
        sampler = DirectionSampler(np.random.RandomState(17), random.Random(17))     # same streams on every rank

        p_dev = None
        if args.device_rng:
            if sharded:
                raise SystemExit("--device-rng is a single-GPU option")
            from argon_monte_carlo_amd.energised import device_rng_config
            p_dev = device_rng_config(c, 17)

        def step(k):
            tot = None
            for _ in range(k):
                st = (eng.temp_timestep_device(c["dt"], p_dev) if p_dev is not None else
                      eng.temp_timestep(c["dt"], sampler, energies))[0]
                tot = st if tot is None else {kk: tot[kk] + st[kk] for kk in st}
            return tot
        parallelism = ("single GPU" if not sharded else f"index-range shards x{world}, hits concatenated in index order, "
                       "one all-gather per step: positions + changed velocities (RCCL), list build + p-p sweep replicated on every rank") + \
            (" + device-side Philox sampling (opt-in, NOT the reference's random streams)" if args.device_rng else
             " + host RNG/mpmath hand-over per energised case")
        engines = [eng.engine if sharded else eng]
    elif world == 1 and not args.force_sharded:
        p, c, init = make_workload(args.workload, n_per_gpu, device=local_rank)
        if args.sorted_ic:                              # experiment: storage order = the detection grid's cell order
            x, y, z = (np.asarray(a) for a in init[:3])
            cr = p.collision_range
            if kind == "cube":
                vol, xlo, zlo = p.cube_x * p.cube_y * p.cube_z, 0.0, 0.0
            else:
                vol = 2 * np.pi * p.R_oa ** 2 * p.h_oa + np.pi * p.R_g ** 2 * (p.H - 2 * p.h_oa)
                xlo, zlo = -p.R_oa, 0.0
            h = max(0.63 * (vol / len(x)) ** (1.0 / 3.0), 2.01 * cr)
            cx = np.floor((x - (xlo - h)) / h).astype(np.int64)
            cy = np.floor((y - (xlo - h)) / h).astype(np.int64)
            cz = np.floor((z - (zlo - h)) / h).astype(np.int64)
            order = np.lexsort((cx, cy, cz))
            init = tuple(np.asarray(a)[order] for a in init)
        eng = Engine(p)
        eng.set_stream(stream_ptr)
        eng.upload(*init)
        step = lambda k: eng.run(c["dt"], k)          # noqa: E731
        n_total = int(p.n)
        parallelism = "single GPU"
        engines = [eng]
    else:
        from argon_monte_carlo_amd.dist import ShardedSimulation
        n_total = n_per_gpu if args.strong else n_per_gpu * world       # weak scaling (default): per-GPU particles fixed
        p, c, init = make_workload(args.workload, n_total, device=local_rank)
        sim = ShardedSimulation(p, rank, world, backend=args.backend, stream_ptr=stream_ptr)
        driver = sim
        sim.upload(*init)
        step = lambda k: sim.run(c["dt"], k)          # noqa: E731
        parallelism = (f"index-range shards x{world}: streaming pass (drift, walls, bounds) on the shard, one all-gather per step "
                       "(positions + changed velocities, RCCL), list build of the WHOLE system on every rank, " +
                       ("detection of the whole system REPLICATED on every rank, " if sim.replicated_detect else
                        "detection SHARDED by index (own particles against everybody) + a second all-gather of the candidate pairs, ") +
                       "ordered resolve REPLICATED on every rank (DESIGN.md 6)")
        engines = [sim.engine]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    step(args.warmup)
    sync()
    t0 = time.perf_counter()
    stats = step(args.steps)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # per-kernel durations: the same K steps again with every launch bracketed by hipEvents on the launch stream
    eng0 = engines[0]
    eng0.profile(True)
    if driver is not None:
        driver.profile_collective(True)
    t0 = time.perf_counter()
    step(args.steps)
    sync()
    el_prof = time.perf_counter() - t0
    kt = eng0.kernel_times()
    eng0.profile(False)
    if driver is not None:
        kt["allgather"] = driver.collective_times()       # the step's collective, same (total ms, count) form
        driver.profile_collective(False)

    if rank == 0:
        value = n_total * args.steps / el
        n_local = n_total // world if world > 1 else n_total
        dom = max(((k, v) for k, v in kt.items() if v[1] > 0 and k != "allgather"), key=lambda kv: kv[1][0], default=(None, (0, 0)))
        roof = None
        if dom[0] is not None and "allpairs" in args.workload and kt.get("detect", (0, 0))[1] > 0:
            dom = ("detect", kt["detect"])      # these workloads exist to report the all-pairs detector, dominant or not
        if dom[0] is not None:
            k, (ms, cnt) = dom
            avg_s = ms / cnt * 1e-3
            allpairs = "allpairs" in args.workload
            if k == "detect" and allpairs:
                # the all-pairs detector: fp64 vector work, 9 flop per unordered pair
                flops = 9.0 * n_total * (n_total - 1) / 2.0
                ach = flops / avg_s / 1e12
                tiled = n_total >= 16 * 1024
                roof = {"kernel": "k_detect_allpairs_tiled" if tiled else "k_detect_allpairs", "kernel_class": k, "bound": "fp64_valu",
                        "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
                        "traffic": None, "avg_launch_us": avg_s * 1e6, "algorithmic_flops_per_launch": flops}
                if tiled:
                    # the tiled kernel tests d^2 in the expanded form |ri|^2 + |rj|^2 - 2 ri.rj: three fused multiply-adds
                    # and a comparison per pair — 7 floating-point operations issued in 4 instructions for the 9 the
                    # direct form counts.  Both figures, so that nobody has to guess which one a fraction means.
                    ex = 7.0 * n_total * (n_total - 1) / 2.0 / avg_s / 1e12
                    roof["executed"] = {"flop_per_pair": 7, "instructions_per_pair": 4, "TFLOP/s": ex, "frac_of_peak": ex / FP64_VECTOR_PEAK_TFLOPS,
                                        "vector_instruction_issue_frac": 4.0 * n_total * (n_total - 1) / 2.0 / avg_s / (FP64_VECTOR_PEAK_TFLOPS * 1e12 / 2.0)}
            else:
                # algorithmic bytes of the dominant kernel: a complete timestep for the streaming pass, the positions
                # (24 B per particle) for every kernel of the pair sweep — whose wide-cluster / ordered-workgroup
                # kernels touch a few hundred KB and are bound by LATENCY (chains of dependent round trips), so that this
                # fraction prices their time against the sweep's data, not their own traffic
                per_particle = ALGO_BYTES.get(k, 24)
                units = n_local if k in ("drift_walls", "bounds") else n_total
                ach = per_particle * units / avg_s / 1e9
                roof = {"kernel": KERNEL_OF_CLASS.get(k, k), "kernel_class": k, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": avg_s * 1e6,
                        "algorithmic_bytes_per_launch": per_particle * units}
                if k in ("clusters_wide", "resolve", "commit"):
                    roof["note"] = "latency-bound kernel (dependent scattered round trips on a few hundred candidates): priced at the pair sweep's 24 B per particle"
            roof.update(class_report(kt, args.steps, n_total, n_local, value / world, el_prof / args.steps * 1e3))
        if roof is not None and roof["bound"] == "hbm":
            roof["traffic"], roof["traffic_source"] = committed_traffic(args.workload, roof["kernel_class"], args.round_tag)
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "geometry": kind, "n_particles": n_total, "n_per_gpu": n_local,
                       "dt": c["dt"], "parallelism": parallelism,
                       "pp_collisions_per_step": stats["n_pp"] / args.steps if stats else None},
            "roofline": roof,
        }
        extras = args.extra_workloads
        if extras == "auto":
            extras = "cube_1e6,pore_1e6" if (world == 1 and not args.force_sharded and args.workload == "cube_1e5" and not args.n) else "none"
        if extras != "none" and world == 1:
            out["extra_workloads"] = [extra_workload(w, args.steps, args.warmup, local_rank, stream_ptr, sync) for w in extras.split(",") if w]
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline(args.workload))
            if not args.no_python_mp_baseline:
                out["cpu_baseline_python_mp"] = cpu_baseline_python_mp()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or args.force_sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
