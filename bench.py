#!/usr/bin/env python3
"""bench.py — drone-steps/s of the fused fly() rollout on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 3, the one the metric is quoted on): 65 536 drones per GPU x 2 000
timesteps, 15-32-32-4 controller, 3-coin chain, crash penalty on, float32.  A "step" is one whole
episode of the population = the fused-step launches of ld_fly_rollout_f32 (four tiles of 16 384 drones, 64
timesteps per launch with the drone state in registers in between: 128 launches) + the finalize launch (+ the RCCL all-gather of the
16-byte fitness records when N > 1).  Population shards are independent, so the headline line is weak
scaling: every rank owns 65 536 drones.  With N > 1 the same run ALSO times the strong-scaling case
BASELINE's target describes (ONE 65 536-drone population split over the N GPUs) and reports it in the
`strong_scaling` object of the same JSON line, together with `ranks_seen` and a checksum of the gathered
records, so that the line proves the collective saw N ranks.  `--strong` makes the strong case the
headline instead.

Genomes are synthetic: hover controllers (learning-drones_amd/controllers.py) with small per-drone
noise, so that nearly every drone is alive for all 2 000 steps and every counted drone-step does the
full work (frozen blocks skip their gene traffic; the JSON line reports the live fraction).

Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events on the launch stream,
`cpu_baseline` times the CPU oracle (the stand-in for the reference's Python loop) on the host
cores, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
WORKLOAD = dict(drones_per_gpu=65536, timesteps=2000, hidden=(32, 32), n_coins=3)


def make_genomes(torch, n, hidden, device, seed):
    from learning_drones_amd import controllers
    poles = (1.2, 1.6, 2.0, 2.5)
    bases = np.stack([np.concatenate([w.ravel() for w in controllers.hover_net(hidden, gains=controllers.hover_gains(pole_xy=p))])
                      for p in poles])
    g = torch.Generator(device=device)
    g.manual_seed(0x1D20 + seed)
    base = torch.as_tensor(bases, dtype=torch.float32, device=device)[torch.arange(n, device=device) % len(poles)]
    return base + 2e-4 * torch.randn(base.shape, generator=g, device=device, dtype=torch.float32)


def cpu_model() -> str:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hidden, n_coins, timesteps):
    """The oracle (CPU restatement of the reference loop, float64 with the reference's rounding
    points) on one host thread and on all of this GPU's host cores, on bounded samples of the same
    workload (~4 s + ~12 s of wall time)."""
    from learning_drones_amd import controllers
    from learning_drones_amd.params import EpisodeParams, MlpSpec, SimParams, coin_chain_reference
    from oracle import oracle as O
    O.build()
    cores = min(O.max_threads(), 16)      # the GPU box's CPU share for one GPU
    spec = MlpSpec.from_hidden(hidden)
    base = np.concatenate([w.ravel() for w in controllers.hover_net(hidden)])
    rng = np.random.default_rng(0)
    sim, coins = SimParams.reference(), coin_chain_reference(n_coins)
    ep = EpisodeParams.reference(n_coins=n_coins, coin_respawn=False)

    def timed(n, T, threads):
        genes = base[None, :] + rng.normal(0, 2e-4, (n, spec.param_count))
        t0 = time.perf_counter()
        O.fly(sim, ep, spec, genes, T, coins=coins, n_threads=threads)
        return n * T / (time.perf_counter() - t0), time.perf_counter() - t0

    rate1, _ = timed(16, 200, 1)                                        # probes, to size the samples
    n1 = int(min(4096, max(8, rate1 * 4.0 / timesteps)))
    one, dt1 = timed(n1, timesteps, 1)
    rate, _ = timed(64 * cores, 200, cores)
    n = int(min(65536, max(256, rate * 12.0 / timesteps)))
    value, dt = timed(n, timesteps, cores)
    return {"value": value, "unit": "drone-steps/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"{n} drones x {timesteps} steps, same controller/coins, float64 oracle, {cores} threads, {dt:.1f} s",
            "one_thread": {"value": one, "cores": 1, "sample": f"{n1} drones x {timesteps} steps, {dt1:.1f} s"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--drones", type=int, default=WORKLOAD["drones_per_gpu"],
                    help="drones per GPU (weak scaling); with --strong, and for the strong_scaling object, the TOTAL population")
    ap.add_argument("--timesteps", type=int, default=WORKLOAD["timesteps"])
    ap.add_argument("--hidden", type=str, default=",".join(map(str, WORKLOAD["hidden"])))
    ap.add_argument("--coins", type=int, default=WORKLOAD["n_coins"])
    ap.add_argument("--strong", action="store_true", help="make the strong-scaling case the headline: --drones is the TOTAL "
                    "population, split over the ranks (BASELINE's 8-GPU target is quoted for a 65 536-drone population in all)")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="timesteps per fused launch (0 = the library's choice)")
    ap.add_argument("--waves-per-block", type=int, default=0, help="wavefronts per 64-drone genome block (0 = the library's choice)")
    ap.add_argument("--tile-drones", type=int, default=0, help="drones per tile of the rollout (0 = the library's choice: tiles of "
                    "~120 MB of genes when the population's genes exceed the Infinity Cache; the --drones value = no tiling)")
    ap.add_argument("--specialise", dest="specialise", action="store_true", default=None, help="compile the unrolled kernels for "
                    "--hidden at first use if the library does not ship them (learning_drones_amd.specialise); the default is "
                    "Population's: automatic for large float32 populations when hipcc is there")
    ap.add_argument("--no-specialise", dest="specialise", action="store_false", help="stay on the run-time-spec kernels")
    ap.add_argument("--walls", type=int, default=0, help="number of wall boxes (BASELINE config 5 has walls: objects/walls.py is "
                    "empty in the reference, so the boxes are this build's: slabs beside the flight corridor, tested every step, never hit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline rollout and its roofline (for rocprofv3 "
                    "runs: keeps every fused launch in the trace a full, live-population launch of the headline geometry)")
    ap.add_argument("--exchange", default="allgather", choices=("allgather", "alltoall"), help="parent-genome exchange of the "
                    "generation step when N > 1: all-gather of the genome shards (the default here: the two most ordinary "
                    "collectives, so that an unproven path cannot cost the scaling record) or the all-to-all of exactly the "
                    "selected parents (1/N of the bytes; unverified on RCCL until a multi-GPU record exists)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' lets several ranks share one "
                    "GPU to rehearse the N > 1 control flow on a one-GPU box")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        limit = datetime.timedelta(minutes=5)      # a wedged collective ends the job in minutes, not at the default 10+
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=limit)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=limit)
        # bring the communicator up before anything is timed (also when --warmup 0)
        dist.all_reduce(torch.zeros(1, device=device))
        torch.cuda.synchronize()

    from learning_drones_amd.parallel import ShardedPopulation
    from learning_drones_amd.params import coin_chain_reference
    hidden = tuple(int(h) for h in args.hidden.split(","))
    T = args.timesteps
    if args.drones % world and (args.strong or world > 1):
        raise SystemExit("--drones must be divisible by the number of GPUs (strong-scaling split)")

    walls = None
    if args.walls:
        import numpy as np
        walls = np.array([[60.0 + 3 * k, -50.0, 0.0, 61.0 + 3 * k, 50.0, 80.0] for k in range(args.walls)])

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fly_kwargs():
        if args.steps_per_launch or args.waves_per_block or args.tile_drones:
            return dict(steps_per_launch=args.steps_per_launch or None, waves_per_block=args.waves_per_block,
                        tile_drones=args.tile_drones or None)
        return {}

    def measure(n_local, steps, warmup, force_tile=None):
        """`steps` timed episodes of a population of n_local drones per rank: barrier + synchronize on both
        sides, wall time = max over ranks; the fused launches' average duration from HIP events on the
        launch stream over the same region."""
        pop = ShardedPopulation(n_local * world, hidden, coins=coin_chain_reference(args.coins), precision="f32", walls=walls,
                                device=device, process_group=dist.group.WORLD if dist else None, specialise=args.specialise)
        assert pop.local.n == n_local
        pop.local.set_genomes(make_genomes(torch, n_local, hidden, device, seed=rank))
        geo = pop.local.rollout_geometry(T)
        kwargs = fly_kwargs()
        if force_tile:
            kwargs = dict(steps_per_launch=args.steps_per_launch or None, waves_per_block=args.waves_per_block, tile_drones=force_tile)
        if args.steps_per_launch or args.waves_per_block or args.tile_drones or force_tile:      # explicit geometry
            k = args.steps_per_launch or geo["steps_per_launch"]
            tile = min(force_tile or args.tile_drones or geo["tile_drones"], n_local)
            geo.update(steps_per_launch=k, tile_drones=tile, launches=-(-n_local // tile) * -(-T // k))
            # the wavefronts per block the library picks for launches of `tile` drones (or the explicit value)
            geo["waves_per_block"] = args.waves_per_block or pop.local.auto_waves_per_block(tile, k)
            geo["cached_drones"] = 0                       # an explicit geometry is flown as given: one kernel per launch

        def episode():
            pop.local.reset()
            return pop.fly(T, **kwargs)                # fused launches + finalize (+ all-gather of records)

        for _ in range(warmup):
            episode()
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()                                   # HIP events on the launch stream, over the timed region
        records = None
        for _ in range(steps):
            records = episode()
        ev1.record()
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        launch_ms = ev0.elapsed_time(ev1) / (steps * geo["launches"]) if steps else float("nan")
        # algorithmic bytes of the AVERAGE launch of the episode: every drone streams its genes T times and reads and
        # writes its state once per launch (the last launch of a tile is shorter when K does not divide T, the last
        # tile smaller when the tile does not divide n: averaging over the episode's launches accounts for both)
        launches_per_tile = -(-T // geo["steps_per_launch"])
        bytes_per_launch = (pop.local.bytes_per_launch(T, n_local) - pop.local.bytes_per_launch(0, n_local)
                            + launches_per_tile * pop.local.bytes_per_launch(0, n_local)) / geo["launches"]
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "kernel": ("fly_split_kernel<float, L=%d, %s>" % (geo["waves_per_block"], "-".join(map(str, hidden)))
                           if geo["steps_per_launch"] > 1 or geo["waves_per_block"] > 1 or pop.local.has_resident_engine()
                           else "fly_step_kernel<float, GenericNet> (run-time layer spec)"),
                "bytes_per_launch": bytes_per_launch, "launch_us": launch_ms * 1e3,
                "timesteps_per_launch": geo["steps_per_launch"], "wavefronts_per_block": geo["waves_per_block"],
                "launches_per_episode": geo["launches"], "drones_per_launch": geo["tile_drones"],
                "cached_drones_per_launch": geo.get("cached_drones", 0),
                "tiles": -(-n_local // geo["tile_drones"]),
                "launch_note": ("a launch here = two concurrent kernels: the first %d drones of the tile by fly_split_kernel<L=8, two "
                                "workgroups per block, CACHEABLE gene loads> on the library's side stream, the rest by the streaming "
                                "kernel named above; launch_us = HIP-event time of the episode / launches" % geo.get("cached_drones", 0))
                               if geo.get("cached_drones", 0) else None,
                "tiling_note": "a population whose genes exceed the 256 MiB Infinity Cache is flown tile by tile (a tile's whole "
                               "episode, then the next tile): the genes, re-streamed every timestep, then come from HBM once per "
                               "episode and from the cache afterwards, so `achieved` (algorithmic bytes / time) can exceed the HBM "
                               "peak; `traffic` is what the memory-side counters saw",
                "bytes_model": "per launch, averaged over the episode's launches: drones x (timesteps x 4P + 200): genes once per "
                               "timestep, the 25 state words read and written once per launch (they stay in registers in between); "
                               "4P + 200 per drone-step at one timestep per launch"}
        return pop, records, dt, roof

    n_total_strong = args.drones
    n_local = args.drones // world if args.strong else args.drones
    pop, records, dt, roof = measure(n_local, args.steps, args.warmup)

    # proof that the collective saw every rank: the ranks themselves, gathered, and a checksum of the
    # gathered records (identical on every rank by construction; rank 0 prints it)
    ranks_seen = [0]
    if dist:
        seen = torch.empty((world,), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(seen, torch.tensor([rank], dtype=torch.int64, device=device))
        ranks_seen = [int(r) for r in seen.tolist()]

    def checksum(rec):
        w = torch.tensor([1, 3, 5, 7], dtype=torch.int64, device=rec.device)
        idx = torch.arange(rec.shape[0], dtype=torch.int64, device=rec.device) % 1021 + 1
        return int(((rec.to(torch.int64) * w).sum(dim=1) * idx).sum().item() % (2 ** 61 - 1))

    # live fraction: flight_time / T over the whole population (records are gathered on every rank)
    live = float(records[:, 1].double().sum().item()) / (records.shape[0] * T)
    out_records = {"rows": int(records.shape[0]), "checksum": checksum(records)}

    # ---- the strong-scaling case in the same run (N > 1): ONE population of --drones in all
    strong = None
    if world > 1 and not args.strong and not args.no_extras:
        pop_s, rec_s, dt_s, roof_s = measure(n_total_strong // world, args.steps, 1)
        strong = {"value": n_total_strong * T * args.steps / dt_s, "unit": "drone-steps/s", "scaling": "strong",
                  "population": n_total_strong, "drones_per_gpu": n_total_strong // world, "ms_per_step": dt_s / args.steps * 1e3,
                  "roofline": roof_s, "records": {"rows": int(rec_s.shape[0]), "checksum": checksum(rec_s)}}
        del pop_s

    extras = {}
    if not args.no_extras:
        # one timestep per launch (the literal contract: one launch advances the population one timestep)
        pop.local.reset()
        for _ in range(20):
            pop.local.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400):
            pop.local.step()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 400 * 1e3
        extras["one_timestep_per_launch"] = {"launch_us": us, "achieved": pop.local.bytes_per_launch(1) / us / 1e3, "unit": "GB/s",
                                             "frac": pop.local.bytes_per_launch(1) / us / 1e3 / HBM_PEAK_GBS,
                                             "bytes_per_launch": pop.local.bytes_per_launch(1)}
        # Practical ceiling next to the spec peak: a device-to-device copy of 1 GiB on this GPU
        src_buf = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=device).normal_()
        dst_buf = torch.empty_like(src_buf)
        for _ in range(3):
            dst_buf.copy_(src_buf)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst_buf.copy_(src_buf)
        c1.record()
        torch.cuda.synchronize()
        roof["copy_ceiling_gbs"] = 10 * 2 * src_buf.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src_buf, dst_buf

    # The same population in whole-population launches: every timestep's genes from HBM (the HBM-bound figure next
    # to the cache-blocked headline), when the headline run was tiled
    if not args.no_extras and roof["tiles"] > 1:
        _, _, dt_w, roof_w = measure(n_local, args.steps, 1, force_tile=n_local)
        extras["whole_population_launches"] = {"value": n_local * world * T * args.steps / dt_w, "unit": "drone-steps/s",
                                               "ms_per_step": dt_w / args.steps * 1e3,
                                               "roofline": {k: roof_w[k] for k in ("achieved", "frac", "launch_us", "bytes_per_launch",
                                                                                   "timesteps_per_launch", "wavefronts_per_block",
                                                                                   "drones_per_launch", "launches_per_episode")}}

    # The shard one GPU owns when BASELINE's 65 536-drone population is spread over 8 (N = 1 only: with
    # N > 1 the strong_scaling object measures the real thing)
    if world == 1 and not args.no_extras and n_local == WORKLOAD["drones_per_gpu"]:
        _, _, dt_8, roof_8 = measure(n_local // 8, args.steps, 1)
        extras["shard_of_8"] = {"drones": n_local // 8, "value": n_local // 8 * T * args.steps / dt_8, "unit": "drone-steps/s",
                                "ms_per_step": dt_8 / args.steps * 1e3, "roofline": roof_8}

    # The reference's own kind of population (random genomes, evolve/brain.py:24): every drone tumbles
    # and crashes within a few hundred steps; frozen blocks skip their gene traffic, so an episode
    # costs far less than N*T full drone-steps.  Informational (live drone-steps only are real work).
    random_pop = {}
    engines = () if args.no_extras else (("step", "resident") if pop.local.has_resident_engine() else ("step",))
    if engines:
        pop.local.init_genomes_reference(seed=7 + rank)
    for engine in engines:
        pop.local.reset()
        fence()
        q0 = time.perf_counter()
        rec = pop.fly(T, engine=engine)
        fence()
        random_pop[engine + "_episode_ms"] = (time.perf_counter() - q0) * 1e3
    if engines:
        random_pop["live_drone_step_fraction"] = float(rec[:, 1].double().sum().item()) / (rec.shape[0] * T)
        pop.local.set_genomes(make_genomes(torch, n_local, hidden, device, seed=rank))

    # HBM bytes per launch from the PMC counters and the profiled launch duration: collected by separate
    # rocprofv3 passes of this same command and committed under profiles/ (they cannot be read live);
    # used only when the profiled launch configuration is the one being run.
    for f in sorted((REPO / "profiles").glob("r*_pmc.json"), reverse=True):
        prof = json.loads(f.read_text())
        cfg = prof.get("config")
        if not cfg or "hbm_traffic_bytes_per_launch" not in prof:
            continue
        if (cfg["drones"] == n_local and tuple(cfg["hidden"]) == hidden and cfg["coins"] == args.coins
                and cfg.get("timesteps_per_launch", 1) == roof["timesteps_per_launch"]
                and cfg.get("wavefronts_per_block", 1) == roof["wavefronts_per_block"]
                and cfg.get("drones_per_launch", cfg["drones"]) == roof["drones_per_launch"]
                and cfg.get("cached_drones_per_launch", 0) == roof["cached_drones_per_launch"]):
            roof["traffic"], roof["traffic_source"] = prof["hbm_traffic_bytes_per_launch"], f"profiles/{f.name}"
            if prof.get("launch_us_profiled"):
                roof["launch_us_profiled"] = prof["launch_us_profiled"]
                roof["frac_profiled"] = roof["bytes_per_launch"] / prof["launch_us_profiled"] / 1e3 / HBM_PEAK_GBS
            break
    roof.setdefault("traffic", None)

    # The genome-resident engine (one launch per episode, genes held in registers: beyond the contract
    # of streaming the genes every timestep, bit-identical results) on the same workload, for reference.
    resident = None
    if pop.local.has_resident_engine() and not args.no_extras:
        pop.local.reset()
        pop.fly(T, engine="resident")
        fence()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            pop.local.reset()
            pop.fly(T, engine="resident")
        fence()
        rdt = max_over_ranks(time.perf_counter() - r0)
        resident = {"value": n_local * world * T * args.steps / rdt, "unit": "drone-steps/s", "ms_per_step": rdt / args.steps * 1e3,
                    "speedup_over_step_engine": dt / rdt,
                    "note": "ld_fly_rollout_resident_f32: whole episode in one launch, genome in VGPRs (VALU-bound, "
                            "HBM traffic ~1/T of the step engine); not the kernel the roofline object describes"}

    # generations/s of the whole loop on the device: rollout + all-gather of records + argmax/elite +
    # parent exchange (N > 1) + select/breed kernels + state reset.  Mutated children of stable flyers
    # crash early and frozen blocks skip their gene traffic, so this is faster than 1 / episode time; the
    # live fraction of the measured generations is reported next to it.
    from learning_drones_amd.parallel import DeviceEvolution
    evo = DeviceEvolution(pop, seed=0x1D20, exchange=args.exchange)
    n_gen, gen_live, gen_dt, gen_error = 3, 0.0, float("nan"), ("skipped (--no-extras)" if args.no_extras else None)
    if not args.no_extras:
        ok = 1
        try:
            pop.local.reset()
            evo.step(T)
            fence()
            g0 = time.perf_counter()
            for _ in range(n_gen):
                _, _, rec = evo.step(T)
                gen_live += float(rec[:, 1].double().sum().item()) / (rec.shape[0] * T) / n_gen
            fence()
            gen_dt = max_over_ranks(time.perf_counter() - g0)
        except RuntimeError as err:      # reported, and fatal for the whole job below (no rank may carry on alone)
            gen_error, ok = str(err)[:200], 0
        if dist:                         # every rank learns whether ANY rank failed: nobody carries on alone
            flag = torch.tensor([ok], dtype=torch.int64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                gen_error = gen_error or "the generation step failed on another rank"
        if gen_error:                    # the headline above stands; the line reports generations_per_s null + the error
            print(f"[bench] rank {rank}: generation step failed: {gen_error}", file=sys.stderr, flush=True)
    # the generation step's own kernels (select + breed, no communication): local shard bred from itself
    from learning_drones_amd.params import GaParams
    breed_ms = breed_gbs = None
    if not args.no_extras and not gen_error:
        fit_local = pop.local.records[:, 0].contiguous()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pop.local.breed(fit_local, GaParams.reference(seed=3))
        b0.record()
        for g in range(3):
            pop.local.breed(fit_local, GaParams.reference(seed=3, generation=g + 1))
        b1.record()
        torch.cuda.synchronize()
        breed_ms = b0.elapsed_time(b1) / 3
        breed_gbs = 2 * 4 * pop.local.P * n_local / (breed_ms * 1e-3) / 1e9  # parent words read + child words written

    known = {(65536, 2000, (32, 32), 3): "BASELINE config 3", (1024, 2000, (16, 16), 1): "BASELINE config 2",
             (32768, 4000, (32, 32), 3): "BASELINE config 4 (one GPU's shard)",
             (8192, 2000, (32, 32), 3): "BASELINE target population (65 536 drones) spread over 8 GPUs: one GPU's shard",
             (131072, 4000, (64, 64, 64), 5): "BASELINE config 5 (one GPU's shard)"}
    config_name = known.get((n_local, T, hidden, args.coins), "custom workload") + (f", {args.walls} wall boxes" if args.walls else "")

    if rank == 0:
        n_total = n_local * world
        value = n_total * T * args.steps / dt
        out = {
            "metric": "drone-steps/sec (whole node) + generations/sec, 65 536-drone population",
            "value": value, "unit": "drone-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{config_name}: {n_local} drones/GPU x {T} timesteps, 15-{'-'.join(map(str, hidden))}-4 "
                                   f"controller per drone, {args.coins}-coin chain, crash penalty; step = one episode "
                                   f"({roof['launches_per_episode']} fused launches of {roof['timesteps_per_launch']} timesteps x "
                                   f"{roof['drones_per_launch']} drones + finalize"
                                   + (" + RCCL all-gather of fitness records)" if world > 1 else ")"),
                       "population": n_total, "timesteps": T, "hidden": list(hidden), "coins": args.coins, "walls": args.walls,
                       "parallelism": f"population sharded x{world}, no data-path collective inside the rollout"},
            "ranks_seen": ranks_seen, "records": out_records,
            "strong_scaling": strong,
            "evaluations_per_s": args.steps / dt,
            "live_drone_step_fraction": live,
            "resident_engine": resident,
            "random_init_population": random_pop,
            "generations_per_s": None if gen_error else n_gen / gen_dt,
            "generation": {"measured": n_gen, "includes": "rollout, record all-gather, elite, parent exchange (N>1), "
                           "device select+breed (evolve/util.py + Drone.mutate semantics), reset; rollout engine: "
                           + evo.engine + " (work-queue resident kernel where available)"
                           + ", parent exchange: " + (evo.exchange if world > 1 else "local"),
                           "live_drone_step_fraction": gen_live, "error": gen_error,
                           "select_breed_ms": breed_ms, "select_breed_gbs": breed_gbs,
                           "select_breed_note": "gather of 16-byte parent words at random drone indices + streaming "
                                                "child stores; algorithmic 8P bytes per drone"},
            "roofline": roof,
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(hidden, args.coins, T)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
