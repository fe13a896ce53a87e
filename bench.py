#!/usr/bin/env python3
"""bench.py — drone-steps/s of the fused fly() rollout on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 3, the one the metric is quoted on): 65 536 drones per GPU x 2 000
timesteps, 15-32-32-4 controller, 3-coin chain, crash penalty on, float32.  A "step" is one whole
episode of the population = the fused-step launches of ld_fly_rollout_f32 (4 launches of 512 timesteps over all
65 536 drones, the drone state in registers in between; workgroups of 8 wavefronts fly 32 drones each, both halves of
every wavefront summing rows of the controller, and hold their drones' genes in registers and LDS for the launch) + the finalize launch (+ the RCCL all-gather of the
16-byte fitness records when N > 1).  Population shards are independent, so the headline line is weak
scaling: every rank owns 65 536 drones.  With N > 1 the same run ALSO times the strong-scaling case
BASELINE's target describes (ONE 65 536-drone population split over the N GPUs) and reports it in the
`strong_scaling` object of the same JSON line, together with `ranks_seen` and a checksum of the gathered
records, so that the line proves the collective saw N ranks.  `--strong` makes the strong case the
headline instead.

Genomes are synthetic: hover controllers (learning-drones_amd/controllers.py) with small per-drone
noise, so that nearly every drone is alive for all 2 000 steps and every counted drone-step does the
full work (frozen blocks skip their gene traffic; the JSON line reports the live fraction).

Rank 0 prints ONE JSON line.  `roofline` prices the headline launch, timed live with HIP events on the launch
stream, at every level its bytes can come from -- HBM, the Infinity-Cache fabric, the L2s, LDS -- and at the vector
issue rate, and names the level with the highest fraction `bound` (so `frac` <= 1 by construction); the counter-backed
levels are read from the rocprofv3 summaries committed under profiles/ for this same configuration.
`roofline_hbm_bound` is the same kernel family with nothing kept on chip (every timestep's genes from HBM), the figure
BASELINE's ">= 60 % of the HBM roofline" asks about.  `cpu_baseline` times the CPU oracle (the stand-in for the
reference's Python loop) on the host cores, rank 0, N = 1 only.
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
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md, L2 (per XCD): ~34.5 TB/s aggregate
LDS_PEAK_GBS = 150000.0          # MI355X_MICROARCH.md, LDS: ~150 TB/s aggregate for ds_read_b64 / b128 with every CU streaming
FABRIC_PEAK_GBS = 7900.0         # Infinity Cache -> L2 stream: the best a plain reader got on this pool (profiles/r02_mall_read.json)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2   # wave64 vector instructions per ns, chip-wide: one per 2 cycles per SIMD-32 at 2.4 GHz
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
    ap.add_argument("--no-lds-keep", action="store_true", help="fly without the genes kept in LDS (ld_debug_lds_keep(0)): the rollout "
                    "then tiles the population for the Infinity Cache; with --tile-drones = --drones every timestep's genes come from HBM")
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

    from learning_drones_amd import _native
    from learning_drones_amd.parallel import ShardedPopulation
    from learning_drones_amd.params import coin_chain_reference
    if args.no_lds_keep:
        _native.lib().ld_debug_lds_keep(0)
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

    def measure(n_local, steps, warmup, force_tile=None, keep=None):
        """`steps` timed episodes of a population of n_local drones per rank: barrier + synchronize on both
        sides, wall time = max over ranks; the fused launches' average duration from HIP events on the
        launch stream over the same region.  keep=False: without the genes kept in LDS."""
        if keep is not None:
            _native.lib().ld_debug_lds_keep(1 if keep else 0)
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
            geo["lds_words"], geo["register_words"], geo["streamed_words"] = pop.local.kept_words(geo["waves_per_block"], min(k, T))

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
        if keep is not None:
            _native.lib().ld_debug_lds_keep(0 if args.no_lds_keep else -1)
        launch_ms = ev0.elapsed_time(ev1) / (steps * geo["launches"]) if steps else float("nan")
        launch_s = launch_ms * 1e-3
        # algorithmic bytes of the AVERAGE launch of the episode: every drone streams its genes T times and reads and
        # writes its state once per launch (the last launch of a tile is shorter when K does not divide T, the last
        # tile smaller when the tile does not divide n: averaging over the episode's launches accounts for both)
        launches_per_tile = -(-T // geo["steps_per_launch"])
        state_bytes = pop.local.bytes_per_launch(0, n_local)                  # 200 B per drone, once per launch
        gene_bytes = pop.local.bytes_per_launch(T, n_local) - state_bytes     # 4P per drone-step, whole episode
        tiles = -(-n_local // geo["tile_drones"])
        bytes_per_launch = (gene_bytes + launches_per_tile * state_bytes) / geo["launches"]
        steps_per_launch_avg = T * tiles / geo["launches"]                    # timesteps of the average launch
        drones_per_launch_avg = n_local / tiles
        # where the average launch's gene bytes come from.  A gene word is 16 bytes; per drone the launch keeps `lds_words` in
        # LDS and `register_words` in registers (both read from memory once per launch; the LDS ones read back from LDS every
        # timestep) and requests `streamed_words` per timestep from the L2s, which pass on to the fabric / HBM what they do
        # not hold.
        in_lds, in_regs, streamed = geo["lds_words"], geo["register_words"], geo["streamed_words"]
        kept = in_lds + in_regs
        lds_bytes = drones_per_launch_avg * steps_per_launch_avg * 16 * in_lds
        l2_request_bytes = drones_per_launch_avg * (steps_per_launch_avg * 16 * streamed + 16 * kept) + state_bytes / tiles
        # HBM, modelled: a launch's genes are new to the caches when they exceed the Infinity Cache between two launches
        # over the same drones (the next launch of a tile comes right behind it; of the whole population, after 446 MB)
        tile_gene_bytes = drones_per_launch_avg * 4 * pop.local.P
        hbm_bytes = (tile_gene_bytes if tile_gene_bytes > 200e6 else tile_gene_bytes / launches_per_tile) + state_bytes / tiles
        if steps_per_launch_avg < 2 and tile_gene_bytes > 200e6:
            hbm_bytes = bytes_per_launch                                      # one timestep per launch: everything from HBM
        levels = {
            "hbm": {"bytes_per_launch": hbm_bytes, "source": "model: a launch's genes from HBM once (they exceed the Infinity Cache "
                    "between launches) + the state", "achieved": hbm_bytes / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"},
            "l2": {"bytes_per_launch": l2_request_bytes, "source": "model: the words the launch does not keep in LDS, requested "
                   "or in registers, requested from the L2s every timestep (+ the kept words once per launch)", "achieved": l2_request_bytes / launch_s / 1e9,
                   "peak": L2_PEAK_GBS, "unit": "GB/s"},
            "lds": {"bytes_per_launch": lds_bytes, "source": "model: the words kept in LDS, read back every timestep (the words held "
                    "in registers cost no traffic at all)",
                    "achieved": lds_bytes / launch_s / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s"},
        }
        paired = kept > 0 and geo["waves_per_block"] == 16 and any(h % 16 == 0 for h in hidden)
        wide = kept > 0 and streamed == 0 and geo["waves_per_block"] == 8 and pop.local.flies_wide()
        if wide:
            # the layer outputs (and the angles' sin / cos) that cross wavefronts through LDS: every wavefront reads every hidden
            # layer's outputs, four read the last hidden layer's, every one the 4 thrusts, the 6 sin / cos values, 1 / cos(pitch) and the coin flag; written once
            xch_words = 8 * sum(hidden[:-1]) + 4 * hidden[-1] + 8 * 12 + sum(hidden) + 4 + 8
            lds_bytes += drones_per_launch_avg * steps_per_launch_avg * 4 * xch_words
            levels["lds"]["bytes_per_launch"] = lds_bytes
            levels["lds"]["achieved"] = lds_bytes / launch_s / 1e9
            levels["lds"]["source"] = ("model: the gene words kept in LDS, read back every timestep, + the layer outputs, thrusts and sin / cos "
                                       "values that cross the workgroup's 8 wavefronts through LDS (every wavefront reads every hidden layer's outputs)")
            kernel_name = ("fly_wide_kernel<%s> (workgroups of 8 wavefronts x 64 drones, lane = drone; %d gene words per drone in registers, "
                           "%d in LDS, none loaded per timestep; sin / cos of the angles computed beside the output layer)"
                           % ("-".join(map(str, hidden)), in_regs, in_lds))
        elif paired:
            kernel_name = ("fly_pair_kernel<%s> (workgroups of 8 wavefronts x 32 drones, both halves of a wavefront summing rows; %d gene "
                           "words per drone in registers, %d in LDS)" % ("-".join(map(str, hidden)), in_regs, in_lds))
        elif geo["steps_per_launch"] > 1 or geo["waves_per_block"] > 1 or pop.local.has_resident_engine():
            kernel_name = ("fly_split_kernel<float, %d wavefronts per block, %d + %d gene words per drone kept in LDS + registers, %s>"
                           % (geo["waves_per_block"], in_lds, in_regs, "-".join(map(str, hidden))))
        else:
            kernel_name = "fly_step_kernel<float, GenericNet> (run-time layer spec)"
        roof = {"kernel": kernel_name,
                "launch_us": launch_ms * 1e3, "bytes_per_launch": bytes_per_launch, "algorithmic_gbs": bytes_per_launch / launch_s / 1e9,
                "timesteps_per_launch": geo["steps_per_launch"], "wavefronts_per_block": geo["waves_per_block"],
                "launches_per_episode": geo["launches"], "drones_per_launch": geo["tile_drones"], "tiles": tiles,
                "lds_words_per_drone": in_lds, "register_words_per_drone": in_regs, "kept_words_per_drone": kept, "streamed_words_per_drone": streamed,
                "levels": levels,
                "bytes_model": "algorithmic bytes per launch, averaged over the episode's launches: drones x (timesteps x 4P + 200): "
                               "genes once per timestep, the 25 state words read and written once per launch (they stay in registers "
                               "in between); 4P + 200 per drone-step at one timestep per launch.  algorithmic_gbs is that over the "
                               "launch time: what an implementation that fetched every gene from memory every timestep would have to "
                               "stream; the launch serves most of it on chip, so it is NOT priced against a memory ceiling -- "
                               "`levels` is"}
        return pop, records, dt, roof

    def close_roofline(roof, n_drones):
        """Add the counter-backed levels from the committed profile of this configuration (profiles/r*_pmc.json: they
        cannot be collected inside this run), then name the binding level."""
        for f in sorted((REPO / "profiles").glob("r*_pmc.json"), reverse=True):
            prof = json.loads(f.read_text())
            cfg = prof.get("config")
            if not cfg or "per_launch" not in prof:
                continue
            if (cfg["drones"] == n_drones and tuple(cfg["hidden"]) == hidden and cfg["coins"] == args.coins
                    and cfg.get("timesteps_per_launch", 1) == roof["timesteps_per_launch"]
                    and cfg.get("wavefronts_per_block", 1) == roof["wavefronts_per_block"]
                    and cfg.get("drones_per_launch", cfg["drones"]) == roof["drones_per_launch"]
                    and cfg.get("kept_words_per_drone", 0) == roof["kept_words_per_drone"]):
                per, launch_s = prof["per_launch"], roof["launch_us"] * 1e-6
                roof["counters_source"] = f"profiles/{f.name}"
                roof["launch_us_profiled"] = prof.get("launch_us_profiled")
                if per.get("fabric_bytes") is not None:
                    roof["levels"]["fabric"] = {"bytes_per_launch": per["fabric_bytes"], "source": "PMC: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 "
                                                "correction), the L2s' memory side: Infinity-Cache hits included", "achieved":
                                                per["fabric_bytes"] / launch_s / 1e9, "peak": FABRIC_PEAK_GBS, "unit": "GB/s"}
                    roof["traffic"] = per["fabric_bytes"]
                if per.get("l2_hit_rate") is not None:
                    roof["levels"]["l2"]["hit_rate_pmc"] = per["l2_hit_rate"]
                if per.get("valu_wave_instructions") is not None:
                    roof["levels"]["valu_issue"] = {"wave_instructions_per_launch": per["valu_wave_instructions"], "source": "PMC: SQ_INSTS_VALU",
                                                    "achieved": per["valu_wave_instructions"] / launch_s / 1e9, "peak": VALU_PEAK_GINST,
                                                    "unit": "G wave-instructions/s"}
                if per.get("lds_busy_fraction") is not None:
                    roof["levels"]["lds"].update(array_busy_fraction_pmc=per["lds_busy_fraction"], bank_conflict_cycles_pmc=per.get("lds_bank_conflict_cycles"),
                                                 array_cycles_pmc=per.get("lds_array_cycles"))
                for key in ("sq_busy_fraction", "sq_issue_fraction", "sq_wait_fraction"):
                    if per.get(key) is not None:
                        roof[key] = per[key]
                break
        roof.setdefault("traffic", None)
        for lv in roof["levels"].values():
            lv["frac"] = lv["achieved"] / lv["peak"]
        name = max(roof["levels"], key=lambda k: roof["levels"][k]["frac"])
        top = roof["levels"][name]
        roof.update(bound={"hbm": "hbm", "fabric": "infinity-cache fabric", "l2": "l2", "lds": "lds", "valu_issue": "valu issue"}[name],
                    achieved=top["achieved"], peak=top["peak"], unit=top["unit"], frac=top["frac"])
        return roof

    n_total_strong = args.drones
    n_local = args.drones // world if args.strong else args.drones
    pop, records, dt, roof = measure(n_local, args.steps, args.warmup)
    roof = close_roofline(roof, n_local)

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
                  "roofline": close_roofline(roof_s, n_total_strong // world),
                  "records": {"rows": int(rec_s.shape[0]), "checksum": checksum(rec_s)}}
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
        extras["one_timestep_per_launch"] = {"launch_us": us, "bound": "hbm", "achieved": pop.local.bytes_per_launch(1) / us / 1e3, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": pop.local.bytes_per_launch(1) / us / 1e3 / HBM_PEAK_GBS,
                                             "bytes_per_launch": pop.local.bytes_per_launch(1),
                                             "note": "ld_fly_step_f32: the literal contract, one launch advances the whole population one "
                                                     "timestep; nothing can be kept on chip, every gene comes from HBM every launch"}
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

    # The HBM-bound form of the fused step is the literal contract launch above -- one timestep per launch: the population's
    # genes (446 MB) exceed the 256 MiB Infinity Cache, so every launch streams all of them from HBM (PMC: traffic = 1.001 x
    # algorithmic) -- and that is what BASELINE's ">= 60 % of the HBM roofline on the fused step kernel" asks about.
    hbm_bound = None
    if "one_timestep_per_launch" in extras and pop.local.bytes_per_launch(1) > 300e6:
        hbm_bound = dict(extras["one_timestep_per_launch"])
        hbm_bound["kernel"] = "fly_split_kernel<float, %d wavefront(s) per block, nothing kept, %s>, one timestep per launch" % (
            pop.local.auto_waves_per_block(n_local, 1), "-".join(map(str, hidden)))
        for f in sorted((REPO / "profiles").glob("r*_c3_k1_pmc.json"), reverse=True):
            prof = json.loads(f.read_text())
            if prof.get("config", {}).get("drones") == n_local and tuple(prof["config"]["hidden"]) == hidden and prof.get("per_launch", {}).get("fabric_bytes"):
                hbm_bound["traffic"], hbm_bound["counters_source"] = prof["per_launch"]["fabric_bytes"], f"profiles/{f.name}"
                break
        hbm_bound.setdefault("traffic", None)
    # Between the two: 64 timesteps per launch with NOTHING kept in LDS or registers (ld_debug_lds_keep(0)) -- whole-population
    # launches, whose resident workgroups re-stream 111 MB of genes from the Infinity Cache, and the round-2 rollout
    # (Infinity-Cache tiles): both bound by the fabric between that cache and the L2s.
    if not args.no_extras and roof["kept_words_per_drone"] > 0 and 4 * pop.local.P * n_local > 300e6:
        _, _, dt_w, roof_w = measure(n_local, args.steps, 1, force_tile=n_local, keep=False)
        close_roofline(roof_w, n_local)
        extras["nothing_kept_on_chip"] = {"value": n_local * world * T * args.steps / dt_w, "unit": "drone-steps/s", "ms_per_step": dt_w / args.steps * 1e3,
                                          "roofline": {k: roof_w.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "launch_us", "bytes_per_launch",
                                                                                  "algorithmic_gbs", "traffic", "counters_source", "kernel",
                                                                                  "timesteps_per_launch", "wavefronts_per_block", "drones_per_launch")},
                                          "note": "whole-population launches of 64 timesteps, ld_debug_lds_keep(0): every gene word requested from the "
                                                  "L2s every timestep; 256 workgroups are resident and re-stream their 111 MB from the Infinity Cache"}
        _, _, dt_t, roof_t = measure(n_local, args.steps, 1, keep=False)
        extras["without_kept_genes"] = {"value": n_local * world * T * args.steps / dt_t, "unit": "drone-steps/s", "ms_per_step": dt_t / args.steps * 1e3,
                                        "launch_us": roof_t["launch_us"], "drones_per_launch": roof_t["drones_per_launch"],
                                        "wavefronts_per_block": roof_t["wavefronts_per_block"], "algorithmic_gbs": roof_t["algorithmic_gbs"],
                                        "note": "the round-2 rollout: tiles of ~112 MB of genes that the Infinity Cache holds, nothing in LDS"}

    # The shard one GPU owns when BASELINE's 65 536-drone population is spread over 8 (N = 1 only: with
    # N > 1 the strong_scaling object measures the real thing)
    if world == 1 and not args.no_extras and n_local == WORKLOAD["drones_per_gpu"]:
        _, _, dt_8, roof_8 = measure(n_local // 8, args.steps, 1)
        extras["shard_of_8"] = {"drones": n_local // 8, "value": n_local // 8 * T * args.steps / dt_8, "unit": "drone-steps/s",
                                "ms_per_step": dt_8 / args.steps * 1e3, "roofline": close_roofline(roof_8, n_local // 8)}

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
    steady_live, steady_dt = 0.0, float("nan")
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
            # ... and the same loop once the population is the GA's own (four more generations on, then six timed): the
            # first generations above still carry the hand-made stable flyers' long-lived children
            for _ in range(4):
                evo.step(T)
            fence()
            g1 = time.perf_counter()
            for _ in range(6):
                _, _, rec = evo.step(T)
                steady_live += float(rec[:, 1].double().sum().item()) / (rec.shape[0] * T) / 6
            fence()
            steady_dt = max_over_ranks(time.perf_counter() - g1)
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
                           "exchange": evo.exchange if world > 1 else "local",
                           "exchange_bytes_per_generation_per_rank": evo.exchange_bytes(),
                           "live_drone_step_fraction": gen_live, "error": gen_error,
                           "steady_state": None if gen_error else {
                               "generations_per_s": 6 / steady_dt, "measured": 6, "after_generations": 8, "live_drone_step_fraction": steady_live,
                               "note": "the same loop from the ninth generation on: generations_per_s above is quoted on the first "
                                       "three generations after the hand-made stable flyers (comparable across rounds)"},
                           "select_breed_ms": breed_ms, "select_breed_gbs": breed_gbs,
                           "select_breed_note": "gather of 16-byte parent words at random drone indices + streaming "
                                                "child stores; algorithmic 8P bytes per drone"},
            "roofline": roof,
            "roofline_hbm_bound": hbm_bound,
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
