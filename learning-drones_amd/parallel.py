"""Population sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Drones are independent within a generation (the reference evaluates them one after another,
evolve/evolution.py:64-73), so rank r owns the contiguous drone range shard_bounds(N, W, r) — state
and genomes of that range live only on GPU r — and the T-step rollout needs no communication.  The
only exchange is ONE all-gather per generation of the 16-byte per-drone result records
(ld_fitness_record: reward, flight_time, coins|crashed, distance), after which every rank holds all
N fitness values and can run the identical seeded selection.  At 65 536 drones per GPU a rank
contributes 1 MiB; on the fully connected xGMI mesh that is latency-bound (~tens of microseconds)
against a ~0.2 s rollout, so nothing is bucketed or overlapped.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of rank's drones: contiguous, sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_counts(n_total: int, world: int):
    return [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]


class DistComm:
    """The collectives this module needs, over a torch.distributed group (RCCL on GPUs, gloo in CPU tests).  Anything
    with the same five members (rank, world, all_gather, all_to_all, broadcast, device_alltoall) can stand in for it:
    the single-GPU tests run eight "ranks" as threads of one process over a mailbox (tests/conftest.py, ThreadComm),
    because a GPU box admits only a few processes on its card at a time."""

    def __init__(self, group):
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # gloo has no all-to-all on device tensors; nccl (= RCCL on ROCm) does.  Read from the backend's NAME, identically
        # on every rank and before any collective is issued: probing with a try/except could let one rank fall back
        # while another does not (collective errors surface asynchronously), after which the ranks would wait in
        # different collectives.
        self.device_alltoall = dist.get_backend(group) == "nccl"

    def all_gather(self, tensor):
        """Equal shards [k, ...] -> [world * k, ...] in rank order (ONE all_gather_into_tensor)."""
        import torch
        import torch.distributed as dist
        out = torch.empty((self.world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        dist.all_gather_into_tensor(out, tensor.contiguous(), group=self.group)
        return out

    def broadcast(self, tensor, src: int):
        """In place, from the group's rank `src`."""
        import torch.distributed as dist
        dist.broadcast(tensor, src=dist.get_global_rank(self.group, src), group=self.group)
        return tensor

    def all_to_all(self, send):
        """send[r]: tensor for rank r (first dimension may differ per destination, trailing dimensions
        equal).  Returns recv with recv[r] = what rank r sent here."""
        import torch
        import torch.distributed as dist
        dev = send[0].device
        counts = torch.tensor([t.shape[0] for t in send], dtype=torch.int64, device=dev)
        incoming = torch.empty_like(counts)
        dist.all_to_all_single(incoming, counts, group=self.group)
        incoming = [int(c) for c in incoming.tolist()]
        out = torch.empty((sum(incoming),) + tuple(send[0].shape[1:]), dtype=send[0].dtype, device=dev)
        dist.all_to_all_single(out, torch.cat(send).contiguous(), output_split_sizes=incoming,
                               input_split_sizes=[t.shape[0] for t in send], group=self.group)
        return list(out.split(incoming))


def as_comm(group_or_comm):
    """None (one rank), a communicator object as described at DistComm, or a torch.distributed process group."""
    if group_or_comm is None or hasattr(group_or_comm, "all_gather"):
        return group_or_comm
    return DistComm(group_or_comm)


def gather_records(local_records, n_total: int, process_group=None):
    """All-gather the per-drone result records ([n_local, 4] int32) into [n_total, 4] in global drone
    order on every rank.  Works on any device/backend pair torch.distributed supports."""
    import torch
    comm = as_comm(process_group)
    if comm is None or comm.world == 1:
        return local_records
    counts = shard_counts(n_total, comm.world)
    width = local_records.shape[1]
    if len(set(counts)) == 1:
        return comm.all_gather(local_records)
    pad = max(counts)                                          # unequal shards: pad to the longest, gather, cut
    buf = torch.zeros((pad, width), dtype=local_records.dtype, device=local_records.device)
    buf[: local_records.shape[0]] = local_records
    parts = comm.all_gather(buf).view(comm.world, pad, width)
    return torch.cat([parts[r, :c] for r, c in enumerate(counts)], dim=0)


def exchange_parent_genomes(genomes_local, n_local: int, parents_global, comm):
    """Fetch exactly the parent genomes this rank's children need (selection draws parents
    uniformly over the whole population, so ~(W-1)/W of them live on other ranks).

    genomes_local: this rank's genome array [blocks, W, 64, 4]; parents_global: int tensor of global
    drone indices (one per local child); every rank owns n_local consecutive drones.  Two all-to-alls:
    the requested indices, then the genome rows.  Returns (compact, parents_compact): a genome array
    holding only the distinct needed parents, and the parents re-indexed into it — exactly what
    ld_ga_breed takes as src / parents with one segment."""
    import torch
    from .params import GENOME_BLOCK
    needed = torch.unique(parents_global.long())                          # sorted, distinct
    owner = torch.div(needed, n_local, rounding_mode="floor")
    requests = [needed[owner == r] for r in range(comm.world)]            # sorted by owner already
    to_serve = comm.all_to_all(requests)                                  # indices other ranks want from me
    rows_out = []
    for idx in to_serve:
        slot = idx - comm.rank * n_local
        rows_out.append(genomes_local[torch.div(slot, GENOME_BLOCK, rounding_mode="floor"), :, slot % GENOME_BLOCK, :])
    rows = torch.cat(comm.all_to_all(rows_out))                           # [len(needed), W, 4], in `needed` order
    W = genomes_local.shape[1]
    blocks = (needed.numel() + GENOME_BLOCK - 1) // GENOME_BLOCK
    padded = torch.zeros((blocks * GENOME_BLOCK, W, 4), dtype=genomes_local.dtype, device=genomes_local.device)
    padded[: needed.numel()] = rows
    compact = padded.view(blocks, GENOME_BLOCK, W, 4).permute(0, 2, 1, 3).contiguous()
    parents_compact = torch.searchsorted(needed, parents_global.long()).to(torch.int32)
    return compact, parents_compact


class ShardedPopulation:
    """A population of n_total drones split over the ranks of `process_group`; `local` is this
    rank's Population shard (learning-drones_amd/population.py)."""

    def __init__(self, n_total: int, hidden, *, process_group=None, device=None, **population_kwargs):
        from .population import Population
        self.n_total = int(n_total)
        self.comm = as_comm(process_group)                     # a torch.distributed group, a communicator object, or None
        self.group = process_group
        self.rank, self.world = (0, 1) if self.comm is None else (self.comm.rank, self.comm.world)
        self.begin, self.end = shard_bounds(self.n_total, self.world, self.rank)
        self.local = Population(self.end - self.begin, hidden, device=device, **population_kwargs)

    def set_genomes(self, rows_global) -> None:
        """rows_global: [n_total, P] on every rank (host array or tensor); each rank keeps its slice."""
        self.local.set_genomes(rows_global[self.begin:self.end])

    def fly(self, timesteps: int, engine: str = "step", **geometry):
        """Rollout of the local shard, then the generation-end all-gather.  Returns the [n_total, 4]
        int32 record tensor (global drone order) on every rank.  `geometry`: steps_per_launch /
        waves_per_block of Population.fly."""
        local = self.local.fly(timesteps, engine=engine, **geometry)
        return gather_records(local, self.n_total, self.comm)

    @staticmethod
    def records_to_numpy(records) -> np.ndarray:
        from .params import CRASHED_BIT, FITNESS_DTYPE
        raw = records.cpu().numpy().view(FITNESS_DTYPE).reshape(-1)
        out = np.zeros(len(raw), dtype=[("reward", "<i4"), ("flight_time", "<i4"), ("coins", "<i4"),
                                        ("crashed", "?"), ("distance_to_coin", "<f4")])
        out["reward"], out["flight_time"] = raw["reward"], raw["flight_time"]
        out["coins"] = raw["coins_crashed"] & (CRASHED_BIT - 1)
        out["crashed"] = (raw["coins_crashed"] & CRASHED_BIT) != 0
        out["distance_to_coin"] = raw["distance_to_coin"]
        return out


class DeviceEvolution:
    """The GA loop of evolve/evolution.py:50-111 with every stage on the GPU(s): batched rollout,
    all-gather of the fitness records, argmax / global best, device-side selection + crossover +
    mutation + elitism (Population.breed), state reset.  One instance per rank.

    Multi-GPU: tournament selection draws parents uniformly over the WHOLE population
    (evolve/util.py:9-14), so children on rank r need genomes that live on other ranks.  Every rank
    runs the same Philox-keyed selection for its own children on the same gathered fitness vector and
    then fetches exactly the distinct parents it needs (exchange="alltoall": four all_to_all_single
    calls -- request counts, requested indices, row counts, genome rows -- about one shard's worth of
    genes in and out per rank, 1/W of an all-gather).  exchange="allgather" is the simple variant:
    all-gather every shard (cfg3: 446 MB per rank) and breed from the gathered copy; it is what runs on
    backends without a device all-to-all (gloo).  Collectives per generation: one all_gather_into_tensor
    of the 16-byte records, one broadcast of the elite genome when the global best improves, and the
    parent exchange above.  The all-to-all path has run on gloo-backed CPU tensors and in threaded
    single-GPU rehearsals only; it is unverified on RCCL until a multi-GPU record exists, so the default is
    exchange="allgather" (the two most ordinary collectives) and "alltoall" is opt-in; exchange_bytes() says what
    either moves per generation."""

    def __init__(self, population: "ShardedPopulation", ga=None, seed: int = 0, engine: str = "auto",
                 exchange: str = "allgather", place: Optional[str] = None):
        from .params import GaParams
        if exchange not in ("alltoall", "allgather"):
            raise ValueError("exchange must be 'alltoall' or 'allgather'")
        self.pop = population
        self.engine = engine
        self.exchange = exchange
        if place not in (None, "parent", "lifetime"):
            raise ValueError("place must be None, 'parent' or 'lifetime'")
        # Population.breed: where the children go.  Same children either way (tested); measured at 65 536 x 32-32:
        # "parent" does turn the gather into neighbouring reads, but the breed kernel is bound by its Philox /
        # Box-Muller arithmetic (624 VALU instructions per wavefront and word), not by the gather, and the sort
        # costs 0.25 ms; "lifetime" cut a rollout's masked work by 8 % before the resident kernel had its work
        # queue and nothing since.  So the default is no re-ordering.
        self.place = place
        self._alltoall_checked = False
        self.ga = ga if ga is not None else GaParams.reference(seed=seed)
        self.generation = 0
        self.global_best_score = None
        self.elite = None                      # W x 4 genome words of the best drone so far
        self.history = []
        self.slot_of = None                    # canonical drone index -> global genome slot (None = identity)

    def exchange_bytes(self) -> int:
        """Genome bytes one rank RECEIVES per generation for the parent exchange: all-gather = every other shard;
        all-to-all = the distinct parents its children need that live elsewhere (about (W-1)/W of one shard: tournament
        selection draws ~0.8 n_local distinct parents per rank)."""
        pop = self.pop
        if pop.world == 1:
            return 0
        shard = pop.local.genomes.numel() * pop.local.genomes.element_size()
        return shard * (pop.world - 1) if self.exchange == "allgather" else int(shard * 0.8 * (pop.world - 1) / pop.world)

    def _gather_genomes(self):
        """All-gather variant of the parent exchange: every rank receives every shard."""
        local = self.pop.local.genomes
        if self.pop.world == 1:
            return local, self.pop.local.n, 1
        return self.pop.comm.all_gather(local), self.pop.local.n, self.pop.world      # shards back to back = the segments of ld_ga_breed

    def step(self, timesteps: int):
        """One generation.  Returns (best_score, global_best_score, records)."""
        import torch
        pop = self.pop
        records = pop.fly(timesteps, engine=self.engine)              # rollout + all-gather of records (slot order)
        # The GA itself works on CANONICAL drone indices: child c of a generation is drone c of the next
        # one wherever its genome was placed (slot_of), so a run is the same run with or without the
        # sorted placement of the children below.
        slot_of = self.slot_of
        canon = records if slot_of is None else records[slot_of]
        fitness = canon[:, 0].contiguous()
        lifetimes = records[:, 1].contiguous() if self.place == "lifetime" else None   # indexed by slot, as breed() reads it
        place = "parent" if self.place == "parent" else None
        best_score = int(fitness.max().item())
        # FIRST maximum, as np.argmax (evolution.py:77); torch.argmax does not promise which of several
        # equal maxima it returns, and every rank must name the same owner for the broadcast below
        best_index = int((fitness == best_score).nonzero()[0].item())
        best_slot = best_index if slot_of is None else int(slot_of[best_index].item())
        if self.global_best_score is None or best_score > self.global_best_score:   # evolution.py:80-83
            self.global_best_score = best_score
            elite = torch.empty((pop.local.W, 4), dtype=pop.local.dtype, device=pop.local.device)
            owner = next(r for r in range(pop.world) if shard_bounds(pop.n_total, pop.world, r)[0] <= best_slot
                         < shard_bounds(pop.n_total, pop.world, r)[1])
            if owner == pop.rank:
                elite.copy_(pop.local.genome_of(best_slot - pop.begin))
            if pop.world > 1:
                pop.comm.broadcast(elite, owner)
            self.elite = elite
        self.ga.generation = self.generation
        if pop.world > 1 and pop.n_total % pop.world:
            raise ValueError("the device generation step needs equal shards (n_total divisible by the world size)")
        if pop.world > 1 and self.exchange == "alltoall" and not self._alltoall_checked:
            if not pop.comm.device_alltoall:                   # decided from the backend's name, identically on every rank (DistComm)
                self.exchange = "allgather"
            self._alltoall_checked = True
        if pop.world > 1 and self.exchange == "alltoall":
            # only the selected parents travel (~1/W of an all-gather)
            pop.local.breed(fitness, self.ga, elite=self.elite, child_begin=pop.begin, lifetimes=lifetimes, slot_of=slot_of, place=place,
                            fetch_parents=lambda parents: exchange_parent_genomes(pop.local.genomes, pop.local.n, parents, pop.comm))
        else:
            src, n_per_segment, n_segments = self._gather_genomes()
            pop.local.breed(fitness, self.ga, elite=self.elite, child_begin=pop.begin, src=src,
                            n_per_segment=n_per_segment, n_segments=n_segments, lifetimes=lifetimes, slot_of=slot_of, place=place)
        # where each child went: local placement -> global canonical-index -> slot map for the next generation
        order = pop.local.last_order
        if order is None:
            self.slot_of = None
        else:
            local_slot_of = torch.empty_like(order, dtype=torch.int64)
            local_slot_of[order.long()] = torch.arange(order.numel(), device=order.device) + pop.begin
            if pop.world > 1:
                local_slot_of = pop.comm.all_gather(local_slot_of)                     # 8 bytes per drone
            self.slot_of = local_slot_of
        pop.local.reset()                                             # Drone.reset for every child
        self.generation += 1
        self.history.append((best_score, self.global_best_score))
        return best_score, self.global_best_score, records

    # ---- resume (the reference never saves the population, RNG state or generation counter,
    # evolve/evolution.py:85-87; the Philox draws are keyed by (seed, generation), so these few
    # values plus the genomes continue a run bit for bit)
    def save(self, path: str) -> None:
        import numpy as _np
        pop = self.pop.local
        _np.savez_compressed(path, generation=self.generation, seed=int(self.ga.seed),
                             ga=_np.array([self.ga.p_tournament, self.ga.p_crossover, self.ga.mutation_rate,
                                           self.ga.mutation_sigma]),
                             global_best_score=-2**62 if self.global_best_score is None else self.global_best_score,
                             elite=_np.zeros((0, 4)) if self.elite is None else self.elite.cpu().numpy(),
                             history=_np.array(self.history, dtype=_np.int64).reshape(-1, 2),
                             genes=pop.get_genomes().cpu().numpy(), rank=self.pop.rank, world=self.pop.world,
                             slot_of=_np.zeros((0,), _np.int64) if self.slot_of is None else self.slot_of.cpu().numpy())

    def load(self, path: str) -> None:
        import numpy as _np
        import torch
        from .params import GaParams
        z = _np.load(path, allow_pickle=False)
        if int(z["world"]) != self.pop.world or int(z["rank"]) != self.pop.rank:
            raise ValueError("checkpoint was written by a different rank / world size")
        g = z["ga"]
        self.ga = GaParams.reference(seed=int(z["seed"]), p_tournament=g[0], p_crossover=g[1], mutation_rate=g[2],
                                     mutation_sigma=g[3])
        self.generation = int(z["generation"])
        gb = int(z["global_best_score"])
        self.global_best_score = None if gb == -2**62 else gb
        self.elite = None if len(z["elite"]) == 0 else torch.as_tensor(z["elite"]).to(self.pop.local.device, self.pop.local.dtype)
        self.history = [tuple(int(v) for v in row) for row in z["history"]]
        slot_of = z["slot_of"] if "slot_of" in z else _np.zeros((0,), _np.int64)      # (absent in round-1 checkpoints)
        self.slot_of = None if len(slot_of) == 0 else torch.as_tensor(slot_of).to(self.pop.local.device)
        self.pop.local.set_genomes(z["genes"])
        self.pop.local.reset()
