"""Phases of DeviceEvolution.step at config 3 (steady state): rollout, fitness bookkeeping, select+breed, reset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation
from learning_drones_amd.params import coin_chain_reference, GaParams
from bench import make_genomes
n, hidden, T = 65536, (32, 32), 2000
pop = ShardedPopulation(n, hidden, coins=coin_chain_reference(3))
pop.local.set_genomes(make_genomes(torch, n, hidden, pop.local.device, 0))
evo = DeviceEvolution(pop, seed=0x1D20)
for _ in range(4): evo.step(T)
sync = torch.cuda.synchronize
for g in range(4):
    sync(); t0 = time.perf_counter()
    rec = pop.fly(T, engine="auto"); sync(); t1 = time.perf_counter()
    fit = rec[:, 0].contiguous(); best = int(fit.max().item()); idx = int((fit == best).nonzero()[0].item()); sync(); t2 = time.perf_counter()
    pop.local.breed(fit, GaParams.reference(seed=1, generation=g), elite=pop.local.genome_of(idx)); sync(); t3 = time.perf_counter()
    pop.local.reset(); sync(); t4 = time.perf_counter()
    print({"rollout_ms": round((t1 - t0) * 1e3, 3), "fitness_ms": round((t2 - t1) * 1e3, 3), "breed_ms": round((t3 - t2) * 1e3, 3), "reset_ms": round((t4 - t3) * 1e3, 3)}, flush=True)
sync(); t0 = time.perf_counter()
for _ in range(10): evo.step(T)
sync(); print("evo.step ms", round((time.perf_counter() - t0) * 100, 3))
