"""Where a GA generation's time goes (one GPU): rollout variants on the current children, then the generation step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes
n, hidden, T = 65536, (32, 32), 2000
pop = ShardedPopulation(n, hidden, coins=coin_chain_reference(3))
pop.local.set_genomes(make_genomes(torch, n, hidden, pop.local.device, 0))
evo = DeviceEvolution(pop, seed=0x1D20)
def sync(): torch.cuda.synchronize()
for g in range(6):
    genes = pop.local.genomes.clone()
    out = {}
    ref = None
    for name, kw in (("resident", dict(engine="resident")), ("step", dict(engine="step")), ("compacting_64", dict(engine="compacting", steps_per_launch=64)),
                     ("compacting_128", dict(engine="compacting")), ("compacting_256", dict(engine="compacting", steps_per_launch=256))):
        pop.local.reset(); pop.local.fly(T, **kw); sync()                  # (warm: the first call allocates the list)
        pop.local.reset(); sync(); t0 = time.perf_counter()
        rec = pop.local.fly(T, **kw); sync()
        out[name] = round((time.perf_counter() - t0) * 1e3, 2)
        ref = rec.clone() if ref is None else ref
        out["same_" + name] = bool(torch.equal(ref, rec))
    life = rec[:, 1].float()
    out["live"] = round(float(life.mean()) / T, 3)
    out["waves_with_survivor"] = round(float((life.view(-1, 8).max(dim=1).values == T).float().mean()), 3)
    pop.local.reset(); sync(); t0 = time.perf_counter()
    evo.step(T); sync()
    out["generation_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    fit = pop.local.records[:, 0].contiguous()
    from learning_drones_amd.params import GaParams
    for place in (None, "parent"):
        keep = pop.local.genomes
        sync(); t0 = time.perf_counter()
        pop.local.breed(fit, GaParams.reference(seed=3, generation=g), place=place); sync()
        out[f"breed_{place}_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        pop.local.genomes = keep
    print(g, out, flush=True)
