import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference
lib = _native.lib()
K = 64
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for hidden in ((10, 10), (8, 8), (8, 10)):
    for n in (65536, 8192, 1024):
        pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
        pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
        row = {"hidden": hidden, "n": n}
        ref = None
        for waves in (0, 2, 4, 8, 16):
            pop.reset(); pop.steps(K, waves); pop.steps(K, waves); torch.cuda.synchronize()
            st = (pop.fstate.clone(), pop.istate.clone())
            ref = ref or st
            same = torch.equal(ref[0], st[0]) and torch.equal(ref[1], st[1])
            row[f"w{waves}"] = round(min(timed(lambda: pop.steps(K, waves)) for _ in range(2)) / K, 3)
            row[f"same{waves}"] = same
        row["auto"] = pop.auto_waves_per_block(n, K)
        print(json.dumps(row), flush=True)
        del pop
