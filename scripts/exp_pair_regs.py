import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference
lib = _native.lib()
K = 64
tag = os.path.basename(os.environ.get("LD_DRONES_LIB", "shipped"))
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
hidden = (32, 32)
for n in (65536, 8192, 1024):
    pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    lib.ld_debug_lds_keep(0)
    pop.reset(); pop.steps(K, 0); pop.steps(K, 0); torch.cuda.synchronize()
    ref = (pop.fstate.clone(), pop.istate.clone())
    lib.ld_debug_lds_keep(-1)
    pop.reset(); pop.steps(K, 16); pop.steps(K, 16); torch.cuda.synchronize()
    same = bool(torch.equal(ref[0], pop.fstate) and torch.equal(ref[1], pop.istate))
    us = round(min(timed(lambda: pop.steps(K, 16)) for _ in range(3)) / K, 3)
    print(json.dumps({"lib": tag, "n": n, "us_per_timestep": us, "same": same, "words": pop.kept_words(16, K)}), flush=True)
    del pop
