import json, sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference
lib = _native.lib()
K = 64
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for hidden in ((32, 32), (16, 16), (64,), (32,), (16,)):
    for n in (65536, 8192, 1024):
        pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
        pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
        row = {"hidden": hidden, "n": n}
        ref = None
        for mode, name in ((0, "nokeep"), (2, "unpaired"), (1, "paired")):
            lib.ld_debug_lds_keep(mode)
            pop.reset(); pop.steps(K, 16 if mode else 0); pop.steps(K, 16 if mode else 0); torch.cuda.synchronize()
            st = (pop.fstate.clone(), pop.istate.clone())
            ref = ref or st
            row["same_" + name] = bool(torch.equal(ref[0], st[0]) and torch.equal(ref[1], st[1]))
            if mode:
                row["us_" + name] = round(min(timed(lambda: pop.steps(K, 16)) for _ in range(3)) / K, 3)
                if n == 65536:
                    row["us_w8_" + name] = round(min(timed(lambda: pop.steps(K, 8)) for _ in range(2)) / K, 3)
        lib.ld_debug_lds_keep(-1)
        print(json.dumps(row), flush=True)
        del pop
