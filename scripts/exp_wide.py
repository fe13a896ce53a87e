"""fly_wide_kernel (workgroups of 64 drones, every gene on chip, the angles' sin / cos beside the output layer) against
the paired kernel (workgroups of 32) and the unwide 64-drone workgroups: bit-identity and microseconds per timestep."""
import json, sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference
import os
lib = _native.lib()
tag = os.path.basename(os.environ.get("LD_DRONES_LIB", "shipped"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
specs = [tuple(int(x) for x in s.split("-")) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32-32", "16-16", "64", "24-24", "32", "10-10"])]
sizes = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["65536", "16384", "8192", "1024", "999"])]
for hidden in specs:
    for n in sizes:
        pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
        pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
        row = {"lib": tag, "hidden": hidden, "n": n}
        ref = None
        # (mode, waves_per_block): nothing kept; the paired kernel; the unwide 64-drone workgroups; the wide kernel
        for mode, waves, name in ((0, 0, "nokeep"), (1, 16, "pair32"), (2, 8, "split64"), (1, 8, "wide64")):
            lib.ld_debug_lds_keep(mode)
            pop.reset(); pop.steps(K, waves); pop.steps(7, waves); pop.steps(K, waves); torch.cuda.synchronize()
            st = (pop.fstate.clone(), pop.istate.clone())
            ref = ref or st
            row["same_" + name] = bool(torch.equal(ref[0], st[0]) and torch.equal(ref[1], st[1]))
            if mode:
                row["us_" + name] = round(min(timed(lambda: pop.steps(K, waves)) for _ in range(3)) / K, 3)
        lib.ld_debug_lds_keep(-1)
        print(json.dumps(row), flush=True)
        del pop
