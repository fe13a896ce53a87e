"""Resident engine on the bench population: ms per episode for a few (n, spec).  LD_DRONES_LIB selects the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes
cases = (((32, 32), 65536, 2000), ((16, 16), 65536, 2000), ((64, 64, 64), 16384, 1000), ((10, 10), 65536, 2000), ((8, 10), 65536, 2000),
         ((16,), 65536, 2000), ((32,), 65536, 2000), ((64,), 65536, 2000), ((8, 8), 65536, 2000), ((24, 24), 65536, 2000),
         ((48, 48), 32768, 1000), ((64, 64), 32768, 1000), ((32, 32, 32), 32768, 1000))
if len(sys.argv) > 1:
    cases = [c for c in cases if "-".join(map(str, c[0])) in sys.argv[1:]]
for hidden, n, T in cases:
    pop = Population(n, hidden, coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    pop.fly(T, engine="resident"); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        pop.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        pop.fly(T, engine="resident"); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(os.environ.get("LD_DRONES_LIB", "default"), hidden, n, T, "ms", round(min(ts), 2), "drone-steps/s %.3g" % (n * T / min(ts) * 1e3), flush=True)
