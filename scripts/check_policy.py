"""us per timestep (64 per launch) at the library's own choice of wavefronts per block next to fixed choices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes
K = 64
for hidden, n in (((10, 10), 1024), ((10, 10), 8192), ((8, 10), 8192), ((8, 8), 8192), ((10, 10), 65536), ((16, 16), 65536), ((32, 32), 65536),
                  ((32, 32), 8192), ((64, 64, 64), 131072), ((24, 24), 65536), ((16,), 131072)):
    pop = Population(n, hidden, coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    row = {}
    for waves in (0, 1, 2, 4, 8, 16):
        best = 1e9
        for rep in range(3):
            pop.reset(); pop.steps(K, waves); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): pop.steps(K, waves)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * K))
        row[waves] = round(best, 2)
    print(hidden, n, "auto =", pop.auto_waves_per_block(n, K), row, flush=True)
    del pop
