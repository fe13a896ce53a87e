"""Step engine (default geometry) on small shards: us per timestep.  LD_DRONES_LIB selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes
for n in (1024, 4096, 8192, 16384):
    pop = Population(n, (32, 32), coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, (32, 32), pop.device, 0))
    best = 1e9
    for rep in range(3):
        pop.reset()
        for _ in range(3): pop.steps(16, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): pop.steps(16, 0)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 640)
    print(os.path.basename(os.environ.get("LD_DRONES_LIB", "default")), n, round(best, 2), "us/step", round(pop.bytes_per_launch(16) / 16 / best / 1e6, 2), "TB/s", flush=True)
