"""us per timestep of ld_fly_steps_f32 (64 timesteps per launch) on LARGE launches by wavefronts per block: the policy
for launches that fill the chip at one wavefront per block already.  usage: ab_large.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes

K = 64
for hidden, n in (((24, 24), 65536), ((32,), 65536), ((64,), 65536), ((48, 48), 65536), ((32, 32, 32), 65536), ((8, 10), 65536), ((8, 8), 131072), ((16,), 131072), ((16, 16), 131072), ((32, 32), 262144),
                  ):
    pop = Population(n, hidden, coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    wide = pop.spec.words_per_drone > 512
    row = {}
    for waves in (1, 2, 4, 8) + ((16,) if wide else ()):
        best = 1e9
        for rep in range(3):
            pop.reset(); pop.steps(K, waves); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): pop.steps(K, waves)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * K))
        row[waves] = round(best, 2)
    print(hidden, n, row, "auto", pop.auto_waves_per_block(n, K), "TB/s at best %.2f" % (pop.bytes_per_launch(K, n) / K / min(row.values()) / 1e6), flush=True)
    del pop
