"""Grid A/B of ld_fly_steps_f32 on one GPU: per-step time (us) and algorithmic TB/s for
(drones, layer spec) x wavefronts per block x steps per launch.  HIP events on the launch stream."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes

def measure(pop, L, K, steps=320):
    pop.reset()
    for _ in range(3): pop.steps(K, L)
    pop.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps // K): pop.steps(K, L)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (steps // K * K)

cases = [((32, 32), n) for n in (1024, 8192, 16384, 32768, 65536)] + [((64, 64, 64), n) for n in (8192, 16384, 131072)] + [((16, 16), 1024)]
for hidden, n in cases:
    pop = Population(n, hidden, coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    for L in (1, 2, 4, 8):
        if L > 1 and n * L > 64 * 2048: continue
        row = {"hidden": hidden, "n": n, "L": L}
        for K in (1, 2, 4, 16, 64):
            us = measure(pop, L, K)
            row[f"K{K}"] = (round(us, 2), round(pop.bytes_per_drone_step * n / us / 1e6, 2))
        print(json.dumps(row), flush=True)
