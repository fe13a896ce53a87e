"""A/B of the fused step kernels on one GPU: per-launch time and algorithmic TB/s for
(drones, layer spec, wavefronts per block, steps per launch), HIP events on the launch stream.
    python scripts/ab_split.py [--quick]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_genomes

def measure(pop, fn, launches, steps_per_launch=1):
    pop.reset()
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (launches * steps_per_launch)
    return us, pop.bytes_per_drone_step * pop.n / us / 1e6

cases = [((32, 32), n) for n in (1024, 4096, 8192, 16384, 32768, 49152, 65536, 131072)] + \
        [((16, 16), 1024), ((16, 16), 8192), ((64, 64, 64), 8192), ((64, 64, 64), 16384), ((64, 64, 64), 131072), ((10, 10), 65536)]
if "--quick" in sys.argv: cases = [((32, 32), 8192), ((32, 32), 32768), ((32, 32), 65536), ((16, 16), 1024)]
legacy = False                                   # (the one-wavefront float32 step kernel this compared against is gone)
out = []
for hidden, n in cases:
    pop = Population(n, hidden, coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    row = {"hidden": hidden, "n": n}
    us, tbs = measure(pop, pop.step, 300)
    row["step" + ("_legacy" if legacy else "_auto")] = (round(us, 2), round(tbs, 2))
    if not legacy:
        for L in (1, 2, 4, 8):
            if n * L > 64 * 2048 and L > 1: continue
            us, tbs = measure(pop, lambda: pop.steps(1, L), 300)
            row[f"L{L}"] = (round(us, 2), round(tbs, 2))
        for K in (4, 16):
            if n > 16384: continue
            us, tbs = measure(pop, lambda: pop.steps(K, 0), max(300 // K, 20), K)
            row[f"K{K}"] = (round(us, 2), round(tbs, 2))
    print(json.dumps(row), flush=True)
