"""Sub-block workgroups: us per timestep of ld_fly_steps_f32 at waves_per_block 4/8/16/32 on small shards (32-32, K=64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes
hidden = tuple(int(h) for h in sys.argv[1].split(",")) if len(sys.argv) > 1 else (32, 32)
for n in (1024, 2048, 4096, 8192, 16384):
    pop = Population(n, hidden, coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    row = {}
    ref = None
    for waves in ((4, 8, 16) if pop.W > 512 else (4, 8, 16, 32)):
        best = 1e9
        for rep in range(3):
            pop.reset()
            pop.steps(64, waves)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): pop.steps(64, waves)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 640)
        row[waves] = round(best, 2)
        pop.reset(); pop.steps(200, waves)
        st = (pop.fstate.clone(), pop.istate.clone())
        if ref is None: ref = st
        else: assert torch.equal(st[0], ref[0]) and torch.equal(st[1], ref[1]), (n, waves)
    print(hidden, n, row, "auto", pop.auto_waves_per_block(n, 64), "TB/s best %.2f" % (pop.bytes_per_launch(64) / 64 / min(row.values()) / 1e6), flush=True)
