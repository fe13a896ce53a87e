"""Experiment 5 on the LDS-kept genes: every shipped layer spec of up to 512 gene words, whole-population launches of 64
timesteps by workgroups per block, next to the library's own choice and to the rollout without keeping (tiles and, where
the round-2 policy took it, the cached tile head on the side stream).

    python scripts/exp_lds_keep5.py [out.json]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference

out_path = sys.argv[1] if len(sys.argv) > 1 else None
lib = _native.lib()
rows = []
K = 64


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def episode_ms(pop, T, reps=2, **kw):
    pop.reset(); pop.fly(T, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pop.reset(); pop.fly(T, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for hidden in ((32, 32), (16, 16), (24, 24), (64,), (32,), (16,)):
    for n in (65536, 8192, 1024):
        pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
        pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
        lib.ld_debug_lds_keep(1)
        row = {"hidden": list(hidden), "words": pop.W, "drones": n, "us_per_timestep": {}}
        for waves in (8, 16, 32):
            pop.reset()
            row["us_per_timestep"][str(waves)] = min(timed(lambda: pop.steps(K, waves)) for _ in range(2)) / K
        lib.ld_debug_lds_keep(-1)
        row["auto_waves"] = pop.auto_waves_per_block(n, K)
        T = 1000
        row["rollout_ms_keep"] = min(episode_ms(pop, T) for _ in range(2))
        rec = pop.records.clone()
        lib.ld_debug_lds_keep(0)
        row["rollout_ms_no_keep"] = min(episode_ms(pop, T) for _ in range(2))
        row["records_identical"] = bool(torch.equal(rec, pop.records))
        lib.ld_debug_lds_keep(-1)
        row["drone_steps_per_s_keep"] = n * T / row["rollout_ms_keep"] * 1e3
        rows.append(row)
        print(json.dumps(row), flush=True)
        del pop
if out_path:
    with open(out_path, "w") as f:
        json.dump(rows, f, indent=1)
