"""Experiment 3 on the LDS-kept genes: workgroups of 8 wavefronts that fly 48 drones each (waves_per_block = 11), so
that a 12 288-drone tile gives each of the 256 CUs exactly one workgroup with 43 % of its drones' genes in LDS.

    python scripts/exp_lds_keep3.py [out.json]
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
hidden, K = (32, 32), 64
rows = []


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


lib.ld_debug_lds_keep(1)
for n in (12288, 16384, 8192, 10240, 24576, 6144, 4096):
    pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    ref = None
    for waves in (8, 11, 16, 32):
        pop.reset()
        pop.steps(K, waves); pop.steps(K, waves)
        torch.cuda.synchronize()
        state = (pop.fstate.clone(), pop.istate.clone())
        ref = ref or state
        same = bool(torch.equal(ref[0], state[0]) and torch.equal(ref[1], state[1]))
        us = min(timed(lambda: pop.steps(K, waves)) for _ in range(3))
        row = {"drones": n, "waves_per_block": waves, "us_per_launch": us, "us_per_timestep": us / K,
               "drones_per_us": n / (us / K), "bit_identical": same}
        rows.append(row)
        print(json.dumps(row), flush=True)
    del pop

n, T = 65536, 2000
pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
ref = None
for name, kw in (("default", {}), ("tile 12288 x 48-drone workgroups", dict(tile_drones=12288, waves_per_block=11)),
                 ("tile 16384 x 48-drone workgroups", dict(tile_drones=16384, waves_per_block=11)),
                 ("tile 16384 x 8", dict(tile_drones=16384, waves_per_block=8)),
                 ("tile 13120 x 48-drone workgroups", dict(tile_drones=13120, waves_per_block=11)), ("default", {})):
    pop.reset(); pop.fly(T, **kw); torch.cuda.synchronize()
    rec = pop.records.clone()
    ref = rec if ref is None else ref
    t0 = time.perf_counter()
    for _ in range(3):
        pop.reset(); pop.fly(T, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    row = {"rollout": name, "ms_per_episode": ms, "drone_steps_per_s": n * T / ms * 1e3, "records_identical": bool(torch.equal(ref, rec))}
    rows.append(row)
    print(json.dumps(row), flush=True)
lib.ld_debug_lds_keep(-1)
if out_path:
    with open(out_path, "w") as f:
        json.dump(rows, f, indent=1)
