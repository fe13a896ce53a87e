"""Experiment: genes kept in LDS for the whole launch (fly_split_kernel<..., KeepKiB>, ld_debug_lds_keep).

    python scripts/exp_lds_keep.py [hidden] [out.json]

Times launches of 64 timesteps (ld_fly_steps_f32) by drones per launch x wavefronts per block x keep on / off,
checks that keep on and keep off leave bit-identical states, and times the default rollout (tiles + cached head)
of the bench population both ways.  Results: profiles/r03_lds_keep.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference

hidden = tuple(int(h) for h in (sys.argv[1] if len(sys.argv) > 1 else "32,32").split(","))
out_path = sys.argv[2] if len(sys.argv) > 2 else None
lib = _native.lib()
K = 64
results = {"hidden": list(hidden), "timesteps_per_launch": K, "launches": [], "rollouts": []}


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


for n in (16384, 8192, 65536, 4096):
    pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    ref = None
    for waves in (8, 16, 32, 4):
        for keep in (0, 1):
            if waves == 4 and keep:
                continue
            lib.ld_debug_lds_keep(keep)
            pop.reset()
            pop.steps(K, waves)
            pop.steps(K, waves)
            torch.cuda.synchronize()
            state = (pop.fstate.clone(), pop.istate.clone())
            if ref is None:
                ref = state
            same = bool(torch.equal(ref[0], state[0]) and torch.equal(ref[1], state[1]))
            us = [timed(lambda: pop.steps(K, waves)) for _ in range(3)]
            row = {"drones": n, "waves_per_block": waves, "keep": keep, "us_per_launch": us, "us_per_timestep": min(us) / K,
                   "algorithmic_tbs": pop.bytes_per_launch(K) / min(us) / 1e6, "bit_identical": same}
            results["launches"].append(row)
            print(json.dumps(row), flush=True)
    del pop

# the default rollout of the bench population (config 3), both ways
n, T = 65536, 2000
pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
ref = None
for keep in (0, 1, 0, 1):
    lib.ld_debug_lds_keep(keep)
    pop.reset(); pop.fly(T); torch.cuda.synchronize()
    rec = pop.records.clone()
    if ref is None:
        ref = rec
    t0 = time.perf_counter()
    for _ in range(3):
        pop.reset(); pop.fly(T)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    row = {"rollout": "default geometry", "drones": n, "timesteps": T, "keep": keep, "ms_per_episode": ms,
           "drone_steps_per_s": n * T / ms * 1e3, "records_identical": bool(torch.equal(ref, rec))}
    results["rollouts"].append(row)
    print(json.dumps(row), flush=True)
    for tile in (16384, 8192, 65536):
        t0 = time.perf_counter()
        for _ in range(2):
            pop.reset(); pop.fly(T, tile_drones=tile)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        row = {"rollout": f"tile_drones={tile}, one kernel per launch", "drones": n, "timesteps": T, "keep": keep, "ms_per_episode": ms,
               "drone_steps_per_s": n * T / ms * 1e3, "records_identical": bool(torch.equal(ref, pop.records))}
        results["rollouts"].append(row)
        print(json.dumps(row), flush=True)
lib.ld_debug_lds_keep(-1)
if out_path:
    with open(out_path, "w") as f:
        json.dump(results, f, indent=1)
