"""Experiment 2 on the LDS-kept genes: which rollout geometry of config 3 (and of the other BASELINE shard sizes)
is fastest once the 8-wavefront workgroups keep 136 KiB of genes in LDS -- with and without the cached tile head on
the side stream, by tile size and workgroups per block.

    python scripts/exp_lds_keep2.py [out.json]
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


def episode_ms(pop, T, reps=3, **kw):
    pop.reset(); pop.fly(T, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pop.reset(); pop.fly(T, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run(n, T, hidden, coins, label, **kw):
    pop = Population(n, hidden, coins=coin_chain_reference(coins), specialise=False)
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    ref = None
    for keep in (0, 1):
        lib.ld_debug_lds_keep(keep)
        for name, args in kw.items():
            ms = min(episode_ms(pop, T, **args) for _ in range(2))
            rec = pop.records.clone()
            if ref is None:
                ref = rec
            row = {"workload": label, "geometry": name, "keep": keep, "ms_per_episode": ms, "drone_steps_per_s": n * T / ms * 1e3,
                   "records_identical": bool(torch.equal(ref, rec))}
            rows.append(row)
            print(json.dumps(row), flush=True)
    lib.ld_debug_lds_keep(-1)
    del pop


run(65536, 2000, (32, 32), 3, "config 3",
    default={}, tile16k_w8=dict(tile_drones=16384, waves_per_block=8), tile16k_w16=dict(tile_drones=16384, waves_per_block=16),
    tile8k_w16=dict(tile_drones=8192, waves_per_block=16), tile32k_w8=dict(tile_drones=32768, waves_per_block=8),
    tile12k_w8=dict(tile_drones=12288, waves_per_block=8), tile4k_w32=dict(tile_drones=4096, waves_per_block=32),
    whole_w8=dict(tile_drones=65536, waves_per_block=8))
run(8192, 2000, (32, 32), 3, "config 3 shard of 8",
    default={}, w8=dict(waves_per_block=8), w16=dict(waves_per_block=16), w32=dict(waves_per_block=32),
    tile4k_w32=dict(tile_drones=4096, waves_per_block=32))
run(32768, 4000, (32, 32), 3, "config 4 shard",
    default={}, tile16k_w8=dict(tile_drones=16384, waves_per_block=8), whole_w8=dict(tile_drones=32768, waves_per_block=8))
run(1024, 2000, (16, 16), 1, "config 2",
    default={}, w8=dict(waves_per_block=8), w16=dict(waves_per_block=16), w32=dict(waves_per_block=32), w4=dict(waves_per_block=4))
run(65536, 1000, (16, 16), 3, "16-16 x 65 536",
    default={}, whole_w8=dict(tile_drones=65536, waves_per_block=8), whole_w4=dict(tile_drones=65536, waves_per_block=4))
run(65536, 1000, (24, 24), 3, "24-24 x 65 536",
    default={}, whole_w8=dict(tile_drones=65536, waves_per_block=8), tile16k_w8=dict(tile_drones=16384, waves_per_block=8))
run(65536, 1000, (64,), 3, "64 x 65 536",
    default={}, whole_w8=dict(tile_drones=65536, waves_per_block=8), tile16k_w8=dict(tile_drones=16384, waves_per_block=8))
if out_path:
    with open(out_path, "w") as f:
        json.dump(rows, f, indent=1)
