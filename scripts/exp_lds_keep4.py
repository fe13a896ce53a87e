"""Experiment 4 on the LDS-kept genes: launches of K timesteps by drones per launch x geometry, for the library named by
LD_DRONES_LIB (a variant build, e.g. with another ring depth) or the shipped one.

    [LD_DRONES_LIB=variants/libld_d8.so] python scripts/exp_lds_keep4.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import make_genomes
from learning_drones_amd import Population, _native
from learning_drones_amd.params import coin_chain_reference

lib = _native.lib()
hidden = (32, 32)
tag = os.path.basename(os.environ.get("LD_DRONES_LIB", "shipped"))


def timed(fn, reps=8):
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
for n, cases in ((16384, ((8, 64),)), (12288, ((11, 64),)), (10240, ((11, 64),)), (8192, ((16, 64), (11, 64))), (6144, ((16, 64),)),
                 (4096, ((32, 64), (16, 64))), (2048, ((32, 64),)), (24576, ((11, 64),)), (65536, ((11, 64), (8, 64), (16, 64)))):
    pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
    pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
    for waves, K in cases:
        pop.reset()
        us = min(timed(lambda: pop.steps(K, waves)) for _ in range(3))
        print(json.dumps({"lib": tag, "drones": n, "waves_per_block": waves, "timesteps_per_launch": K, "us_per_launch": us,
                          "us_per_timestep": us / K, "drones_per_us": n / (us / K)}), flush=True)
    del pop
