"""One-wavefront-per-block kernels (float64; float32 run-time layer specs): ms per episode at one timestep per launch
vs the default 64, on the bench population."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes

for precision, hidden, n, T in (("f64", (32, 32), 65536, 500), ("f64", (32, 32), 1024, 2000), ("f64", (64, 64, 64), 16384, 200),
                                ("f32", (12, 12), 65536, 2000), ("f32", (12, 12), 1024, 2000), ("f32", (40, 40, 40), 65536, 500)):
    pop = Population(n, hidden, precision=precision, coins=coin_chain_reference(3), specialise=False)
    g = make_genomes(torch, n, hidden, pop.device, 0)
    pop.set_genomes(g.double() if precision == "f64" else g)
    out = {}
    for k in (1, 64):
        ts = []
        for rep in range(3):
            pop.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            done = 0
            while done < T:
                pop.steps(min(k, T - done)); done += min(k, T - done)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out[k] = min(ts)
    print(precision, hidden, n, T, {k: round(v, 2) for k, v in out.items()}, "us/step", {k: round(v / T * 1e3, 2) for k, v in out.items()},
          "TB/s", {k: round(pop.bytes_per_drone_step * n * T / v / 1e9, 2) for k, v in out.items()}, flush=True)
