"""Step-kernel timing of the float64 reference-tracking kernels (validation mode).  usage: time_f64.py N hidden"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from learning_drones_amd.population import Population
from learning_drones_amd.params import coin_chain_reference
n = int(sys.argv[1]); hidden = tuple(int(h) for h in sys.argv[2].split(','))
for prec in ("f32", "f64"):
    pop = Population(n, hidden, precision=prec, coins=coin_chain_reference(3))
    pop.set_genomes(bench.make_genomes(torch, n, hidden, pop.device, 0))
    for _ in range(20): pop.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): pop.step()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"{prec} n={n} {hidden}: {us:8.2f} us per step  {pop.bytes_per_drone_step * n / us / 1e3:8.1f} GB/s algorithmic")
    del pop
