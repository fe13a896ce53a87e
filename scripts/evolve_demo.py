"""End-to-end demo: the reference's GA (evolve/evolution.py defaults: p_tournament 0.6, p_crossover
0.05, mutation 0.05 x N(0, 0.3), elitism) on a large population, every stage on the GPU.
usage: evolve_demo.py [n_drones] [hidden csv] [generations] [timesteps]"""
import sys, time
sys.path.insert(0, '.')
import torch
from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
hidden = tuple(int(h) for h in (sys.argv[2] if len(sys.argv) > 2 else "32,32").split(','))
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 30
T = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
pop = ShardedPopulation(n, hidden)
pop.local.init_genomes_reference(seed=0)                  # N(0, 1/sqrt(n_out)) like evolve/brain.py:24
evo = DeviceEvolution(pop, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for g in range(gens):
    tg = time.perf_counter()
    best, gbest, rec = evo.step(T)
    torch.cuda.synchronize()
    ft = rec[:, 1].float()
    coins = (rec[:, 2] & 0x3fffffff).float()
    print(f"gen {g:3d}  best {best:7d}  global best {gbest:7d}  mean flight time {ft.mean().item():7.1f}  "
          f"max {int(ft.max().item()):5d}  mean coins {coins.mean().item():.3f}  crashed {((rec[:, 2] >> 30) & 1).float().mean().item():.3f}  "
          f"[{(time.perf_counter() - tg) * 1e3:6.1f} ms]", flush=True)
print(f"{gens} generations of {n} drones x {T} steps: {time.perf_counter() - t0:.2f} s")
