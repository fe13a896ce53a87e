import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from learning_drones_amd.population import Population
from learning_drones_amd.params import coin_chain_reference
n = int(sys.argv[1]); hidden = tuple(int(h) for h in sys.argv[2].split(',')); T = int(sys.argv[3])
pop = Population(n, hidden, coins=coin_chain_reference(3))
pop.set_genomes(bench.make_genomes(torch, n, hidden, pop.device, 0))
for engine in ("step", "resident", "step", "resident"):
    pop.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    pop.fly(T, engine=engine); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = pop.results()
    print(f"{engine:9s} n={n} {hidden} T={T}: {dt*1e3:9.2f} ms  {n*T/dt/1e9:8.3f} G drone-steps/s   alive {1-r['crashed'].mean():.3f} coins {r['coins'].mean():.2f}")
