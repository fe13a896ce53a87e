"""Soak test: every engine and launch geometry against the run-time-spec step kernel on randomly drawn
populations (sizes, layer specs, episode lengths, coin tables with and without the respawn rule, wall
boxes, wind, initial states): the default rollout (64 timesteps per launch, wavefronts per block chosen
from n), a random (wavefronts per block, timesteps per launch) pair, and the work-queue resident engine.
Any difference in state, bookkeeping or records is an error.  usage: soak_engines.py [rounds]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from learning_drones_amd.population import Population
from learning_drones_amd import _native
from learning_drones_amd.params import SimParams, coin_chain_reference
from test_gpu_parity import _mixed_genomes

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2024)
bad = 0
for r in range(rounds):
    hidden = Population.RESIDENT_SPECS[rng.integers(len(Population.RESIDENT_SPECS))]
    n = int(rng.choice([1, 5, 8, 63, 64, 65, 200, 1000, 4097]))
    T = int(rng.integers(1, 1500))
    coins = coin_chain_reference(int(rng.integers(1, 65)))
    walls = None
    if rng.random() < 0.5:
        lo = rng.uniform(-10, 30, (int(rng.integers(1, 17)), 3)); lo[:, 2] = rng.uniform(0, 60, len(lo))
        walls = np.concatenate([lo, lo + rng.uniform(0.5, 4, lo.shape)], axis=1)
    init = np.array([rng.normal(0, 2), rng.normal(0, 2), rng.uniform(5, 40), *rng.normal(0, 1, 3), *rng.normal(0, 0.2, 3), *rng.normal(0, 1, 3)])
    spec, genes, cls = _mixed_genomes(hidden, max(n, 3), seed=1000 + r)
    genes = genes[:n]
    respawn = bool(rng.random() < 0.5)
    sim = SimParams.reference(wind=rng.normal(0, 0.5, 3)) if rng.random() < 0.3 else None
    waves, k = int(rng.choice([1, 2, 4, 8, 16])), int(rng.choice([1, 3, 16, 50, T]))
    tile = int(rng.choice([64, 128, 192, 1024, 4096]))
    out = []
    f64 = hidden in Population.RESIDENT_SPECS_F64 and rng.random() < 0.25
    for kw in ("generic", dict(), dict(steps_per_launch=k, waves_per_block=waves, tile_drones=tile), dict(engine="resident"),
               dict(engine="compacting", steps_per_launch=int(rng.choice([0, 8, 50, 300])))):
        pop = Population(n, hidden, coins=coins, walls=walls, init_state=init, coin_respawn=respawn, sim=sim)
        pop.set_genomes(genes)
        if kw == "generic":
            _native.lib().ld_debug_force_generic(1)
            for _ in range(T):
                pop.step()
            pop.finalize()
            _native.lib().ld_debug_force_generic(0)
        else:
            pop.fly(T, **kw)
        out.append((pop.fstate.clone(), pop.istate.clone(), pop.records.clone()))
    same = all(torch.equal(a, b) for other in out[1:] for a, b in zip(out[0], other))
    if f64:                                   # the float64 twins: resident engine against one launch per timestep
        o64 = []
        for engine in ("step", "resident"):
            pop = Population(n, hidden, precision="f64", coins=coins, walls=walls, init_state=init, coin_respawn=respawn, sim=sim)
            pop.set_genomes(genes)
            pop.fly(min(T, 400), engine=engine)
            o64.append((pop.fstate.clone(), pop.istate.clone(), pop.records.clone(), pop.distance.clone()))
        same = same and all(torch.equal(a, b) for a, b in zip(*o64))
    crashed = float((out[0][1][2] & 1).float().mean())
    print(f"round {r:3d} spec {str(hidden):14s} n={n:5d} T={T:5d} coins={len(coins):2d}{'+' if respawn else ' '} L={waves} K={k:4d} tile={tile:4d} walls={0 if walls is None else len(walls):2d} crashed {crashed:.2f}  {'ok' if same else 'MISMATCH'}", flush=True)
    bad += not same
print("mismatches:", bad)
sys.exit(1 if bad else 0)
