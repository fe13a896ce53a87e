"""Diagnostic: how float32 closed-loop trajectories of the GPU kernel, the float32 oracle and the
float64 oracle diverge per genome class (chaos in tumbling drones); the numbers behind the L4 rung
of tests/test_gpu_parity.py.  Run on a GPU box: python scripts/diag_f32.py"""
import sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import torch
from learning_drones_amd.population import Population
from learning_drones_amd.params import *
from oracle import oracle as O
from test_gpu_parity import _mixed_genomes
for hidden, n, T in (((16,16),1024,2000), ((32,32),1000,600)):
    spec, genes, cls = _mixed_genomes(hidden, n, seed=n)
    pop = Population(n, hidden, precision="f32"); pop.set_genomes(genes); pop.fly(T); got = pop.results()
    ref = O.fly(pop.sim, pop.episode, spec, genes, T, n_threads=O.max_threads(), mode=O.CLOSED_F32)
    ref64 = O.fly(pop.sim, pop.episode, spec, genes, T, n_threads=O.max_threads(), mode=O.FAITHFUL_F64)
    for name, m in (("stable", cls == 0), ("marginal", cls == 1), ("random", cls == 2)):
        same = got["reward"][m] == ref.reward[m]
        same64 = got["reward"][m] == ref64.reward[m]
        cpu32_vs_64 = ref.reward[m] == ref64.reward[m]
        dft = np.abs(got["flight_time"][m] - ref.flight_time[m])
        drw = np.abs(got["reward"][m] - ref.reward[m])
        print(hidden, name, "gpu32==cpu32 %.3f gpu32==cpu64 %.3f cpu32==cpu64 %.3f" % (same.mean(), same64.mean(), cpu32_vs_64.mean()),
              "flight_time |d| median %.0f p90 %.0f max %.0f" % (np.median(dft), np.percentile(dft, 90), dft.max()),
              "reward |d| median %.0f p90 %.0f" % (np.median(drw), np.percentile(drw, 90)),
              "crashed gpu %.3f cpu %.3f" % (got["crashed"][m].mean(), (ref.crashed[m] != 0).mean()),
              "mean reward gpu %.1f cpu32 %.1f cpu64 %.1f" % (got["reward"][m].mean(), ref.reward[m].mean(), ref64.reward[m].mean()))
    # per-step divergence for a few random drones: first step where |state diff| > 1e-3
    k = 64
    popk = Population(k, hidden, precision="f32"); popk.set_genomes(genes[:k])
    tr = O.fly(pop.sim, pop.episode, spec, genes[:k], 400, mode=O.CLOSED_F32, trace=True)
    first = np.full(k, 400)
    for t in range(400):
        popk.step()
        s = popk.state_numpy()[:, :12]
        d = np.max(np.abs(s - tr.trace_state[t, :, :12]) / np.maximum(np.abs(tr.trace_state[t, :, :12]), 1e-2), axis=1)
        upd = (d > 1e-3) & (first == 400)
        first[upd] = t
    print("first step with rel diff > 1e-3: stable", first[0::3][:10], "marginal", first[1::3][:10], "random", first[2::3][:10])
