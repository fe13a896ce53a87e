"""A/B timing of ld_fly_step_f32 across builds of libld_drones.so, interleaved in one process
(cdna_hip_programming.md rule 24).  usage: ab_step.py N hidden(csv) lib1.so lib2.so ..."""
import ctypes as C, sys, statistics
sys.path.insert(0, '.')
import numpy as np, torch
from learning_drones_amd.params import *
import bench

alt = '--alt' in sys.argv
if alt: sys.argv.remove('--alt')
n = int(sys.argv[1]); hidden = tuple(int(h) for h in sys.argv[2].split(',')); libs = sys.argv[3:]
spec = MlpSpec.from_hidden(hidden); P, W = spec.param_count, spec.words_per_drone
dev = torch.device('cuda:0')
sim, ep = SimParams.reference(), EpisodeParams.reference(n_coins=3, coin_respawn=False)
coins = torch.from_numpy(coin_chain_reference(3)).to(dev)
rows = bench.make_genomes(torch, n, hidden, dev, 0)
fstate = torch.zeros((22, n), dtype=torch.float32, device=dev); istate = torch.zeros((3, n), dtype=torch.int32, device=dev)
genomes = torch.zeros(((n + 63) // 64, W, 64, 4), dtype=torch.float32, device=dev)
genomes2 = torch.zeros_like(genomes) if alt else genomes
vp = C.c_void_p
import os
handles = {path.split('/')[-1]: C.CDLL(path.split(':')[0]) for path in libs}
first = next(iter(handles.values()))
stream = vp(torch.cuda.current_stream().cuda_stream)
assert first.ld_genomes_pack_f32(vp(rows.data_ptr()), C.c_int64(P), vp(genomes.data_ptr()), C.c_int64(n), C.c_int64(P), stream) == 0
if alt: genomes2.copy_(genomes)
def reset():
    assert first.ld_population_reset_f32(vp(fstate.data_ptr()), vp(istate.data_ptr()), C.c_int64(n), None, stream) == 0
flip = [0]
def step(h):
    flip[0] ^= 1
    g = genomes2 if (alt and flip[0]) else genomes
    rc = h.ld_fly_step_f32(C.byref(sim), C.byref(ep), C.byref(spec), vp(coins.data_ptr()), None, vp(g.data_ptr()),
                           vp(fstate.data_ptr()), vp(istate.data_ptr()), C.c_int64(n), stream)
    assert rc == 0, rc
res = {k: [] for k in handles}
L = 200
for rnd in range(7):
    for name, h in handles.items():
        h.ld_debug_force_generic(1 if name.endswith(':generic') else 0)
        reset()
        for _ in range(10): step(h)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(L): step(h)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / L * 1e3)
B = (4 * P + 200) * n
for name, v in res.items():
    med, mn = statistics.median(v), min(v)
    print(f"{name:16s} median {med:8.2f} us  min {mn:8.2f} us   {B/med/1e3:8.1f} GB/s (median)  {B/mn/1e3:8.1f} GB/s (best)")
