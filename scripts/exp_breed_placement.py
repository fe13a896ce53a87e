import json, sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import make_genomes
from learning_drones_amd import Population
from learning_drones_amd.params import GaParams, coin_chain_reference
n, hidden = 65536, (32, 32)
pop = Population(n, hidden, coins=coin_chain_reference(3), specialise=False)
pop.set_genomes(make_genomes(torch, n, hidden, pop.device, 0))
pop.fly(200); torch.cuda.synchronize()
fit = pop.records[:, 0].contiguous()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for place in (None, "parent"):
    ms = min(timed(lambda: pop.breed(fit, GaParams.reference(seed=3), place=place)) for _ in range(3))
    print(json.dumps({"place": place, "select_breed_ms": ms}), flush=True)
# the pieces of the placement
parents = pop.last_parents
keys64 = parents.long().clone()
print("sort int64 stable ms", min(timed(lambda: torch.sort(keys64, stable=True)) for _ in range(3)))
print("sort int64 ms", min(timed(lambda: torch.sort(keys64)) for _ in range(3)))
k32 = ((parents >> 6) << 17 | torch.arange(n, device=parents.device, dtype=torch.int32)).to(torch.int32)
print("sort int32 composite ms", min(timed(lambda: torch.sort(k32)) for _ in range(3)))
k16 = (parents >> 6).to(torch.int16)
print("argsort int16 block ms", min(timed(lambda: torch.argsort(k16, stable=True)) for _ in range(3)))
print("argsort int32 parents ms", min(timed(lambda: torch.argsort(parents)) for _ in range(3)))

# the kernel alone, with and without a placement, through the C ABI
import ctypes as C
from learning_drones_amd import _native
lib = _native.lib()
ga = GaParams.reference(seed=3)
n_layers = pop.spec.n_hidden + 1
parents = torch.empty((n,), dtype=torch.int32, device=pop.device)
cuts = torch.empty((n // 2, n_layers), dtype=torch.int32, device=pop.device)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
assert lib.ld_ga_select(C.byref(ga), C.byref(pop.spec), p(fit), n, 0, n, p(parents), p(cuts), st) == 0
children = torch.empty_like(pop.genomes)
def breed(order):
    assert lib.ld_ga_breed_f32(C.byref(ga), C.byref(pop.spec), p(pop.genomes), n, 1, p(parents), p(cuts), None, 0, n, p(order), p(children), st) == 0
print("select kernel ms", min(timed(lambda: lib.ld_ga_select(C.byref(ga), C.byref(pop.spec), p(fit), n, 0, n, p(parents), p(cuts), st)) for _ in range(3)))
print("breed kernel, no order ms", min(timed(lambda: breed(None)) for _ in range(3)))
order = torch.argsort(parents).to(torch.int32).contiguous()
print("breed kernel, children sorted by parent ms", min(timed(lambda: breed(order)) for _ in range(3)))
order_blk = torch.argsort((parents >> 6).to(torch.int16), stable=True).to(torch.int32).contiguous()
print("breed kernel, children sorted by parent block ms", min(timed(lambda: breed(order_blk)) for _ in range(3)))
