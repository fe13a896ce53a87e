"""Experiment: can part of a tile's genes be kept in the L2s?  A 16 384-drone tile of 32-32 is flown as two
concurrent launches on two streams: drones [0, R) by a build whose gene loads are cacheable, [R, n) by the usual
non-temporal build -- if the L2s keep the first group's genes while the second group's stream through, a quarter of
the re-streams leave the Infinity-Cache fabric.  Results: profiles/r02_l2_partition.json.  Needs two builds of the
32-32 specialisation: learning_drones_amd.build.build_specialisation((32, 32)) writes build/jit/ld_spec_32_32.hip; compile
it twice with build.FLAGS plus -DLD_GENE_NT=0 / -DLD_GENE_NT=1 into build/jit/libexp_32_32_{cached,nt}.so."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learning_drones_amd import Population
from learning_drones_amd.params import coin_chain_reference
from bench import make_genomes

jit = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "learning-drones_amd", "build", "jit")
libs = {t: C.CDLL(os.path.join(jit, f"libexp_32_32_{t}.so")) for t in ("cached", "nt")}
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
for l in libs.values():
    l.ld_spec_fly_steps_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp]
    l.ld_spec_fly_steps_f32.restype = C.c_int

n, K = 16384, 64
pop = Population(n, (32, 32), coins=coin_chain_reference(3))
pop.set_genomes(make_genomes(torch, n, (32, 32), pop.device, 0))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def launch(tag, begin, end, waves, stream):
    rc = libs[tag].ld_spec_fly_steps_f32(C.cast(C.byref(pop.sim), vp), C.cast(C.byref(pop.episode), vp), pop._ptr(pop.coins_dev), pop._ptr(pop.walls_dev),
                                         pop._ptr(pop.genomes), pop._ptr(pop.fstate), pop._ptr(pop.istate), n, begin, end, K, waves, stream.cuda_stream)
    assert rc == 0, rc

def timed(fn, reps=12):
    pop.reset(); torch.cuda.synchronize()
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

import itertools
def report(name, fn):
    ts = [timed(fn) for _ in range(3)]
    print(f"{name}: " + " ".join("%.1f" % t for t in ts) + " us", flush=True)

report("tile 16384: whole nt waves 4", lambda: launch("nt", 0, n, 4, s1))
report("tile 16384: whole nt waves 8", lambda: launch("nt", 0, n, 8, s1))
report("tile 16384: whole cached waves 4", lambda: launch("cached", 0, n, 4, s1))
for (ta, tb), R, wa, wb in itertools.product((("cached", "nt"), ("nt", "nt")), (3072, 4096, 6144, 8192), (8, 16), (8,)):
    def both():
        launch(ta, 0, R, wa, s1)
        launch(tb, R, n, wb, s2)
    report(f"tile 16384: {ta} [0,{R}) waves {wa} || {tb} rest waves {wb}", both)
