"""CPU-only checks: the C ABI library loads and exports what include/ld_drones.h declares, host-side
parameter logic, and the drop-in boundary fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from learning_drones_amd import _native, params
from learning_drones_amd.params import EpisodeParams, MlpSpec, SimParams

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from learning_drones_amd import build
    build.build(verbose=False)
    return _native.lib()


def test_header_and_library_agree(lib):
    header = (REPO / "include" / "ld_drones.h").read_text()
    declared = set(re.findall(r"\b(ld_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_native.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ld_version() == _native.ABI_VERSION == 3
    assert b"spec" in lib.ld_strerror(-2)


def test_host_side_entry_points(lib):
    sim = SimParams()
    assert lib.ld_sim_params_default(C.byref(sim)) == 0
    ref = SimParams.reference()
    for f, _ in SimParams._fields_:
        a, b = getattr(sim, f), getattr(ref, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), f
    assert sim.gravity_force == 9.81 * (4 * 0.05 + 0.2) and sim.mass_total == 0.4
    ep = EpisodeParams()
    assert lib.ld_episode_params_default(C.byref(ep)) == 0
    ref = EpisodeParams.reference(n_coins=0)
    for f, _ in EpisodeParams._fields_:
        assert getattr(ep, f) == getattr(ref, f), f
    table = np.zeros((7, 5))
    assert lib.ld_coin_chain_reference(table.ctypes.data_as(C.POINTER(C.c_double)), 7) == 0
    assert np.array_equal(table, params.coin_chain_reference(7))
    assert lib.ld_coin_chain_reference(table.ctypes.data_as(C.POINTER(C.c_double)), 65) == -3
    # P(10,10)=314, P(16,16)=596, P(32,32)=1700, P(64,64,64)=9604
    for hidden, P in (((10, 10), 314), ((16, 16), 596), ((32, 32), 1700), ((64, 64, 64), 9604)):
        spec = MlpSpec.from_hidden(hidden)
        assert lib.ld_mlp_param_count(C.byref(spec)) == P == spec.param_count
        assert lib.ld_mlp_words_per_drone(C.byref(spec)) == (P + 3) // 4 == spec.words_per_drone
    bad = MlpSpec.from_hidden((16,))
    bad.n_hidden = 0
    assert lib.ld_mlp_param_count(C.byref(bad)) == -2
    assert lib.ld_mlp_param_count(None) == -1


def test_struct_layouts_match_header():
    assert C.sizeof(SimParams) == 16 * 8
    assert C.sizeof(EpisodeParams) == 4 * 4 + 4 * 8 + 4 * 4
    assert C.sizeof(MlpSpec) == 5 * 4
    assert C.sizeof(params.FitnessRecord) == 16 == params.FITNESS_DTYPE.itemsize


def test_layer_spec_and_weight_flattening():
    spec = MlpSpec.from_layer_spec({1: 10, 2: 10, 3: 4})          # drone/drone.py:65
    assert spec.hidden_widths == (10, 10) and spec.layer_shapes == [(10, 16), (10, 11), (4, 11)]
    assert spec.layer_spec() == {1: 10, 2: 10, 3: 4}
    with pytest.raises(ValueError):
        MlpSpec.from_layer_spec({1: 10, 2: 3})                    # last layer must be the 4 thrusts
    with pytest.raises(ValueError):
        MlpSpec.from_hidden((65,))
    rng = np.random.default_rng(0)
    w = [rng.normal(size=s) for s in spec.layer_shapes]
    flat = params.flatten_weights(w)
    assert flat[16 + 3] == w[0][1, 3] and flat[160 + 11 * 2 + 10] == w[1][2, 10]      # row-major, bias last
    back = params.unflatten_weights(flat, spec)
    assert all(np.array_equal(a, b) for a, b in zip(w, back))
    assert params.spec_of_weights(w).hidden_widths == (10, 10)


def test_no_cpu_fallback(monkeypatch, tmp_path):
    """Without the built library the product raises; it never routes to the oracle."""
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setenv("LD_DRONES_LIB", str(tmp_path / "missing.so"))
    with pytest.raises(_native.NativeLibraryMissing):
        _native.lib()
    monkeypatch.delenv("LD_DRONES_LIB")
    monkeypatch.setattr(_native, "_lib", None)
    import torch
    if not torch.cuda.is_available():
        from learning_drones_amd.population import Population
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            Population(4, (10, 10))
    src = "".join(p.read_text() for p in (REPO / "learning-drones_amd").rglob("*.py"))
    assert "oracle" not in src.replace("oracle's", "").replace("the oracle", "")   # product never imports oracle/


def test_hover_net_is_exact_linear_feedback(oracle):
    """The hand-built controller encodes u = sigmoid(0.4 (z_h + K x / slope)) exactly."""
    from learning_drones_amd import controllers
    K = controllers.hover_gains()
    for hidden in ((10, 10), (32, 32), (16,), (64, 64, 64)):
        w = controllers.hover_net(hidden)
        spec = params.spec_of_weights(w)
        rng = np.random.default_rng(1)
        x = rng.normal(0, 2, (16, 15))
        y = oracle.mlp_forward(spec, params.flatten_weights(w), x, mode=oracle.CLOSED_F64)
        u_h = 0.4 * 9.81 / 20
        z_h = np.log(u_h / (1 - u_h)) / 0.4
        slope = 0.4 * u_h * (1 - u_h)
        pat = np.array([[1, -1, -1, 1], [-1, -1, 1, 1], [-1, 1, -1, 1], [1, 1, 1, 1]], float)
        v = np.c_[x, np.ones(16)] @ K.T
        want = 1 / (1 + np.exp(-0.4 * (z_h + (v @ pat) / slope)))
        assert np.allclose(y, want, rtol=0, atol=1e-12)


def test_argument_validation_without_gpu(lib):
    """Every entry point validates its arguments before it touches HIP, so the error behaviour of
    the boundary can be checked on a CPU-only machine (no kernel is launched here: each call either
    fails validation or is the n == 0 no-op)."""
    from learning_drones_amd.params import GaParams
    sim, ep, ga = SimParams.reference(), EpisodeParams.reference(n_coins=3), GaParams.reference()
    spec, bad = MlpSpec.from_hidden((16, 16)), MlpSpec.from_hidden((16, 16))
    bad.hidden[1] = 0
    vp = C.c_void_p
    fake = vp(0x1000)                      # non-null, 16-byte aligned, never dereferenced
    odd = vp(0x1004)
    OK, NULL, SPEC, SIZE, ALIGN = 0, -1, -2, -3, -4
    for sfx in ("f32", "f64"):
        step = getattr(lib, f"ld_fly_step_{sfx}")
        assert step(None, C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, 8, None) == NULL
        assert step(C.byref(sim), C.byref(ep), C.byref(bad), fake, None, fake, fake, fake, 8, None) == SPEC
        assert step(C.byref(sim), None, C.byref(spec), fake, None, fake, fake, fake, 8, None) == NULL
        assert step(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, -1, None) == SIZE
        assert step(C.byref(sim), C.byref(ep), C.byref(spec), None, None, None, None, None, 0, None) == OK      # empty population
        assert step(C.byref(sim), C.byref(ep), C.byref(spec), None, None, fake, fake, fake, 8, None) == NULL    # no coin table
        assert step(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, odd, fake, fake, 8, None) == ALIGN
        many = EpisodeParams.reference(n_coins=65)
        assert step(C.byref(sim), C.byref(many), C.byref(spec), fake, None, fake, fake, fake, 8, None) == SIZE
        walled = EpisodeParams.reference(n_coins=3, n_walls=2)
        assert step(C.byref(sim), C.byref(walled), C.byref(spec), fake, None, fake, fake, fake, 8, None) == NULL  # walls missing
        roll = getattr(lib, f"ld_fly_rollout_{sfx}")
        args = [C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, fake]
        args += [fake] if sfx == "f64" else []
        assert roll(*args, 8, -5, None) == SIZE
        assert getattr(lib, f"ld_measure_{sfx}")(C.byref(sim), C.byref(ep), None, None, 0, None) == OK
        assert getattr(lib, f"ld_measure_{sfx}")(C.byref(sim), C.byref(ep), None, None, 4, None) == NULL
        assert getattr(lib, f"ld_mlp_forward_{sfx}")(C.byref(bad), fake, fake, fake, 4, None) == SPEC
        assert getattr(lib, f"ld_mlp_forward_{sfx}")(C.byref(spec), odd, fake, fake, 4, None) == ALIGN
        assert getattr(lib, f"ld_population_reset_{sfx}")(None, None, 0, None, None) == OK
        assert getattr(lib, f"ld_population_reset_{sfx}")(None, None, 3, None, None) == NULL
        assert getattr(lib, f"ld_genomes_pack_{sfx}")(fake, 10, fake, 4, 20, None) == SIZE                       # row stride < P
        assert getattr(lib, f"ld_genomes_unpack_{sfx}")(odd, fake, 20, 4, 20, None) == ALIGN
        breed = getattr(lib, f"ld_ga_breed_{sfx}")
        assert breed(C.byref(ga), C.byref(spec), fake, 8, 1, fake, fake, None, 0, 7, None, vp(0x2000), None) == SIZE   # odd children
        assert breed(C.byref(ga), C.byref(spec), fake, 8, 1, fake, fake, None, 0, 8, None, fake, None) == SIZE         # dst aliases src
        assert breed(C.byref(ga), C.byref(spec), fake, 8, 1, fake, fake, None, 0, 0, None, fake, None) == OK
        assert breed(None, C.byref(spec), fake, 8, 1, fake, fake, None, 0, 8, None, vp(0x2000), None) == NULL
        pid = getattr(lib, f"ld_pid_episode_{sfx}")
        assert pid(C.byref(sim), fake, fake, 0, 1.0, 1000, 100, fake, fake, fake, fake, None, 4, None) == SIZE   # no coins
        assert pid(C.byref(sim), None, None, 4, 1.0, 1000, 100, None, None, None, None, None, 0, None) == OK
        assert pid(C.byref(sim), None, fake, 4, 1.0, 1000, 100, fake, fake, fake, fake, None, 4, None) == NULL
    assert lib.ld_ga_select(C.byref(ga), C.byref(spec), fake, 7, 0, 6, fake, fake, None) == SIZE                # odd population
    assert lib.ld_ga_select(C.byref(ga), C.byref(spec), fake, 8, 4, 6, fake, fake, None) == SIZE                # children overrun
    assert lib.ld_ga_select(C.byref(ga), C.byref(spec), fake, 8, 0, 0, None, None, None) == OK
    assert lib.ld_ga_select(C.byref(ga), C.byref(spec), None, 8, 0, 8, fake, fake, None) == NULL
    steps = lib.ld_fly_steps_f32                                 # multi-wavefront / multi-step fused launch
    assert steps(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, 8, -1, 0, None) == SIZE
    assert steps(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, 8, 4, 3, None) == SIZE      # waves not 0/1/2/4/8/11/16/32
    assert steps(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, 0, 4, 8, None) == OK
    assert steps(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, odd, fake, fake, 8, 4, 8, None) == ALIGN
    assert steps(C.byref(sim), C.byref(ep), C.byref(MlpSpec.from_hidden((12,))), fake, None, fake, fake, fake, 8, 4, 4, None) == SPEC   # run-time spec: one wavefront per block only
    ro = lib.ld_fly_rollout_steps_f32
    assert ro(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, fake, 8, 10, -1, 0, 0, None) == SIZE
    assert ro(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, fake, 8, 10, 0, 0, 100, None) == SIZE   # tile not a multiple of 64
    assert ro(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, fake, 0, 10, 0, 0, 0, None) == OK
    k, w, m, tile = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    geo = lambda h, n, T: (lib.ld_fly_rollout_geometry_f32(C.byref(MlpSpec.from_hidden(h)), n, T, C.byref(k), C.byref(w), C.byref(m),
                                                            C.byref(tile)), k.value, w.value, m.value, tile.value)
    # nets of up to 512 gene words keep their genes on chip for the launch: no tiles, 512 timesteps per launch (the genes are
    # loaded once per launch), workgroups of 8 wavefronts x 64 drones (fly_wide_kernel) where every gene word of 64 drones fits
    # on chip -- all the shipped specs of up to 512 words do
    assert geo((32, 32), 65536, 2000) == (OK, 512, 8, 4, 65536) and geo((32, 32), 8192, 2000) == (OK, 512, 8, 4, 8192)
    assert geo((32, 32), 32768, 4000) == (OK, 512, 8, 8, 32768)           # config 4 shard
    assert geo((32, 32), 16384, 2000) == (OK, 512, 8, 4, 16384) and geo((32, 32), 1024, 2000) == (OK, 512, 8, 4, 1024)
    assert geo((16, 16), 65536, 2000) == (OK, 512, 8, 4, 65536) and geo((16, 16), 1024, 2000) == (OK, 512, 8, 4, 1024)   # config 2
    assert geo((64,), 8192, 100) == (OK, 512, 8, 1, 8192) and geo((64,), 65536, 100) == (OK, 512, 8, 1, 65536) and geo((24, 24), 8192, 100)[2] == 8
    in_lds, in_regs, streamed = C.c_int64(), C.c_int64(), C.c_int64()
    words = lambda h, w, steps: (lib.ld_fly_steps_kept_words_f32(C.byref(MlpSpec.from_hidden(h)), w, steps, C.byref(in_lds), C.byref(in_regs),
                                                                 C.byref(streamed)), in_lds.value, in_regs.value, streamed.value)
    # (gene words per drone kept in LDS, held in registers, loaded every timestep): 32-32 has 428 (three words straddle two
    # wavefronts' rows and are loaded by both)
    assert words((32, 32), 11, 64)[1:] == (176, 96, 156)
    # workgroups of 64 drones (fly_wide_kernel): 38 words per wavefront in registers, the rest in LDS, nothing loaded per timestep
    assert words((32, 32), 8, 64)[1:] == (124, 304, 0) and words((64,), 8, 64)[1:] == (44, 280, 0) and words((24, 24), 8, 64)[1:] == (2, 278, 0)
    assert words((16, 16), 8, 64)[1:] == (0, 156, 0) and words((10, 10), 8, 64)[1:] == (0, 88, 0)
    # ... and the run-time restatement of that plan, which the policy asks for specs specialised at run time, agrees with the kernels' own
    wide_lds, wide_regs = C.c_int64(), C.c_int64()
    for h in ((32, 32), (16, 16), (64,), (24, 24), (32,), (16,), (10, 10), (8, 8), (8, 10)):
        assert lib.ld_debug_wide_plan(C.byref(MlpSpec.from_hidden(h)), C.byref(wide_lds), C.byref(wide_regs)) == 1
        assert words(h, 8, 64)[1:] == (wide_lds.value, wide_regs.value, 0), h
    for h in ((48, 48), (64, 64), (32, 32, 32), (64, 64, 64)):              # more gene words than a CU holds for 64 drones
        assert lib.ld_debug_wide_plan(C.byref(MlpSpec.from_hidden(h)), C.byref(wide_lds), C.byref(wide_regs)) == 0
    assert lib.ld_debug_wide_plan(None, None, None) == NULL
    # workgroups of 32 drones pair the halves of a wavefront (ld_pair.hpp): a stream position is two gene words, 32 positions per
    # wavefront stay in registers, LDS keeps the rest: everything on chip (436 = 425 + the words two rows' blocks share)
    assert words((32, 32), 16, 64)[1:] == (12, 424, 0) and words((16, 16), 16, 64)[1:] == (0, 164, 0) and words((64,), 16, 64)[1:] == (100, 224, 0)
    assert words((24, 24), 16, 64)[1:] == (268, 12, 0)                      # 24 rows: no pairing, the unpaired kernel's plan
    assert lib.ld_debug_lds_keep(2) == -1 and words((32, 32), 16, 64)[1:] == (268, 112, 48) and words((32, 32), 8, 64)[1:] == (132, 96, 200)   # (the kernels of before, same geometries)
    assert geo((32, 32), 65536, 2000) == (OK, 512, 16, 4, 65536) and lib.ld_debug_lds_keep(-1) == 2
    assert words((32, 32), 32, 64)[1:] == (428, 0, 0)                       # everything on chip
    assert words((32, 32), 16, 4)[1:] == (0, 0, 425) and words((32, 32), 4, 64)[1:] == (0, 0, 425)   # short launches and other geometries keep nothing
    assert words((64, 64, 64), 8, 64)[1:] == (0, 0, 2401) and words((32, 32), 3, 64)[0] == SIZE
    # ... and without the kept genes (ld_debug_lds_keep(0)): tiles of ~112 MB of genes that the Infinity Cache holds
    assert lib.ld_debug_lds_keep(0) == -1
    assert geo((32, 32), 65536, 2000) == (OK, 64, 4, 128, 16384) and geo((32, 32), 32768, 4000) == (OK, 64, 4, 126, 16384)
    assert words((32, 32), 16, 64)[1:] == (0, 0, 425)
    assert lib.ld_debug_lds_keep(-1) == 0
    assert geo((64, 64, 64), 131072, 4000) == (OK, 64, 8, 63, 131072)     # 38 KB of genes per drone: a tile could not fill the chip; two workgroups of 4
    assert geo((64, 64, 64), 8192, 100) == (OK, 64, 8, 2, 8192) and geo((12,), 1000, 50) == (OK, 64, 1, 1, 1000)   # run-time spec: one wavefront per block, 64 timesteps per launch
    assert geo((10, 10), 65536, 2000) == (OK, 512, 8, 4, 65536)           # the reference's own controller: 79 gene words, all of them kept on chip
    assert geo((10, 10), 1024, 2000)[2] == 8 and geo((8, 8), 1024, 2000)[2] == 8
    assert geo((10, 10), 1048576, 100)[4] == 1048576                      # launches that keep their genes are not tiled
    assert lib.ld_debug_lds_keep(0) == -1
    assert geo((10, 10), 65536, 2000) == (OK, 64, 2, 32, 65536)           # nothing kept: 82 MB fits the Infinity Cache as it is; 10 rows split evenly over 2 wavefronts only
    assert geo((10, 10), 1024, 2000)[2] == 2 and geo((8, 8), 1024, 2000)[2] == 4 and geo((10, 10), 1048576, 100)[4] == 65536
    assert lib.ld_debug_lds_keep(-1) == 0
    assert lib.ld_fly_rollout_geometry_f32(C.byref(spec), 8, 8, None, None, None, None) == NULL
    for sfx in ("f32", "f64"):
        acc = getattr(lib, f"ld_accelerations_{sfx}")
        assert acc(C.byref(sim), fake, fake, fake, None, 4, None) == NULL and acc(C.byref(sim), None, None, None, None, -1, None) == SIZE
        frames = getattr(lib, f"ld_accelerations_frames_{sfx}")
        assert frames(C.byref(sim), fake, fake, fake, None, None, 0, 4, None) == NULL and frames(C.byref(sim), None, None, None, None, None, 1, 0, None) == OK
    assert lib.ld_debug_force_generic(1) == 0 and lib.ld_debug_force_generic(0) == 1
    res = lib.ld_fly_rollout_resident_f32
    unsupported = MlpSpec.from_hidden((12,))                 # no unrolled / resident specialisation
    assert res(C.byref(sim), C.byref(ep), C.byref(unsupported), fake, None, fake, fake, fake, fake, 8, 10, fake, None) == SPEC
    assert res(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, fake, 8, 10, None, None) == NULL     # no queue word
    assert res(C.byref(sim), C.byref(ep), C.byref(spec), None, None, None, None, None, None, 0, 10, None, None) == OK
    assert lib.ld_genome_words(C.byref(spec), 65) == 2 * 64 * 149 and lib.ld_genome_words(C.byref(spec), 0) == 0
    assert lib.ld_genome_words(C.byref(spec), -1) == SIZE
    for code in (0, -1, -2, -3, -4, -5, -6, -99):
        assert lib.ld_strerror(code)


def test_runtime_specialisation_builds_and_registers(lib):
    """learning_drones_amd.specialise(): hipcc builds the unrolled float32 kernels for one more layer spec
    (cross-compiles without a GPU), the object exports the two entry points, and the library's registry takes
    them once; specs with a built-in specialisation, NULL entry points and duplicates are refused."""
    from learning_drones_amd import build as ld_build
    spec = MlpSpec.from_hidden((13, 7))
    assert lib.ld_has_specialisation_f32(C.byref(spec)) == 0
    assert lib.ld_has_specialisation_f32(C.byref(MlpSpec.from_hidden((32, 32)))) == 1
    so_path = ld_build.build_specialisation((13, 7))
    assert so_path.exists() and ld_build.build_specialisation((13, 7)) == so_path            # cached
    so = C.CDLL(str(so_path))
    fly, mlp = C.cast(so.ld_spec_fly_steps_f32, C.c_void_p), C.cast(so.ld_spec_mlp_forward_f32, C.c_void_p)
    res = C.cast(so.ld_spec_fly_resident_f32, C.c_void_p)
    assert lib.ld_register_specialisation_f32(C.byref(spec), None, mlp, res) == -1
    assert lib.ld_register_specialisation_f32(C.byref(MlpSpec.from_hidden((32, 32))), fly, mlp, res) == -3      # ships unrolled
    assert lib.ld_has_resident_f32(C.byref(spec)) == 0 and lib.ld_has_resident_f32(C.byref(MlpSpec.from_hidden((32, 32)))) == 1
    assert lib.ld_has_resident_f64(C.byref(MlpSpec.from_hidden((64, 64, 64)))) == 0 and lib.ld_has_resident_f64(C.byref(MlpSpec.from_hidden((10, 10)))) == 1
    assert _native.specialise((13, 7)) is True and lib.ld_has_specialisation_f32(C.byref(spec)) == 1
    assert lib.ld_register_specialisation_f32(C.byref(spec), fly, mlp, None) == -3                           # already there
    assert lib.ld_has_resident_f32(C.byref(spec)) == 1                   # the specialisation brought a resident kernel
    assert _native.specialise((13, 7)) is True and _native.specialise((16, 16)) is True                       # no-ops
    k, w, m = C.c_int32(), C.c_int32(), C.c_int32()
    tile = C.c_int64()
    assert lib.ld_fly_rollout_geometry_f32(C.byref(spec), 8192, 2000, C.byref(k), C.byref(w), C.byref(m), C.byref(tile)) == 0
    assert (k.value, w.value, m.value, tile.value) == (512, 8, 4, 8192)                    # the specialised spec takes the multi-step, gene-keeping geometry (a small net: workgroups of 64 drones, fly_wide_kernel)
    wide_lds, wide_regs = C.c_int64(), C.c_int64()
    assert lib.ld_debug_wide_plan(C.byref(spec), C.byref(wide_lds), C.byref(wide_regs)) == 1 and (wide_lds.value, wide_regs.value) == (0, 92)
    assert lib.ld_debug_force_generic(1) == 0 and lib.ld_has_specialisation_f32(C.byref(spec)) == 0
    assert lib.ld_debug_force_generic(0) == 1
    # the entry points validate like the built-in ones (n == 0 is a no-op, no kernel is launched without a GPU)
    sim, ep = SimParams.reference(), EpisodeParams.reference(n_coins=1)
    fake = C.c_void_p(0x1000)
    assert lib.ld_fly_steps_f32(C.byref(sim), C.byref(ep), C.byref(spec), fake, None, fake, fake, fake, 0, 4, 0, None) == 0


def test_no_fused_step_kernel_runs_out_of_scratch():
    """scripts/audit_scratch.py over the built library: no float32 multi-wavefront step kernel keeps more than a
    handful of bytes per lane in scratch.  (hipcc once turned the narrow layer specs' kernels into ones that held the
    whole gene ring there, 3-4x slower, without a word at build time; see build.py, NO_SINK.)"""
    import subprocess, sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(repo / "scripts" / "audit_scratch.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "fly_split_kernel<float, false, 4, 1, 0, 0, 32, 32>" in out.stdout and "fly_split_kernel<float, false, 8, 2, 0, 0, 32, 32>" in out.stdout
    # the variants that keep 136 KiB of a workgroup's genes in LDS (workgroups of 64, 32, 48 and 16 drones): none of
    # those the launch policy picks uses scratch
    for spec in ("32, 32", "16, 16", "24, 24", "16", "32", "64", "10, 10", "8, 8", "8, 10"):
        for geometry in ("8, 1", "8, 2", "8, 3", "8, 4"):
            line = [l for l in out.stdout.splitlines() if re.match(rf"fly_split_kernel<float, false, {geometry}, 136, \d+, {spec}>\s", l)]
            assert len(line) == 1 and line[0].endswith("scratch 0"), (spec, geometry, line)
    for spec in ("32, 32", "16, 16", "16", "32", "64"):                          # ... and the kernels that pair the halves of a wavefront
        line = [l for l in out.stdout.splitlines() if re.match(rf"fly_pair_kernel<136, \d+, {spec}>\s", l)]
        assert len(line) == 1 and line[0].endswith("scratch 0"), (spec, line)
    assert "fly_resident_kernel" in out.stdout


def test_build_flags_per_translation_unit():
    """The multi-wavefront step kernels of layer specs with a hidden layer narrower than 16 are built without SimplifyCFG's
    common-code sinking (build.py, NO_SINK); every other translation unit with the plain flags."""
    from pathlib import Path
    from learning_drones_amd import build as B
    no_sink = "-simplifycfg-sink-common=false"
    for name, want in (("ld_split_10_10.hip", True), ("ld_split_8_8.hip", True), ("ld_split_8_10.hip", True), ("ld_split_16.hip", False),
                       ("ld_split_32_32.hip", False), ("ld_split_64_64_64_W8.hip", False), ("ld_resident.hip", False), ("ld_abi.hip", False),
                       # the LDS-keeping variants live in translation units of their own, always built without the sinking
                       ("ld_split_keep_32_32.hip", True), ("ld_split_keep_16.hip", True), ("ld_split_keep_64.hip", True)):
        assert (no_sink in B.flags_for(Path(name))) is want, name
    assert B.keeps_in_lds((32, 32)) and B.keeps_in_lds((12, 12)) and not B.keeps_in_lds((48, 48)) and not B.keeps_in_lds((64, 64, 64))
    # ... one per shipped layer spec of up to 512 gene words, and the main unit of each of those specs calls it
    for src in B.SOURCES:
        if src.name.startswith("ld_split_keep_"):
            main = src.with_name(src.name.replace("ld_split_keep_", "ld_split_"))
            assert "LD_SPLIT_KEEP_TU" in src.read_text() and "LD_SPLIT_HAS_KEEP" in main.read_text(), src.name
    assert B.narrow((12, 12)) and B.narrow((13, 7)) and not B.narrow((20, 20)) and not B.narrow((16,))
    assert "-ffp-contract=off" in B.FLAGS and "-fno-slp-vectorize" in B.FLAGS
