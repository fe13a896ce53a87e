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
    assert lib.ld_version() == 1
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
    assert C.sizeof(EpisodeParams) == 4 * 4 + 4 * 8 + 2 * 4
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
