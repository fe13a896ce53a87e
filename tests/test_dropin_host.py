"""Host-side drop-in parity (no GPU): the mirrors of Drone / NeuralNetwork / evolve.util draw from
numpy's global random stream in the reference's order, so seeded runs reproduce the reference's
genomes, selections, cuts and mutations bit for bit (golden G3 / G6)."""
import json
import numpy as np
import pytest

from learning_drones_amd.drone.drone import Drone
from learning_drones_amd.evolve import util as ga
from learning_drones_amd.evolve.brain import NeuralNetwork
from learning_drones_amd.objects.coins import Coin
from learning_drones_amd.params import MlpSpec, flatten_weights, unflatten_weights


def test_seeded_drone_genomes_equal_reference(g3, g6):
    for seed in range(32):
        np.random.seed(seed)
        d = Drone(visualize=False, n_servers=0)
        assert np.array_equal(d.genes, g3[f"weights_{seed}"]), seed
        assert d.position.shape == (3, 1) and d.thrust.shape == (4, 1) and d.reward == 0 and not d.crashed
        assert len(d.sensors) == 6 and d.sensors[2].previous_pos == 10.0
    np.random.seed(50)
    nn = NeuralNetwork(layer_spec={1: 32, 2: 32, 3: 4})
    assert np.array_equal(flatten_weights(nn.weights), g6["init_seed50_32x32"])
    assert nn.n_layers == 3 and nn.layer_spec[0] == 15


def _population(g6):
    spec = MlpSpec.from_hidden(g6["hidden"])
    drones = []
    for genes, reward in zip(g6["pop_weights"], g6["rewards"]):
        d = Drone(visualize=False, n_servers=0)
        d.controller.weights = unflatten_weights(genes, spec)
        d.reward = int(reward)
        drones.append(d)
    return drones


def test_tournament_select(g6):
    drones = _population(g6)
    np.random.seed(12)
    got = [ga.tournament_select(drones, int(a), int(b), 0.6) for a, b in g6["tour_idx"]]
    assert np.array_equal(got, g6["tour_sel"])


def test_crossover(g6):
    drones = _population(g6)
    for k in range(3):
        np.random.seed(20 + k)
        w1, w2 = ga.crossover(drones[0].controller.weights, drones[1].controller.weights, float(g6[f"cross_p_{k}"]))
        assert np.array_equal(flatten_weights(w1), g6[f"cross_a_{k}"])
        assert np.array_equal(flatten_weights(w2), g6[f"cross_b_{k}"])
        assert [w.shape for w in w1] == [(10, 16), (10, 11), (4, 11)]


def test_selection_and_crossover(g6):
    drones = _population(g6)
    for k in range(2):
        pt, pc = g6[f"sel_params_{k}"]
        np.random.seed(30 + k)
        new = ga.selection_and_crossover(drones, pt, pc)
        got = np.stack([flatten_weights(d.controller.weights) for d in new])
        assert np.array_equal(got, g6[f"sel_weights_{k}"])
    assert np.array_equal(np.stack([d.genes for d in drones]), g6["pop_weights"])      # parents untouched
    with pytest.raises(ValueError):
        ga.selection_and_crossover(drones[:5], 0.6, 0.05)


def test_mutate(g6):
    drones = _population(g6)
    for k in range(2):
        d = Drone(visualize=False, n_servers=0)
        d.controller.weights = [w.copy() for w in drones[2].controller.weights]
        np.random.seed(40 + k)
        d.mutate(float(g6[f"mut_rate_{k}"]))
        assert np.array_equal(d.genes, g6[f"mut_weights_{k}"])


def test_best_and_compare():
    assert ga.get_best_of_generation([3, 9, 9, -1]) == (1, 9)
    a = [np.ones((2, 3)), np.zeros((1, 3))]
    ga.compare_weights(a, [w.copy() for w in a])
    with pytest.raises(AssertionError):
        ga.compare_weights(a, [a[0], np.ones((1, 3))])


def test_coin():
    c = Coin(np.array([[5.0], [5.0], [10.0]]), 2, 2000)
    assert (c.x, c.y, c.z, c.radius, c.value) == (5.0, 5.0, 10.0, 2, 2000)
    assert c.is_in_vicinity(np.array([[5.0], [5.0], [11.999]])) and not c.is_in_vicinity(np.array([[5.0], [5.0], [12.0]]))
    assert c.distance_drone_to_coin(np.array([[2.0], [1.0], [10.0]])) == 5.0
    with pytest.raises(AssertionError):
        Coin(np.zeros(3), 1)
    assert c.row() == [5.0, 5.0, 10.0, 2.0, 2000.0]


def test_reset_restores_reference_defaults():
    d = Drone(visualize=False, n_servers=0, initial_pos=np.array([[1.0], [2.0], [3.0]]))
    assert d.position[2, 0] == 3.0
    d.reward, d.crashed, d.coins, d.flight_time = -5, True, 2, 9
    d.reset()
    assert (d.reward, d.crashed, d.coins, d.flight_time, d.distance_to_coin) == (0, False, 0, 0, np.inf)
    assert np.array_equal(d.position.ravel(), [0, 0, 10]) and np.array_equal(d.thrust.ravel(), [0, 0, 0, 0])


def _jsonpickle_ndarray(a, first):
    """An ndarray the way jsonpickle 1.4.1 writes it without its numpy extension (through ndarray.__reduce__);
    every array after the first refers to the first one's dtype object by py/id."""
    import base64
    dtype = ({"py/reduce": [{"py/type": "numpy.dtype"}, {"py/tuple": ["f8", 0, 1]},
                            {"py/tuple": [3, "<", None, None, None, -1, -1, 0]}, None, None]} if first else {"py/id": 7})
    return {"py/reduce": [{"py/function": "numpy.core.multiarray._reconstruct"},
                          {"py/tuple": [{"py/type": "numpy.ndarray"}, {"py/tuple": [0]}, {"py/b64": "Yg=="}]},
                          {"py/tuple": [1, {"py/tuple": list(a.shape)}, dtype, False,
                                        {"py/b64": base64.b64encode(np.ascontiguousarray(a, dtype="<f8").tobytes()).decode()}]},
                          None, None]}


def test_reference_checkpoint_reader(tmp_path):
    """evolution.py:85-87 writes json.dump(jsonpickle.encode(drone)): read data-only (parity unpinned: the document
    is hand-written to jsonpickle's documented shape; jsonpickle itself is not importable here)."""
    from learning_drones_amd.evolve.evolution import read_reference_checkpoint
    rng = np.random.default_rng(5)
    weights = [rng.normal(size=(8, 16)), rng.normal(size=(10, 9)), rng.normal(size=(4, 11))]
    doc = {"py/object": "drone.drone.Drone", "visualize": False, "reward": 1234,
           "controller": {"py/object": "evolve.brain.NeuralNetwork", "n_layers": 3,
                          "layer_spec": {"1": 8, "2": 10, "3": 4, "0": 15},
                          "weights": [_jsonpickle_ndarray(w, i == 0) for i, w in enumerate(weights)]}}
    path = tmp_path / "best_drone_ref.json"
    path.write_text(json.dumps(json.dumps(doc)))                     # the reference dumps the jsonpickle STRING
    spec, got = read_reference_checkpoint(json.loads(path.read_text()))
    assert spec == {1: 8, 2: 10, 3: 4, 0: 15}
    assert all(np.array_equal(a, b) for a, b in zip(got, weights))
    # the numpy-extension spelling and plain lists
    doc["controller"]["weights"] = [{"py/object": "numpy.ndarray", "dtype": "float64", "values": w.tolist()} for w in weights]
    assert all(np.array_equal(a, b) for a, b in zip(read_reference_checkpoint(doc)[1], weights))
    doc["controller"]["weights"] = [w.tolist() for w in weights]
    assert all(np.array_equal(a, b) for a, b in zip(read_reference_checkpoint(json.dumps(doc))[1], weights))
