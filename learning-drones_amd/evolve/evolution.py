"""The GA driver: evolve/evolution.py:33-119 as a callable.  Same loop shape (evaluate -> argmax ->
keep global best -> selection_and_crossover -> reset + mutate -> elitism in slot 0) and the same
hyper-parameter names (evolution.py:16-20); the per-drone `fly(drone, idx=i)` calls become ONE
batched rollout per generation (flight.fly_population)."""
from __future__ import annotations

import json
from copy import deepcopy
from time import time
from typing import Callable, Optional

import numpy as np

from ..drone.drone import Drone
from ..flight import fly_population
from ..params import flatten_weights
from .util import selection_and_crossover

n_generations = 20
n_drones = 20
mutation_rate = 0.05
p_tournament = 0.6
p_crossover = 0.05


def save_drone(drone: Drone, path: str, **extra) -> None:
    """Checkpoint of one drone as plain JSON (the reference writes jsonpickle, evolution.py:85-87;
    jsonpickle is not a dependency here)."""
    doc = {"layer_spec": {str(k): int(v) for k, v in drone.controller.layer_spec.items() if k != 0},
           "genes": flatten_weights(drone.controller.weights).tolist(), "reward": int(drone.reward),
           "coins": int(drone.coins), "flight_time": int(drone.flight_time), "crashed": bool(drone.crashed),
           "distance_to_coin": float(drone.distance_to_coin), **extra}
    with open(path, "w") as f:
        json.dump(doc, f)


def _jsonpickle_array(node, state) -> np.ndarray:
    """One numpy array out of a jsonpickle document, data only (nothing is imported or called from the file).
    jsonpickle (1.4.1, the reference's pin, without its numpy extension, which the reference never registers)
    writes an ndarray through `ndarray.__reduce__`: {"py/reduce": [_reconstruct, args, STATE, ...]} with
    STATE = {"py/tuple": [version, {"py/tuple": shape}, dtype, fortran_order, {"py/b64": raw bytes}]} and dtype
    either a nested py/reduce whose first argument is the type string ("f8") or a {"py/id": n} back-reference to
    the dtype object of an earlier array.  The numpy-extension spelling ({"py/object": "numpy.ndarray",
    "values": [...]}) and plain nested lists are accepted too."""
    import base64
    if isinstance(node, list):
        return np.asarray(node, dtype=np.float64)
    if "values" in node:
        return np.asarray(node["values"], dtype=node.get("dtype", "float64")).astype(np.float64)
    st = node["py/reduce"][2]
    st = st["py/tuple"] if isinstance(st, dict) else st
    _, shape, dtype, fortran, data = st
    shape = tuple(shape["py/tuple"] if isinstance(shape, dict) else shape)
    if isinstance(dtype, dict) and "py/reduce" in dtype:
        args = dtype["py/reduce"][1]
        code = (args["py/tuple"] if isinstance(args, dict) else args)[0]
        order = dtype["py/reduce"][2]
        order = (order["py/tuple"] if isinstance(order, dict) else order)[1] if order else "<"
        state["dtype"] = np.dtype(code).newbyteorder(order if order in "<>" else "=")
    raw = base64.b64decode(data["py/b64"]) if isinstance(data, dict) else bytes(data)
    a = np.frombuffer(raw, dtype=state.get("dtype", np.dtype("<f8")))
    return a.reshape(shape, order="F" if fortran else "C").astype(np.float64)


def read_reference_checkpoint(doc):
    """(layer_spec, weights) from the reference's `best_drone_*.json` (evolution.py:85-87: a jsonpickle string of
    the whole Drone object, json.dump'ed once more).  Restated from jsonpickle's documented encoding; jsonpickle
    is not importable here and the reference ships no checkpoint file, so this reader is checked against a
    hand-written document of that shape only (tests/test_dropin_host.py) -- parity unpinned."""
    if isinstance(doc, str):
        doc = json.loads(doc)
    ctrl = doc["controller"]
    state = {}
    weights = [_jsonpickle_array(w, state) for w in ctrl["weights"]]
    spec = {int(k): int(v) for k, v in ctrl["layer_spec"].items() if not str(k).startswith("py/")}
    return spec, weights


def load_drone(path: str, precision: str = "f64") -> Drone:
    """A drone from a checkpoint: this package's plain-JSON layout (save_drone) or the reference's jsonpickle
    document (read data-only, see read_reference_checkpoint)."""
    from ..evolve.brain import NeuralNetwork
    from ..params import unflatten_weights
    with open(path) as f:
        doc = json.load(f)
    if isinstance(doc, str) or "py/object" in doc:
        spec, weights = read_reference_checkpoint(doc)
        d = Drone(visualize=False, n_servers=0, precision=precision)
        d.controller = NeuralNetwork(layer_spec={k: v for k, v in spec.items() if k != 0}, precision=precision)
        if [w.shape for w in weights] != [w.shape for w in d.controller.weights]:
            raise ValueError(f"{path}: weight shapes {[w.shape for w in weights]} do not match layer_spec {spec}")
        d.controller.weights = weights
        return d
    d = Drone(visualize=False, n_servers=0, precision=precision)
    d.controller = NeuralNetwork(layer_spec={int(k): v for k, v in doc["layer_spec"].items()}, precision=precision)
    d.controller.weights = unflatten_weights(np.asarray(doc["genes"]), d.controller.spec)
    return d


def evolve(n_generations: int = n_generations, n_drones: int = n_drones, mutation_rate: float = mutation_rate,
           p_tournament: float = p_tournament, p_crossover: float = p_crossover, timesteps: int = 2500,
           precision: str = "f32", checkpoint: Optional[str] = None, log: Callable = print, drones=None):
    """Run the GA; returns (global_best_drone, history) where history[g] = (best_score, global_best,
    scores, times)."""
    drones = drones if drones is not None else [Drone(visualize=False, n_servers=0, precision=precision)
                                                for _ in range(n_drones)]
    global_best_score, global_best_drone, history = -np.inf, None, []
    for gen in range(n_generations):
        t0 = time()
        scores, times = fly_population(drones, timesteps=timesteps, precision=precision)
        best_index = int(np.argmax(scores))
        best_score = scores[best_index]
        if best_score > global_best_score:
            global_best_score, global_best_drone = best_score, deepcopy(drones[best_index])
            if checkpoint:
                save_drone(global_best_drone, checkpoint, generation=gen)
        new_population = selection_and_crossover(drones, p_tournament, p_crossover)
        for drone in new_population:
            drone.reset()
            drone.mutate(mutation_rate)
        log(f"Generation {gen}: Best Score: {best_score}, Global Best: {global_best_score} "
            f"(coins {global_best_drone.coins}, crashed {global_best_drone.crashed}, "
            f"time in air {global_best_drone.flight_time}, distance {global_best_drone.distance_to_coin:.2f}) "
            f"[{time() - t0:.2f} s]")
        history.append((best_score, global_best_score, list(scores), list(times)))
        drones = new_population
        elite = deepcopy(global_best_drone)
        elite.reset()
        drones[0] = elite
    return global_best_drone, history


if __name__ == "__main__":
    # python -m learning_drones_amd.evolve.evolution  ==  python evolve/evolution.py of the reference
    # (same hyper-parameters, evolution.py:16-20), with checkpoints under ./evolution/ as JSON
    import os
    execution_time = int(time())
    os.makedirs("evolution", exist_ok=True)
    best, _ = evolve(checkpoint=f"evolution/best_drone_{execution_time}.json")
    save_drone(best, f"evolution/winning_drone_{execution_time}.json")
    print(f"To reference this training, use execution time {execution_time}")
