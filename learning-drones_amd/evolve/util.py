"""GA operators: mirror of evolve/util.py:5-104 (same names, arguments, return values, and the same
sequence of draws from numpy's global random stream, so a seeded run selects the same parents and
cuts at the same genes as the reference — pinned by tests/golden/g6_ga.npz).

The operators themselves work on flat gene vectors (the reference's flatten() order), which is also
what the device-side genome layout uses; the Drone-object functions are thin wrappers."""
from __future__ import annotations

from copy import deepcopy

import numpy as np

from ..params import flatten_weights, spec_of_weights, unflatten_weights


def tournament_select(drones, idx1, idx2, p_tournament):
    """Binary tournament (util.py:30-45): with probability p the strictly better drone wins; a tie
    counts as 'second is not worse'."""
    r = np.random.random_sample()
    first_better = drones[idx1].reward > drones[idx2].reward
    return idx1 if ((r < p_tournament) == first_better) else idx2


def crossover_genes(genes1: np.ndarray, genes2: np.ndarray, layer_sizes, p_crossover):
    """Per layer, with probability p, a single-point swap of the two flattened layers at a uniform
    cut point (util.py:48-78).  Returns two new gene vectors."""
    a, b = genes1.copy(), genes2.copy()
    off = 0
    for size in layer_sizes:
        if np.random.random_sample() < p_crossover:
            cut = np.random.randint(0, size)
            a[off + cut:off + size], b[off + cut:off + size] = genes2[off + cut:off + size], genes1[off + cut:off + size]
        off += size
    return a, b


def crossover(weights1, weights2, p_crossover):
    spec = spec_of_weights(weights1)
    sizes = [r * c for r, c in spec.layer_shapes]
    a, b = crossover_genes(flatten_weights(weights1), flatten_weights(weights2), sizes, p_crossover)
    return unflatten_weights(a, spec), unflatten_weights(b, spec)


def selection_and_crossover(drones, p_tournament, p_crossover):
    """New population of the same (even) size: for each pair of slots draw four indices, run two
    tournaments, copy the winners and cross them (util.py:5-27)."""
    n = len(drones)
    if n % 2:
        raise ValueError("the population size must be even (the reference pairs slots i, i+1)")
    new_population = [None] * n
    for i in range(0, n, 2):
        idx = [np.random.randint(0, n) for _ in range(4)]
        first = tournament_select(drones, idx[0], idx[1], p_tournament)
        second = tournament_select(drones, idx[2], idx[3], p_tournament)
        new_population[i], new_population[i + 1] = deepcopy(drones[first]), deepcopy(drones[second])
        w1, w2 = crossover(new_population[i].controller.weights, new_population[i + 1].controller.weights, p_crossover)
        new_population[i].controller.weights, new_population[i + 1].controller.weights = w1, w2
    return new_population


def get_best_of_generation(scores):
    """(index, score) of the first maximum (util.py:81-90)."""
    best_index, best_score = -1, -np.inf
    for i, score in enumerate(scores):
        if score > best_score:
            best_index, best_score = i, score
    return best_index, best_score


def compare_weights(weights1, weights2):
    """Assert two genomes are identical (accepts Drones or weights lists; util.py:93-104)."""
    w1 = getattr(getattr(weights1, "controller", None), "weights", weights1)
    w2 = getattr(getattr(weights2, "controller", None), "weights", weights2)
    for i, (l1, l2) in enumerate(zip(w1, w2)):
        bad = np.argwhere(np.asarray(l1) != np.asarray(l2))
        assert len(bad) == 0, f"i: {i}, j: {bad[0][0]}, k: {bad[0][1]}, w1: {l1[tuple(bad[0])]}, w2: {l2[tuple(bad[0])]}"
