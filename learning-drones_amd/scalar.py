"""One-drone populations backing the scalar drop-in objects (Drone, NeuralNetwork).  The scalar API
exists so that code written against the reference keeps working; anything performance-minded should
hold a Population and call fly() on it."""
from __future__ import annotations

from .params import MlpSpec

_cache = {}


def scalar_population(spec: MlpSpec, precision: str):
    """A cached 1-drone Population for this layer spec / precision (GPU required)."""
    from .population import Population
    key = (spec.hidden_widths, precision)
    if key not in _cache:
        _cache[key] = Population(1, spec.hidden_widths, precision=precision)
    return _cache[key]
