"""One-drone engines backing the scalar drop-in objects (Drone, NeuralNetwork).  The scalar API exists so that code
written against the reference keeps working call by call; anything performance-minded should hold a Population and call
fly() on it.  What a call costs here is its host <-> device round trip, so the engine keeps everything resident:

  * one 1-drone Population per (layer spec, precision), its state and genome tensors living on the device;
  * pinned host staging buffers (numpy views of them are filled / read in place), so a call is asynchronous copies + one
    kernel launch + ONE stream synchronisation;
  * the genome is uploaded only when the controller's weights differ from what the device holds.

The reference's own loop body costs 105-120 us per timestep on a CPU core; scripts/time_scalar.py times
Drone.measure() + Drone.translate_input_to_thrust() through this module against that."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native
from .params import F_COUNT, I_COUNT, MlpSpec, flatten_weights

_cache = {}


class ScalarEngine:
    def __init__(self, spec: MlpSpec, precision: str):
        import torch
        from .population import Population
        self.torch = torch
        self.pop = Population(1, spec.hidden_widths, precision=precision)
        self.precision = precision
        dtype = self.pop.dtype
        pin = dict(pin_memory=True)
        self.h_f = torch.zeros(F_COUNT, dtype=dtype, **pin)            # pos3 vel3 angle3 angle_vel3 | acc6 | thrust4
        self.h_i = torch.zeros(I_COUNT, dtype=torch.int32, **pin)       # reward | coins | 2*flight_time + crashed
        self.h_x = torch.zeros(15, dtype=dtype, **pin)
        self.h_y = torch.zeros(4, dtype=dtype, **pin)
        self.f, self.i, self.x, self.y = self.h_f.numpy(), self.h_i.numpy(), self.h_x.numpy(), self.h_y.numpy()
        self.d_x = torch.zeros((15, 1), dtype=dtype, device=self.pop.device)
        self.d_y = torch.zeros((4, 1), dtype=dtype, device=self.pop.device)
        self.genes = None                                               # what the device genome holds (host copy)
        self._measure = getattr(_native.lib(), f"ld_measure_{precision}")
        self._forward = getattr(_native.lib(), f"ld_mlp_forward_{precision}")

    def _stream(self):
        return self.torch.cuda.current_stream(self.pop.device)

    def measure(self, sim, episode, state12, acc6, thrust4):
        """Drone.measure for one drone: returns (row22 float64, reward delta, crashed)."""
        pop, f = self.pop, self.f
        f[0:12], f[12:18], f[18:22] = state12, acc6, thrust4
        self.i[:] = 0
        stream = self._stream()
        pop.fstate.view(-1).copy_(self.h_f, non_blocking=True)
        pop.istate.view(-1).copy_(self.h_i, non_blocking=True)
        _native.check(self._measure(C.byref(sim), C.byref(episode), pop._ptr(pop.fstate), pop._ptr(pop.istate), 1,
                                    C.c_void_p(stream.cuda_stream)), "ld_measure")
        self.h_f.copy_(pop.fstate.view(-1), non_blocking=True)
        self.h_i.copy_(pop.istate.view(-1), non_blocking=True)
        stream.synchronize()
        return f.astype(np.float64), int(self.i[0]), bool(self.i[2] & 1)

    def feed_forward(self, weights, x15):
        """NeuralNetwork.feed_forward for one genome: x (15,) -> (4,) float64."""
        pop = self.pop
        genes = flatten_weights(weights)
        if self.genes is None or not np.array_equal(genes, self.genes):     # upload only what changed
            pop.set_genomes(genes[None, :])
            self.genes = genes
        self.x[:] = x15
        stream = self._stream()
        self.d_x.view(-1).copy_(self.h_x, non_blocking=True)
        _native.check(self._forward(C.byref(pop.spec), pop._ptr(pop.genomes), pop._ptr(self.d_x), pop._ptr(self.d_y), 1,
                                    C.c_void_p(stream.cuda_stream)), "ld_mlp_forward")
        self.h_y.copy_(self.d_y.view(-1), non_blocking=True)
        stream.synchronize()
        return self.y.astype(np.float64)


def scalar_engine(spec: MlpSpec, precision: str) -> ScalarEngine:
    """The cached engine for this layer spec / precision (GPU required)."""
    key = (spec.hidden_widths, precision)
    if key not in _cache:
        _cache[key] = ScalarEngine(spec, precision)
    return _cache[key]


def scalar_population(spec: MlpSpec, precision: str):
    """The engine's 1-drone Population (kept for callers that want the tensors)."""
    return scalar_engine(spec, precision).pop
