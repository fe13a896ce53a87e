"""NeuralNetwork: mirror of evolve/brain.py:10-76 (same constructor, attributes and methods); the
forward pass runs on the GPU through ld_mlp_forward_* (no CPU path)."""
from __future__ import annotations

import numpy as np
import numpy.random as npr

from ..drone.onboard_computer import ControlUnit
from ..params import MlpSpec


class NeuralNetwork(ControlUnit):
    def __init__(self, layer_spec={1: 8, 2: 10, 3: 4}, n_inputs=15, precision="f64"):
        """layer_spec: {layer_number: n_neurons}; weights[i] has shape (n_out, n_in + 1) with the bias
        in the last column, drawn N(0, 1/sqrt(n_out)) from numpy's global stream in the reference's
        order (brain.py:18-24), so np.random.seed reproduces the reference's genomes."""
        layer_spec = dict(layer_spec)
        layer_spec[0] = n_inputs
        self.layer_spec = layer_spec
        self.n_layers = len(layer_spec) - 1
        self.precision = precision
        self.weights = []
        for num in [k for k in layer_spec if k != 0]:
            n_out, n_in = layer_spec[num], layer_spec[num - 1]
            self.weights.append(npr.normal(0.0, 1.0 / np.sqrt(n_out), size=(n_out, n_in + 1)))

    @property
    def spec(self) -> MlpSpec:
        return MlpSpec.from_layer_spec(self.layer_spec, self.layer_spec[0])

    # brain.py:26-39: the activation helpers, for callers that use them directly (the GPU forward pass
    # applies the same two functions inside ld_mlp_forward_*)
    def _relu(self, x, thresh=0, alpha=1.0):
        return 0.0 if x < thresh else alpha * x

    def relu(self, x_vec, thresh=0, alpha=1.0):
        return np.array([[self._relu(x, thresh=thresh, alpha=alpha) for x in np.squeeze(x_vec)]]).T

    def sigmoid(self, x, beta=0.4):
        return 1 / (np.exp(-beta * x) + 1)

    def feed_forward(self, input_vec):
        x = np.asarray(input_vec, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != 1:
            raise AssertionError(x.shape)                    # brain.py:48 column-vector assert
        from ..scalar import scalar_engine
        return scalar_engine(self.spec, self.precision).feed_forward(self.weights, x.ravel()).reshape(-1, 1)

    def translate_input_to_thrust(self, drone_position, drone_velocity, drone_angle, drone_angle_velocity,
                                  coin_position):
        x = np.vstack((drone_position, drone_velocity, drone_angle, drone_angle_velocity, coin_position))
        return self.feed_forward(x)
