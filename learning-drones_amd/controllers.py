"""Hand-built controller genomes (pure numpy, no GPU needed).

The reference only ever creates *random* genomes (``evolve/brain.py:24``), and
random genomes tumble and crash within ~200 steps.  Parity tests and the
"stable flight" benchmark variant need genomes that actually fly to coins, so
this module builds a linear state-feedback law and encodes it *exactly* in the
reference's network format (``weights``: list of ``(n_out, n_in + 1)`` arrays,
bias in the last column, ReLU hidden layers, ``sigmoid(0.4 x)`` output;
``evolve/brain.py:41-54``).

A ReLU network can carry a linear signal v through any number of hidden layers
as the pair ``relu(v), relu(-v)`` (``v = relu(v) - relu(-v)``), so four linear
channels (roll, pitch, yaw, collective) need eight neurons per hidden layer.
"""
from __future__ import annotations

import math

import numpy as np

# input order of evolve/brain.py:65-73
_POS, _VEL, _ANG, _ANGVEL, _COIN = 0, 3, 6, 9, 12
N_INPUTS = 15

# motor sign patterns for a unit roll / pitch / yaw moment
# (drone/quadcopter/physical_model.py:107-109)
_ROLL = np.array([1.0, -1.0, -1.0, 1.0])
_PITCH = np.array([-1.0, -1.0, 1.0, 1.0])
_YAW = np.array([-1.0, 1.0, -1.0, 1.0])
_ALL = np.ones(4)


def hover_gains(pole_xy: float = 1.6, pole_z: float = 2.0, pole_yaw: float = 3.0,
                *, mass: float = 0.4, gravity: float = 9.81, c_f: float = 5.0,
                c_m: float = 0.5, r_m: float = 0.1,
                inertia=(2160e-6, 2160e-6, 4210e-6)) -> np.ndarray:
    """Return K (4 x 16): rows = roll, pitch, yaw, collective channel; columns =
    the 15 network inputs + constant.  Each channel is the thrust *fraction*
    offset applied with its motor sign pattern.

    x and y are chains of four integrators (x'' = g*pitch, pitch'' = M/Iy); all
    four poles are put at ``-pole_xy``.  z and yaw are double integrators.
    """
    K = np.zeros((4, N_INPUTS + 1))
    p = pole_xy
    a0, a1, a2, a3 = p ** 4, 4 * p ** 3, 6 * p ** 2, 4 * p
    # pitch channel: pitch'' = -(a3 q + a2 pitch + a1/g vx + a0/g (x - xc))
    per_acc_pitch = inertia[1] / (4.0 * r_m * c_f)          # fraction per rad/s^2
    K[1, _ANGVEL + 1] = -a3 * per_acc_pitch
    K[1, _ANG + 1] = -a2 * per_acc_pitch
    K[1, _VEL + 0] = -a1 / gravity * per_acc_pitch
    K[1, _POS + 0] = -a0 / gravity * per_acc_pitch
    K[1, _COIN + 0] = +a0 / gravity * per_acc_pitch
    # roll channel: y'' = -g*roll
    per_acc_roll = inertia[0] / (4.0 * r_m * c_f)
    K[0, _ANGVEL + 0] = -a3 * per_acc_roll
    K[0, _ANG + 0] = -a2 * per_acc_roll
    K[0, _VEL + 1] = +a1 / gravity * per_acc_roll
    K[0, _POS + 1] = +a0 / gravity * per_acc_roll
    K[0, _COIN + 1] = -a0 / gravity * per_acc_roll
    # yaw channel: hold yaw = 0
    per_acc_yaw = inertia[2] / (4.0 * math.sqrt(2.0) * r_m * c_m)
    K[2, _ANGVEL + 2] = -2.0 * pole_yaw * per_acc_yaw
    K[2, _ANG + 2] = -pole_yaw ** 2 * per_acc_yaw
    # collective channel: z'' = 4 c_f e / m
    per_acc_z = mass / (4.0 * c_f)
    K[3, _VEL + 2] = -2.0 * pole_z * per_acc_z
    K[3, _POS + 2] = -pole_z ** 2 * per_acc_z
    K[3, _COIN + 2] = +pole_z ** 2 * per_acc_z
    return K


def hover_net(hidden=(10, 10), *, beta: float = 0.4, gains: np.ndarray | None = None,
              mass: float = 0.4, gravity: float = 9.81, c_f: float = 5.0) -> list[np.ndarray]:
    """Weights list (reference format) of a network that flies to the coin.

    ``hidden``: hidden layer widths, each >= 8.  The thrust law is
    ``u = sigmoid(beta * (z_hover + slope^-1 * (pattern . K x)))`` i.e. the
    linear feedback of :func:`hover_gains` through the output sigmoid linearised
    at hover thrust.
    """
    hidden = tuple(int(h) for h in hidden)
    if not hidden or min(hidden) < 8:
        raise ValueError("hover_net needs >= 1 hidden layer of width >= 8")
    K = hover_gains(mass=mass, gravity=gravity, c_f=c_f) if gains is None else np.asarray(gains, float)
    u_h = mass * gravity / (4.0 * c_f)                      # hover thrust fraction
    z_h = math.log(u_h / (1.0 - u_h)) / beta
    slope = beta * u_h * (1.0 - u_h)                        # d sigmoid(beta z)/dz at hover

    weights = []
    n_in = N_INPUTS
    # first hidden layer: (relu(v_c), relu(-v_c)) for the four channels
    w = np.zeros((hidden[0], n_in + 1))
    for c in range(4):
        w[2 * c, :] = K[c]
        w[2 * c + 1, :] = -K[c]
    weights.append(w)
    n_in = hidden[0]
    for h in hidden[1:]:
        w = np.zeros((h, n_in + 1))
        for c in range(4):
            w[2 * c, 2 * c], w[2 * c, 2 * c + 1] = 1.0, -1.0
            w[2 * c + 1, 2 * c], w[2 * c + 1, 2 * c + 1] = -1.0, 1.0
        weights.append(w)
        n_in = h
    out = np.zeros((4, n_in + 1))
    for c, pattern in enumerate((_ROLL, _PITCH, _YAW, _ALL)):
        out[:, 2 * c] = pattern / slope
        out[:, 2 * c + 1] = -pattern / slope
    out[:, n_in] = z_h
    weights.append(out)
    return weights


def perturbed(weights, sigma: float, rng: np.random.Generator) -> list[np.ndarray]:
    """Copy of ``weights`` with N(0, sigma) noise on every gene (a cheap way to get a
    population of *different* genomes that mostly still fly)."""
    return [w + rng.normal(0.0, sigma, size=w.shape) for w in weights]
