"""Host-side mirrors of the POD structs in include/ld_drones.h and the reference's constants.

Nothing here touches the GPU.  Values and their sources:
  * physics constants  -- drone/quadcopter/parameters.py:15-35, QuadcopterPhysics.__init__
                          (drone/quadcopter/physical_model.py:37-73)
  * reward constants   -- drone/drone.py:10-17
  * coin chain         -- flight/__init__.py:25,47-52
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Iterable, Sequence

import numpy as np

N_INPUTS = 15
N_OUTPUTS = 4
MAX_HIDDEN_LAYERS = 4
MAX_WIDTH = 64
MAX_COINS = 64
COIN_STRIDE = 5
MAX_WALLS = 16
WALL_STRIDE = 6
GENOME_BLOCK = 64        # drones per genome block (= wavefront width), ld_device.hpp kGenomeBlock

# fstate / istate field offsets (enum ld_ffield / ld_ifield)
F_POS, F_VEL, F_ANGLE, F_ANGVEL, F_ACC, F_THRUST, F_COUNT = 0, 3, 6, 9, 12, 18, 22
I_REWARD, I_COINS, I_STATUS, I_COUNT = 0, 1, 2, 3

# drone/quadcopter/parameters.py:47-52
INITIAL_STATE = (0.0, 0.0, 10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)


class SimParams(C.Structure):
    """struct ld_sim_params."""
    _fields_ = [
        ("mass_total", C.c_double), ("gravity_force", C.c_double), ("coef_force", C.c_double),
        ("coef_moment", C.c_double), ("coef_wind", C.c_double), ("radius_motor", C.c_double),
        ("inv_inertia", C.c_double * 3), ("moments_payload", C.c_double * 3),
        ("wind", C.c_double * 3), ("dt", C.c_double),
    ]

    @classmethod
    def reference(cls, *, mass_center=0.200, mass_motor=0.050, radius_motor_center=0.1,
                  I_x=2160e-6, I_y=2160e-6, I_z=4210e-6, coef_force=5, coef_moment=0.5,
                  coef_wind=0.1, gravity=9.81, mass_payload=0.0, x_payload=0.0, y_payload=0.0,
                  wind=(0.0, 0.0, 0.0), dt=0.01) -> "SimParams":
        """Same argument names as QuadcopterPhysics.__init__ (physical_model.py:37-52)."""
        p = cls()
        p.mass_total = mass_center + 4 * mass_motor + mass_payload        # physical_model.py:114
        G = gravity * (4 * mass_motor + mass_center)                      # :60
        mp = [0.0, 0.0, 0.0]
        if mass_payload != 0:                                             # :68-73
            force_payload = gravity * mass_payload
            G += force_payload
            mp = [x_payload * force_payload, y_payload * force_payload, 0.0]
        p.gravity_force = G
        p.coef_force, p.coef_moment, p.coef_wind = coef_force, coef_moment, coef_wind
        p.radius_motor = radius_motor_center
        p.inv_inertia[:] = [1 / I_x, 1 / I_y, 1 / I_z]                    # :63
        p.moments_payload[:] = mp
        p.wind[:] = [float(w) for w in np.asarray(wind, dtype=float).ravel()]
        p.dt = dt                                                         # drone/drone.py:10
        return p


class EpisodeParams(C.Structure):
    """struct ld_episode_params."""
    _fields_ = [
        ("reward_time", C.c_int32), ("reward_crashed", C.c_int32), ("reward_unstable", C.c_int32),
        ("reward_distance", C.c_int32), ("angle_limit", C.c_double), ("angle_vel_limit", C.c_double),
        ("z_floor", C.c_double), ("z_ceiling", C.c_double), ("n_coins", C.c_int32), ("n_walls", C.c_int32),
        ("coin_respawn", C.c_int32), ("reserved", C.c_int32),
    ]

    @classmethod
    def reference(cls, *, punish_time=False, reward_time=True, n_coins=1, n_walls=0, coin_respawn=None) -> "EpisodeParams":
        """drone/drone.py:12-17; punish_time / reward_time as in fly() (flight/__init__.py:41-44).
        coin_respawn=True continues the chain beyond the table by the reference's rule
        (flight/__init__.py:47-52: +(5, 2.5, 0), radius 1, value 1000), without bound; False ends the
        chain with the table's last row (BASELINE's "1 coin" / "3 coins" / "5 coins" configurations).  None (the
        default) = not stated: the field is set to the reference's behaviour (1) and `respawn_explicit` is False, so
        that Population can tell a stated choice from the default (its precedence: its own coin_respawn argument, then
        an episode block that states one, then "a coin table given explicitly is a finite chain")."""
        e = cls()
        e.reward_time = 2 * (int(bool(reward_time)) - int(bool(punish_time)))
        e.reward_crashed = -3000
        e.reward_unstable = -1
        e.reward_distance = -100
        e.angle_limit = math.pi / 4
        e.angle_vel_limit = 20.0
        e.z_floor, e.z_ceiling = 0.0, 80.0
        e.n_coins, e.n_walls = n_coins, n_walls
        e.coin_respawn, e.reserved = (1 if coin_respawn is None else int(bool(coin_respawn))), 0
        e.respawn_explicit = coin_respawn is not None
        return e


class MlpSpec(C.Structure):
    """struct ld_mlp_spec."""
    _fields_ = [("n_hidden", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN_LAYERS)]

    @classmethod
    def from_hidden(cls, hidden: Iterable[int]) -> "MlpSpec":
        hidden = [int(h) for h in hidden]
        if not 1 <= len(hidden) <= MAX_HIDDEN_LAYERS:
            raise ValueError(f"need 1..{MAX_HIDDEN_LAYERS} hidden layers, got {len(hidden)}")
        if any(h < 1 or h > MAX_WIDTH for h in hidden):
            raise ValueError(f"hidden widths must be 1..{MAX_WIDTH}, got {hidden}")
        s = cls()
        s.n_hidden = len(hidden)
        for i, h in enumerate(hidden):
            s.hidden[i] = h
        return s

    @classmethod
    def from_layer_spec(cls, layer_spec: dict, n_inputs: int = N_INPUTS) -> "MlpSpec":
        """From the reference's ``layer_spec`` dict (evolve/brain.py:10-24): keys 1..L map to
        neuron counts, the last one being the 4 thrust outputs."""
        if n_inputs != N_INPUTS:
            raise ValueError(f"the fused kernels take {N_INPUTS} inputs (pos, vel, angle, angle_vel, coin)")
        layers = [int(layer_spec[k]) for k in sorted(k for k in layer_spec if k != 0)]
        if not layers or layers[-1] != N_OUTPUTS:
            raise ValueError(f"last layer must have {N_OUTPUTS} neurons (rotor thrusts), got {layers}")
        return cls.from_hidden(layers[:-1])

    @property
    def hidden_widths(self) -> tuple:
        return tuple(self.hidden[i] for i in range(self.n_hidden))

    @property
    def layer_shapes(self) -> list:
        """[(n_out, n_in + 1), ...] -- shapes of NeuralNetwork.weights (evolve/brain.py:24)."""
        shapes, n_in = [], N_INPUTS
        for h in self.hidden_widths + (N_OUTPUTS,):
            shapes.append((h, n_in + 1))
            n_in = h
        return shapes

    @property
    def param_count(self) -> int:
        return sum(a * b for a, b in self.layer_shapes)

    @property
    def words_per_drone(self) -> int:
        return (self.param_count + 3) // 4

    def layer_spec(self) -> dict:
        d = {i + 1: h for i, h in enumerate(self.hidden_widths)}
        d[self.n_hidden + 1] = N_OUTPUTS
        return d


class GaParams(C.Structure):
    """struct ld_ga_params; defaults are the hyper-parameters of evolve/evolution.py:16-20 and the
    mutation width of drone/drone.py:172."""
    _fields_ = [("p_tournament", C.c_double), ("p_crossover", C.c_double), ("mutation_rate", C.c_double),
                ("mutation_sigma", C.c_double), ("seed", C.c_uint64), ("generation", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def reference(cls, seed: int = 0, generation: int = 0, p_tournament=0.6, p_crossover=0.05, mutation_rate=0.05,
                  mutation_sigma=0.3) -> "GaParams":
        return cls(p_tournament, p_crossover, mutation_rate, mutation_sigma, seed, generation, 0)


class FitnessRecord(C.Structure):
    """struct ld_fitness_record (16 bytes)."""
    _fields_ = [("reward", C.c_int32), ("flight_time", C.c_int32), ("coins_crashed", C.c_int32),
                ("distance_to_coin", C.c_float)]


FITNESS_DTYPE = np.dtype([("reward", "<i4"), ("flight_time", "<i4"), ("coins_crashed", "<i4"),
                          ("distance_to_coin", "<f4")])
CRASHED_BIT = 1 << 30


def coin_chain_reference(n_coins: int = MAX_COINS) -> np.ndarray:
    """Coin table [n][5] = x, y, z, radius, value by the rule of flight/__init__.py:25,47-52:
    first coin (5, 5, 10) r=2 worth 2000, each next one at +(5, 2.5, 0), r=1, worth 1000.
    The reference's chain is unbounded: with EpisodeParams.coin_respawn the kernels continue any table by
    that rule, so ONE row is the whole reference chain; longer tables are the finite chains of the
    BASELINE configurations (coin_respawn=False)."""
    if not 1 <= n_coins <= MAX_COINS:
        raise ValueError(f"n_coins must be 1..{MAX_COINS}")
    k = np.arange(n_coins, dtype=np.float64)
    table = np.empty((n_coins, COIN_STRIDE))
    table[:, 0] = 5.0 + 5.0 * k
    table[:, 1] = 5.0 + 2.5 * k
    table[:, 2] = 10.0
    table[:, 3] = np.where(k == 0, 2.0, 1.0)
    table[:, 4] = np.where(k == 0, 2000.0, 1000.0)
    return table


def coin_table(coins: Sequence) -> np.ndarray:
    """Coin table from objects with .position (3,1) / .radius / .value (objects/coins.py:5-15)
    or from rows (x, y, z, radius, value)."""
    rows = []
    for c in coins:
        if hasattr(c, "position"):
            p = np.asarray(c.position, dtype=float).ravel()
            rows.append([p[0], p[1], p[2], float(c.radius), float(c.value)])
        else:
            rows.append([float(v) for v in c])
    table = np.asarray(rows, dtype=np.float64).reshape(-1, COIN_STRIDE)
    if not 1 <= len(table) <= MAX_COINS:
        raise ValueError(f"need 1..{MAX_COINS} coins")
    return table


def flatten_weights(weights: Sequence[np.ndarray]) -> np.ndarray:
    """NeuralNetwork.weights -> gene vector: per layer the row-major (n_out, n_in+1) matrix
    (the order evolve/util.py:56-57 cuts in), layers concatenated."""
    return np.concatenate([np.asarray(w, dtype=np.float64).ravel() for w in weights])


def unflatten_weights(genes: np.ndarray, spec: MlpSpec) -> list:
    genes = np.asarray(genes, dtype=np.float64).ravel()
    if genes.size != spec.param_count:
        raise ValueError(f"expected {spec.param_count} genes, got {genes.size}")
    out, off = [], 0
    for a, b in spec.layer_shapes:
        out.append(genes[off:off + a * b].reshape(a, b).copy())
        off += a * b
    return out


def spec_of_weights(weights: Sequence[np.ndarray]) -> MlpSpec:
    """Infer the layer spec from a reference-format weights list."""
    shapes = [np.asarray(w).shape for w in weights]
    if len(shapes) < 2 or shapes[0][1] != N_INPUTS + 1 or shapes[-1][0] != N_OUTPUTS:
        raise ValueError(f"not a 15-input / 4-output controller: {shapes}")
    for (a, _), (_, d) in zip(shapes[:-1], shapes[1:]):
        if d != a + 1:
            raise ValueError(f"inconsistent layer shapes {shapes}")
    return MlpSpec.from_hidden([s[0] for s in shapes[:-1]])


def controller_spec(controller) -> MlpSpec:
    """The layer spec of a controller object.  The reference's contract for a controller is an object with a
    ``.weights`` list of (n_out, n_in + 1) arrays (evolve/brain.py:10-24; drone/drone.py:65,83-86): this package's
    NeuralNetwork also carries ``.spec``, anything else (the reference's own NeuralNetwork, a hand-made stand-in) is
    read through its weights."""
    spec = getattr(controller, "spec", None)
    if isinstance(spec, MlpSpec):
        return spec
    if not hasattr(controller, "weights"):
        raise TypeError(f"{type(controller).__name__} has no .weights: not a network controller (evolve/brain.py:10-24)")
    return spec_of_weights(controller.weights)

