"""fly(): mirror of flight/__init__.py:12-99, plus fly_population() which evaluates a whole list of
Drones in one batched rollout (what evolve/evolution.py's loop over `fly(drone, idx=i)` becomes)."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from ..params import EpisodeParams, SimParams, controller_spec, flatten_weights


def _check_episode_args(delta_t, visualize, n_servers):
    if visualize or n_servers:
        raise NotImplementedError("visualisation / data-scraper servers are out of scope of learning-drones_amd")
    if abs(delta_t - 0.01) > 0:
        # in the reference delta_t only labels the returned time; the integrator step is drone.py:10
        pass


# A drone is any object with the reference's public attributes (drone/drone.py:32-76): position, velocity, angle,
# angle_vel (3,1), thrust (4,1), sensors (six objects with current_acceleration / last_acceleration / previous_pos /
# previous_vel, drone/sensor/sensor.py:25-31), reward, coins, flight_time, crashed, distance_to_coin and a controller
# with .weights.  Nothing below touches anything else, so the reference's own Drone class works as well as this
# package's.
def _state12(d) -> np.ndarray:
    return np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in (d.position, d.velocity, d.angle, d.angle_vel)])


def _acc6(d) -> np.ndarray:
    return np.array([float(s.current_acceleration) for s in d.sensors])


def _store(d, row22) -> None:
    d.position, d.velocity = row22[0:3].reshape(3, 1).copy(), row22[3:6].reshape(3, 1).copy()
    d.angle, d.angle_vel = row22[6:9].reshape(3, 1).copy(), row22[9:12].reshape(3, 1).copy()
    for i, s in enumerate(d.sensors):
        s.last_acceleration = s.current_acceleration
        s.current_acceleration = float(row22[12 + i])
        s.previous_pos = float(row22[i]) if i < 3 else float(row22[3 + i])
        s.previous_vel = float(row22[3 + i]) if i < 3 else float(row22[6 + i])
    d.thrust = row22[18:22].reshape(4, 1).copy()


def fly_population(drones: Sequence, timesteps: int = 2500, delta_t: float = 0.01, punish_time=False,
                   reward_time=True, precision: str | None = None, coins=None, walls=None):
    """Evaluate every drone of the list with ONE population rollout on the GPU and write the outcome
    back into the Drone objects (reward, coins, flight_time, crashed, distance_to_coin, position ...).
    Returns (scores, times): what the reference collects from fly(drone, idx=i) for i in range(n)."""
    from ..population import Population
    if not drones:
        return [], []
    precision = precision or getattr(drones[0], "precision", "f64")      # a foreign Drone: the reference-tracking kernels
    # any object with .weights is a controller (evolve/brain.py:10-24): the layer spec is read from the weights
    # when the object does not carry one
    spec = controller_spec(drones[0].controller)
    if any(controller_spec(d.controller).hidden_widths != spec.hidden_widths for d in drones[1:]):
        raise ValueError("fly_population needs one layer spec for the whole list (one genome array per population)")
    n = len(drones)
    pop = Population(n, spec.hidden_widths, precision=precision, coins=coins, walls=walls,
                     episode=EpisodeParams.reference(punish_time=punish_time, reward_time=reward_time),
                     sim=SimParams.reference())
    pop.set_genomes(np.stack([flatten_weights(d.controller.weights) for d in drones]))
    pop.set_state(np.stack([_state12(d) for d in drones]), np.stack([_acc6(d) for d in drones]),
                  np.stack([np.asarray(d.thrust, dtype=np.float64).ravel() for d in drones]))
    start = np.array([[d.reward, d.coins, 2 * d.flight_time + int(d.crashed)] for d in drones], np.int32)
    import torch
    pop.istate.copy_(torch.from_numpy(start.T.copy()))
    pop.fly(timesteps, engine="auto")
    res, state = pop.results(), pop.state_numpy()
    scores, times = [], []
    for i, d in enumerate(drones):
        _store(d, state[i])
        d.reward, d.coins, d.flight_time = int(res["reward"][i]), int(res["coins"][i]), int(res["flight_time"][i])
        d.crashed = bool(res["crashed"][i])
        d.distance_to_coin = float(res["distance_to_coin"][i])
        last_step = (d.flight_time - 1) if d.crashed else (timesteps - 1)      # flight/__init__.py:40,87,99
        scores.append(d.reward)
        times.append(last_step * delta_t)
    return scores, times


def fly(drone, timesteps: int = 2500, delta_t: float = 0.01, visualize=False, n_servers=0, punish_time=False,
        reward_time=True, idx=None):
    """One episode for one Drone.  Returns (reward, real_time, idx) like the reference."""
    _check_episode_args(delta_t, visualize, n_servers)
    scores, times = fly_population([drone], timesteps, delta_t, punish_time, reward_time)
    return scores[0], times[0], idx
