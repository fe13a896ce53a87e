"""Batched PID policy (BASELINE config 1 on the device): the cascaded PID controller of
drone/controller/pid.py + drone/onboard_computer.py flying the coin course of
flight/collect_coins.py:261-266, one drone per gain set."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native
from .params import F_COUNT, INITIAL_STATE, SimParams

# target coins of flight/collect_coins.py:261-266 (radius 1, value 1000 each)
COLLECT_COINS_COURSE = np.array([[0.0, 0.0, 5.0], [25.0, 25.0, 5.0], [-10.0, 35.0, 5.0], [0.0, 0.0, 5.0]])

# gain order of drone/quadcopter/choose_paramset.py:convert_paramset_2_float
GAIN_NAMES = ("kp_x", "kp_y", "kp_z", "ki_x", "ki_y", "ki_z", "kd_x", "kd_y", "kd_z",
              "kp_rol", "kp_pit", "kp_yaw", "ki_rol", "ki_pit", "ki_yaw", "kd_rol", "kd_pit", "kd_yaw")


def pid_episodes(gains, timesteps: int = 1000, *, coins=COLLECT_COINS_COURSE, coin_radius: float = 1.0,
                 coin_value: int = 1000, precision: str = "f64", sim: SimParams | None = None, trace: bool = False,
                 device=None):
    """Fly one PID-controlled episode per row of `gains` ([n, 18]).  Returns a dict with score,
    coins (collected), steps (executed), state ([n, 22]) and, if trace, trajectory [steps, n, 16]."""
    import torch
    _native.require_gpu()
    gains = np.asarray(gains, dtype=np.float64).reshape(-1, 18)
    n = len(gains)
    dtype = torch.float64 if precision == "f64" else torch.float32
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    sim = sim if sim is not None else SimParams.reference()
    g = torch.as_tensor(gains.T.copy()).to(dev, dtype).contiguous()
    c = torch.as_tensor(np.asarray(coins, np.float64).reshape(-1, 3)).to(dev).contiguous()
    fstate = torch.zeros((F_COUNT, n), dtype=dtype, device=dev)
    fstate[:12] = torch.as_tensor(np.asarray(INITIAL_STATE)).to(dev, dtype)[:, None]
    score, target, steps = (torch.zeros((n,), dtype=torch.int32, device=dev) for _ in range(3))
    tr = torch.zeros((timesteps, 16, n), dtype=dtype, device=dev) if trace else None
    p = lambda t: C.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())
    fn = getattr(_native.lib(), f"ld_pid_episode_{precision}")
    _native.check(fn(C.byref(sim), p(g), p(c), len(c), float(coin_radius), int(coin_value), int(timesteps), p(fstate),
                     p(score), p(target), p(steps), p(tr), n, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                  "ld_pid_episode")
    out = {"score": score.cpu().numpy(), "coins": target.cpu().numpy(), "steps": steps.cpu().numpy(),
           "state": fstate.t().double().cpu().numpy()}
    if trace:
        out["trace"] = tr.permute(0, 2, 1).double().cpu().numpy()
    return out
