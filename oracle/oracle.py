"""ctypes binding of oracle/libld_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.  The oracle is pinned against golden vectors generated from the
imported reference (tests/golden/make_golden.py; tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from learning_drones_amd.params import (COIN_STRIDE, EpisodeParams, MlpSpec, SimParams,  # noqa: E402
                                        WALL_STRIDE, coin_chain_reference)

FAITHFUL_F64, CLOSED_F64, CLOSED_F32 = 0, 1, 2
LIB_PATH = HERE / "libld_oracle.so"

_lib = None


def build(force: bool = False) -> Path:
    srcs = [HERE / "ld_oracle.c", HERE / "ld_oracle_generic.inc", HERE / "ld_oracle.h",
            HERE.parent / "include" / "ld_drones.h"]
    if force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(HERE), "-B", "libld_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.ldo_max_threads.restype = C.c_int
    return _lib


def _p(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}")


def max_threads() -> int:
    return int(lib().ldo_max_threads())


def set_dot_variant(v: int) -> None:
    lib().ldo_set_dot_variant(C.c_int(v))


def accelerations(sim: SimParams, angle, thrust, mode=FAITHFUL_F64):
    angle, thrust = _d(angle).reshape(-1, 3), _d(thrust).reshape(-1, 4)
    lin, rot = np.empty_like(angle), np.empty_like(angle)
    for i in range(len(angle)):
        _check(lib().ldo_accelerations(C.byref(sim), _p(angle[i], C.c_double), _p(thrust[i], C.c_double),
                                       _p(lin[i], C.c_double), _p(rot[i], C.c_double), C.c_int(mode)),
               "ldo_accelerations")
    return lin, rot


def accelerations_frames(sim: SimParams, angle, thrust, lab_frame: bool):
    """calculate_accelerations with lin_acc_drone_2_lab = lab_frame: (lin [n,3], rot [n,3], Rot [n,3,3])."""
    angle, thrust = np.atleast_2d(_d(angle)), np.atleast_2d(_d(thrust))
    n = len(angle)
    lin, rot, mat = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 9))
    for i in range(n):
        _check(lib().ldo_accelerations_frames(C.byref(sim), _p(angle[i], C.c_double), _p(thrust[i], C.c_double),
                                              _p(lin[i], C.c_double), _p(rot[i], C.c_double), _p(mat[i], C.c_double),
                                              1 if lab_frame else 0), "ldo_accelerations_frames")
    return lin, rot, mat.reshape(n, 3, 3)


def measure(sim: SimParams, ep: EpisodeParams, state, acc, thrust, mode=FAITHFUL_F64):
    """Returns (state', acc', reward_delta, crashed) for [n] drones."""
    state, acc, thrust = _d(state).reshape(-1, 12).copy(), _d(acc).reshape(-1, 6).copy(), _d(thrust).reshape(-1, 4)
    n = len(state)
    reward = np.zeros(n, np.int64)
    crashed = np.zeros(n, np.int32)
    _check(lib().ldo_measure(C.byref(sim), C.byref(ep), C.c_int64(n), _p(state, C.c_double),
                             _p(acc, C.c_double), _p(thrust, C.c_double), _p(reward, C.c_int64),
                             _p(crashed, C.c_int32), C.c_int(mode)), "ldo_measure")
    return state, acc, reward, crashed


def mlp_forward(spec: MlpSpec, genes, x, mode=FAITHFUL_F64):
    genes, x = _d(genes).ravel(), _d(x).reshape(-1, 15)
    assert genes.size == spec.param_count
    y = np.empty((len(x), 4))
    _check(lib().ldo_mlp_forward(C.byref(spec), _p(genes, C.c_double), C.c_int64(len(x)), _p(x, C.c_double),
                                 _p(y, C.c_double), C.c_int(mode)), "ldo_mlp_forward")
    return y


class FlyResult:
    __slots__ = ("state", "acc", "thrust", "reward", "coins", "flight_time", "crashed", "distance",
                 "trace_state", "trace_reward", "trace_coins")


def fly(sim: SimParams, ep: EpisodeParams, spec: MlpSpec, genes, timesteps, *, coins=None, walls=None,
        init_state=None, n_threads=1, mode=FAITHFUL_F64, trace=False, state=None, acc=None, thrust=None):
    """fly() for genomes genes[n][P].  Returns a FlyResult (arrays indexed by drone)."""
    genes = _d(genes).reshape(-1, spec.param_count)
    n = len(genes)
    coins = coin_chain_reference(ep.n_coins) if coins is None else _d(coins).reshape(-1, COIN_STRIDE)
    assert len(coins) == ep.n_coins
    walls = None if walls is None else _d(walls).reshape(-1, WALL_STRIDE)
    r = FlyResult()
    if state is None:
        init = np.array([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0], float) if init_state is None else _d(init_state)
        state = np.broadcast_to(init.reshape(-1, 12), (n, 12))
    r.state = _d(state).reshape(n, 12).copy()
    r.acc = np.zeros((n, 6)) if acc is None else _d(acc).reshape(n, 6).copy()
    r.thrust = np.zeros((n, 4)) if thrust is None else _d(thrust).reshape(n, 4).copy()
    r.reward = np.zeros(n, np.int64)
    r.coins = np.zeros(n, np.int32)
    r.flight_time = np.zeros(n, np.int32)
    r.crashed = np.zeros(n, np.int32)
    r.distance = np.zeros(n)
    r.trace_state = np.zeros((timesteps, n, 22)) if trace else None
    r.trace_reward = np.zeros((timesteps, n), np.int64) if trace else None
    r.trace_coins = np.zeros((timesteps, n), np.int32) if trace else None
    _check(lib().ldo_fly(C.byref(sim), C.byref(ep), C.byref(spec), _p(coins, C.c_double), _p(walls, C.c_double),
                         _p(genes, C.c_double), C.c_int64(n), C.c_int32(timesteps), _p(r.state, C.c_double),
                         _p(r.acc, C.c_double), _p(r.thrust, C.c_double), _p(r.reward, C.c_int64),
                         _p(r.coins, C.c_int32), _p(r.flight_time, C.c_int32), _p(r.crashed, C.c_int32),
                         _p(r.distance, C.c_double), _p(r.trace_state, C.c_double), _p(r.trace_reward, C.c_int64),
                         _p(r.trace_coins, C.c_int32), C.c_int(n_threads), C.c_int(mode)), "ldo_fly")
    return r


def open_loop(sim: SimParams, thrust_seq, state=None, acc=None, mode=FAITHFUL_F64):
    thrust_seq = _d(thrust_seq).reshape(-1, 4)
    T = len(thrust_seq)
    state = np.array([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0], float) if state is None else _d(state).copy()
    acc = np.zeros(6) if acc is None else _d(acc).copy()
    trace = np.zeros((T, 12))
    _check(lib().ldo_open_loop(C.byref(sim), C.c_int32(T), _p(thrust_seq, C.c_double), _p(state, C.c_double),
                               _p(acc, C.c_double), _p(trace, C.c_double), C.c_int(mode)), "ldo_open_loop")
    return trace


def pid_episode(sim: SimParams, gains, coins, timesteps, coin_radius=1.0, coin_value=1000):
    gains, coins = _d(gains).ravel(), _d(coins).reshape(-1, 3)
    assert gains.size == 18
    ts = np.zeros((timesteps, 12)); tu = np.zeros((timesteps, 4))
    tsc = np.zeros(timesteps, np.int64); ttg = np.zeros(timesteps, np.int32)
    steps = C.c_int32(0); score = C.c_int64(0)
    _check(lib().ldo_pid_episode(C.byref(sim), _p(gains, C.c_double), _p(coins, C.c_double), C.c_int32(len(coins)),
                                 C.c_double(coin_radius), C.c_int32(coin_value), C.c_int32(timesteps),
                                 _p(ts, C.c_double), _p(tu, C.c_double), _p(tsc, C.c_int64), _p(ttg, C.c_int32),
                                 C.byref(steps), C.byref(score)), "ldo_pid_episode")
    k = steps.value
    return ts[:k], tu[:k], tsc[:k], ttg[:k], int(score.value)
