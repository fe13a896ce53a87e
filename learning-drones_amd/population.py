"""Population: the batched analogue of a list of reference ``Drone`` objects.

All per-drone data lives in HBM as struct-of-arrays PyTorch-ROCm tensors (PyTorch is used for
device memory and streams only); every computation is a call through the C ABI of
include/ld_drones.h into the hand-written HIP kernels.  There is no CPU path.

Reference objects this replaces, for n drones at once:
  Drone state + reward bookkeeping   drone/drone.py:21-76, 129-165, 177-218
  NeuralNetwork weights              evolve/brain.py:10-24 (per-drone genome)
  fly()                              flight/__init__.py:12-99
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _native
from .params import (COIN_STRIDE, CRASHED_BIT, F_ACC, F_ANGLE, F_ANGVEL, F_COUNT, F_POS, F_THRUST, F_VEL,
                     FITNESS_DTYPE, GENOME_BLOCK, I_COINS, I_COUNT, I_REWARD, I_STATUS, INITIAL_STATE, MAX_COINS,
                     WALL_STRIDE,
                     EpisodeParams, GaParams, MlpSpec, SimParams, coin_chain_reference, flatten_weights,
                     unflatten_weights)


def _torch():
    import torch
    return torch


def _auto_specialise_enabled() -> bool:
    """The default (specialise=None) compiles missing layer specs when hipcc is on the machine and the
    environment does not say LD_SPECIALISE=0."""
    import os
    import shutil
    if os.environ.get("LD_SPECIALISE", "1") == "0":
        return False
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


class Population:
    """n drones with one controller genome each, resident on one GPU.

    precision: "f32" (production kernels) or "f64" (reference-tracking kernels that reproduce the
    reference's float32 rounding points; validation mode, 2x the bytes).
    """

    AUTO_SPECIALISE_MIN_DRONES = 4096      # below this a 12-60 s compile does not pay for itself

    def __init__(self, n: int, hidden: Iterable[int] = (10, 10), *, precision: str = "f32",
                 device: Optional[str] = None, sim: Optional[SimParams] = None,
                 episode: Optional[EpisodeParams] = None, coins: Optional[np.ndarray] = None,
                 walls: Optional[np.ndarray] = None, init_state: Optional[Sequence[float]] = None,
                 coin_respawn: Optional[bool] = None, specialise: Optional[bool] = None):
        """coins=None is the reference's episode: the first coin (5, 5, 10) and the unbounded respawn
        chain behind it (flight/__init__.py:25,47-52).  A coin table given explicitly is a finite chain
        (nothing respawns after its last row) unless coin_respawn=True -- the argument, or an `episode` built with
        EpisodeParams.reference(coin_respawn=True) -- continues it by the reference rule.  (Beyond the table the kernels
        compute last_row + k * (5, 2.5, 0), which equals the reference's repeated additions exactly when the last row's x
        and y are multiples of 2.5, as in every table the reference's rule itself produces; for other tables the two can
        differ in the last bit.)
        specialise=True compiles the unrolled float32 kernels for this layer spec if the library does not ship
        them (_native.specialise: hipcc at first use, cached; identical results, about twice the throughput of
        the run-time-spec kernels) and raises if that fails; False never does; None (the default) does it for
        float32 populations of AUTO_SPECIALISE_MIN_DRONES or more when hipcc is there, and carries on with the
        run-time-spec kernels (with a warning) when the compile fails.  LD_SPECIALISE=0 in the environment turns
        the automatic case off."""
        torch = _torch()
        _native.require_gpu()
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be 'f32' or 'f64'")
        self.n = int(n)
        if self.n < 0:
            raise ValueError("n must be >= 0")
        self.spec = hidden if isinstance(hidden, MlpSpec) else MlpSpec.from_hidden(hidden)
        if precision == "f32" and specialise is None:
            specialise = "auto" if self.n >= self.AUTO_SPECIALISE_MIN_DRONES and _auto_specialise_enabled() else False
        if specialise and precision == "f32":
            try:
                _native.specialise(self.spec.hidden_widths)
            except Exception as exc:                              # noqa: BLE001 -- hipcc missing / compile error
                if specialise != "auto":
                    raise
                import warnings
                warnings.warn(f"learning-drones_amd: no run-time specialisation for layer spec {self.spec.hidden_widths} "
                              f"({exc}); using the run-time-spec kernels", RuntimeWarning)
        self.precision = precision
        self.dtype = torch.float32 if precision == "f32" else torch.float64
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.sim = sim if sim is not None else SimParams.reference()
        if coin_respawn is None:
            # precedence: the coin_respawn argument; then an episode block that STATES its coin_respawn (built with
            # EpisodeParams.reference(coin_respawn=...), filled by hand, or read back from a checkpoint); then the rule of
            # the docstring: the default episode is the unbounded chain, an explicit table is a finite one
            if episode is not None and getattr(episode, "respawn_explicit", True):
                coin_respawn = bool(episode.coin_respawn)
            else:
                coin_respawn = coins is None
        coins = coin_chain_reference(1) if coins is None else np.ascontiguousarray(coins, np.float64).reshape(-1, COIN_STRIDE)
        if not 1 <= len(coins) <= MAX_COINS:
            raise ValueError(f"need 1..{MAX_COINS} coin rows")
        walls = np.zeros((0, WALL_STRIDE)) if walls is None else np.ascontiguousarray(walls, np.float64).reshape(-1, WALL_STRIDE)
        self.episode = episode if episode is not None else EpisodeParams.reference()
        self.episode.n_coins, self.episode.n_walls = len(coins), len(walls)
        self.episode.coin_respawn = int(bool(coin_respawn))
        self.coins_host, self.walls_host = coins, walls
        self.init_state = np.asarray(INITIAL_STATE if init_state is None else init_state, np.float64).reshape(12).copy()

        P, W = self.spec.param_count, self.spec.words_per_drone
        self.P, self.W = P, W
        dev, n = self.device, self.n
        self.fstate = torch.zeros((F_COUNT, n), dtype=self.dtype, device=dev)
        self.istate = torch.zeros((I_COUNT, n), dtype=torch.int32, device=dev)
        self.n_blocks = (n + GENOME_BLOCK - 1) // GENOME_BLOCK
        self.genomes = torch.zeros((self.n_blocks, W, GENOME_BLOCK, 4), dtype=self.dtype, device=dev)
        self.records = torch.zeros((n, 4), dtype=torch.int32, device=dev)          # ld_fitness_record[n]
        self.distance = torch.zeros((n,), dtype=torch.float64, device=dev) if precision == "f64" else None
        self.queue = torch.zeros((1,), dtype=torch.int32, device=dev)          # work-queue counter of the resident kernel
        self.coins_dev = torch.from_numpy(coins).to(dev)
        self.walls_dev = torch.from_numpy(walls).to(dev) if len(walls) else None
        self._finalized = False
        self.reset()

    # ------------------------------------------------------------------ plumbing
    def _fn(self, name: str):
        return getattr(_native.lib(), f"{name}_{self.precision}")

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t) -> C.c_void_p:
        return C.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())

    # ------------------------------------------------------------------ genomes
    def set_genomes(self, rows) -> None:
        """rows: [n, P] gene vectors in reference order (numpy or torch, any float dtype)."""
        torch = _torch()
        rows = torch.as_tensor(rows)
        if tuple(rows.shape) != (self.n, self.P):
            raise ValueError(f"expected genes of shape ({self.n}, {self.P}), got {tuple(rows.shape)}")
        rows = rows.to(device=self.device, dtype=self.dtype).contiguous()
        if self.n:
            _native.check(self._fn("ld_genomes_pack")(self._ptr(rows), self.P, self._ptr(self.genomes), self.n, self.P,
                                                      self._stream()), "ld_genomes_pack")

    def get_genomes(self):
        """[n, P] gene vectors (torch tensor on the device)."""
        rows = _torch().empty((self.n, self.P), dtype=self.dtype, device=self.device)
        if self.n:
            _native.check(self._fn("ld_genomes_unpack")(self._ptr(self.genomes), self._ptr(rows), self.P, self.n, self.P,
                                                        self._stream()), "ld_genomes_unpack")
        return rows

    def set_controllers(self, controllers: Sequence) -> None:
        """controllers: n objects with ``.weights`` (reference NeuralNetwork format) or n weights lists."""
        if len(controllers) != self.n:
            raise ValueError(f"expected {self.n} controllers")
        rows = np.stack([flatten_weights(getattr(c, "weights", c)) for c in controllers]) if self.n else np.zeros((0, self.P))
        self.set_genomes(rows)

    def get_controllers(self) -> list:
        rows = self.get_genomes().double().cpu().numpy()
        return [unflatten_weights(r, self.spec) for r in rows]

    def init_genomes_reference(self, seed: int) -> None:
        """Genomes ~ N(0, 1/sqrt(n_neurons)) per layer, bias included (evolve/brain.py:24), drawn on
        the device with a Philox stream (statistically, not bitwise, the reference's MT19937 draw)."""
        torch = _torch()
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        cols = []
        for n_out, n_in1 in self.spec.layer_shapes:
            cols.append(torch.randn((self.n, n_out * n_in1), generator=g, device=self.device, dtype=torch.float32)
                        / float(np.sqrt(n_out)))
        self.set_genomes(torch.cat(cols, dim=1) if self.n else torch.zeros((0, self.P)))

    # ------------------------------------------------------------------ state
    def reset(self, init_state: Optional[Sequence[float]] = None) -> None:
        """Drone.reset() for every drone (drone/drone.py:177-218)."""
        if init_state is not None:
            self.init_state = np.asarray(init_state, np.float64).reshape(12).copy()
        init = self.init_state.ctypes.data_as(C.POINTER(C.c_double))
        _native.check(self._fn("ld_population_reset")(self._ptr(self.fstate), self._ptr(self.istate), self.n, init,
                                                      self._stream()), "ld_population_reset")
        self._finalized = False

    def set_state(self, state12=None, acc6=None, thrust4=None) -> None:
        """Overwrite per-drone state from [n, k] arrays (used by tests and by the scalar Drone)."""
        torch = _torch()
        for arr, off, k in ((state12, F_POS, 12), (acc6, F_ACC, 6), (thrust4, F_THRUST, 4)):
            if arr is not None:
                t = torch.as_tensor(np.asarray(arr, np.float64).reshape(self.n, k)).to(self.device, self.dtype)
                self.fstate[off:off + k] = t.t()

    @property
    def position(self): return self.fstate[F_POS:F_POS + 3]
    @property
    def velocity(self): return self.fstate[F_VEL:F_VEL + 3]
    @property
    def angle(self): return self.fstate[F_ANGLE:F_ANGLE + 3]
    @property
    def angle_vel(self): return self.fstate[F_ANGVEL:F_ANGVEL + 3]
    @property
    def acceleration(self): return self.fstate[F_ACC:F_ACC + 6]
    @property
    def thrust(self): return self.fstate[F_THRUST:F_THRUST + 4]
    @property
    def reward(self): return self.istate[I_REWARD]
    @property
    def coins(self): return self.istate[I_COINS]
    @property
    def flight_time(self): return self.istate[I_STATUS] >> 1
    @property
    def crashed(self): return (self.istate[I_STATUS] & 1).bool()

    def state_numpy(self) -> np.ndarray:
        """[n, 22] float64: pos3 vel3 angle3 angle_vel3 acc6 thrust4."""
        return self.fstate.t().double().cpu().numpy()

    # ------------------------------------------------------------------ kernels
    def measure(self) -> None:
        """Drone.measure for all live drones (drone/drone.py:129-165)."""
        _native.check(self._fn("ld_measure")(C.byref(self.sim), C.byref(self.episode), self._ptr(self.fstate),
                                             self._ptr(self.istate), self.n, self._stream()), "ld_measure")

    def feed_forward(self, inputs):
        """NeuralNetwork.feed_forward for every genome: inputs [15, n] -> thrust [4, n]."""
        torch = _torch()
        x = torch.as_tensor(inputs).to(self.device, self.dtype).contiguous()
        if tuple(x.shape) != (15, self.n):
            raise ValueError(f"inputs must be (15, {self.n})")
        y = torch.empty((4, self.n), dtype=self.dtype, device=self.device)
        _native.check(self._fn("ld_mlp_forward")(C.byref(self.spec), self._ptr(self.genomes), self._ptr(x), self._ptr(y),
                                                 self.n, self._stream()), "ld_mlp_forward")
        return y

    def step(self) -> None:
        """One iteration of fly()'s loop for the whole population = ONE kernel launch."""
        _native.check(self._fn("ld_fly_step")(C.byref(self.sim), C.byref(self.episode), C.byref(self.spec),
                                              self._ptr(self.coins_dev), self._ptr(self.walls_dev),
                                              self._ptr(self.genomes), self._ptr(self.fstate), self._ptr(self.istate),
                                              self.n, self._stream()), "ld_fly_step")

    def steps(self, n_steps: int, waves_per_block: int = 0) -> None:
        """`n_steps` iterations of fly()'s loop in ONE launch (ld_fly_steps_f32 / _f64), the drone state in
        registers in between.  float32 layer specs with an unrolled specialisation: the 64 drones of a genome block
        served by `waves_per_block` wavefronts (0 = chosen from n); float64 and run-time layer specs: one wavefront
        per block (waves_per_block 0 or 1).  Bit-identical to n_steps calls of step()."""
        if self.precision == "f32":
            rc = _native.lib().ld_fly_steps_f32(C.byref(self.sim), C.byref(self.episode), C.byref(self.spec),
                                                self._ptr(self.coins_dev), self._ptr(self.walls_dev),
                                                self._ptr(self.genomes), self._ptr(self.fstate), self._ptr(self.istate),
                                                self.n, int(n_steps), int(waves_per_block), self._stream())
        else:
            if waves_per_block not in (0, 1):
                raise ValueError("the float64 kernels have one geometry: one wavefront per genome block")
            rc = _native.lib().ld_fly_steps_f64(C.byref(self.sim), C.byref(self.episode), C.byref(self.spec),
                                                self._ptr(self.coins_dev), self._ptr(self.walls_dev),
                                                self._ptr(self.genomes), self._ptr(self.fstate), self._ptr(self.istate),
                                                self.n, int(n_steps), self._stream())
        _native.check(rc, "ld_fly_steps")

    def finalize(self) -> None:
        """Tail of fly(): distance penalty + result records (flight/__init__.py:83-87,97-99)."""
        if self._finalized:
            raise RuntimeError("finalize() already applied the distance penalty for this episode; reset() first")
        if self.precision == "f32":
            rc = self._fn("ld_fly_finalize")(C.byref(self.episode), self._ptr(self.coins_dev), self._ptr(self.fstate),
                                             self._ptr(self.istate), self._ptr(self.records), self.n, self._stream())
        else:
            rc = self._fn("ld_fly_finalize")(C.byref(self.episode), self._ptr(self.coins_dev), self._ptr(self.fstate),
                                             self._ptr(self.istate), self._ptr(self.records), self._ptr(self.distance),
                                             self.n, self._stream())
        _native.check(rc, "ld_fly_finalize")
        self._finalized = True

    # layer specs with an unrolled step kernel and a resident engine (csrc/ld_kernels.hpp, ld_resident.hip)
    RESIDENT_SPECS = ((8, 10), (10, 10), (16, 16), (32, 32), (64, 64, 64), (16,), (32,), (64,), (8, 8), (24, 24),
                      (48, 48), (64, 64), (32, 32, 32))
    # ... and the subset whose genome fits in registers as doubles (ld_fly_rollout_resident_f64)
    RESIDENT_SPECS_F64 = ((8, 10), (10, 10), (16, 16), (32, 32), (16,), (32,), (64,), (8, 8), (24, 24))

    def has_resident_engine(self) -> bool:
        """Built in (RESIDENT_SPECS / RESIDENT_SPECS_F64) or, float32, compiled by specialise()."""
        fn = _native.lib().ld_has_resident_f32 if self.precision == "f32" else _native.lib().ld_has_resident_f64
        return fn(C.byref(self.spec)) == 1

    def fly(self, timesteps: int = 2500, engine: str = "step", steps_per_launch: Optional[int] = None,
            waves_per_block: int = 0, tile_drones: Optional[int] = None):
        """fly(drone, timesteps) for every drone, enqueued from C on the current stream.  Returns the
        device tensor of result records ([n, 4] int32 view of ld_fitness_record); see results().

        engine="step": the fused step kernel + finalize.  By default (float32, unrolled specs) a launch advances the
        population 64 timesteps with the drone state in registers in between -- 512 where the launch keeps its drones' genes
        on chip (nets of up to 512 gene words: workgroups of 64 drones, fly_wide_kernel) --, and a genome block's 64 drones
        get as many wavefronts as fill the chip; steps_per_launch / waves_per_block make that geometry explicit
        (ld_fly_rollout_steps_f32; steps_per_launch=1 is one launch per timestep).  A population whose genes exceed the
        256 MiB Infinity Cache and are not kept on chip is flown tile by tile (a tile's whole episode, then the next tile), so
        that the re-streamed genes come from the cache after a tile's first timestep (tile_drones: a multiple of 64; None = the
        library's choice, self.n = no tiling).  Results do not depend on the geometry.
        engine="resident" (specs RESIDENT_SPECS in float32, RESIDENT_SPECS_F64 in float64): one launch for the whole episode with the genome
        held in registers and drones handed to lane groups from a work queue (ld_fly_rollout_resident_f32);
        bit-identical results; a population in which most drones crash early (a GA generation) costs about
        its live drone-steps: there it is 1.7-2x faster than the step engine, and on a few dozen drones its one launch is
        the shorter latency chain; a large population of drones that fly the whole episode is 1.6x faster on the step engine.
        engine="compacting" (float32): the step engine flown 128 timesteps at a time over a list of the drones that are still
        flying, re-made before every launch (ld_fly_rollout_compacting_f32): a population in which most drones crash early
        costs about its live drone-steps, at the step engine's (lower) price per drone-step; bit-identical results.
        engine="auto": resident where available, else step."""
        if self._finalized:
            raise RuntimeError("episode already finalized; reset() first")
        args = [C.byref(self.sim), C.byref(self.episode), C.byref(self.spec), self._ptr(self.coins_dev),
                self._ptr(self.walls_dev), self._ptr(self.genomes), self._ptr(self.fstate), self._ptr(self.istate),
                self._ptr(self.records)]
        if engine == "auto":
            engine = "resident" if self.has_resident_engine() else "step"
        if engine == "resident":
            if not self.has_resident_engine():
                raise ValueError(f"no resident engine for spec {self.spec.hidden_widths} in {self.precision}: float32 covers "
                                 f"{self.RESIDENT_SPECS}, float64 {self.RESIDENT_SPECS_F64}")
            if self.precision == "f64":
                args.append(self._ptr(self.distance))
            args += [self.n, int(timesteps), self._ptr(self.queue), self._stream()]
            _native.check(self._fn("ld_fly_rollout_resident")(*args), "ld_fly_rollout_resident")
            self._finalized = True
            return self.records
        if engine == "compacting":
            if self.precision != "f32":
                raise ValueError("engine='compacting' is a float32 engine")
            if getattr(self, "_list_scratch", None) is None:
                self._list_scratch = _torch().empty((self.n + 16,), dtype=_torch().int32, device=self.device)
            args += [self.n, int(timesteps), int(steps_per_launch or 0), self._ptr(self._list_scratch), self._stream()]
            _native.check(_native.lib().ld_fly_rollout_compacting_f32(*args), "ld_fly_rollout_compacting_f32")
            self._finalized = True
            return self.records
        if engine != "step":
            raise ValueError("engine must be 'step', 'resident', 'compacting' or 'auto'")
        if steps_per_launch is not None or waves_per_block or tile_drones is not None:
            if self.precision != "f32":
                raise ValueError("steps_per_launch / waves_per_block / tile_drones are float32 options")
            args += [self.n, int(timesteps), int(steps_per_launch or 0), int(waves_per_block), int(tile_drones or 0), self._stream()]
            _native.check(_native.lib().ld_fly_rollout_steps_f32(*args), "ld_fly_rollout_steps_f32")
            self._finalized = True
            return self.records
        if self.precision == "f64":
            args.append(self._ptr(self.distance))
        args += [self.n, int(timesteps), self._stream()]
        _native.check(self._fn("ld_fly_rollout")(*args), "ld_fly_rollout")
        self._finalized = True
        return self.records

    def rollout_geometry(self, timesteps: int) -> dict:
        """What fly(timesteps) (step engine, default geometry) launches: loop iterations per launch,
        wavefronts per genome block, fused-step launches per episode (ld_fly_rollout_geometry_f32)."""
        k, w, m, tile = C.c_int32(1), C.c_int32(1), C.c_int32(int(timesteps)), C.c_int64(self.n)
        _native.check(_native.lib().ld_fly_rollout_geometry_f32(C.byref(self.spec), self.n, int(timesteps), C.byref(k),
                                                                C.byref(w), C.byref(m), C.byref(tile)),
                      "ld_fly_rollout_geometry_f32")
        if self.precision == "f64":             # 64 timesteps per launch (nothing is kept on chip), one wavefront per block, no tiles
            return {"steps_per_launch": 64, "waves_per_block": 1, "launches": -(-int(timesteps) // 64), "tile_drones": self.n,
                    "lds_words": 0, "register_words": 0, "streamed_words": self.W}
        lds, regs, streamed = self.kept_words(w.value, min(k.value, int(timesteps)))
        return {"steps_per_launch": k.value, "waves_per_block": w.value, "launches": m.value, "tile_drones": tile.value,
                "lds_words": lds, "register_words": regs, "streamed_words": streamed}

    def flies_wide(self) -> bool:
        """Whether this population's workgroups of 64 drones (waves_per_block 8, launches that keep their genes) run
        fly_wide_kernel: every gene word on chip, the angles' sin / cos computed beside the output layer (csrc/ld_wide.hpp)."""
        return self.precision == "f32" and _native.lib().ld_debug_wide_plan(C.byref(self.spec), None, None) == 1

    def kept_words(self, waves_per_block: int, n_steps: int):
        """Gene words per drone a launch of `n_steps` timesteps in the geometry `waves_per_block` (keeps in LDS, holds in
        registers, loads from memory every timestep) (ld_fly_steps_kept_words_f32)."""
        if self.precision != "f32":
            return 0, 0, self.W
        lds, regs, streamed = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _native.check(_native.lib().ld_fly_steps_kept_words_f32(C.byref(self.spec), int(waves_per_block), int(n_steps), C.byref(lds),
                                                                C.byref(regs), C.byref(streamed)), "ld_fly_steps_kept_words_f32")
        return lds.value, regs.value, streamed.value

    def auto_waves_per_block(self, drones: int, steps_per_launch: int) -> int:
        """Wavefronts per genome block the library chooses (waves_per_block = 0) for a launch of `drones` drones
        and `steps_per_launch` timesteps (ld_fly_steps_auto_waves_f32)."""
        w = _native.lib().ld_fly_steps_auto_waves_f32(C.byref(self.spec), int(drones), int(steps_per_launch))
        if w < 0:
            _native.check(w, "ld_fly_steps_auto_waves_f32")
        return w

    def results(self) -> np.ndarray:
        """Host copy of the result records as a structured array with the extra convenience fields
        of fly()/Drone: reward, flight_time, coins, crashed, distance_to_coin."""
        raw = self.records.cpu().numpy().view(FITNESS_DTYPE).reshape(self.n)
        out = np.zeros(self.n, dtype=[("reward", "<i4"), ("flight_time", "<i4"), ("coins", "<i4"),
                                      ("crashed", "?"), ("distance_to_coin", "<f8")])
        out["reward"], out["flight_time"] = raw["reward"], raw["flight_time"]
        out["coins"] = raw["coins_crashed"] & (CRASHED_BIT - 1)
        out["crashed"] = (raw["coins_crashed"] & CRASHED_BIT) != 0
        out["distance_to_coin"] = raw["distance_to_coin"] if self.distance is None else self.distance.cpu().numpy()
        return out

    # ------------------------------------------------------------------ generation step (device GA)
    def genome_of(self, slot: int):
        """The W x 4 genome words of one drone (device tensor; the elite buffer of breed())."""
        return self.genomes[slot // GENOME_BLOCK, :, slot % GENOME_BLOCK, :].contiguous()

    def breed(self, fitness_all, ga: GaParams, *, elite=None, child_begin: int = 0, src=None,
              n_per_segment: Optional[int] = None, n_segments: int = 1, fetch_parents=None, lifetimes=None,
              slot_of=None, place: Optional[str] = None) -> None:
        """Replace this population's genomes by the next generation (evolve/util.py:5-78 +
        drone/drone.py:167-172 + the elitism of evolve/evolution.py:105-107), on the device.

        fitness_all: int32 device tensor with the fitness of ALL n_total drones (this shard's
        children are global slots [child_begin, child_begin + n)); src: the genome words the parents
        are read from — n_segments back-to-back arrays of n_per_segment drones (default: this
        population's own genomes, one segment); elite: W x 4 genome words copied unmutated into
        global child 0, or None; fetch_parents: optional callable(parents_global) -> (src genome array,
        parents re-indexed into it) used by the multi-GPU all-to-all exchange (parallel.py).

        Placement of the children (ld_ga_breed's `order`; a child's parents, cuts and mutation draws are keyed
        by the child, so any placement produces the same set of children, and the reference attaches no
        meaning to list positions beyond slot 0: evolve/util.py:5-27, evolve/evolution.py:89-107):
        place="parent" sorts the children by the slot of their first parent (neighbouring reads in the kernel's
        gather of 16-byte parent words); lifetimes (int32 device tensor, flight_time of ALL n_total drones of
        the generation just flown) sorts them by that parent's flight time instead, longest first.  Neither
        pays at present (see DeviceEvolution): the breed kernel is bound by its random-number arithmetic and
        the resident kernel balances lifetimes itself.  The elite stays in slot 0 either way.

        slot_of: optional int64 device tensor [n_total]: the genome slot that holds the drone with canonical
        index i.  Selection always works on canonical indices (fitness_all is then in canonical order and the
        selected parents are translated to slots through it), which keeps a whole GA run independent of how
        earlier generations were placed.  After the call, last_order holds this shard's placement
        (slot -> local child index) or None."""
        torch = _torch()
        if self.n % 2:
            raise ValueError("the generation step pairs slots (i, i+1): the shard size must be even")
        fitness_all = fitness_all.to(device=self.device, dtype=torch.int32).contiguous()
        n_total = fitness_all.numel()
        src = self.genomes if src is None else src
        n_per_segment = self.n if n_per_segment is None else int(n_per_segment)
        n_layers = self.spec.n_hidden + 1
        parents = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        cuts = torch.empty((max(self.n // 2, 1), n_layers), dtype=torch.int32, device=self.device)
        lib = _native.lib()
        _native.check(lib.ld_ga_select(C.byref(ga), C.byref(self.spec), self._ptr(fitness_all), n_total, child_begin,
                                       self.n, self._ptr(parents), self._ptr(cuts), self._stream()), "ld_ga_select")
        if slot_of is not None:                                              # canonical parent ids -> genome slots
            parents = slot_of.to(self.device)[parents.long()].to(torch.int32).contiguous()
        self.last_parents = parents
        order = None
        if self.n > 2 and (lifetimes is not None or place == "parent"):
            if lifetimes is not None:                                         # longest-lived first
                key = -lifetimes.to(device=self.device).long()[parents.long()]
            else:                                                             # neighbours in the gather
                key = parents.long().clone()
            if elite is not None and child_begin == 0:
                key[0] = torch.iinfo(torch.int64).min                         # the elite keeps slot 0
            order = torch.sort(key, stable=True).indices.to(torch.int32).contiguous()
        self.last_order = order
        if fetch_parents is not None:
            src, parents = fetch_parents(parents)
            src, parents = src.contiguous(), parents.to(device=self.device, dtype=torch.int32).contiguous()
            n_per_segment, n_segments = src.shape[0] * GENOME_BLOCK, 1
        children = torch.empty_like(self.genomes)
        if elite is not None:
            elite = elite.to(device=self.device, dtype=self.dtype).contiguous()
        _native.check(self._fn("ld_ga_breed")(C.byref(ga), C.byref(self.spec), self._ptr(src), n_per_segment, n_segments,
                                              self._ptr(parents), self._ptr(cuts), self._ptr(elite), child_begin, self.n,
                                              self._ptr(order), self._ptr(children), self._stream()), "ld_ga_breed")
        self.genomes = children
        self.last_cuts = cuts

    # ------------------------------------------------------------------ checkpoint / replay
    def state_dict(self) -> dict:
        """Everything needed to continue this population elsewhere (host numpy arrays)."""
        d = {"n": self.n, "hidden": np.array(self.spec.hidden_widths, np.int32), "precision": self.precision,
             "genes": self.get_genomes().cpu().numpy(), "fstate": self.fstate.cpu().numpy(),
             "istate": self.istate.cpu().numpy(), "records": self.records.cpu().numpy(),
             "coins": self.coins_host, "walls": self.walls_host, "init_state": self.init_state,
             "finalized": self._finalized,
             # the uniform blocks as the raw bytes of struct ld_sim_params / ld_episode_params (wind, payload,
             # dt; reward signs, limits, crash planes, coin_respawn), with the ABI version that laid them out
             "sim": np.frombuffer(bytes(self.sim), np.uint8).copy(),
             "episode": np.frombuffer(bytes(self.episode), np.uint8).copy(),
             "abi_version": _native.ABI_VERSION}
        if self.distance is not None:
            d["distance"] = self.distance.cpu().numpy()
        return d

    def save(self, path: str) -> None:
        np.savez_compressed(path, **self.state_dict())

    @classmethod
    def load(cls, path: str, device: Optional[str] = None) -> "Population":
        torch = _torch()
        z = np.load(path, allow_pickle=False)
        if "sim" not in z or int(z["abi_version"]) != _native.ABI_VERSION:
            raise ValueError(f"{path}: checkpoint without the physics / episode blocks of ABI version {_native.ABI_VERSION} "
                             "(written by an older build); it cannot be resumed faithfully")
        sim = SimParams.from_buffer_copy(z["sim"].tobytes())
        episode = EpisodeParams.from_buffer_copy(z["episode"].tobytes())
        pop = cls(int(z["n"]), tuple(int(h) for h in z["hidden"]), precision=str(z["precision"]), device=device,
                  sim=sim, episode=episode, coins=z["coins"], walls=z["walls"] if len(z["walls"]) else None,
                  init_state=z["init_state"], coin_respawn=bool(episode.coin_respawn))
        pop.set_genomes(z["genes"])
        pop.fstate.copy_(torch.from_numpy(z["fstate"]))
        pop.istate.copy_(torch.from_numpy(z["istate"]))
        pop.records.copy_(torch.from_numpy(z["records"]))
        if pop.distance is not None and "distance" in z:
            pop.distance.copy_(torch.from_numpy(z["distance"]))
        pop._finalized = bool(z["finalized"])
        return pop

    RESULTS_CSV_HEADER = ("Time, Roll, Pitch, Yaw, X, Y, Z, T1, T2, T3, T4, Wind X, Wind Y, PID Roll, PID Pitch, PID Yaw, "
                          "PID X, PID Y, PID Z, Set Roll, Set Pitch, Set Yaw, Set X, Set Y, Set Z")

    def record_trajectory(self, timesteps: int, slots: Sequence[int] = (0,), csv_prefix: Optional[str] = None) -> np.ndarray:
        """Fly `timesteps` steps (from the current state) and return the per-step log of the chosen
        drones in the 25-column layout of the reference's Results.csv (datahandling/data_scraper.py:
        285-307): time [s]; roll, pitch, yaw [deg]; x, y, z [m]; thrust 1-4; wind x, y; six PID columns
        (zero: the controller here is the network); set-points roll, pitch, yaw (zero) and x, y, z (the
        coin the drone is flying to).  Shape [len(slots), timesteps, 25]; rows after a crash repeat the
        frozen state.  With csv_prefix, also writes <csv_prefix><slot>.csv with the reference's header,
        which the reference's own viewers / plot scripts read."""
        torch = _torch()
        idx = torch.as_tensor(list(slots), device=self.device, dtype=torch.long)
        T = int(timesteps)
        log = torch.empty((T, F_COUNT, len(idx)), dtype=self.dtype, device=self.device)
        coin = torch.empty((T, len(idx)), dtype=torch.int32, device=self.device)
        for t in range(T):
            self.step()
            log[t] = self.fstate.index_select(1, idx)
            coin[t] = self.istate[I_COINS].index_select(0, idx)
        a = log.permute(2, 0, 1).double().cpu().numpy()
        k, last = coin.t().cpu().numpy(), len(self.coins_host) - 1
        target = self.coins_host[np.minimum(k, last)][:, :, :3].copy()
        if self.episode.coin_respawn:                                     # beyond the table: flight/__init__.py:47-52
            extra = np.maximum(k - last, 0)
            target[:, :, 0] += 5.0 * extra
            target[:, :, 1] += 2.5 * extra
        out = np.zeros((len(idx), T, 25))
        out[:, :, 0] = np.arange(T)[None, :] * float(self.sim.dt)
        out[:, :, 1:4] = np.degrees(a[:, :, F_ANGLE:F_ANGLE + 3])
        out[:, :, 4:7] = a[:, :, F_POS:F_POS + 3]
        out[:, :, 7:11] = a[:, :, F_THRUST:F_THRUST + 4]
        out[:, :, 11], out[:, :, 12] = float(self.sim.wind[0]), float(self.sim.wind[1])
        out[:, :, 22:25] = target
        if csv_prefix is not None:
            for k, slot in enumerate(slots):
                np.savetxt(f"{csv_prefix}{slot}.csv", out[k], delimiter=",", header=self.RESULTS_CSV_HEADER)
        return out

    # ------------------------------------------------------------------ accounting
    @property
    def bytes_per_drone_step(self) -> int:
        """Algorithmic HBM bytes of one fused drone-step in a one-step launch: genes once + 25 state
        words read and written (4P + 200 for float32)."""
        word = 4 if self.precision == "f32" else 8
        return word * (self.P + 2 * F_COUNT) + 4 * 2 * I_COUNT

    def bytes_per_launch(self, steps_per_launch: int = 1, drones: Optional[int] = None) -> int:
        """Algorithmic bytes of one fused launch of `steps_per_launch` timesteps over `drones` drones (a tile;
        default: the whole population): the genes once per timestep, the state read and written once per
        LAUNCH (it stays in registers in between)."""
        word = 4 if self.precision == "f32" else 8
        drones = self.n if drones is None else int(drones)
        return drones * (steps_per_launch * word * self.P + word * 2 * F_COUNT + 4 * 2 * I_COUNT)
