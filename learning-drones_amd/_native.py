"""ctypes binding of libld_drones.so (the C ABI of include/ld_drones.h).

There is no CPU fallback: if the library has not been built, or no GPU is visible, the
functions here raise.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m learning_drones_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from .params import EpisodeParams, FitnessRecord, GaParams, MlpSpec, SimParams

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libld_drones.so"

# every symbol include/ld_drones.h declares
EXPORTS = (
    "ld_version", "ld_strerror", "ld_last_hip_error", "ld_device_count", "ld_sim_params_default",
    "ld_episode_params_default", "ld_coin_chain_reference", "ld_mlp_param_count", "ld_mlp_words_per_drone",
    "ld_genome_words",
    "ld_genomes_pack_f32", "ld_genomes_unpack_f32", "ld_genomes_pack_f64", "ld_genomes_unpack_f64",
    "ld_population_reset_f32", "ld_population_reset_f64", "ld_measure_f32", "ld_measure_f64",
    "ld_mlp_forward_f32", "ld_mlp_forward_f64", "ld_fly_step_f32", "ld_fly_step_f64", "ld_fly_steps_f64",
    "ld_fly_finalize_f32", "ld_fly_finalize_f64", "ld_fly_rollout_f32", "ld_fly_rollout_f64",
    "ld_ga_select", "ld_ga_breed_f32", "ld_ga_breed_f64", "ld_pid_episode_f32", "ld_pid_episode_f64",
    "ld_fly_rollout_resident_f32", "ld_debug_force_generic", "ld_accelerations_f32", "ld_accelerations_f64",
    "ld_fly_steps_f32", "ld_fly_rollout_steps_f32", "ld_fly_rollout_geometry_f32", "ld_fly_steps_kept_words_f32", "ld_accelerations_frames_f32", "ld_accelerations_frames_f64", "ld_fly_steps_resident_f32",
    "ld_fly_rollout_resident_f64", "ld_register_specialisation_f32", "ld_has_specialisation_f32",
    "ld_has_resident_f32", "ld_has_resident_f64", "ld_fly_steps_auto_waves_f32", "ld_debug_lds_keep", "ld_debug_wide_plan", "ld_fly_rollout_compacting_f32",
)
ABI_VERSION = 3


class NativeLibraryMissing(RuntimeError):
    pass


class LdError(RuntimeError):
    def __init__(self, what: str, status: int, detail: str = ""):
        self.status = status
        super().__init__(f"{what} failed: status {status} ({detail})")


_lib = None


def lib() -> C.CDLL:
    """Load libld_drones.so (once).  Raises NativeLibraryMissing if it was never built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("LD_DRONES_LIB", LIB_PATH))
        if not path.exists():
            raise NativeLibraryMissing(
                f"{path} not found: the HIP extension has not been built (run __graft_entry__.build()). "
                "learning-drones_amd has no CPU fallback.")
        handle = C.CDLL(str(path))
        missing = [s for s in EXPORTS if not hasattr(handle, s)]
        if missing:
            raise NativeLibraryMissing(f"{path} lacks symbols {missing}; rebuild it")
        _declare(handle)
        if handle.ld_version() != ABI_VERSION:
            raise NativeLibraryMissing(f"{path}: ABI version {handle.ld_version()} != {ABI_VERSION}; rebuild it")
        _lib = handle
    return _lib


def _declare(h: C.CDLL) -> None:
    vp, i64, i32, dp = C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)
    sim, ep, spec = C.POINTER(SimParams), C.POINTER(EpisodeParams), C.POINTER(MlpSpec)
    h.ld_version.restype = C.c_int
    h.ld_strerror.restype = C.c_char_p
    h.ld_strerror.argtypes = [C.c_int]
    h.ld_last_hip_error.restype = C.c_char_p
    h.ld_device_count.restype = C.c_int
    h.ld_sim_params_default.argtypes = [sim]
    h.ld_episode_params_default.argtypes = [ep]
    h.ld_coin_chain_reference.argtypes = [dp, i32]
    h.ld_mlp_param_count.restype = i64
    h.ld_mlp_param_count.argtypes = [spec]
    h.ld_mlp_words_per_drone.restype = i64
    h.ld_mlp_words_per_drone.argtypes = [spec]
    h.ld_genome_words.restype = i64
    h.ld_genome_words.argtypes = [spec, i64]
    for sfx in ("f32", "f64"):
        getattr(h, f"ld_genomes_pack_{sfx}").argtypes = [vp, i64, vp, i64, i64, vp]
        getattr(h, f"ld_genomes_unpack_{sfx}").argtypes = [vp, vp, i64, i64, i64, vp]
        getattr(h, f"ld_population_reset_{sfx}").argtypes = [vp, vp, i64, dp, vp]
        getattr(h, f"ld_measure_{sfx}").argtypes = [sim, ep, vp, vp, i64, vp]
        getattr(h, f"ld_mlp_forward_{sfx}").argtypes = [spec, vp, vp, vp, i64, vp]
        getattr(h, f"ld_fly_step_{sfx}").argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, i64, vp]
    h.ld_fly_finalize_f32.argtypes = [ep, vp, vp, vp, vp, i64, vp]
    h.ld_fly_finalize_f64.argtypes = [ep, vp, vp, vp, vp, vp, i64, vp]
    h.ld_fly_rollout_f32.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    h.ld_fly_rollout_f64.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    h.ld_fly_rollout_resident_f32.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]
    h.ld_fly_rollout_resident_f64.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]
    h.ld_fly_steps_f32.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, i64, i32, i32, vp]
    h.ld_fly_steps_f64.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, i64, i32, vp]
    h.ld_fly_rollout_steps_f32.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i64, vp]
    h.ld_fly_rollout_compacting_f32.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]
    h.ld_accelerations_f32.argtypes = [sim, vp, vp, vp, vp, i64, vp]
    h.ld_accelerations_f64.argtypes = [sim, vp, vp, vp, vp, i64, vp]
    h.ld_debug_force_generic.argtypes = [C.c_int]
    h.ld_debug_lds_keep.argtypes = [C.c_int]
    h.ld_debug_wide_plan.argtypes = [spec, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    h.ld_register_specialisation_f32.argtypes = [spec, vp, vp, vp]
    h.ld_has_specialisation_f32.argtypes = [spec]
    h.ld_has_resident_f32.argtypes = [spec]
    h.ld_has_resident_f64.argtypes = [spec]
    h.ld_fly_steps_auto_waves_f32.argtypes = [spec, i64, i32]
    h.ld_fly_steps_kept_words_f32.argtypes = [spec, i32, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    h.ld_accelerations_frames_f32.argtypes = [sim, vp, vp, vp, vp, vp, i32, i64, vp]
    h.ld_accelerations_frames_f64.argtypes = [sim, vp, vp, vp, vp, vp, i32, i64, vp]
    h.ld_fly_steps_resident_f32.argtypes = [sim, ep, spec, vp, vp, vp, vp, vp, i64, i32, vp, vp]
    pi32 = C.POINTER(C.c_int32)
    h.ld_fly_rollout_geometry_f32.argtypes = [spec, i64, i32, pi32, pi32, pi32, C.POINTER(C.c_int64)]
    ga = C.POINTER(GaParams)
    h.ld_ga_select.argtypes = [ga, spec, vp, i64, i64, i64, vp, vp, vp]
    h.ld_ga_breed_f32.argtypes = [ga, spec, vp, i64, i32, vp, vp, vp, i64, i64, vp, vp, vp]
    h.ld_ga_breed_f64.argtypes = [ga, spec, vp, i64, i32, vp, vp, vp, i64, i64, vp, vp, vp]
    for sfx in ("f32", "f64"):
        getattr(h, f"ld_pid_episode_{sfx}").argtypes = [sim, vp, vp, i32, C.c_double, i32, i32, vp, vp, vp, vp, vp, i64, vp]
    for name in EXPORTS:
        if name not in ("ld_strerror", "ld_last_hip_error", "ld_mlp_param_count", "ld_mlp_words_per_drone",
                        "ld_genome_words"):
            getattr(h, name).restype = C.c_int


_specialised = {}          # hidden widths -> CDLL of the run-time specialisation (kept alive: the registry holds its entry points)


def specialise(hidden) -> bool:
    """Make the unrolled float32 kernels available for one more layer spec: compile them with hipcc
    (build.build_specialisation; cached under build/jit/), load the object and register it with the library
    (ld_register_specialisation_f32).  Returns True if the spec now runs unrolled kernels (also when it always
    did), raises if hipcc fails.  Results do not change (same arithmetic as the generic kernels, bit for
    bit); throughput does: 47-68 % -> 85-90 % of the HBM roofline."""
    from . import build as ld_build
    spec = MlpSpec.from_hidden(hidden)
    key = spec.hidden_widths
    h = lib()
    if h.ld_has_specialisation_f32(C.byref(spec)) == 1:
        return True
    so = C.CDLL(str(ld_build.build_specialisation(key)))
    fly = C.cast(so.ld_spec_fly_steps_f32, C.c_void_p)
    mlp = C.cast(so.ld_spec_mlp_forward_f32, C.c_void_p)
    res = C.cast(so.ld_spec_fly_resident_f32, C.c_void_p) if hasattr(so, "ld_spec_fly_resident_f32") else None
    check(h.ld_register_specialisation_f32(C.byref(spec), fly, mlp, res), "ld_register_specialisation_f32")
    _specialised[key] = so
    return True


def check(status: int, what: str) -> None:
    if status != 0:
        h = lib()
        detail = h.ld_strerror(status).decode()
        hip = h.ld_last_hip_error().decode()
        # a run-time specialisation is a shared object of its own (loaded RTLD_LOCAL) with its own copy of the
        # per-thread error text: a launch failure inside one of its kernels is reported there
        for key, so in _specialised.items():
            if hasattr(so, "ld_spec_last_hip_error"):
                so.ld_spec_last_hip_error.restype = C.c_char_p
                text = so.ld_spec_last_hip_error().decode()
                if text:
                    hip += f"{'; ' if hip else ''}[specialisation {key}] {text}"
        raise LdError(what, status, detail + (f"; {hip}" if hip else ""))


def require_gpu() -> None:
    """Raise unless the library is loadable and a HIP device is visible."""
    n = lib().ld_device_count()
    if n <= 0:
        raise RuntimeError("learning-drones_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback")


__all__ = ["lib", "check", "require_gpu", "specialise", "EXPORTS", "LIB_PATH", "NativeLibraryMissing", "LdError",
           "FitnessRecord"]
