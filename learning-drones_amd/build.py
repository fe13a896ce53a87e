"""Build libld_drones.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m learning_drones_amd.build [--force]

Flags that matter for correctness and speed (see csrc/ld_device.hpp, csrc/ld_nets.hpp):
  -ffp-contract=off     fused multiply-adds only where the source says fma(): the float64 kernels
                        reproduce the reference's rounding points, the float32 kernels match the
                        oracle's float arithmetic operation for operation
  -fno-slp-vectorize    the SLP vectoriser pairs neighbouring neurons into v_pk_fma_f32 and in doing
                        so regroups the gene loads into bursts, destroying the load pipeline
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB = PKG_DIR / "libld_drones.so"
SOURCES = [CSRC / "ld_kernels.hip"]
HEADERS = [CSRC / "ld_device.hpp", CSRC / "ld_nets.hpp", CSRC / "ld_ga.hpp", PKG_DIR.parent / "include" / "ld_drones.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found (need ROCm)")
    return exe


def up_to_date() -> bool:
    if not LIB.exists():
        return False
    t = LIB.stat().st_mtime
    return all(p.stat().st_mtime <= t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and up_to_date():
        return LIB
    cmd = [hipcc(), *FLAGS, "-o", str(LIB), *map(str, SOURCES)]
    if verbose:
        print("[learning-drones_amd] " + " ".join(cmd), flush=True)
    env = dict(os.environ)
    subprocess.check_call(cmd, env=env)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
