"""Build libld_drones.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m learning_drones_amd.build [--force]

One object per translation unit, compiled in parallel (the unrolled 64-64-64 kernels alone take
about two minutes), then linked into the in-tree shared library.  Flags that matter for correctness
and speed (see csrc/ld_device.hpp, csrc/ld_nets.hpp):
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
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = PKG_DIR / "build"
LIB = PKG_DIR / "libld_drones.so"
SOURCES = sorted(CSRC.glob("*.hip"))
HEADERS = sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("*.inc")) + [PKG_DIR.parent / "include" / "ld_drones.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found (need ROCm)")
    return exe


def _newest_input() -> float:
    return max(p.stat().st_mtime for p in HEADERS + [Path(__file__)])


def _deps(obj: Path):
    """Files the object was built from, read from the compiler's -MD dependency file (None if absent)."""
    dep = obj.with_suffix(".d")
    if not dep.exists():
        return None
    text = dep.read_text().replace("\\\n", " ")
    return [Path(t) for t in text.split(":", 1)[1].split() if t.startswith(str(PKG_DIR.parent))]


def _compile(src: Path, verbose: bool) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    if obj.exists():
        deps = _deps(obj)
        newest = _newest_input() if deps is None else max([Path(__file__).stat().st_mtime] + [d.stat().st_mtime for d in deps if d.exists()])
        if obj.stat().st_mtime >= max(src.stat().st_mtime, newest) and (deps is None or all(d.exists() for d in deps)):
            return obj
    cmd = [hipcc(), *FLAGS, "-MD", "-MF", str(obj.with_suffix(".d")), "-c", str(src), "-o", str(obj)]
    if verbose:
        print("[learning-drones_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def up_to_date() -> bool:
    if not LIB.exists():
        return False
    t = LIB.stat().st_mtime
    return all(p.stat().st_mtime <= t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and up_to_date():
        return LIB
    OBJ_DIR.mkdir(exist_ok=True)
    if force:
        for o in list(OBJ_DIR.glob("*.o")) + list(OBJ_DIR.glob("*.d")):
            o.unlink()
    jobs = min(len(SOURCES), os.cpu_count() or 4, 8)
    with ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, verbose), SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print("[learning-drones_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
