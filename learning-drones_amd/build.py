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
  -mllvm -simplifycfg-sink-common=false   (the multi-wavefront step kernels of layer specs with a hidden layer
                        narrower than 16, shipped or compiled at run time)  the wavefronts of a block run different arms of one switch; SimplifyCFG
                        sinks their look-alike `ring[i] = load(...)` statements into a common block that
                        stores through a phi of ring-slot addresses, which keeps the whole gene ring from being
                        promoted to registers: 10-10, 8-10, 12-12, 13-7 from four wavefronts per block and 8-8
                        at eight then ran out of scratch, 3-4x slower
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
NO_SINK = ["-mllvm", "-simplifycfg-sink-common=false"]


def narrow(hidden) -> bool:
    """Layer specs with a hidden layer narrower than 16: the ones whose multi-wavefront kernels need NO_SINK (wider ones
    are clean without it, and 15-16-4 at eight wavefronts per block spills WITH it)."""
    return min(int(h) for h in hidden) < 16


def keeps_in_lds(hidden) -> bool:
    """Layer specs of up to 512 gene words: their 8-wavefront step kernels exist in a variant that keeps part of the
    genes in LDS for the whole launch (csrc/ld_split.hpp, KeepPlan), in a translation unit of its own
    (ld_split_keep_<tag>.hip) that is always built with NO_SINK: the fill and read-back statements look alike in every
    wavefront's arm of the switch as well (32-32 without it: the gene ring in 272 bytes of scratch per lane)."""
    n_in, genes = 15, 0
    for h in tuple(int(h) for h in hidden) + (4,):
        genes += h * (n_in + 1)
        n_in = h
    return (genes + 3) // 4 <= 512


def flags_for(src: Path):
    if src.name.startswith("ld_split_keep_"):
        return FLAGS + NO_SINK
    if src.name.startswith("ld_split_"):
        widths = [t for t in src.stem[len("ld_split_"):].split("_") if t.isdigit()]
        return FLAGS + (NO_SINK if narrow(widths) else [])
    return FLAGS


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
    cmd = [hipcc(), *flags_for(src), "-MD", "-MF", str(obj.with_suffix(".d")), "-c", str(src), "-o", str(obj)]
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


JIT_DIR = OBJ_DIR / "jit"


def resident_lanes(hidden) -> int:
    """Lanes per drone for the genome-resident kernel of a spec specialised at run time: the smallest power of
    two that leaves at most ~160 gene registers per lane (two wavefronts per SIMD), the regime the built-in
    specs' measured optima sit in (profiles/r02_resident_lanes.json); 0 if even 64 lanes do not fit."""
    n_in, genes = 15, 0
    for h in tuple(hidden) + (4,):
        genes += h * (n_in + 1)
        n_in = h
    lanes = 2
    while lanes < 64 and genes / lanes > 160:
        lanes *= 2
    return lanes if genes / lanes <= 230 else 0


def build_specialisation(hidden, verbose: bool = False) -> Path:
    """Compile the unrolled float32 kernels (fused step in every launch geometry, feed_forward, and the
    genome-resident engine where the genome fits in registers) for ONE more layer spec into its own shared
    object, build/jit/libld_spec_<widths>.so, and return its path.  The translation unit is a dozen lines: it
    instantiates csrc/ld_split.inc, csrc/ld_unrolled.inc and ld::launch_fly_resident for the spec and exports
    the C entry points ld_register_specialisation_f32 takes.  Needs hipcc (20-60 s for nets up to ~3 000
    genes; the object is reused afterwards)."""
    hidden = tuple(int(h) for h in hidden)
    tag = "_".join(map(str, hidden))
    widths = ", ".join(map(str, hidden))
    lanes = resident_lanes(hidden)
    JIT_DIR.mkdir(parents=True, exist_ok=True)
    src, lib = JIT_DIR / f"ld_spec_{tag}.hip", JIT_DIR / f"libld_spec_{tag}.so"
    keep = keeps_in_lds(hidden)
    text = (f"// generated by learning_drones_amd.build.build_specialisation({hidden})\n"
            f"#define LD_TAG jit_{tag}\n#define LD_WIDTHS {widths}\n#define LD_ONLY_F32\n"
            + ("#define LD_SPLIT_HAS_KEEP\n" if keep else "") +
            f'#include "{CSRC}/ld_unrolled.inc"\n#include "{CSRC}/ld_split.inc"\n#include "{CSRC}/ld_resident.hpp"\n'
            'extern "C" int ld_spec_fly_steps_f32(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, '
            "const double* walls, const float* genomes, float* fstate, int32_t* istate, int64_t n, int64_t begin, int64_t end, "
            "int32_t n_steps, int32_t waves_per_block, void* stream)\n"
            f"{{ return ld_u_split_f32_jit_{tag}(sim, ep, coins, walls, genomes, fstate, istate, n, begin, end, n_steps, "
            "waves_per_block, static_cast<hipStream_t>(stream)); }\n"
            'extern "C" int ld_spec_mlp_forward_f32(const ld_mlp_spec* spec, const float* genomes, const float* inputs, '
            "float* outputs, int64_t n, void* stream)\n"
            f"{{ return ld_u_mlp_f32_jit_{tag}(spec, genomes, inputs, outputs, n, static_cast<hipStream_t>(stream)); }}\n"
            # the text of the last HIP error seen by THIS object's launches (its own copy of ld::g_hip_error: the object
            # is loaded RTLD_LOCAL), for _native.check()
            'extern "C" const char* ld_spec_last_hip_error(void) { return ld::g_hip_error; }\n')
    if lanes:
        text += ('extern "C" int ld_spec_fly_resident_f32(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, '
                 "const double* walls, const float* genomes, float* fstate, int32_t* istate, int64_t n, int32_t n_steps, "
                 "int32_t* queue, void* stream)\n"
                 f"{{ return ld::launch_fly_resident<float, false, {lanes}, {widths}>(sim, ep, coins, walls, genomes, fstate, istate, n, "
                 "n_steps, queue, static_cast<hipStream_t>(stream)); }\n")
    keep_text = (f"// generated by learning_drones_amd.build.build_specialisation({hidden}): the LDS-keeping step kernels\n"
                 f"#define LD_TAG jit_{tag}\n#define LD_WIDTHS {widths}\n#define LD_SPLIT_KEEP_TU\n"
                 f'#include "{CSRC}/ld_split.inc"\n') if keep else ""
    text += "// ---- second translation unit\n" + keep_text if keep else ""
    newest = max(p.stat().st_mtime for p in HEADERS + [Path(__file__)])
    if lib.exists() and src.exists() and src.read_text() == text and lib.stat().st_mtime >= newest:
        return lib
    # several ranks of one job may get here at once: each compiles its own copy and renames it into place
    # (os.replace is atomic; the copies are identical; the source goes first, so that nobody sees a new library
    # beside an old source and compiles again)
    pid = os.getpid()
    main_text = text.split("// ---- second translation unit\n")[0]
    units = [(JIT_DIR / f"ld_spec_{tag}.{pid}.a.hip", main_text, FLAGS + (NO_SINK if narrow(hidden) else []))]
    if keep:
        units.append((JIT_DIR / f"ld_spec_{tag}.{pid}.b.hip", keep_text, FLAGS + NO_SINK))
    mine = JIT_DIR / f"ld_spec_{tag}.{pid}.hip"
    tmp = JIT_DIR / f"libld_spec_{tag}.{pid}.so"
    objs = [u[0].with_suffix(".o") for u in units]
    try:
        for path, unit_text, _ in units:
            path.write_text(unit_text)

        def compile_unit(unit):
            cmd = [hipcc(), *unit[2], "-c", str(unit[0]), "-o", str(unit[0].with_suffix(".o"))]
            if verbose:
                print("[learning-drones_amd] " + " ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(len(units)) as pool:
            list(pool.map(compile_unit, units))
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *map(str, objs)]
        if verbose:
            print("[learning-drones_amd] " + " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        mine.write_text(text)
        os.replace(mine, src)
        os.replace(tmp, lib)
    finally:
        for leftover in [tmp, mine] + objs + [u[0] for u in units]:
            if leftover.exists():
                leftover.unlink()
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
