#!/usr/bin/env python3
"""Every kernel in libld_drones.so (or the shared object given) with its VGPR count and scratch bytes per lane; exit
code 1 if a float32 fused-step kernel the launch policy can pick uses scratch.  Reads the code objects out of the
library's .hip_fatbin section (objcopy, clang-offload-bundler, llvm-readelf --notes)."""
import os, re, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "learning-drones_amd", "libld_drones.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "--dump-section", f".hip_fatbin={fat}", lib])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", data)]
    rows = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        one, co = os.path.join(tmp, f"b{n}.bin"), os.path.join(tmp, f"b{n}.co")
        open(one, "wb").write(data[a:b])
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={one}", f"--output={co}"], capture_output=True)
        if not os.path.exists(co):
            continue
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if name:
                demangled = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                rows.append((demangled.split("(")[0].replace("void ld::", ""), int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                             int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))))
bad = 0
for name, vgpr, scratch in sorted(rows):
    print(f"{name:80s} vgpr {vgpr:3d} scratch {scratch}")
    bad += scratch > 64 and name.startswith("fly_split_kernel<float")
print(f"{len(rows)} kernels, {sum(1 for r in rows if r[2])} with scratch")
sys.exit(1 if bad else 0)
