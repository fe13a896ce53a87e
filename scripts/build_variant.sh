#!/bin/bash
# A variant of libld_drones.so with ONE translation unit rebuilt with extra -D flags, for A/B runs on the GPU box
# (LD_DRONES_LIB=variants/libld_<name>.so python3 scripts/exp_*.py):
#   scripts/build_variant.sh <name> <translation unit stem, e.g. ld_split_keep_32_32> [-DLD_WIDE_REG_WORDS=38 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; tu=$2; shift 2
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function"
case $tu in ld_split_keep_*) FLAGS="$FLAGS -mllvm -simplifycfg-sink-common=false";; esac
hipcc $FLAGS "$@" -c learning-drones_amd/csrc/$tu.hip -o variants/${name}_$tu.o
objs=$(ls learning-drones_amd/build/*.o | grep -v "/$tu.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libld_$name.so $objs variants/${name}_$tu.o
echo variants/libld_$name.so
