#!/bin/bash
# Builds a variant of the native library from ONE source compiled with extra flags: pandora_amd/libvar_<name>.so (kept out of
# history like every .so, shipped to the GPU box by gpurun).  Usage: tools/build_variant.sh <name> <file.hip> <flags...>
# A/B on one box: tools/ab_variants.sh <name> ...
set -e
cd "$(dirname "$0")/../pandora_amd/csrc"
name=$1; src=$2; shift 2
make -s -j16 ../libpandora_amd.so
obj=/tmp/var_${name}_${src%.hip}.o
extra=""
case $src in k_sgmfam.hip|k_sgm.hip) extra="-fno-honor-nans";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-value -Wno-unused-result $extra "$@" -c $src -o $obj
objs=""
for f in *.o; do if [ "$f" = "${src%.hip}.o" ]; then objs="$objs $obj"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libvar_${name}.so $objs -ldl
echo "built pandora_amd/libvar_${name}.so"
