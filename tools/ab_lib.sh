#!/bin/bash
# A/B of two builds of the native library on ONE box: pandora_amd/libpandora_amd.so (base) against pandora_amd/libpandora_amd_exp.so
# (build it from edited sources: make -C pandora_amd/csrc && cp pandora_amd/libpandora_amd.so pandora_amd/libpandora_amd_exp.so, then
# restore the sources and rebuild the base).  Usage (on the GPU box): bash tools/ab_lib.sh [bench_tiles arguments]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS=${@:---only headline --ranks 1}
cp pandora_amd/libpandora_amd.so /tmp/base.so
for i in 1 2 3; do
  cp /tmp/base.so pandora_amd/libpandora_amd.so; echo "base"; python tools/bench_tiles.py $ARGS 2>&1 | grep -v "^#\|^ranks" | cut -c1-230
  cp pandora_amd/libpandora_amd_exp.so pandora_amd/libpandora_amd.so; echo "exp"; python tools/bench_tiles.py $ARGS 2>&1 | grep -v "^#\|^ranks" | cut -c1-230
done
cp /tmp/base.so pandora_amd/libpandora_amd.so
