#!/bin/bash
# A/B of library variants on ONE box (tools/build_variant.sh): the base library and pandora_amd/libvar_<name>.so alternated REPS times.
# Usage (GPU box): [REPS=2] [CMD="python tools/bench_tiles.py --only headline --ranks 1"] bash tools/ab_variants.sh <name> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD=${CMD:-python tools/bench_tiles.py --only headline --ranks 1}
cp pandora_amd/libpandora_amd.so /tmp/base.so
for i in $(seq 1 ${REPS:-2}); do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/base.so pandora_amd/libpandora_amd.so; else cp pandora_amd/libvar_$v.so pandora_amd/libpandora_amd.so; fi
    echo "== $v"; timeout 300 $CMD 2>&1 | grep -v "^#\|^ranks" | cut -c1-400
  done
done
cp /tmp/base.so pandora_amd/libpandora_amd.so
