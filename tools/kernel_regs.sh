#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel of a HIP object file (from its gfx950 code object's metadata).
# Usage: tools/kernel_regs.sh pandora_amd/csrc/k_sgmfam8.o [name filter]
LLVM=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat && $LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co
$LLVM/llvm-readelf --notes $T/co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.agpr_count:/{a=$2} /\.sgpr_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.group_segment_fixed_size:/{l=$2} /\.wavefront_size:/{print n, "vgpr", v, "agpr", a, "sgpr", s, "scratch", p, "lds", l}' | grep "${2:-.}"
rm -rf $T
