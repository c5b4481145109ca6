#!/bin/bash
# Registers / LDS / scratch of every kernel in one HIP source (cross-compiles for gfx950; no GPU needed).
# usage: bash tools/kernel_regs.sh backscrub_amd/csrc/kernels_nn.hip [grep pattern]
SRC=$1; PAT=${2:-.}
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$(dirname $SRC) -I$(dirname $SRC)/../../include --cuda-device-only -S -o $OUT/k.s $SRC 2>/dev/null || { echo "compile failed"; exit 1; }
awk '/^[ \t]*\.amdhsa_kernel /{name=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_accum_offset/{a=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{s=$2} /^[ \t]*\.end_amdhsa_kernel/{printf "%-110s vgpr+agpr %4s accum_offset %4s lds %6s scratch %5s\n", substr(name,1,110), v, a, l, s}' $OUT/k.s | grep -E "$PAT"
echo "asm: $OUT/k.s"
