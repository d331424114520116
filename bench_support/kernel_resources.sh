#!/bin/bash
# Registers, spills, scratch and LDS of the matcher's kernel instances, from the code object's metadata:
#   bench_support/kernel_resources.sh [W=4] [filter-regex]
W=${1:-4}; PAT=${2:-.}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/rh_w${W}.co
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$ROOT/real_amd/csrc -ffp-contract=off \
    -DRH_W=$W --offload-device-only --no-gpu-bundle-output -c $ROOT/real_amd/csrc/match_kernel.hip -o $OUT 2>/dev/null || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT | awk '
/\.group_segment_fixed_size:/ {lds=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2}
/\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vs=$2} /\.agpr_count:/ {ag=$2}
/\.name:/ {name=$2}
/\.wavefront_size:/ {printf "%-60s vgpr %3d agpr %3d vgpr_spill %3d sgpr %3d sgpr_spill %3d scratch %4d lds %6d\n", name, vg, ag, vs, sg, ss, scr, lds}' | c++filt | grep -E "$PAT"
