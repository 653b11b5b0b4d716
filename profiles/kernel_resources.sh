#!/bin/bash
# Register, spill, scratch and LDS figures of every kernel in rt_kernels.hip, from the code object
# metadata of a device-only compile with the library's own flags.
# Usage: bash profiles/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -O1 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 --cuda-device-only -S \
    "$ROOT/raytracing-in-one-weekend_amd/csrc/rt_kernels.hip" -o "$TMP/k.s" 2>/dev/null
echo "# rt_kernels.hip sha256 $(sha256sum "$ROOT/raytracing-in-one-weekend_amd/csrc/rt_kernels.hip" | cut -c1-16)"
awk '/^amdhsa.kernels:/,/^amdhsa.target:/' "$TMP/k.s" | awk '
  /\.name:/ {name=$2}
  /\.group_segment_fixed_size:/ {lds=$2}
  /\.private_segment_fixed_size:/ {scratch=$2}
  /\.sgpr_count:/ {sg=$2}
  /\.sgpr_spill_count:/ {ss=$2}
  /\.vgpr_count:/ {vg=$2}
  /\.vgpr_spill_count:/ {vs=$2; printf "%-70s vgpr %3d  vgpr_spill %3d  sgpr %3d  sgpr_spill %3d  scratch %4d B  static_lds %5d B\n", name, vg, vs, sg, ss, scratch, lds}'
rm -rf "$TMP"
