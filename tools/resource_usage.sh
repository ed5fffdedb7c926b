#!/bin/bash
# Diagnostic: register use / occupancy of every kernel in one HIP source (compile only, no GPU)
# usage: tools/resource_usage.sh icp_slam_prototype_amd/csrc/kernels_grid.hip [extra hipcc flags]
src=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
inc=${ICPK_CSRC:-$root/icp_slam_prototype_amd/csrc}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -I "$root/include" -I "$inc" "$@" -x hip -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/ \[-Rpass[^]]*\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name, "vgpr", v, "scratch", s, "occ", o, "lds", $NF}' |
  c++filt | sort -u
