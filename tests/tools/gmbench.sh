#!/bin/bash
# usage: tests/tools/gmbench.sh [name "extra -D flags"] ...   builds tests/tools/gmbench[_name].bin (see gmbench.hip)
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -Ilibavif_amd/csrc -w"
[ $# -eq 0 ] && set -- "" ""
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs tests/tools/gmbench.hip -x hip libavif_amd/csrc/gainmap_plan.cpp -o tests/tools/gmbench${name:+_$name}.bin &
done
wait
