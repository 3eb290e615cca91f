#!/bin/bash
# usage: [PKB_SRC=pkbench_wide.hip] tests/tools/pkbench.sh name "extra -D flags" ...   (pairs); builds one binary per pair into tests/tools/pkb_<name>.bin
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude -Ilibavif_amd/csrc -w"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs tests/tools/${PKB_SRC:-pkbench.hip} -x hip libavif_amd/csrc/plan.cpp -o tests/tools/pkb_$name.bin &
done
wait
