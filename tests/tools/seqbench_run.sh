#!/bin/bash
# the sequence-launch sweep of tests/tools/seqbench.hip on the GPU box (binaries built here by tests/tools/pkbench.sh with PKB_SRC=seqbench.hip)
cd "$(dirname "$0")"
F=${1:-4}
for v in n2 n1w8 n1w8s n2w8 n1w16; do
  b=./pkb_sq_$v.bin
  [ -x $b ] || continue
  for geo in "0 1" "0 0" "0 2" "0 4"; do
    set -- $geo
    $b "$v wavesXlog2=$1 chunkRows=$2 F=$F" $1 $2 $F
  done
done
