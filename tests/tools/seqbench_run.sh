#!/bin/bash
# the sequence-launch sweep of tests/tools/seqbench.hip on the GPU box (binaries built here by tests/tools/pkbench.sh with PKB_SRC=seqbench.hip)
cd "$(dirname "$0")"
for F in 4; do
for v in n2 n2s n4s n1 n1s; do
  b=./pkb_sq_$v.bin
  [ -x $b ] || continue
  for geo in "0 1" "0 0" "1 0" "1 1" "2 0" "0 2"; do
    set -- $geo
    $b "$v wavesXlog2=$1 chunkRows=$2 F=$F" $1 $2 $F
  done
  $b "$v private halo, raster F=$F" 0 0 $F 0x10
  $b "$v private halo, wavesXlog2=1 raster F=$F" 1 0 $F 0x10
done
done
