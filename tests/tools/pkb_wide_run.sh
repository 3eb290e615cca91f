#!/bin/bash
# runs the cfg5x64_8 ablation binaries built by: PKB_SRC=pkbench_wide.hip tests/tools/pkbench.sh wfull "" wnomatrix -DAVIFHIP_ABLATE_MATRIX ...
cd "$(dirname "$0")/../.."
for v in wfull wnomatrix wnofilter wnostage wnomf wnone wnearest wfull2 wnone2; do for geo in "0 0" "0 1" "1 0" "2 0"; do timeout 60 tests/tools/pkb_$v.bin "$v wavesXLog2,chunk=$geo" $geo; done; done
