#!/bin/bash
# fuzz_round6.sh -- the round's last tree under the self-consistency fuzzers and seed rotations (run via gpurun; output: gpurun_out/r06_fuzz.txt)
{
echo "== round 6's last tree (sequence kernels, lossless IDENT kernels, RGB565 / mixed-arithmetic alpha in the tiles, the gain-map computation's search and offsets kernel)"
for s in 611 612; do
  echo "== python tests/tools/fuzz_tiled_vs_generic.py 600 $s"
  timeout 900 python tests/tools/fuzz_tiled_vs_generic.py 600 $s 2>&1 | tail -3
done
echo "== python tests/tools/fuzz_fused_tail.py 400 611"
timeout 900 python tests/tools/fuzz_fused_tail.py 400 611 2>&1 | tail -3
for r in 1 2 3 4 5 6; do
  echo "== pytest tests/test_gainmap.py tests/test_gpu_sequence.py -m gpu --seed-rotation $r"
  timeout 600 python -m pytest tests/test_gainmap.py tests/test_gpu_sequence.py -m gpu -q -p no:cacheprovider --seed-rotation $r 2>&1 | tail -1
done
echo "== the whole GPU tier under --seed-rotation 21"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --seed-rotation 21 2>&1 | tail -2
} > gpurun_out/r06_fuzz.txt 2>&1
cat gpurun_out/r06_fuzz.txt | cut -c1-250
