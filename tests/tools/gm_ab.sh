#!/bin/bash
# gm_ab.sh -- gain-map computation, this build against gpurun_in/libavifhip_prev.so on ONE box: parity first, then interleaved cfg_bench rows,
# the device-resident call's phases, and the kernels' own durations under rocprofv3 for both builds
python -m pytest tests/test_gainmap.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in new prev; do
    if [ $lib = prev ]; then export AVIFHIP_BENCH_LIB=gpurun_in/libavifhip_prev.so; else unset AVIFHIP_BENCH_LIB; fi
    python tests/tools/cfg_bench.py gmcompute4k_dev gmcompute4k 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('   $lib', r['config'], r['us'])"
  done
done
unset AVIFHIP_BENCH_LIB
AVIFHIP_GAINMAP_TRACE=1 python tests/tools/cfg_bench.py gmcompute4k_dev 2>&1 | grep "avifhip compute" | tail -3
bash tests/tools/profile_cfgs.sh gmab_new gmcompute4k_dev > /dev/null 2>&1
AVIFHIP_BENCH_LIB=$PWD/gpurun_in/libavifhip_prev.so bash tests/tools/profile_cfgs.sh gmab_prev gmcompute4k_dev > /dev/null 2>&1
for t in new prev; do echo "== $t"; cut -c1-190 gpurun_out/gmab_${t}_cfgs_kernel_stats.txt | tail -6; done
