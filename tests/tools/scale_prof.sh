#!/bin/bash
# kernel-level timing of the side-path kernels (plane scaling, transforms): wall clock of the API calls, then one rocprofv3
# kernel trace per configuration, digested by tests/tools/profile_digest.py.  Usage: scale_prof.sh <tag> <cfg_bench names...>
set -u
TAG=${1:-scale}; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 5 120 python tests/tools/cfg_bench.py "$@" > "$OUT/bench.txt" 2>&1 < /dev/null
echo "bench rc=$?"; tail -8 "$OUT/bench.txt"
export TMPDIR=/tmp
for name in "$@"; do
    (cd /tmp && timeout -k 5 100 rocprofv3 --kernel-trace --stats -d "$OUT/$name/stats" -o stats -- python $R/tests/tools/cfg_bench.py $name > "$OUT/$name.log" 2>&1 < /dev/null)
    echo "$name rocprof rc=$?"
    timeout -k 5 60 python tests/tools/profile_digest.py "$OUT/$name" "$OUT/$name" < /dev/null > /dev/null 2>&1
    head -6 "$OUT/${name}_kernel_stats.txt" < /dev/null | cut -c1-60,105-180
done
