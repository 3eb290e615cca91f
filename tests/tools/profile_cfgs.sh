#!/bin/bash
# profile_cfgs.sh <tag> [cfg ...] -- run on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats around tests/tools/cfg_bench.py, one
# configuration per run (several configurations share kernel instantiations), digested into gpurun_out/<tag>_cfgs_kernel_stats.txt:
# per configuration the kernels by total time with calls / average / min / max duration.  The batch configurations (cfg5x64*) are timed by
# cfg_bench.py with a host clock around API calls; this file holds their kernels' own durations.
set -u
TAG=${1:-r02}; shift
CFGS=${@:-cfg2 cfg2n cfg3 cfg4 cfg4rgb cfg4_601 cfg5 cfg5_8 cfg5x64 cfg5x64_8 cfg2_565 f16_420 f16_444a ident8 ident8rgb}
R=$PWD
OUT=$R/gpurun_out/${TAG}_cfgs
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  rocprofv3 --kernel-trace --stats -d "$OUT/$c" -o stats -- python $R/tests/tools/cfg_bench.py $c > "$OUT/$c.jsonl" 2> "$OUT/$c.log"
done
cd "$R"
python - "$OUT" $CFGS > "gpurun_out/${TAG}_cfgs_kernel_stats.txt" <<'PY'
import sys
sys.path.insert(0, "tests/tools")
from profile_digest import kernel_stats
import os
out = sys.argv[1]
print("rocprofv3 --kernel-trace --stats -- python tests/tools/cfg_bench.py <cfg>   (one run per configuration; durations in microseconds)")
for c in sys.argv[2:]:
    stats = kernel_stats(os.path.join(out, c))
    print(f"\n== {c}")
    try:
        for line in open(os.path.join(out, c + ".jsonl")):
            print("   cfg_bench: " + line.strip()[:230])
    except OSError:
        pass
    print(f"   {'kernel':118s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print(f"   {k[:118]:118s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f}")
PY
find "$OUT" -name "*.db" -delete
cat "gpurun_out/${TAG}_cfgs_kernel_stats.txt" | cut -c1-200
