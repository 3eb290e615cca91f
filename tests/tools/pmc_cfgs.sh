#!/bin/bash
# pmc_cfgs.sh <tag> <cfg ...> -- run on the GPU box (via gpurun): instruction counters of the kernels behind cfg_bench.py configurations
# (own rocprofv3 pass: --kernel-trace --pmc only), digested into gpurun_out/<tag>_cfgs_pmc.txt as per-dispatch averages.
set -u
TAG=${1:-r02}; shift
R=$PWD
OUT=$R/gpurun_out/${TAG}_pmc_cfgs
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  AVIFHIP_BENCH_PREHEAT_MS=5 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d "$OUT/$c" -o pmc -- python $R/tests/tools/cfg_bench.py $c > /dev/null 2> "$OUT/$c.log"
done
cd "$R"
python - "$OUT" "$@" > "gpurun_out/${TAG}_cfgs_pmc.txt" <<'PY'
import os, sys
sys.path.insert(0, "tests/tools")
from profile_digest import counters
out = sys.argv[1]
print("rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -- python tests/tools/cfg_bench.py <cfg>")
print("(per-dispatch averages; wave-instructions: x 64 lanes / pixels of the configuration = instructions per pixel)")
for c in sys.argv[2:]:
    print(f"\n== {c}")
    acc = counters(os.path.join(out, c))
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", {}).values()))[:3]:
        print("   " + k[:150])
        for name, per in sorted(cs.items()):
            print(f"      {name:20s} dispatches={len(per):6d} avg_per_dispatch={sum(per.values()) / len(per):16.1f}")
PY
find "$OUT" -name "*.db" -delete
cat "gpurun_out/${TAG}_cfgs_pmc.txt" | cut -c1-170
