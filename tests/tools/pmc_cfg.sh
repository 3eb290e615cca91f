#!/bin/bash
# SQ counter passes over one cfg_bench configuration: pmc_cfg.sh <tag> <cfg_bench name>
set -u
TAG=$1; NAME=$2
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 5 100 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$OUT/pmc_sq1" -o sq1 -- python $R/tests/tools/cfg_bench.py $NAME > "$OUT/sq1.log" 2>&1 < /dev/null
echo "sq1 rc=$?"
timeout -k 5 100 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d "$OUT/pmc_sq2" -o sq2 -- python $R/tests/tools/cfg_bench.py $NAME > "$OUT/sq2.log" 2>&1 < /dev/null
echo "sq2 rc=$?"
cd $R
timeout -k 5 60 python tests/tools/profile_digest.py "$OUT" "$OUT/digest" < /dev/null > /dev/null 2>&1
cat "$OUT/digest_pmc.txt" < /dev/null | head -40
# (third, optional pass: memory-pipeline counters; names from `rocprofv3 -L`, saved once as gpurun_out/<tag>/counters_avail.txt)
cd /tmp
timeout -k 5 60 rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1 < /dev/null
timeout -k 5 100 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d "$OUT/pmc_mem" -o mem -- python $R/tests/tools/cfg_bench.py $NAME > "$OUT/mem.log" 2>&1 < /dev/null
echo "mem rc=$?"; tail -3 "$OUT/mem.log"
cd $R
python - "$OUT" <<'PY'
import glob, sqlite3, sys
from collections import defaultdict
for f in glob.glob(sys.argv[1] + "/pmc_mem/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    acc = defaultdict(lambda: defaultdict(float))
    try:
        for k, c, v, d in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            acc[(k[:60], c)][d] += v
    except Exception as e:
        print("no counters:", e)
    for (k, c), per in sorted(acc.items()):
        xs = list(per.values()); print(k, c, len(xs), sum(xs) / len(xs))
PY
