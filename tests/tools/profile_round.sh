#!/bin/bash
# profile_round.sh <tag> -- run on the GPU box (via gpurun): collects for `python bench.py` (the bench command itself)
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE ...          -> HBM read traffic   (own pass, no trace domains besides kernel-trace)
#   3. rocprofv3 --pmc WRITE_SIZE ...          -> HBM write traffic
#   4. rocprofv3 --pmc SQ_* (two passes)       -> instruction mix / stall picture of the dominant kernel
# and writes gpurun_out/<tag>/*.db plus gpurun_out/<tag>/bench.json.  tests/tools/profile_digest.py turns the
# databases into the text/JSON summaries committed under profiles/.
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1000 --warmup 100 --repeats 3 --streams 1 --no-second-stream-count --no-cpu-baseline --headline-only"
SHORT="python $R/bench.py --steps 300 --warmup 100 --repeats 1 --preheat-ms 50 --streams 1 --no-second-stream-count --no-cpu-baseline --headline-only"   # counter passes: smaller databases
$BENCH > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o fetch -- $SHORT > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o write -- $SHORT > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$OUT/pmc_sq1" -o sq1 -- $SHORT > /dev/null 2> "$OUT/pmc_sq1.log"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA -d "$OUT/pmc_sq2" -o sq2 -- $SHORT > /dev/null 2> "$OUT/pmc_sq2.log"
# digest on the box, keep the text, drop the databases (gpurun_out/ travels back only below 64 MiB)
cd "$R"
python tests/tools/profile_digest.py "$OUT" "gpurun_out/${TAG}_bench" > "$OUT/digest.log" 2>&1
cp "$OUT/bench_plain.json" "gpurun_out/${TAG}_bench_line.json"
cp "$OUT/bench_stats.json" "gpurun_out/${TAG}_bench_line_under_rocprof.json"
find "$OUT" -name "*.db" -delete
tail -5 "$OUT/digest.log"
cut -c1-600 "$OUT/bench_plain.json"
