#!/bin/bash
# evidence_round6.sh [tag] -- round 6's committed figures from ONE box (run via gpurun; everything lands in gpurun_out/, copy what is to be judged into profiles/):
#   <tag>_bench_*                bench.py --headline-only under rocprofv3 --kernel-trace --stats, the HBM-traffic passes, the SQ counter passes (profile_round.sh)
#   <tag>_bench_line_*.json      bench.py as the driver runs it (default flags; --steps 20 --warmup 5), and with every multi-GPU block on a one-GPU box
#   <tag>_4k_rows.jsonl         cfg_bench.py's 4K rows without the profiler (what the bench line's planes_4k block is compared with)
#   <tag>_sequence.txt / .json   sequence launches and their one-frame-per-launch twins in the HBM regime: HIP events, rocprofv3 averages, FETCH / WRITE traffic (seq_evidence.sh)
#   <tag>_cfgs_*                 cfg_bench.py configurations, each run once under rocprofv3 (event-timed row and profiler average from the same launches)
#   <tag>_gainmap_compute.txt    the gain-map computation's kernels (device-resident call), <tag>_gainmap_pmc.txt the apply kernel's counters
#   <tag>_tsan.txt               the concurrency stress under ThreadSanitizer / AddressSanitizer (sanitizer_run.sh)
#   <tag>_fault_regression.txt   farm -> gain maps -> farm, the two files of round 5's fault in the order that died, 20 fresh processes
# EVIDENCE_ONLY="bench seq cfgs gainmap san fault e2e generic" runs a subset (default: all); EVIDENCE_CFGS="cfg ..." replaces the configuration list
set -u
TAG=${1:-r06}
R=$PWD
mkdir -p gpurun_out
want() { [ -z "${EVIDENCE_ONLY:-}" ] || [[ " $EVIDENCE_ONLY " == *" $1 "* ]]; }
if want bench; then
bash tests/tools/profile_round.sh "$TAG" > "gpurun_out/${TAG}_profile_round.log" 2>&1
python bench.py > "gpurun_out/${TAG}_bench_line_default_run.json" 2> "gpurun_out/${TAG}_bench_default.err"
python bench.py --steps 20 --warmup 5 > "gpurun_out/${TAG}_bench_line_driver_flags.json" 2>> "gpurun_out/${TAG}_bench_default.err"
AVIFHIP_BENCH_ALL_BLOCKS=1 AVIFHIP_BENCH_DEVICES=0,0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "gpurun_out/${TAG}_bench_line_all_blocks_one_gpu.json" 2>> "gpurun_out/${TAG}_bench_default.err"
fi
if want bench; then
# the 4K rows WITHOUT the profiler (the configuration table below times its rows under rocprofv3, whose interception adds up to a microsecond to
# launches this short: table and bench line are compared on these -- VERDICT r05 item 7)
python tests/tools/cfg_bench.py cfg2_4k cfg2_4k_cold cfg2_4k_seq cfg2_4k_seq_cold 2>/dev/null | grep '^{' > "gpurun_out/${TAG}_4k_rows.jsonl"
fi
if want seq; then
bash tests/tools/seq_evidence.sh "$TAG" > "gpurun_out/${TAG}_seq_evidence.log" 2>&1
fi
CFGS="cfg2 cfg2_cold cfg2_seq cfg2_seq_cold cfg2_4k cfg2_4k_seq cfg2_4k_cold cfg2_4k_seq_cold cfg2n cfg2_rgb cfg2_565 cfg2_565_odd cfg2_565_alpha cfg2_565_10 cfg2_alpha cfg2_premul cfg2_unpremul cfg3 cfg3_unpremul cfg4 cfg4_cycled cfg4_seq cfg4rgb cfg4_601 cfg4_8k cfg4rgb_8k cfg4_444_8k cfg4_premul_8k cfg4_unpremul_8k cfg4_ycgco_8k ident8_enc gray_enc_8k graya_enc_8k cfg5 cfg5_8 cfg5x64 cfg5x64_8 f16_420 f16_444a ident8 ident8rgb gray8 graya16 premul8 unpremul8 unpremul16 tail0 tail180 tail90 tail90_two_pass tail0_10 tail90_10 tail0_rgba10 tail180_rgba10 tail90_rgba10 tail90_rgba10_two_pass cfg5grid cfg5grid_link cfg5grid_8 cfg5grid_8_pass photo_grid photo_grid_pass cfg2_keep cfg2_keep16 cfg5x64_rot xform90 xform180 scale_box4 scale_up2 scale_down_1_5 gainmap4k gainmap4k_photo gainmap4k_same gainmap4k_half gmcompute4k gmcompute4k_dev"
CFGS=${EVIDENCE_CFGS:-$CFGS}
if want cfgs; then
bash tests/tools/profile_cfgs.sh "$TAG" $CFGS > "gpurun_out/${TAG}_profile_cfgs.log" 2>&1
for c in $CFGS; do cat "gpurun_out/${TAG}_cfgs/$c.jsonl" 2>/dev/null | grep '^{' ; done > "gpurun_out/${TAG}_cfgs_bench.jsonl"
python - "$TAG" > "gpurun_out/${TAG}_gainmap_compute.txt" <<'PY'
import sys
tag = sys.argv[1]
keep, out = False, []
for line in open(f"gpurun_out/{tag}_cfgs_kernel_stats.txt"):
    if line.startswith("== "):
        keep = line.split()[1] in ("gmcompute4k", "gmcompute4k_dev", "gainmap4k", "gainmap4k_half")
    if keep or line.startswith("rocprofv3"):
        out.append(line)
sys.stdout.write("".join(out))
PY
fi
if want gainmap; then
bash tests/tools/pmc_cfg.sh "${TAG}_gainmap" gainmap4k > "gpurun_out/${TAG}_gainmap_pmc.log" 2>&1
cp "gpurun_out/${TAG}_gainmap/digest_pmc.txt" "gpurun_out/${TAG}_gainmap_pmc.txt" 2>/dev/null
# ... and on a photograph-like pair (neighbouring pixels hold neighbouring codes): what the table gathers' bank conflicts are on real images
bash tests/tools/pmc_cfg.sh "${TAG}_gainmap_photo" gainmap4k_photo > "gpurun_out/${TAG}_gainmap_photo_pmc.log" 2>&1
cp "gpurun_out/${TAG}_gainmap_photo/digest_pmc.txt" "gpurun_out/${TAG}_gainmap_photo_pmc.txt" 2>/dev/null
fi
if want san; then
bash tests/tools/sanitizer_run.sh "$TAG" 10 > /dev/null 2>&1
fi
if want fault; then
{ echo "== pytest tests/test_gpu_device_farm.py tests/test_gainmap.py -m gpu (the order that died in round 5), 20 fresh processes, scratch poisoned =="
  for k in $(seq 1 20); do
    timeout 300 python -m pytest tests/test_gpu_device_farm.py tests/test_gainmap.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/run $k: /"
  done; } > "gpurun_out/${TAG}_fault_regression.txt" 2>&1
fi
if want e2e; then
python tests/tools/e2e_bench.py > "gpurun_out/${TAG}_e2e.jsonl" 2> "gpurun_out/${TAG}_e2e.err"
fi
if want generic; then
for r in 0 1 2 3; do echo "== rotation $r"; AVIFHIP_TEST_SEED_ROTATION=$r python tests/tools/list_generic.py 2>&1; done > "gpurun_out/${TAG}_generic_rest.txt"
fi
rm -rf "gpurun_out/${TAG}_cfgs" "gpurun_out/$TAG" "gpurun_out/${TAG}_gainmap" "gpurun_out/${TAG}_gainmap_photo"
ls -la gpurun_out | tail -30
