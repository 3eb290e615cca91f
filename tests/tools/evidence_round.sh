#!/bin/bash
# evidence_round.sh <tag> -- run on the GPU box (via gpurun) in ONE call, so that every committed figure of a round comes from the same box:
#   <tag>_bench_*            bench.py under rocprofv3 --kernel-trace --stats, the two HBM-traffic passes, the SQ counter passes (profile_round.sh)
#   <tag>_bench_line_*.json  bench.py as the driver runs it (default flags; --steps 20 --warmup 5) and the cfg5 workload
#   <tag>_cfgs_kernel_stats.txt + <tag>_cfgs_bench.jsonl   cfg_bench.py configurations, each run ONCE under rocprofv3: the event-timed row and
#                            the profiler's average come from the same launches (tests/test_profiles.py holds them to 5 %)
#   <tag>_cfgs_pmc.txt       instruction counters of the fp32 kernels the round worked on
#   <tag>_gainmap_*          the gain-map application: kernel durations, instruction counters, probe timings; <tag>_valu_rate.txt
#   <tag>_e2e.jsonl          host-to-host rows (C ABI and seam B)
# Everything lands in gpurun_out/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r05}
R=$PWD
mkdir -p gpurun_out
bash tests/tools/profile_round.sh "$TAG" > "gpurun_out/${TAG}_profile_round.log" 2>&1
python bench.py > "gpurun_out/${TAG}_bench_line_default_run.json" 2> "gpurun_out/${TAG}_bench_default.err"
python bench.py --steps 20 --warmup 5 > "gpurun_out/${TAG}_bench_line_driver_flags.json" 2>> "gpurun_out/${TAG}_bench_default.err"
python bench.py --workload cfg5 --no-cpu-baseline > "gpurun_out/${TAG}_bench_cfg5_line.json" 2>> "gpurun_out/${TAG}_bench_default.err"
CFGS="cfg2 cfg2_4k cfg2n cfg2_rgb cfg2_565 cfg2_alpha cfg2_premul cfg2_unpremul cfg3 cfg3_unpremul cfg4 cfg4rgb cfg4_601 cfg4_8k cfg4_premul_8k cfg4_unpremul_8k cfg4_ycgco_8k ident8_enc gray_enc_8k graya_enc_8k cfg5 cfg5_8 cfg5x64 cfg5x64_8 f16_420 f16_444a ident8 ident8rgb gray8 graya16 premul8 unpremul8 unpremul16 tail0 tail180 tail90 tail90_two_pass tail0_10 tail90_10 tail0_rgba10 tail180_rgba10 tail90_rgba10 tail90_rgba10_two_pass cfg5grid cfg5grid_link cfg5grid_8 cfg5grid_8_pass photo_grid photo_grid_pass cfg2_keep cfg2_keep16 cfg5x64_rot xform90 xform180 scale_box4 scale_up2 scale_down_1_5 gainmap4k gainmap4k_half gmcompute4k"
bash tests/tools/profile_cfgs.sh "$TAG" $CFGS > "gpurun_out/${TAG}_profile_cfgs.log" 2>&1
# the event-timed rows of those very runs, one file
for c in $CFGS; do cat "gpurun_out/${TAG}_cfgs/$c.jsonl" 2>/dev/null | grep '^{' ; done > "gpurun_out/${TAG}_cfgs_bench.jsonl"
bash tests/tools/pmc_cfgs.sh "$TAG" cfg2 cfg2_premul cfg2_unpremul cfg3 cfg4 cfg4_8k > "gpurun_out/${TAG}_pmc_cfgs.log" 2>&1
# the gain-map application on its own: kernel durations of the 4K case (same runs as the rows above), its instruction counters, the kernel with one
# of its parts taken out (tests/tools/gmbench.hip), and the issue rates its instruction mix is priced with
python - "$TAG" > "gpurun_out/${TAG}_gainmap_kernel_stats.txt" <<'PY'
import sys
tag = sys.argv[1]
keep, out = False, []
for line in open(f"gpurun_out/{tag}_cfgs_kernel_stats.txt"):
    if line.startswith("== "):
        keep = line.split()[1] in ("gainmap4k", "gainmap4k_half", "gmcompute4k")
    if keep or line.startswith("rocprofv3"):
        out.append(line)
sys.stdout.write("".join(out))
PY
bash tests/tools/pmc_cfg.sh "${TAG}_gainmap" gainmap4k > "gpurun_out/${TAG}_gainmap_pmc.log" 2>&1
cp "gpurun_out/${TAG}_gainmap/digest_pmc.txt" "gpurun_out/${TAG}_gainmap_pmc.txt" 2>/dev/null
{ echo "== tests/tools/gmbench.bin (probe masks: 4 no locator, 8 no fp64 matrix, 16 no base / gain / alpha tables, 32 no stores, 64 no table copy; HIP events, 25 launches)";
  tests/tools/gmbench.bin 4 8 16 32 64 124; echo "== the kernel as the library builds it (no probes), 4K and 4 x 4K rows"; tests/tools/gmbench_np.bin; GM_H=8640 tests/tools/gmbench_np.bin; } > "gpurun_out/${TAG}_gainmap_probes.txt" 2>&1
tests/tools/valu_rate.bin > "gpurun_out/${TAG}_valu_rate.txt" 2>&1
python tests/tools/e2e_bench.py > "gpurun_out/${TAG}_e2e.jsonl" 2> "gpurun_out/${TAG}_e2e.err"
# round 5: the configurations that stream under the tuning knobs with the byte-movement ceiling of each shape (one process: the rows compare), the
# interleaved A/Bs behind the round's launch rules, HBM traffic and wait counters of cfg3 / cfg4, and the in-process device farm (a C consumer
# checking itself against the oracle; the host-to-host bench lines of one process driving two workers)
S=tests/tools/stream_sweep.py
timeout 600 python $S cfg3 cfg3same f16_444a cfg4 cfg5x64 cfg5grid cfg5grid8 cfg2cold cfg2cold_fp32 cfg2warm cfg2warm_fp32 batch1080 > "gpurun_out/${TAG}_stream_sweep.jsonl" 2> "gpurun_out/${TAG}_stream_sweep.err"
{ timeout 200 python $S ab cfg2cold 0x1 0x401; timeout 200 python $S ab cfg2warm 0x1 0x401;
  timeout 300 python $S ab cfg5grid8 0x1 0x1000001 0x1000201 0x201; timeout 300 python $S ab cfg5grid 0x1 0x1000001 0x1000009 0x2000001;
  timeout 200 python $S ab photo_grid 0x1 0x1000201 0x2000001; } > "gpurun_out/${TAG}_stream_sweep_ab.jsonl" 2>> "gpurun_out/${TAG}_stream_sweep.err"
bash tests/tools/traffic_cfgs.sh "$TAG" cfg3 cfg4 > "gpurun_out/${TAG}_traffic.log" 2>&1
rm -rf "gpurun_out/${TAG}_traffic"
{ for w in "7680 4320 8 cfg2" "15360 8640 10 cfg5" "3840 2160 8 cfg4" "4099 3001 8 cfg2 0" "4099 3001 8 cfg4 0"; do
    AVIFHIP_DEVICES=0,0 timeout 300 tests/c/farm_check $w; echo "exit $?"; timeout 300 tests/c/farm_check $w; echo "exit $?"; done; } > "gpurun_out/${TAG}_farm_check.txt" 2>&1
AVIFHIP_BENCH_DEVICES=0,0 timeout 300 python bench.py --in-process --gpus 2 > "gpurun_out/${TAG}_bench_inprocess_cfg2_line.json" 2>> "gpurun_out/${TAG}_bench_default.err"
AVIFHIP_BENCH_DEVICES=0,0 timeout 300 python bench.py --in-process --gpus 2 --workload cfg5 > "gpurun_out/${TAG}_bench_inprocess_cfg5_line.json" 2>> "gpurun_out/${TAG}_bench_default.err"
timeout 300 python tests/tools/list_generic.py > "gpurun_out/${TAG}_generic_rest.txt" 2>&1
# the gain map's conversion inside the apply kernel against the conversion launch + RGBA copy (interleaved in one process), the host-resident calls
# with their phases, and the encode kernels' strips-per-wave rule with every pattern of the byte-movement ceiling
timeout 300 python tests/tools/gm_call_bench.py 5 > "gpurun_out/${TAG}_gainmap_call_ab.jsonl" 2> "gpurun_out/${TAG}_gainmap_call_ab.err"
AVIFHIP_GAINMAP_TRACE=1 timeout 300 python tests/tools/gm_trace.py > "gpurun_out/${TAG}_gainmap_host_trace.txt" 2>&1
AVIFHIP_CEILING_TRACE=1 timeout 300 python tests/tools/spw_ab.py > "gpurun_out/${TAG}_encode_strips_ab.txt" 2>&1
rm -rf "gpurun_out/${TAG}_cfgs" "gpurun_out/${TAG}_pmc_cfgs" "gpurun_out/$TAG" "gpurun_out/${TAG}_gainmap"
ls -la gpurun_out | tail -20
