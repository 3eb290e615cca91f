#!/bin/bash
# seq_evidence.sh <tag> [cfg ...] -- run on the GPU box (via gpurun): the sequence launches (round 6) and their one-frame-per-launch twins where
# every byte comes from and goes to HBM, each configuration as `stream_sweep.py run <cfg>` (nothing but that kernel's launches) in FOUR runs:
# no profiler (HIP events), rocprofv3 --kernel-trace --stats (the profiler's own average duration), --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate
# passes).  Digest: gpurun_out/<tag>_sequence.txt (per configuration: event-timed row, profiler average / min / max, fraction of 8 TB/s on the
# profiler's average, HBM bytes per launch against the algorithmic bytes) and gpurun_out/<tag>_sequence.json (the same, for bench.py / tests).
set -u
TAG=${1:-r06}; shift
CFGS=${@:-cfg2seq cfg2cold cfg2seq_fp32 cfg2cold_fp32 4kseq 4kcold 4kseq_fp32 4kcold_fp32 cfg4seq cfg4}
R=$PWD
OUT=$R/gpurun_out/${TAG}_sequence_runs
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
S="python $R/tests/tools/stream_sweep.py run"
for c in $CFGS; do
  $S $c > "$OUT/$c.row.json" 2> "$OUT/$c.row.err"
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$OUT/$c/stats" -o stats -- $S $c > "$OUT/$c.row_under_rocprof.json" 2> "$OUT/$c.stats.log"
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/$c/fetch" -o fetch -- $S $c > /dev/null 2> "$OUT/$c.fetch.log"
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/$c/write" -o write -- $S $c > /dev/null 2> "$OUT/$c.write.log"
done
cd "$R"
python - "$OUT" "gpurun_out/${TAG}_sequence.json" $CFGS > "gpurun_out/${TAG}_sequence.txt" <<'PY'
import json, os, sys
sys.path.insert(0, "tests/tools")
from profile_digest import counters, kernel_stats
out, jpath = sys.argv[1], sys.argv[2]
print("python tests/tools/stream_sweep.py run <cfg>: (1) no profiler, HIP events; (2) rocprofv3 --kernel-trace --stats; (3) --kernel-trace --pmc FETCH_SIZE; (4) --kernel-trace --pmc WRITE_SIZE")
print("bytes: FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte requests at 64, MI355X_MICROARCH.md), WRITE_SIZE x 1024; fractions of 8 TB/s on the ALGORITHMIC bytes of a launch")
result = {}
for c in sys.argv[3:]:
    def row_of(suffix):
        try:
            return json.loads(open(os.path.join(out, c + suffix)).read().strip().splitlines()[-1])
        except Exception:
            return {}
    row, prow = row_of(".row.json"), row_of(".row_under_rocprof.json")
    print(f"\n== {c}: {row.get('config', '')}")
    print(f"   HIP events, no profiler : {row.get('us', '?')} us per launch = {row.get('us_per_frame', row.get('us', '?'))} us per frame, frac {row.get('frac', '?')}   kernel {row.get('kernel', '?')}")
    print(f"   HIP events, under rocprofv3 --stats: {prow.get('us', '?')} us per launch")
    stats = kernel_stats(os.path.join(out, c, "stats"))
    alg = row.get("algorithmic_read_bytes", 0) + row.get("algorithmic_write_bytes", 0)
    entry = {"config": row.get("config"), "kernel_family": row.get("kernel"), "frames_per_launch": row.get("frames_per_launch", 1), "event_us": row.get("us"),
             "event_us_under_rocprof": prow.get("us"), "algorithmic_bytes_per_launch": alg}
    product = lambda name: "streamCeiling" not in name and "streamMove" not in name  # (the clocks are raised with ~60 ms of the byte mover first)
    stats = {k: v for k, v in stats.items() if product(k)}
    if stats:
        k, v = max(stats.items(), key=lambda kv: sum(kv[1]))
        avg = sum(v) / len(v) / 1e3
        print(f"   {k[:150]}")
        print(f"   rocprofv3: calls {len(v)}  avg {avg:.2f} us  min {min(v)/1e3:.2f}  max {max(v)/1e3:.2f}   -> {alg / (avg * 1e-6) / 8e12:.4f} of 8 TB/s on {alg / 1e6:.1f} MB per launch")
        entry.update(kernel=k, rocprof_calls=len(v), rocprof_avg_us=round(avg, 3), rocprof_frac=round(alg / (avg * 1e-6) / 8e12, 4))
    acc = {}
    for sub in ("fetch", "write"):
        for k, cs in counters(os.path.join(out, c, sub)).items():
            acc.setdefault(k, {}).update(cs)
    acc = {k: v for k, v in acc.items() if product(k)}
    if acc:
        k, cs = max(acc.items(), key=lambda kv: len(kv[1].get("FETCH_SIZE", {})) + len(kv[1].get("WRITE_SIZE", {})))
        avg = {name: sum(per.values()) / len(per) for name, per in cs.items() if per}
        rd, wr = avg.get("FETCH_SIZE", 0.0) * 1024 * 2, avg.get("WRITE_SIZE", 0.0) * 1024
        ar, aw = row.get("algorithmic_read_bytes", 0), row.get("algorithmic_write_bytes", 0)
        for name in sorted(avg):
            print(f"      {name:14s} dispatches={len(cs[name]):6d} avg_per_dispatch={avg[name]:18.1f}")
        if not avg.get("FETCH_SIZE") or not avg.get("WRITE_SIZE"):
            print("      (one of the two counter passes collected nothing for this kernel: no traffic figure)")
        elif ar and aw:
            print(f"      read {rd / 1e6:.1f} MB = {rd / ar:.3f} x algorithmic; write {wr / 1e6:.1f} MB = {wr / aw:.3f} x; total {(rd + wr) / 1e6:.1f} MB per launch = {(rd + wr) / (ar + aw):.3f} x algorithmic")
        entry.update(FETCH_SIZE_avg=avg.get("FETCH_SIZE"), WRITE_SIZE_avg=avg.get("WRITE_SIZE"),
                     traffic_bytes_per_launch=int(rd + wr) if avg.get("FETCH_SIZE") and avg.get("WRITE_SIZE") else None,
                     traffic_note="FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md), separate rocprofv3 --pmc passes")
    result[c] = entry
open(jpath, "w").write(json.dumps(result, indent=1) + "\n")
PY
find "$OUT" -name "*.db" -delete
rm -rf "$OUT"
cat "gpurun_out/${TAG}_sequence.txt" | cut -c1-220
