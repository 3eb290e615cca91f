#!/bin/bash
# traffic_cfgs.sh <tag> <cfg3|cfg4 ...> -- run on the GPU box (via gpurun): HBM traffic and wait counters of the kernels behind the BASELINE
# configurations that stream (VERDICT r04 next #1b): separate rocprofv3 passes (--kernel-trace --pmc only) for FETCH_SIZE, WRITE_SIZE and the
# SQ wait counters over `stream_sweep.py run <cfg>` (frames cycled, default kernel), digested into gpurun_out/<tag>_cfgs_traffic.txt:
# per-dispatch averages, bytes with the guide's gfx950 correction (FETCH_SIZE x 1024 x 2; WRITE_SIZE x 1024), traffic / algorithmic.
set -u
TAG=${1:-r05}; shift
R=$PWD
OUT=$R/gpurun_out/${TAG}_traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  python $R/tests/tools/stream_sweep.py run $c > "$OUT/$c.row.json" 2> "$OUT/$c.row.err"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/$c/fetch" -o fetch -- python $R/tests/tools/stream_sweep.py run $c > /dev/null 2> "$OUT/$c.fetch.log"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/$c/write" -o write -- python $R/tests/tools/stream_sweep.py run $c > /dev/null 2> "$OUT/$c.write.log"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d "$OUT/$c/sq" -o sq -- python $R/tests/tools/stream_sweep.py run $c > /dev/null 2> "$OUT/$c.sq.log"
done
cd "$R"
python - "$OUT" "$@" > "gpurun_out/${TAG}_cfgs_traffic.txt" <<'PY'
import json, os, sys
sys.path.insert(0, "tests/tools")
from profile_digest import counters
out = sys.argv[1]
print("rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_*> -- python tests/tools/stream_sweep.py run <cfg>   (separate passes; per-dispatch averages of the dominant kernel)")
print("bytes: FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte requests at 64, MI355X_MICROARCH.md), WRITE_SIZE x 1024")
for c in sys.argv[2:]:
    row = {}
    try:
        row = json.loads(open(os.path.join(out, c + ".row.json")).read().strip().splitlines()[-1])
    except Exception as e:
        print(f"\n== {c}: no event-timed row ({e})")
    print(f"\n== {c}: {row.get('config', '')}  kernel {row.get('kernel', '?')}  {row.get('us', '?')} us per launch (HIP events, no profiler)")
    acc = {}
    for sub in ("fetch", "write", "sq"):
        for k, cs in counters(os.path.join(out, c, sub)).items():
            acc.setdefault(k, {}).update(cs)
    if not acc:
        print("   no counters collected")
        continue
    # the dominant kernel = the one with the most dispatches carrying FETCH_SIZE
    k, cs = max(acc.items(), key=lambda kv: len(kv[1].get("FETCH_SIZE", {})) + len(kv[1].get("WRITE_SIZE", {})))
    print("   " + k[:160])
    avg = {name: sum(per.values()) / len(per) for name, per in cs.items() if per}
    for name in sorted(avg):
        print(f"      {name:22s} dispatches={len(cs[name]):6d} avg_per_dispatch={avg[name]:18.1f}")
    rd, wr = avg.get("FETCH_SIZE", 0.0) * 1024 * 2, avg.get("WRITE_SIZE", 0.0) * 1024
    ar, aw = row.get("algorithmic_read_bytes", 0), row.get("algorithmic_write_bytes", 0)
    if ar and aw:
        print(f"      read  {rd / 1e6:9.1f} MB per launch = {rd / ar:.3f} x algorithmic ({ar / 1e6:.1f} MB)")
        print(f"      write {wr / 1e6:9.1f} MB per launch = {wr / aw:.3f} x algorithmic ({aw / 1e6:.1f} MB)")
        print(f"      total {(rd + wr) / 1e6:9.1f} MB per launch = {(rd + wr) / (ar + aw):.3f} x algorithmic")
    if avg.get("SQ_WAVE_CYCLES") and avg.get("SQ_WAIT_ANY"):
        print(f"      waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) = {avg['SQ_WAIT_ANY'] / avg['SQ_WAVE_CYCLES']:.3f}; issuing VALU (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) = "
              f"{avg.get('SQ_ACTIVE_INST_VALU', 0.0) / avg['SQ_WAVE_CYCLES']:.3f}")
PY
find "$OUT" -name "*.db" -delete
cat "gpurun_out/${TAG}_cfgs_traffic.txt" | cut -c1-200
