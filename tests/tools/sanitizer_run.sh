#!/bin/bash
# sanitizer_run.sh <tag> [seconds] -- on the GPU box: tests/c/stress over the product's host code built with ThreadSanitizer and with
# AddressSanitizer (libavif_amd/csrc: make tsan asan; tests/c: make stress_tsan stress_asan -- built in the container, run here).
# 8 threads x {yuv->rgb, rgb->yuv, premultiply, gain map, scale} x 4 sizes, buffers allocated and freed per call, the device set switched
# every 300 ms (farm on / off), half of the threads replaced every second.  Writes gpurun_out/<tag>_tsan.txt: the program's own summary, the
# number of reports with and without tests/c/tsan.supp, and every report that survives the suppressions in full.
set -u
TAG=${1:-r06}; SEC=${2:-10}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out /tmp/san
OUT=gpurun_out/${TAG}_tsan.txt
{
echo "== ThreadSanitizer, no suppressions (the HIP / HSA runtimes are not instrumented: their internal synchronisation is invisible) =="
rm -f /tmp/san/raw.*
TSAN_OPTIONS="log_path=/tmp/san/raw history_size=4 second_deadlock_stack=1" timeout 300 tests/c/stress_tsan $SEC; echo "exit $?"
cat /tmp/san/raw.* 2>/dev/null > /tmp/san/raw_all.txt
python3 - <<'PY'
import re, collections
txt = open('/tmp/san/raw_all.txt', errors='replace').read()
reps = [r for r in txt.split('==================') if 'WARNING: ThreadSanitizer' in r]
both = []
kinds = collections.Counter()
for r in reps:
    kinds[re.search(r'WARNING: ThreadSanitizer: ([^(\n]+)', r).group(1).strip()] += 1
    blocks = re.split(r'\n\s*\n', r)
    def top_ours(b):
        return any(('libavifhip_tsan' in l or 'stress.c' in l) for l in b.split('\n') if re.match(r'\s+#[01] ', l))
    if len(blocks) > 1 and top_ours(blocks[0]) and top_ours(blocks[1]):
        both.append(r)
print(f"reports: {len(reps)} {dict(kinds)}; with the top frames of BOTH accesses in instrumented code: {len(both)}")
for r in both:
    print(r.strip()[:3000]); print('------')
PY
echo
echo "== ThreadSanitizer with tests/c/tsan.supp =="
rm -f /tmp/san/supp.*
TSAN_OPTIONS="suppressions=$PWD/tests/c/tsan.supp log_path=/tmp/san/supp history_size=4 second_deadlock_stack=1 print_suppressions=1" timeout 300 tests/c/stress_tsan $SEC; echo "exit $?"
cat /tmp/san/supp.* 2>/dev/null > /tmp/san/supp_all.txt
echo "reports that survive the suppressions: $(grep -c 'WARNING: ThreadSanitizer' /tmp/san/supp_all.txt)"
grep -v '^$' /tmp/san/supp_all.txt | cut -c1-400 | head -400
echo
echo "== AddressSanitizer (lifetime errors around QuiesceOnExit, the download helper thread, contexts leased and returned) =="
ASAN_OPTIONS="detect_leaks=0 log_path=/tmp/san/asan" timeout 300 tests/c/stress_asan $SEC; echo "exit $?"
cat /tmp/san/asan.* 2>/dev/null | cut -c1-400 | head -200
echo
echo "== the uninstrumented product, same program =="
timeout 300 tests/c/stress $SEC; echo "exit $?"
} > $OUT 2>&1
tail -30 $OUT
