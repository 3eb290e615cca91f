#!/bin/bash
# cfg5 (64 tiles of 1920x1080 10-bit 4:2:0 in one batch launch, 1.46 GB per launch: nothing stays in the Infinity Cache): the
# wave-private kernels over waves side by side x tile order x strips per wave, against the cooperative runs (tuning 5)
echo -n "cooperative (default policy): "; timeout 120 python tests/tools/cfg_bench.py cfg5x64 2>&1 | python3 -c "
import sys,json
print('  '.join('%s/%s %.1f' % (json.loads(l)['config'], json.loads(l)['arithmetic'][:3], json.loads(l)['us']) for l in sys.stdin if l.startswith('{')))"
for ns in 4 2; do for wx in 1 2 3; do for order in 0:0 1:1 1:2 1:4; do band=${order%%:*}; chunk=${order##*:}; t=$((band | 8 | (ns<<8) | (wx<<16) | (chunk<<20))); echo -n "solo ns=$ns wavesX=$((1<<(wx-1))) band=$band chunk=$chunk: "; AVIFHIP_TUNING=$t timeout 120 python tests/tools/cfg_bench.py cfg5x64 2>&1 | python3 -c "
import sys,json
print('  '.join('%s/%s %.1f' % (json.loads(l)['config'], json.loads(l)['arithmetic'][:3], json.loads(l)['us']) for l in sys.stdin if l.startswith('{')))"; done; done; done
