"""Sweep of the tiled kernels' launch geometry (strips per wave, tiles per workgroup run: avifhipSetTuning) at 1080p / 4K / 8K."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import cfg_bench as B
from libavif_amd import abi
lib = B.lib
DEFAULT = None
for size, name in (((3840, 2160), "4k"), ((1920, 1080), "1080p"), ((7680, 4320), "8k")):
    lib.avifhipSetArithmetic(0)
    pair = B.y2r(size[0], size[1], 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, up=B.BIL, avoid=False)
    res = []
    for ns in (0, 1, 2):
        for run in (0, 1, 2, 3, 4, 6, 8):
            lib.avifhipSetTuning(1 | (ns << 8) | (run << 12))
            res.append((round(B.time_y2r(pair, 40) * 1e3, 2), ns, run))
    lib.avifhipSetTuning(1)
    res.sort()
    print(name, "default(0,0):", [r for r in res if r[1] == 0 and r[2] == 0][0][0], "best:", res[:4])
