"""Launches one configuration N times (for rocprofv3 runs). Usage: run_one.py <cfg> [launches] [tuning] [tiled]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from ab_bench import lib, native, setup  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib.avifhipSetTuning(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
lib.avifhipSetTiledKernels(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
dimg, drgb, bpp, px = setup(name)
for _ in range(n):
    native.check(lib.avifhipImageYUVToRGBAsync(dimg.struct, drgb.struct, None))
native.check(lib.avifhipSynchronize(None))
print(name, native.last_kernel(), n, "launches")
