"""Self-consistency fuzz on the GPU box (not part of the test suite): the bandwidth-tuned kernels against the universal ones -- two
independent implementations, each pinned against the oracles by tests/ -- on seeded random configurations at sizes the C oracles would
take minutes for.  Prints every disagreement; exit code 1 if there is one.
    python tests/tools/fuzz_tiled_vs_generic.py [cases] [seed]"""
import random
import sys
from dataclasses import replace
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import harness as H  # noqa: E402
from libavif_amd import native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
lib = native.load()
sizes = [(1920, 1080), (1281, 723), (2048, 64), (4100, 38), (516, 1030), (259, 517), (1024, 1024), (3841, 19), (640, 481),
         (3840, 2160), (3850, 1702)]  # the last two: above 6 megapixels, where 8-bit bilinear frames take the wave-private fp32 kernels
bad = 0
kernels = {}
for arithmetic in (0, 1):
    lib.avifhipSetArithmetic(arithmetic)
    be = H.HipDeviceBackend()
    # the default arithmetic's own domain (what libavif hands to libyuv) in the first pass, the whole configuration space in the second
    y2r = H.libyuv_y2r_cases(sizes, n_random=n, seed=seed) if arithmetic == 0 else H.y2r_sweep(sizes, n_random=n, seed=seed + 1)
    rnd.shuffle(y2r)
    for c in y2r[:n]:
        c = replace(c, avoid_libyuv=bool(arithmetic))
        lib.avifhipSetTiledKernels(1)
        r1, p1 = H.run_y2r(be, c)
        k = native.last_kernel()
        kernels[k.split("<")[0]] = kernels.get(k.split("<")[0], 0) + 1
        lib.avifhipSetTiledKernels(0)
        r2, p2 = H.run_y2r(be, c)
        if r1 != r2 or not np.array_equal(p1, p2):
            bad += 1
            print("Y2R", c.ident(), k, r1, r2, "" if r1 != r2 else H.describe_diff(p2, p1), flush=True)
    r2y = H.r2y_sweep(sizes, n_random=n // 2, seed=seed + 7 + arithmetic)
    rnd.shuffle(r2y)
    for c in r2y[: n // 2]:
        c = replace(c, avoid_libyuv=bool(arithmetic))
        lib.avifhipSetTiledKernels(1)
        r1, i1 = H.run_r2y(be, c)
        k = native.last_kernel()
        kernels[k.split("<")[0]] = kernels.get(k.split("<")[0], 0) + 1
        lib.avifhipSetTiledKernels(0)
        r2, i2 = H.run_r2y(be, c)
        d = None if r1 != r2 else H.planes_equal(i2, i1, padding=False)
        if r1 != r2 or d:
            bad += 1
            print("R2Y", c.ident(), k, r1, r2, d, flush=True)
lib.avifhipSetTiledKernels(1)
print(f"{bad} disagreements; kernels seen: {kernels}")
sys.exit(1 if bad else 0)
