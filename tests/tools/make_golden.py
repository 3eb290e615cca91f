"""make_golden.py -- writes tests/golden/*.npz: inputs and outputs of the REFERENCE ITSELF (oracle/_ref/libavif_ref.so,
compiled from /root/reference by oracle/Makefile: libyuv OFF, -O3, x86-64, no FMA) for a fixed list of configurations.
Run in the build container (where /root/reference exists); the fixtures then pin the oracle and the HIP path on machines
that have no reference (the GPU box).  Each file holds the case parameters, every input plane with its row padding, and
the exact output buffer."""
import json
import os
import sys
from dataclasses import asdict
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import harness as H  # noqa: E402
import oracle_lib  # noqa: E402
from libavif_amd import abi  # noqa: E402

Y2R = [
    H.Y2RCase(256, 256, matrix=6, yuv_range=1, upsampling=0),                                   # BASELINE cfg1 (API defaults)
    H.Y2RCase(320, 24, matrix=1, yuv_range=0, upsampling=4),                                    # cfg2 in miniature
    H.Y2RCase(320, 24, matrix=1, yuv_range=0, upsampling=3),
    H.Y2RCase(264, 10, yuv_depth=10, yuv_format=1, matrix=9, yuv_range=1, alpha=True, rgb_depth=16, rgb_premultiplied=True),  # cfg3
    H.Y2RCase(300, 22, yuv_depth=10, yuv_format=3, matrix=1, yuv_range=0, rgb_depth=10, upsampling=4),  # cfg5 tile
    H.Y2RCase(300, 22, yuv_depth=10, yuv_format=3, matrix=1, yuv_range=0, rgb_depth=8, upsampling=4),
    H.Y2RCase(127, 9, yuv_format=2, matrix=5, yuv_range=0, rgb_format=abi.AVIF_RGB_FORMAT_BGR, upsampling=4),
    H.Y2RCase(65, 7, yuv_depth=12, yuv_format=3, matrix=9, yuv_range=0, rgb_depth=12, rgb_format=abi.AVIF_RGB_FORMAT_ARGB, alpha=True, row_pad=6),
    H.Y2RCase(64, 6, yuv_format=1, matrix=0, yuv_range=1, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565),
    H.Y2RCase(70, 5, yuv_depth=10, yuv_format=1, matrix=16, yuv_range=1, rgb_depth=8),
    H.Y2RCase(70, 5, yuv_format=4, matrix=1, yuv_range=0, rgb_format=abi.AVIF_RGB_FORMAT_GRAYA, alpha=True, rgb_premultiplied=True),
    H.Y2RCase(96, 6, yuv_depth=10, yuv_format=3, matrix=1, yuv_range=0, rgb_depth=16, is_float=True, alpha=True, upsampling=4),
    H.Y2RCase(1, 1), H.Y2RCase(3, 5, yuv_format=3, upsampling=4, alpha=True, rgb_premultiplied=True),
]
R2Y = [
    H.R2YCase(320, 18),                                                                          # cfg4 in miniature (random alpha)
    H.R2YCase(320, 18, opaque=True),
    H.R2YCase(127, 9, rgb_format=abi.AVIF_RGB_FORMAT_BGR, yuv_format=2, matrix=6, yuv_range=1),
    H.R2YCase(65, 7, rgb_depth=16, rgb_format=abi.AVIF_RGB_FORMAT_ARGB, yuv_depth=12, yuv_format=1, matrix=9, rgb_premultiplied=True),
    H.R2YCase(33, 5, rgb_format=abi.AVIF_RGB_FORMAT_GRAY, yuv_format=4),
    H.R2YCase(3, 3, yuv_format=3), H.R2YCase(64, 4, rgb_depth=8, yuv_depth=10, yuv_format=1, matrix=16, yuv_range=1, rgb_format=0),
]


def main():
    ref = oracle_lib.ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libavif_ref.so is missing: run `make -C oracle` where /root/reference exists")
    be = H.libavif_backend(ref, "reference")
    out = ROOT / "tests" / "golden"
    out.mkdir(exist_ok=True)
    for k, c in enumerate(Y2R):
        img = H.make_y2r_inputs(c)
        rgb = H.make_y2r_output(c)
        res = be.yuv_to_rgb(img.struct, rgb.struct)
        arrays = {f"plane{p}": a for p, a in enumerate(img.planes + [img.alpha]) if a is not None}
        np.savez_compressed(out / f"y2r_{k:02d}.npz", case=json.dumps(asdict(c)), result=res, output=rgb.pixels, **arrays)
    for k, c in enumerate(R2Y):
        rgb = H.make_r2y_inputs(c)
        img = H.make_r2y_output(c)
        res = be.rgb_to_yuv(img.struct, rgb.struct)
        arrays = {f"plane{p}": a for p, a in enumerate(img.planes + [img.alpha]) if a is not None}
        np.savez_compressed(out / f"r2y_{k:02d}.npz", case=json.dumps(asdict(c)), result=res, pixels=rgb.pixels, **arrays)
    print(f"wrote {len(Y2R)} + {len(R2Y)} fixtures to {out} from {os.path.basename(ref._name)} ({sum(f.stat().st_size for f in out.glob('*.npz'))} bytes)")


if __name__ == "__main__":
    main()
