"""make_golden_libyuv.py -- writes tests/golden/yuvlib_*.npz: inputs and outputs of a libavif BUILT WITH LIBYUV for a
fixed list of configurations of the reference's integer path.  libyuv's source is absent from /root/reference
(third-party, pinned 1949); the generating binary is the only libyuv-enabled libavif available offline: Pillow's
bundled libavif 1.4.1 + libyuv 1922, reached through its public avifImageYUVToRGB / avifImageRGBToYUV /
avifRGBImage{Pre,Unpre}multiplyAlpha.  Run in the build container; the fixtures then pin the integer-path oracle and
the HIP integer kernels on machines without that binary (the GPU box)."""
import json
import sys
from dataclasses import asdict
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import harness as H  # noqa: E402
import oracle_lib  # noqa: E402
from libavif_amd import abi, synth  # noqa: E402

A = abi
Y2R = [
    H.Y2RCase(256, 256, matrix=6, yuv_range=1, upsampling=0, avoid_libyuv=False),                      # BASELINE cfg1, API defaults
    H.Y2RCase(320, 24, matrix=1, yuv_range=0, upsampling=4, avoid_libyuv=False),                       # cfg2 in miniature: I420ToARGBMatrixFilter
    H.Y2RCase(320, 24, matrix=1, yuv_range=0, upsampling=3, avoid_libyuv=False),                       # nearest
    H.Y2RCase(301, 23, matrix=9, yuv_range=1, upsampling=4, avoid_libyuv=False, rgb_format=A.AVIF_RGB_FORMAT_BGRA, alpha=True),
    H.Y2RCase(300, 22, yuv_depth=10, yuv_format=3, matrix=1, yuv_range=0, rgb_depth=8, upsampling=4, avoid_libyuv=False),  # cfg5 -> RGBA8: I010
    H.Y2RCase(127, 9, yuv_depth=10, yuv_format=2, matrix=9, yuv_range=0, upsampling=2, avoid_libyuv=False, alpha=True),    # I210Alpha
    H.Y2RCase(66, 6, yuv_depth=10, yuv_format=1, matrix=9, yuv_range=1, avoid_libyuv=False, alpha=True, rgb_premultiplied=True),  # I410Alpha + ARGBAttenuate
    H.Y2RCase(65, 7, yuv_depth=12, yuv_format=3, matrix=1, yuv_range=0, upsampling=0, avoid_libyuv=False),                  # I012 (nearest under AUTOMATIC)
    H.Y2RCase(65, 7, yuv_depth=12, yuv_format=3, matrix=1, yuv_range=0, upsampling=4, avoid_libyuv=False, alpha=True),      # downshift + I420AlphaToARGBMatrixFilter
    H.Y2RCase(64, 8, yuv_depth=10, yuv_format=3, matrix=6, yuv_range=0, upsampling=1, avoid_libyuv=False, rgb_format=A.AVIF_RGB_FORMAT_ARGB),  # downshift + I420ToRGBAMatrix
    H.Y2RCase(127, 9, yuv_format=2, matrix=5, yuv_range=0, rgb_format=A.AVIF_RGB_FORMAT_BGR, upsampling=4, avoid_libyuv=False),  # I422ToRGB24MatrixFilter
    H.Y2RCase(63, 5, yuv_format=3, matrix=2, yuv_range=1, rgb_format=A.AVIF_RGB_FORMAT_RGB_565, upsampling=3, avoid_libyuv=False),
    H.Y2RCase(70, 5, yuv_format=4, matrix=0, yuv_range=1, rgb_format=A.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False),               # I400 with identity -> BT.601
    H.Y2RCase(70, 5, yuv_format=1, matrix=12, color_primaries=9, yuv_range=0, rgb_format=A.AVIF_RGB_FORMAT_RGB, avoid_libyuv=False),
    H.Y2RCase(64, 6, yuv_format=1, matrix=1, yuv_range=0, rgb_format=A.AVIF_RGB_FORMAT_ABGR, avoid_libyuv=False),               # no libyuv entry: fp32
    H.Y2RCase(48, 6, yuv_format=3, matrix=1, yuv_range=0, alpha=True, image_premultiplied=True, upsampling=4, avoid_libyuv=False),  # ARGBUnattenuate post-pass
    H.Y2RCase(48, 6, yuv_format=1, matrix=1, yuv_range=0, alpha=True, rgb_premultiplied=True, avoid_libyuv=True),               # fp32 colour, libyuv attenuate
    H.Y2RCase(1, 1, avoid_libyuv=False), H.Y2RCase(3, 5, yuv_format=3, upsampling=4, alpha=True, avoid_libyuv=False),
    H.Y2RCase(2, 1, yuv_format=3, upsampling=4, avoid_libyuv=False), H.Y2RCase(1, 2, yuv_format=2, upsampling=4, avoid_libyuv=False),
]
R2Y = [
    H.R2YCase(320, 18, matrix=6, avoid_libyuv=False),                                                     # ABGRToI420
    H.R2YCase(319, 17, matrix=6, yuv_range=1, avoid_libyuv=False, opaque=True),                           # ABGRToJ420, odd size
    H.R2YCase(127, 9, rgb_format=A.AVIF_RGB_FORMAT_BGR, yuv_format=2, matrix=5, yuv_range=1, avoid_libyuv=False),  # two-step RGB24 -> J422
    H.R2YCase(65, 7, rgb_format=A.AVIF_RGB_FORMAT_ARGB, yuv_format=1, matrix=6, yuv_range=0, avoid_libyuv=False),  # two-step BGRA -> I444
    H.R2YCase(65, 7, rgb_format=A.AVIF_RGB_FORMAT_RGB, yuv_format=1, matrix=6, yuv_range=1, avoid_libyuv=False),   # RAWToJ444
    H.R2YCase(33, 5, rgb_format=A.AVIF_RGB_FORMAT_ABGR, yuv_format=4, matrix=6, yuv_range=1, avoid_libyuv=False),  # RGBAToJ400
    H.R2YCase(33, 5, rgb_format=A.AVIF_RGB_FORMAT_BGRA, yuv_format=4, matrix=6, yuv_range=0, avoid_libyuv=False),  # ARGBToI400
    H.R2YCase(64, 4, rgb_format=A.AVIF_RGB_FORMAT_RGBA, yuv_format=1, matrix=6, yuv_range=1, avoid_libyuv=False),  # no entry: fp32
    H.R2YCase(64, 4, matrix=1, avoid_libyuv=False),                                                       # BT.709: fp32 (cfg4's arithmetic)
    H.R2YCase(3, 3, yuv_format=3, matrix=6, avoid_libyuv=False), H.R2YCase(1, 1, yuv_format=3, matrix=6, avoid_libyuv=False),
]
MUL = [(A.AVIF_RGB_FORMAT_RGBA, 8), (A.AVIF_RGB_FORMAT_BGRA, 8), (A.AVIF_RGB_FORMAT_ARGB, 8), (A.AVIF_RGB_FORMAT_RGBA, 16)]


def main():
    lib = oracle_lib.pillow()
    if lib is None:
        raise SystemExit("no libyuv-enabled libavif binary found (Pillow's bundled libavif)")
    be = H.libavif_backend(lib, "pillow")
    out = ROOT / "tests" / "golden"
    out.mkdir(exist_ok=True)
    for k, c in enumerate(Y2R):
        img = H.make_y2r_inputs(c)
        rgb = H.make_y2r_output(c)
        res = be.yuv_to_rgb(img.struct, rgb.struct)
        arrays = {f"plane{p}": a for p, a in enumerate(img.planes + [img.alpha]) if a is not None}
        np.savez_compressed(out / f"yuvlib_y2r_{k:02d}.npz", case=json.dumps(asdict(c)), result=res, output=rgb.pixels, **arrays)
    for k, c in enumerate(R2Y):
        rgb = H.make_r2y_inputs(c)
        img = H.make_r2y_output(c)
        res = be.rgb_to_yuv(img.struct, rgb.struct)
        arrays = {f"plane{p}": a for p, a in enumerate(img.planes + [img.alpha]) if a is not None}
        np.savez_compressed(out / f"yuvlib_r2y_{k:02d}.npz", case=json.dumps(asdict(c)), result=res, pixels=rgb.pixels, **arrays)
    for k, (fmt, depth) in enumerate(MUL):
        # every (colour, alpha) pair for RGBA; a 24-alpha slice for the other layouts
        w, h = (256, 256 if k == 0 else 24) if depth == 8 else (64, 32)
        src = abi.make_rgb(w, h, depth, fmt)
        if depth == 8:
            ch = src.channels()
            a_first = fmt in (A.AVIF_RGB_FORMAT_ARGB, A.AVIF_RGB_FORMAT_ABGR)
            cols = [q for q in range(4) if q != (0 if a_first else 3)]
            ch[:, :, cols[0]] = np.arange(256)[None, :]
            ch[:, :, cols[1]] = np.arange(256)[None, :]
            ch[:, :, cols[2]] = np.arange(256)[None, :]
            ch[:, :, 0 if a_first else 3] = ((np.arange(h) * (1 if h == 256 else 37)) % 256)[:, None]
        else:
            synth.fill_rgb(src, 0xA11CE)
        outs = {}
        for which in ("premultiply", "unpremultiply"):
            work = abi.make_rgb(w, h, depth, fmt)
            work.pixels[...] = src.pixels
            outs[which + "_result"] = getattr(be, which)(work.struct)
            outs[which] = work.pixels.copy()
        np.savez_compressed(out / f"yuvlib_mul_{k:02d}.npz", case=json.dumps({"format": fmt, "depth": depth}), pixels=src.pixels, **outs)
    files = list(out.glob("yuvlib_*.npz"))
    print(f"wrote {len(files)} fixtures to {out} from {lib._name} (libyuv {lib.avifLibYUVVersion()}), {sum(f.stat().st_size for f in files)} bytes")


if __name__ == "__main__":
    main()
