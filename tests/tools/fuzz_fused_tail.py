"""Self-consistency fuzz on the GPU box (not part of the test suite): the fused decode-side tail (avifhipImageYUVToRGBTransformedAsync: the
conversion's stores go through the crop / rotate / mirror map, only what the crop keeps is converted, quarter turns choose where their tile
grid starts) against the same result in two passes (avifhipImageYUVToRGBAsync into a canvas, then avifhipRGBImageTransformAsync) -- two
routes through different kernels, each pinned against the oracles at small sizes by tests/test_fused_tail.py and tests/test_transform.py --
on seeded random crops, turns, mirrors, depths and pixel formats at sizes the C oracles would take minutes for.
    python tests/tools/fuzz_fused_tail.py [cases] [seed]"""
import ctypes as C
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from libavif_amd import abi, device, native, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = native.load()
BIL, NEAR = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, abi.AVIF_CHROMA_UPSAMPLING_NEAREST
sizes = [(3840, 2160), (2048, 858), (1921, 1083), (4100, 260), (516, 2050), (1280, 720)]
bad, kernels = 0, {}
for case in range(n):
    w, h = rnd.choice(sizes)
    depth = rnd.choice([8, 8, 10, 12])
    fmt = rnd.choice([abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_PIXEL_FORMAT_YUV422, abi.AVIF_PIXEL_FORMAT_YUV444])
    arith = rnd.choice([0, 1])
    rgb_depth = rnd.choice([8, depth]) if arith else (8 if rnd.random() < 0.7 else depth)
    rgb_fmt = rnd.choice([abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGRA, abi.AVIF_RGB_FORMAT_ARGB, abi.AVIF_RGB_FORMAT_RGB])
    alpha = rnd.random() < 0.3
    lib.avifhipSetArithmetic(arith)
    img = abi.make_yuv(w, h, depth, fmt, rnd.choice([abi.AVIF_RANGE_LIMITED, abi.AVIF_RANGE_FULL]), rnd.choice([1, 6, 9]), with_alpha=alpha)
    synth.fill_yuv(img, 0x1234 + case)
    dimg = device.DeviceYUV(img)
    cw, ch = rnd.randint(64, w), rnd.randint(2, h)
    cx, cy = rnd.randint(0, w - cw), rnd.randint(0, h - ch)
    if rnd.random() < 0.25:
        cx, cy, cw, ch = 0, 0, w, h
    if fmt != abi.AVIF_PIXEL_FORMAT_YUV444:
        cx &= ~1  # (the crop is what an application passes: any origin; the reference's clap rule asks for even ones on subsampled images)
    if fmt == abi.AVIF_PIXEL_FORMAT_YUV420:
        cy &= ~1
    cw, ch = min(cw, w - cx), min(ch, h - cy)
    crop = abi.avifCropRect(cx, cy, cw, ch)
    angle, mirror = rnd.choice([0, 1, 2, 3]), rnd.choice([-1, 0, 1])
    dw, dh = (ch, cw) if angle & 1 else (cw, ch)
    up = rnd.choice([BIL, BIL, NEAR])
    mk = lambda ww, hh: abi.make_rgb(ww, hh, rgb_depth, rgb_fmt, upsampling=up, avoid_libyuv=bool(arith), allocate=False)
    fused, mid, two = device.DeviceRGB(mk(dw, dh)), device.DeviceRGB(mk(w, h)), device.DeviceRGB(mk(dw, dh))
    px = abi.rgb_pixel_size(rgb_fmt, rgb_depth)
    for d in (fused, two):
        native.check(lib.avifhipDeviceMemset(d.buffer.ptr, 0x5a, d.pitch * dh))
    r1 = lib.avifhipImageYUVToRGBTransformedAsync(dimg.struct, fused.struct, C.byref(crop), int(angle != 0), angle, int(mirror >= 0), max(mirror, 0), None)
    k = native.last_kernel()
    r2 = lib.avifhipImageYUVToRGBAsync(dimg.struct, mid.struct, None)
    if r2 == 0:
        r2 = lib.avifhipRGBImageTransformAsync(two.struct, mid.struct, C.byref(crop), int(angle != 0), angle, int(mirror >= 0), max(mirror, 0), None)
    native.check(lib.avifhipSynchronize(None))
    kernels[k.split("<")[0] + (",mapped" if "mapped" in k else "")] = kernels.get(k.split("<")[0] + (",mapped" if "mapped" in k else ""), 0) + 1
    a = fused.buffer.download(fused.pitch * dh).reshape(dh, fused.pitch)[:, : dw * px]
    b = two.buffer.download(two.pitch * dh).reshape(dh, two.pitch)[:, : dw * px]
    if r1 != r2 or not np.array_equal(a, b):
        bad += 1
        where = np.argwhere(a != b)
        print("TAIL", (w, h), depth, fmt, "arith", arith, "rgb", rgb_fmt, rgb_depth, "alpha", alpha, "crop", (cx, cy, cw, ch), "angle", angle, "mirror", mirror, k, r1, r2,
              f"{len(where)} bytes differ, first at {tuple(where[0]) if len(where) else None}", flush=True)
    for d in (fused, mid, two):
        d.buffer.free()
    for b in dimg.buffers:
        if b is not None:
            b.free()
print(f"{bad} disagreements in {n} cases; kernels of the fused route: {kernels}")
sys.exit(1 if bad else 0)
