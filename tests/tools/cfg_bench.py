"""Kernel timings of every BASELINE configuration on the GPU box (HIP events on the launch stream, device-resident
buffers), in both arithmetic families.  Prints one JSON line per (configuration, arithmetic):
    python tests/tools/cfg_bench.py [cfg2 cfg2n cfg3 cfg4 cfg4rgb cfg5 cfg5x64 ...]
Algorithmic bytes per pixel are SURVEY.md 8d's figures."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, device, native, synth  # noqa: E402

if os.environ.get("AVIFHIP_BENCH_LIB"):  # A/B measurements: a variant build of the library (e.g. different store policy)
    native.LIB_PATH = Path(os.environ["AVIFHIP_BENCH_LIB"]).resolve()
lib = native.load()
if os.environ.get("AVIFHIP_BENCH_SLAB"):
    # A/B measurement: every device buffer carved out of ONE allocation (value = alignment in bytes) instead of one hipMalloc per plane --
    # shows what many small allocations (1 MiB chroma planes of 1080p tiles) cost in address translation
    _slab = {"base": None, "off": 0, "size": 6 << 30, "align": int(os.environ["AVIFHIP_BENCH_SLAB"], 0)}

    def _slab_init(self, nbytes):
        if _slab["base"] is None:
            _slab["base"] = lib.avifhipDeviceAlloc(_slab["size"])
            assert _slab["base"]
        a = _slab["align"]
        off = (_slab["off"] + a - 1) // a * a
        self.nbytes = max(int(nbytes), 1)
        assert off + self.nbytes <= _slab["size"]
        self.ptr = _slab["base"] + off
        _slab["off"] = off + self.nbytes

    device.DeviceBuffer.__init__ = _slab_init
    device.DeviceBuffer.free = lambda self: None
if os.environ.get("AVIFHIP_TUNING"):  # A/B measurements: plan.h TuningBits (e.g. 5 = round 1's cooperative runs for the fp32 / 10-12-bit families)
    lib.avifhipSetTuning(int(os.environ["AVIFHIP_TUNING"], 0))
BIL, NEAR = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, abi.AVIF_CHROMA_UPSAMPLING_NEAREST


def y2r(w, h, depth, fmt, rng, mc, rgb_depth, up=BIL, alpha=False, premult=False, avoid=True, rgb_format=abi.AVIF_RGB_FORMAT_RGBA, seed=0x12345678):
    img = abi.make_yuv(w, h, depth, fmt, rng, mc, with_alpha=alpha)
    synth.fill_yuv(img, seed)
    rgb = abi.make_rgb(w, h, rgb_depth, rgb_format, upsampling=up, alpha_premultiplied=premult, avoid_libyuv=avoid, allocate=False)
    return device.DeviceYUV(img), device.DeviceRGB(rgb)


PREHEAT_MS = float(os.environ.get("AVIFHIP_BENCH_PREHEAT_MS", "60"))  # an idle chip sits at ~95 MHz and needs tens of milliseconds of work to ramp (bench.py does the same)


CLOCK = "events"  # how the row being measured is timed: "events" (HIP events around kernel launches inside the library) or "host" (wall clock around API calls)


def preheat(fn):
    """Calls fn(iters) -> ms per launch until PREHEAT_MS of GPU work have run."""
    spent = 0.0
    while spent < PREHEAT_MS:
        spent += max(fn(200), 1e-3) * 200


def settled(fn):
    """Median of 9 event-timed bursts (not the best one: the rows must agree with a profiler's average over the same launches; a kernel that
    is hard on the vector ALUs runs its first ~10 ms after a change of kernel up to 25 % slower: profiles/r03_sustain_probe.txt)."""
    global CLOCK
    CLOCK = "events"
    xs = sorted(fn() for _ in range(9))
    return xs[4]


def host_clock(call, burst=100):
    """Milliseconds per call of an asynchronous entry point that launches more than one kernel (or whose kernel is not behind avifhipTime*):
    wall clock around bursts of calls, each closed by a synchronisation -- PREHEAT_MS of the same calls first, then the median of 7 bursts,
    so that the bursts, not the first slow milliseconds after a change of kernel, carry a profiler's average over the run."""
    global CLOCK
    CLOCK = "host"
    end = time.perf_counter() + PREHEAT_MS * 1e-3
    while time.perf_counter() < end:
        for _ in range(20):
            call()
        native.check(lib.avifhipSynchronize(None))
    xs = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(burst):
            call()
        native.check(lib.avifhipSynchronize(None))
        xs.append((time.perf_counter() - t0) / burst * 1e3)
    return sorted(xs)[3]


def time_y2r(pair, iters=100):
    preheat(lambda n: lib.avifhipTimeYUVToRGB(pair[0].struct, pair[1].struct, 0, n, None))
    return settled(lambda: lib.avifhipTimeYUVToRGB(pair[0].struct, pair[1].struct, 4, iters, None))


def time_y2r_two_frames(pair, iters=20):
    """Frames larger than the 256 MB Infinity Cache (8K 10-bit 4:4:4 + alpha -> RGBA16: 530 MB) timed over TWO frames cycled (a second set of
    buffers with the same samples): relaunching ONE such frame finds part of it in the cache, which no decoder's next frame does (round 5: the
    library now launches them for the streaming regime, bench.py's cfg3 has cycled two frames since round 4)."""
    twin_img = device.DeviceYUV(pair[0].host)
    host_rgb = pair[1].host
    twin_rgb = device.DeviceRGB(host_rgb)
    twin_rgb.struct.isFloat = pair[1].struct.isFloat
    imgs = (C.POINTER(abi.avifImage) * 2)(C.pointer(pair[0].struct), C.pointer(twin_img.struct))
    rgbs = (C.POINTER(abi.avifRGBImage) * 2)(C.pointer(pair[1].struct), C.pointer(twin_rgb.struct))
    preheat(lambda n: lib.avifhipTimeYUVToRGBCycle(2, imgs, rgbs, 0, n, None))
    return settled(lambda: lib.avifhipTimeYUVToRGBCycle(2, imgs, rgbs, 4, iters, None))


def time_y2r_cycle(pairs, per=1, iters=40):
    """`len(pairs)` frames cycled, `per` frames per launch (1: avifhipImageYUVToRGBAsync; more: a sequence launch): ms per FRAME -- bench.py's own regimes"""
    n = len(pairs)
    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(q[0].struct) for q in pairs])
    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(q[1].struct) for q in pairs])
    if per == 1:
        preheat(lambda k: lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 0, k, None))
        return settled(lambda: lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 4, iters, None))
    preheat(lambda k: lib.avifhipTimeYUVToRGBBatchCycle(n, imgs, rgbs, per, 0, max(1, k // per), None) / per)
    return settled(lambda: lib.avifhipTimeYUVToRGBBatchCycle(n, imgs, rgbs, per, 2, iters, None)) / per


def run(name):
    global CLOCK
    out = []
    for arith, avoid in (("float", True), ("integer", False)):
        lib.avifhipSetArithmetic(1 if arith == "float" else 0)
        px, bpp, ms = 0, 0.0, None
        extra = {}
        if name == "cfg2_565":
            # Android's bitmap format (android_jni/.../libavif_jni.cc:206-223): 8K 8-bit 4:2:0 -> RGB565, nearest (libyuv has no filtering 565 entry)
            pair = y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, up=NEAR, avoid=avoid, rgb_format=abi.AVIF_RGB_FORMAT_RGB_565)
            px, bpp, ms = 7680 * 4320, 3.5, time_y2r(pair)
        elif name in ("cfg2_565_odd", "cfg2_565_alpha", "cfg2_565_10"):
            # round 6: what had gone through the one-lane-per-pixel kernels -- RGB565 rows that are 2-byte aligned only (an odd width: every other
            # row's 8-byte stores straddle), an alpha plane multiplied in inside the loop (the format drops it, src/reformat.c:1503-1511: 1.5 + 1 + 2
            # B/px, the fp32 loops in both arithmetics), and 10-bit planes (integer: Convert16To8Plane + I420ToRGB565Matrix; 3 + 2 B/px)
            w = 7679 if name == "cfg2_565_odd" else 7680
            depth = 10 if name == "cfg2_565_10" else 8
            pair = y2r(w, 4320, depth, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, up=NEAR, alpha=(name == "cfg2_565_alpha"), avoid=avoid,
                       rgb_format=abi.AVIF_RGB_FORMAT_RGB_565)
            px, bpp, ms = w * 4320, {"cfg2_565_odd": 3.5, "cfg2_565_alpha": 4.5, "cfg2_565_10": 5.0}[name], time_y2r(pair)
        elif name in ("f16_420", "f16_444a"):
            # half-float outputs (what an HDR compositor takes): 8K 10-bit -> RGBA F16; 4:2:0 bilinear without alpha (1.5*2 + 8 B/px), 4:4:4 with alpha (4*2 + 8)
            if arith == "integer":
                continue  # libyuv has no 16-bit outputs: one arithmetic
            a444 = name == "f16_444a"
            img = abi.make_yuv(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444 if a444 else abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 9, with_alpha=a444)
            synth.fill_yuv(img, 0x4242)
            rgb = abi.make_rgb(7680, 4320, 16, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)
            rgb.struct.isFloat = 1
            pair = (device.DeviceYUV(img), device.DeviceRGB(rgb))
            px, bpp, ms = 7680 * 4320, (16.0 if a444 else 11.0), (time_y2r_two_frames(pair) if a444 else time_y2r(pair, 20))
        elif name in ("ident8", "ident8rgb"):
            # lossless RGB stored as 8-bit 4:4:4 GBR planes (identity matrix, full range): a byte shuffle, 3 + 4 (or 3 + 3) B/px
            fmt = abi.AVIF_RGB_FORMAT_RGB if name == "ident8rgb" else abi.AVIF_RGB_FORMAT_RGBA
            pair = y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 0, 8, avoid=avoid, rgb_format=fmt)
            px, bpp, ms = 7680 * 4320, (6.0 if name == "ident8rgb" else 7.0), time_y2r(pair)
        elif name in ("gray8", "graya16"):
            # gray layouts: 8K 8-bit 4:2:0 -> GRAY8 (luma only: 1 + 1 B/px); 8K 10-bit 4:2:0 + alpha -> GRAYA16 (2 + 2 + 4 B/px)
            if arith == "integer":
                continue  # libyuv has no gray entries: one arithmetic
            if name == "gray8":
                pair = y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=avoid, rgb_format=abi.AVIF_RGB_FORMAT_GRAY)
                px, bpp, ms = 7680 * 4320, 2.0, time_y2r(pair)
            else:
                pair = y2r(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_FULL, 9, 16, alpha=True, avoid=avoid, rgb_format=abi.AVIF_RGB_FORMAT_GRAYA)
                px, bpp, ms = 7680 * 4320, 8.0, time_y2r(pair)
        elif name in ("premul8", "premul16", "unpremul8", "premul10", "unpremul10", "unpremul16"):
            # avifRGBImagePremultiplyAlpha / UnpremultiplyAlpha in place on a device-resident 8K RGBA image: every pixel read and written once
            # (10-bit: the integer un-premultiply in 16-bit containers; 16-bit: the IEEE division stays)
            depth = int(name[-2:]) if name[-2:] in ("10", "16") else 8
            rgb = abi.make_rgb(7680, 4320, depth, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=avoid)
            synth.fill_rgb(rgb, 0x5151)
            if depth == 10:
                rgb.pixels.view(np.uint16)[...] &= 1023
            drgb = device.DeviceRGB(rgb, upload=True)
            fn = lib.avifhipRGBImageUnpremultiplyAlphaAsync if name.startswith("unpremul") else lib.avifhipRGBImagePremultiplyAlphaAsync
            best = host_clock(lambda: native.check(fn(drgb.struct, None)))
            px, bpp, ms = 7680 * 4320, (8.0 if depth == 8 else 16.0), best
        elif name in ("cfg2_alpha", "cfg2_premul"):
            # images with an alpha plane: 8K 8-bit 4:2:0 + A -> RGBA8 bilinear, straight or premultiplied (what Android's bitmaps take:
            # android_jni/.../libavif_jni.cc); 1.5 + 1 + 4 B/px
            pair = y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, alpha=True, premult=(name == "cfg2_premul"), avoid=avoid)
            px, bpp, ms = 7680 * 4320, 6.5, time_y2r(pair)
        elif name in ("cfg2_keep", "cfg2_keep16"):
            # rgb->ignoreAlpha on RGBA (an application that wants RGBX): the destination's alpha samples stay as they are (src/reformat.c:1449-1450);
            # the fp32 arithmetic only -- libyuv writes 255 whatever the flag says.  8K 8-bit 4:2:0 -> RGBA8 (1.5 + 4 B/px, the alpha byte read
            # back inside the pixel's own cache line), or 10-bit -> RGBA(10) (3 + 8)
            if arith == "integer":
                continue
            deep = name == "cfg2_keep16"
            pair = y2r(7680, 4320, 10 if deep else 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 10 if deep else 8, avoid=avoid)
            pair[1].struct.ignoreAlpha = 1
            # (bytes: the planes + the destination pixels READ for their alpha + the pixels written: 1.5 + 4 + 4, or 3 + 8 + 8)
            px, bpp, ms = 7680 * 4320, (19.0 if deep else 9.5), time_y2r(pair)
        elif name == "cfg2_rgb":
            # 3-byte pixels (what avifdec hands to its JPEG / opaque PNG writers): 8K 8-bit 4:2:0 -> RGB8, bilinear, 1.5 + 3 B/px
            pair = y2r(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=avoid, rgb_format=abi.AVIF_RGB_FORMAT_RGB)
            px, bpp, ms = 7680 * 4320, 4.5, time_y2r(pair)
        elif name in ("cfg2_4k", "cfg2_4k_seq", "cfg2_4k_cold", "cfg2_4k_seq_cold", "cfg2_seq", "cfg2_seq_cold", "cfg2_cold"):
            # bench.py's regimes (round 6: the table's 4K row relaunched ONE frame -- 45.6 MB, pixels included, all of it cache-resident: 8.2 us --
            # where bench.py cycles four -- 9.0 us; same kernel, different regime, now the same measurement): frames cycled x frames per launch
            w, h = (3840, 2160) if "4k" in name else (7680, 4320)
            frames = (24 if "4k" in name else 12) if name.endswith("cold") else 4
            per = 4 if "_seq" in name else 1
            pairs = [y2r(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, avoid=avoid, seed=0x12345678 + k % 4) for k in range(frames)]
            px, bpp, ms = w * h, 5.5, time_y2r_cycle(pairs, per)
            extra["frames_cycled"], extra["frames_per_launch"] = frames, per
            extra["regime"] = "HBM (nothing cache-resident)" if name.endswith("cold") else "planes L3-resident"
        elif name in ("cfg2", "cfg2n"):
            w, h = 7680, 4320
            pair = y2r(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, up=NEAR if name == "cfg2n" else BIL, avoid=avoid)
            px, bpp, ms = w * h, 5.5, time_y2r(pair)
        elif name in ("cfg2_unpremul", "cfg3_unpremul"):
            # images stored PREMULTIPLIED converted to straight-alpha pixels: the un-multiply inside the conversion (src/reformat.c:894-947);
            # cfg2's planes + alpha -> RGBA8 (6.5 B/px), cfg3's -> RGBA16 (16 B/px)
            if name == "cfg2_unpremul":
                img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True, alpha_premultiplied=True)
                rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)
                px, bpp = 7680 * 4320, 6.5
            else:
                img = abi.make_yuv(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, with_alpha=True, alpha_premultiplied=True)
                rgb = abi.make_rgb(7680, 4320, 16, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)
                px, bpp = 7680 * 4320, 16.0
            synth.fill_yuv(img, 0x12345678)
            pair = (device.DeviceYUV(img), device.DeviceRGB(rgb))
            ms = time_y2r_two_frames(pair) if name == "cfg3_unpremul" else time_y2r(pair, 100)
        elif name == "cfg3":
            pair = y2r(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, alpha=True, premult=True, avoid=avoid)
            px, bpp, ms = 7680 * 4320, 16.0, time_y2r_two_frames(pair)
        elif name in ("cfg4", "cfg4rgb", "cfg4_601", "cfg4_8k", "cfg4rgb_8k", "ident8_enc", "cfg4_premul_8k", "cfg4_unpremul_8k", "cfg4_ycgco_8k", "cfg4_444_8k"):
            fmt = abi.AVIF_RGB_FORMAT_RGB if name.startswith("cfg4rgb") else abi.AVIF_RGB_FORMAT_RGBA
            mc = 6 if name == "cfg4_601" else 1
            w, h = (7680, 4320) if name.endswith("_8k") or name == "ident8_enc" else (3840, 2160)  # the encode direction on the headline's frame size
            # pending alpha multiply (straight RGBA into a premultiplied image: what avifenc --premultiply asks for) / un-multiply on the encode side
            mul = {"cfg4_premul_8k": 1, "cfg4_unpremul_8k": 2}.get(name, 0)
            if mul and arith == "integer":
                continue  # (libyuv is never asked when alpha is pending: one arithmetic)
            rgb = abi.make_rgb(w, h, 8, fmt, avoid_libyuv=avoid, alpha_premultiplied=(mul == 2))
            synth.fill_rgb(rgb, 0x12345678, opaque=(mul == 0))
            if name == "ident8_enc":  # lossless encode (avifenc -l): 8K RGBA8 -> GBR planes 8-bit 4:4:4 full range + alpha, 4 + 3 + 1 B/px
                img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 0, with_alpha=True)
            elif name == "cfg4_444_8k":  # BT.709 limited 8-bit 4:4:4 + alpha (avifenc -y 444), 4 + 3 + 1 B/px
                img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
            elif name == "cfg4_ycgco_8k":  # YCgCo 8-bit 4:4:4 full range + alpha, 4 + 3 + 1 B/px
                if arith == "integer":
                    continue
                img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 8, with_alpha=True)
            else:
                img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, mc, with_alpha=(fmt == abi.AVIF_RGB_FORMAT_RGBA), alpha_premultiplied=(mul == 1))
            dimg, drgb = device.DeviceYUV(img), device.DeviceRGB(rgb, upload=True)
            px, bpp = w * h, (8.0 if name in ("ident8_enc", "cfg4_ycgco_8k", "cfg4_444_8k") else (6.5 if fmt == abi.AVIF_RGB_FORMAT_RGBA else 4.5))
            preheat(lambda n: lib.avifhipTimeRGBToYUV(dimg.struct, drgb.struct, 0, n, None))
            ms = settled(lambda: lib.avifhipTimeRGBToYUV(dimg.struct, drgb.struct, 4, 100, None))
        elif name in ("cfg4_cycled", "cfg4_seq"):
            # cfg4 as bench.py measures it: 8 frames cycled (431 MB: it streams), one frame per launch or four (avifhipImageRGBToYUVBatchAsync, round 6)
            enc = []
            for k in range(8):
                rgb = abi.make_rgb(3840, 2160, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=avoid)
                synth.fill_rgb(rgb, 0x12345678 + k % 2, opaque=True)
                img = abi.make_yuv(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
                enc.append((device.DeviceYUV(img, upload=False), device.DeviceRGB(rgb, upload=True)))
            imgs = (C.POINTER(abi.avifImage) * 8)(*[C.pointer(q[0].struct) for q in enc])
            rgbs = (C.POINTER(abi.avifRGBImage) * 8)(*[C.pointer(q[1].struct) for q in enc])
            per = 4 if name == "cfg4_seq" else 1
            if per == 1:
                preheat(lambda n: lib.avifhipTimeRGBToYUVCycle(8, imgs, rgbs, 0, n, None))
                ms = settled(lambda: lib.avifhipTimeRGBToYUVCycle(8, imgs, rgbs, 4, 100, None))
            else:
                preheat(lambda n: lib.avifhipTimeRGBToYUVBatchCycle(8, imgs, rgbs, per, 0, max(1, n // per), None) / per)
                ms = settled(lambda: lib.avifhipTimeRGBToYUVBatchCycle(8, imgs, rgbs, per, 2, 40, None)) / per
            px, bpp = 3840 * 2160, 6.5
            extra["frames_cycled"], extra["frames_per_launch"], extra["regime"] = 8, per, "HBM (431 MB cycled)"
        elif name in ("gray_enc_8k", "graya_enc_8k"):
            # gray sources of the encode direction (a grayscale PNG through avifenc): 8K GRAY8 -> 4:0:0 luma (1 + 1 B/px); GRAYA8 -> luma + alpha (2 + 2 B/px)
            if arith == "integer":
                continue  # libyuv is never asked for gray sources: one arithmetic
            with_a = name == "graya_enc_8k"
            rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_GRAYA if with_a else abi.AVIF_RGB_FORMAT_GRAY, avoid_libyuv=avoid)
            synth.fill_rgb(rgb, 0x12345678)
            img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV400, abi.AVIF_RANGE_FULL, 1, with_alpha=with_a)
            dimg, drgb = device.DeviceYUV(img), device.DeviceRGB(rgb, upload=True)
            preheat(lambda n: lib.avifhipTimeRGBToYUV(dimg.struct, drgb.struct, 0, n, None))
            px, bpp, ms = 7680 * 4320, (4.0 if with_a else 2.0), settled(lambda: lib.avifhipTimeRGBToYUV(dimg.struct, drgb.struct, 4, 100, None))
        elif name in ("cfg5", "cfg5_8"):
            pair = y2r(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 10 if name == "cfg5" else 8, avoid=avoid)
            px, bpp, ms = 1920 * 1080, (11.0 if name == "cfg5" else 7.0), time_y2r(pair)
        elif name == "cfg5x64_rot":
            # cfg5x64 with the OUTPUT buffers rotating between two sets: every batch carries a fresh descriptor table to the device (a decoder
            # that does not reuse its tile buffers), against cfg5x64's resident table
            if arith == "integer":
                continue
            pairs = [y2r(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 10, avoid=avoid, seed=0x12345678 + t) for t in range(64)]
            outs_b = [device.DeviceRGB(abi.make_rgb(1920, 1080, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)) for _ in range(64)]
            imgs = (C.POINTER(abi.avifImage) * 64)(*[C.pointer(p[0].struct) for p in pairs])
            rgbs = [(C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(p[1].struct) for p in pairs]), (C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(o.struct) for o in outs_b])]
            k = [0]

            def call():
                k[0] += 1
                native.check(lib.avifhipImageYUVToRGBBatchAsync(64, imgs, rgbs[k[0] & 1], None, None))
            best = host_clock(call, burst=50)
            px, bpp, ms = 64 * 1920 * 1080, 11.0, best
        elif name in ("cfg5x64", "cfg5x64_8"):
            rgb_depth = 10 if name == "cfg5x64" else 8
            pairs = [y2r(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, rgb_depth, avoid=avoid, seed=0x12345678 + t) for t in range(64)]
            imgs = (C.POINTER(abi.avifImage) * 64)(*[C.pointer(p[0].struct) for p in pairs])
            rgbs = (C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(p[1].struct) for p in pairs])
            for _ in range(int(PREHEAT_MS / 0.2) + 3):
                native.check(lib.avifhipImageYUVToRGBBatchAsync(64, imgs, rgbs, None, None))
            native.check(lib.avifhipSynchronize(None))
            CLOCK = "host"
            best, submit = 1e9, 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(10):
                    native.check(lib.avifhipImageYUVToRGBBatchAsync(64, imgs, rgbs, None, None))
                t1 = time.perf_counter()
                native.check(lib.avifhipSynchronize(None))
                best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
                submit = min(submit, (t1 - t0) / 10 * 1e3)
            extra["host_submit_us_per_batch"] = round(submit * 1e3, 1)  # the calling thread's own time per call (plans, table, launches)
            px, bpp, ms = 64 * 1920 * 1080, (11.0 if rgb_depth == 10 else 7.0), best
        elif name in ("scale_box4", "scale_up2", "scale_down_1_5"):
            # avifImageScale on 8-bit 4:2:0 planes: 8K -> 1080p (box), 4K -> 8K (2x upsampler), 8K -> 5120x2880 (bilinear down)
            if arith == "integer":
                continue  # one arithmetic: the vendored libyuv scaler's integers
            (sw, sh), (dw, dh) = {"scale_box4": ((7680, 4320), (1920, 1080)), "scale_up2": ((3840, 2160), (7680, 4320)),
                                  "scale_down_1_5": ((7680, 4320), (5120, 2880))}[name]
            src = abi.make_yuv(sw, sh, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
            synth.fill_yuv(src, 0x77)
            dst = abi.make_yuv(dw, dh, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
            dsrc, ddst = device.DeviceYUV(src), device.DeviceYUV(dst)
            call = lambda: native.check(lib.avifhipImageScaleAsync(dsrc.struct, ddst.struct, None))
            best = host_clock(call)
            kernel_name = native.last_kernel()
            # the job's byte-movement ceiling: every source sample read once, every destination sample written once, nothing computed
            preheat(lambda n: lib.avifhipTimeStreamCeilingScale(dsrc.struct, ddst.struct, 0, n, None))
            ceil_ms = settled(lambda: lib.avifhipTimeStreamCeilingScale(dsrc.struct, ddst.struct, 4, 100, None))
            CLOCK = "host"
            extra["ceiling_us"] = round(ceil_ms * 1e3, 2)
            extra["vs_ceiling"] = round(ceil_ms / best, 3)
            extra["kernel_override"] = kernel_name
            # algorithmic bytes: every source sample read once + every destination sample written once, per luma pixel of the LARGER image
            px = max(sw * sh, dw * dh)
            bpp, ms = 1.5 * (sw * sh + dw * dh) / px, best
        elif name in ("xform90", "xform180"):
            # avifApplyTransforms on the converted 8K RGBA8 image: clap crop + irot + imir in one pass (8 B/pixel)
            if arith == "integer":
                continue  # pure byte movement: no arithmetic family
            angle = 1 if name == "xform90" else 2
            src = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA)
            synth.fill_rgb(src, 0x1234)
            crop = abi.avifCropRect(8, 4, 7664, 4312)
            dw, dh = (4312, 7664) if angle == 1 else (7664, 4312)
            dst = abi.make_rgb(dw, dh, 8, abi.AVIF_RGB_FORMAT_RGBA, allocate=False)
            dsrc, ddst = device.DeviceRGB(src, upload=True), device.DeviceRGB(dst)
            call = lambda: native.check(lib.avifhipRGBImageTransformAsync(ddst.struct, dsrc.struct, C.byref(crop), 1, angle, 1, 1, None))
            best = host_clock(call)
            px, bpp, ms = 7664 * 4312, 8.0, best
        elif name in ("tail0", "tail180", "tail90", "tail90_two_pass", "tail180_two_pass", "tail0_10", "tail90_10", "tail90_10_two_pass",
                      "tail0_rgba10", "tail180_rgba10", "tail90_rgba10", "tail90_rgba10_two_pass", "tail90_nocrop", "tail90_rgba10_nocrop"):
            # the decode-side tail fused (avifhipImageYUVToRGBTransformedAsync): 8K 8-bit 4:2:0 -> RGBA8 bilinear with clap crop +
            # irot + imir, against the same result in two passes (conversion, then avifhipRGBImageTransformAsync).  5.5 B/pixel of
            # the cropped image is what HAS to move.  "_10": 10-bit planes -> RGBA8 (3 + 4 B/pixel); "_rgba10": 10-bit planes -> RGBA at the
            # image's depth, the API default (cfg5's pixels: 3 + 8 B/pixel, fp32 arithmetic in either library setting -- libyuv has no 16-bit outputs)
            wide = "_rgba10" in name
            if arith == "float" and "_10" in name:
                continue  # (covered by the integer row: the packed kernels; the fp32 twin of 10 -> 8 bits is not a default anybody gets)
            if arith == "integer" and wide:
                continue  # one arithmetic
            deep = "_10" in name or wide
            nocrop = name.endswith("_nocrop")  # the whole frame turned (no clap): destination runs start on cache-line boundaries
            angle = {"tail0": 0, "tail180": 2, "tail90": 1, "tail90_two_pass": 1, "tail180_two_pass": 2}[name.replace("_nocrop", "").replace("_rgba10", "").replace("_10", "")]
            img = abi.make_yuv(7680, 4320, 10 if deep else 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
            synth.fill_yuv(img, 0x12345678)
            dimg = device.DeviceYUV(img)
            crop = abi.avifCropRect(0, 0, 7680, 4320) if nocrop else abi.avifCropRect(8, 4, 7664, 4312)
            dw, dh = (crop.height, crop.width) if angle == 1 else (crop.width, crop.height)
            dst = abi.make_rgb(dw, dh, 10 if wide else 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)
            ddst = device.DeviceRGB(dst)
            if name.endswith("two_pass"):
                mid = abi.make_rgb(7680, 4320, 10 if wide else 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)
                dmid = device.DeviceRGB(mid)

                def call():
                    native.check(lib.avifhipImageYUVToRGBAsync(dimg.struct, dmid.struct, None))
                    native.check(lib.avifhipRGBImageTransformAsync(ddst.struct, dmid.struct, C.byref(crop), 1, angle, 1, 1, None))
            else:
                def call():
                    native.check(lib.avifhipImageYUVToRGBTransformedAsync(dimg.struct, ddst.struct, C.byref(crop), int(angle != 0), angle, int(angle != 0), 1, None))
            best = host_clock(call)
            px, bpp, ms = crop.width * crop.height, (11.0 if wide else 7.0 if deep else 5.5), best
        elif name in ("cfg5grid", "cfg5grid_8", "photo_grid", "cfg5grid_link", "cfg5grid_8_pass", "photo_grid_pass"):
            # (_pass: the tile batch + seam pass of rounds 1-3 forced, _link: one launch forced -- AVIFHIP_GRID_SEAM_PASS; without a suffix the
            #  library chooses: one launch for the packed 16-bit kernels and for canvases up to 32 megapixels)
            forced = "1" if name.endswith("_pass") else ("0" if name.endswith("_link") else None)
            gname = name.rsplit("_pass", 1)[0].rsplit("_link", 1)[0]
            if forced is None:
                os.environ.pop("AVIFHIP_GRID_SEAM_PASS", None)
            else:
                os.environ["AVIFHIP_GRID_SEAM_PASS"] = forced
            # BASELINE configs[4]: 8 x 8 grid of decoded 1920x1080 10-bit 4:2:0 tiles -> one 15360x8640 RGBA canvas, tiles
            # converted where they lie (avifhipGridYUVToRGBAsync: no YUV canvas, seams redone across tiles)
            # photo_grid: what a phone camera writes -- 4032 x 3024 8-bit 4:2:0 as 8 x 6 tiles of 512 x 512 (the last row cropped) -> RGBA8
            rgb_depth = 10 if gname == "cfg5grid" else 8
            cols, rows, tw, th, ow, oh, depth = (8, 6, 512, 512, 4032, 3024, 8) if gname == "photo_grid" else (8, 8, 1920, 1080, 15360, 8640, 10)
            tiles = []
            for t in range(cols * rows):
                img = abi.make_yuv(tw, th, depth, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED if depth == 10 else abi.AVIF_RANGE_FULL, 1 if depth == 10 else 6)
                synth.fill_yuv(img, 0x12345678 + t)
                tiles.append(device.DeviceYUV(img))
            rgb = abi.make_rgb(ow, oh, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=avoid, allocate=False)
            drgb = device.DeviceRGB(rgb)
            imgs = (C.POINTER(abi.avifImage) * (cols * rows))(*[C.pointer(t.struct) for t in tiles])
            grid = native.avifhipGrid(rows, cols, ow, oh)
            best = host_clock(lambda: native.check(lib.avifhipGridYUVToRGBAsync(C.byref(grid), imgs, None, 0, drgb.struct, None)))
            px, bpp, ms = ow * oh, (11.0 if rgb_depth == 10 else (7.0 if depth == 10 else 5.5)), best
        elif name in ("gainmap4k", "gainmap4k_half", "gainmap4k_cpu", "gainmap4k_rgb", "gainmap4k_photo", "gainmap4k_same"):
            # avifRGBImageApplyGainMap: 3840x2160 RGBA8 sRGB BT.709 base -> RGBA10 PQ BT.2020 HDR rendition, 8-bit 4:4:4 gain map of the
            # same size (or 4:2:0 at half size, rescaled on the device first).  Algorithmic bytes: base 4 + gain-map planes + output 8.
            CLOCK = "host"
            if arith == "integer":
                continue
            import ctypes
            half = name == "gainmap4k_half"
            w, h = 3840, 2160
            # (_rgb: 3-channel pixels on both sides -- the general kernel)
            gm_fmt = abi.AVIF_RGB_FORMAT_RGB if name == "gainmap4k_rgb" else abi.AVIF_RGB_FORMAT_RGBA
            base = abi.make_rgb(w, h, 8, gm_fmt, avoid_libyuv=False)
            synth.fill_rgb(base, 0x4242)
            gimg = abi.make_yuv(w // 2 if half else w, h // 2 if half else h, 8, abi.AVIF_PIXEL_FORMAT_YUV420 if half else abi.AVIF_PIXEL_FORMAT_YUV444,
                                abi.AVIF_RANGE_FULL, 6)
            synth.fill_yuv(gimg, 0x99)
            if name == "gainmap4k_photo":
                # what a photograph looks like to the table gathers: neighbouring pixels hold neighbouring codes (a smooth ramp across the frame with two
                # codes of noise) -- the lanes of a wave read a handful of table entries, not 64 random ones (round 6: the LDS bank-conflict counters)
                rng = np.random.default_rng(7)
                yy, xx = np.mgrid[0:h, 0:w]
                ramp = ((xx * 200 // w + yy * 55 // h) % 256).astype(np.int32)
                ch = base.channels()
                for k in range(3):
                    ch[:, :, k] = np.clip(ramp + 10 * k + rng.integers(-2, 3, size=ramp.shape), 0, 255).astype(np.uint8)
                ch[:, :, 3] = 255
                for pl in range(3):
                    gimg.planes[pl][:h, :w] = np.clip(128 + ramp // 4 + rng.integers(-1, 2, size=ramp.shape), 0, 255).astype(np.uint8)
            gm = abi.avifGainMap()
            for i in range(3):
                gm.gainMapMin[i].n, gm.gainMapMin[i].d = 0, 1
                gm.gainMapMax[i].n, gm.gainMapMax[i].d = 3, 1
                gm.gainMapGamma[i].n, gm.gainMapGamma[i].d = 1, 1
                gm.baseOffset[i].n, gm.baseOffset[i].d = 1, 64
                gm.alternateOffset[i].n, gm.alternateOffset[i].d = 1, 64
            gm.baseHdrHeadroom.n, gm.baseHdrHeadroom.d, gm.alternateHdrHeadroom.n, gm.alternateHdrHeadroom.d = 0, 1, 3, 1
            gm.useBaseColorSpace = 1
            clli, diag = abi.avifContentLightLevelInformationBox(), abi.avifDiagnostics()
            gain_bytes = (w * h * 3) if not half else (w * h * 3 // 8)
            px, bpp = w * h, ((3 if name == "gainmap4k_rgb" else 4) * w * h + gain_bytes + (6 if name == "gainmap4k_rgb" else 8) * w * h) / (w * h)
            if name == "gainmap4k_cpu":
                # the oracle (= the reference's arithmetic, one thread) on this host, for the ratio
                sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
                import oracle_lib
                o = oracle_lib.oracle()
                gm.image = C.pointer(gimg.struct)
                tone = abi.make_rgb(w, h, 10, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False, allocate=False)
                t0 = time.perf_counter()
                assert o.oracleRGBImageApplyGainMap(base.struct, 1, 13, C.byref(gm), 3.0, 9, 16, tone.struct, C.byref(clli), 1) == 0
                ms = (time.perf_counter() - t0) * 1e3
                ctypes.CDLL(None).free(ctypes.c_void_p(tone.struct.pixels))
            else:
                dbase, dgimg = device.DeviceRGB(base, upload=True), device.DeviceYUV(gimg)
                gm.image = C.pointer(dgimg.struct)
                tone = abi.make_rgb(w, h, 10, gm_fmt, avoid_libyuv=False, allocate=False)
                dout = device.DeviceRGB(tone)
                # (_same: the output keeps the base image's primaries -- no fp64 matrix per pixel, src/gainmap.c:263-266: what the kernel costs without it)
                out_primaries = 1 if name == "gainmap4k_same" else 9
                call = lambda: native.check(lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, 1, 13, C.byref(gm), 3.0, out_primaries, 16, dout.struct, C.byref(clli),
                                                                                 C.byref(diag), None))
                for _ in range(3):
                    call()
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(10):
                        call()  # waits for its stream: result code and CLLI depend on the pixels
                    best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
                ms = best
        elif name in ("gmcompute4k", "gmcompute4k_cpu", "gmcompute4k_dev"):
            # avifRGBImageComputeGainMap: 3840x2160 RGBA8 sRGB base + RGBA10 PQ BT.2020 alternate -> 8-bit 4:4:4 gain map + metadata.
            # The entry point takes HOST images (the encode side is host-driven): the time includes the PCIe transfers both ways.
            if arith == "integer":
                continue
            sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
            import gainmap_cases as G
            c = G.ComputeCase(3840, 2160, alt_primaries=9, seed=3)
            base, alt = G.make_compute_inputs(c)
            gm, img = G.make_compute_gain_map(c)
            diag = abi.avifDiagnostics()
            px, bpp = c.w * c.h, 4 + 8 + 3
            if name == "gmcompute4k_dev":
                # round 6: everything in device memory (avifhipRGBImageComputeGainMapAsync): what the passes cost without the host link.  Wall clock
                # around back-to-back calls (each waits for its own first passes: the metadata is a function of every pixel)
                dbase, dalt = device.DeviceRGB(base, upload=True), device.DeviceRGB(alt, upload=True)
                host_gm = abi.make_yuv(c.w, c.h, c.gm_depth, c.gm_format, c.gm_range, c.gm_matrix)
                dgm = device.DeviceYUV(host_gm, upload=False)
                gm.image = C.pointer(dgm.struct)
                t = lib.avifhipTimeRGBImageComputeGainMap
                assert t(dbase.struct, 1, 13, dalt.struct, 9, 16, C.byref(gm), 3, 10, None) > 0, lib.avifhipLastError()
                ms = sorted(t(dbase.struct, 1, 13, dalt.struct, 9, 16, C.byref(gm), 1, 10, None) for _ in range(5))[2]
                extra["clock_note"] = "wall clock around 10 back-to-back device-resident calls, median of 5"
            elif name == "gmcompute4k_cpu":
                import oracle_lib
                t0 = time.perf_counter()
                assert oracle_lib.oracle().oracleRGBImageComputeGainMap(base.struct, 1, 13, alt.struct, 9, 16, C.byref(gm), 1) == 0
                ms = (time.perf_counter() - t0) * 1e3
            else:
                call = lambda: native.check(lib.avifhipRGBImageComputeGainMap(base.struct, 1, 13, alt.struct, 9, 16, C.byref(gm), C.byref(diag)))
                for _ in range(2):
                    call()
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter()
                    call()
                    best = min(best, (time.perf_counter() - t0) * 1e3)
                ms = best
        else:
            raise SystemExit(f"unknown configuration {name}")
        gbps = bpp * px / (ms * 1e-3) / 1e9
        out.append({"config": name, "arithmetic": arith, "kernel": extra.pop("kernel_override", None) or native.last_kernel(), "clock": CLOCK, "us": round(ms * 1e3, 2), "megapixels_per_s": round(px / 1e6 / (ms * 1e-3)),
                    "algorithmic_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4), **extra})
    lib.avifhipSetArithmetic(0)
    return out


if __name__ == "__main__":
    for n in sys.argv[1:] or ["cfg2", "cfg2n", "cfg3", "cfg4", "cfg4rgb", "cfg4_601", "cfg5", "cfg5_8", "cfg5x64", "cfg5x64_8"]:
        for line in run(n):
            print(json.dumps(line), flush=True)
