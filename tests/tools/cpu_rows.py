"""CPU rows beside the GPU numbers, timed on the host cores of the machine this runs on:
  cfg2 (8K 8-bit 4:2:0 BT.709 limited -> RGBA8 bilinear): the reference compiled from its sources (fp32 path) with maxThreads = 1
  and 8, and -- where Pillow's bundled libavif (built WITH libyuv) is importable -- the default libyuv path, one thread;
  cfg5 (64 tiles 1920x1080 10-bit 4:2:0 -> RGBA, API defaults): one single-threaded conversion per tile over a pool of host
  threads (ctypes releases the GIL), the CPU analogue of the tile farm.
One JSON line per row: python tests/tools/cpu_rows.py"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib  # noqa: E402
from libavif_amd import abi, synth  # noqa: E402


def best_of(fn, n):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    cores = os.cpu_count() or 1
    ref, pil = oracle_lib.ref(), oracle_lib.pillow()
    img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
    synth.fill_yuv(img, 0x12345678)
    mp = 7680 * 4320 / 1e6
    rows = []
    if ref is not None:
        for threads in (1, 8):
            rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=True, max_threads=threads)
            t = best_of(lambda: ref.avifImageYUVToRGB(img.struct, rgb.struct), 3)
            rows.append({"config": "cfg2", "implementation": "reference from source, built-in fp32 path", "maxThreads": threads, "ms": round(t * 1e3, 1),
                         "megapixels_per_s": round(mp / t, 1), "note": "the reference runs 4:2:0 bilinear on one thread whatever maxThreads says"})
        # nearest upsampling does use the threads
        rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_NEAREST, avoid_libyuv=True, max_threads=8)
        t = best_of(lambda: ref.avifImageYUVToRGB(img.struct, rgb.struct), 5)
        rows.append({"config": "cfg2 nearest", "implementation": "reference from source, built-in fp32 path", "maxThreads": 8, "ms": round(t * 1e3, 1),
                     "megapixels_per_s": round(mp / t, 1)})
    if pil is not None:
        rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
        t = best_of(lambda: pil.avifImageYUVToRGB(img.struct, rgb.struct), 5)
        rows.append({"config": "cfg2", "implementation": "libavif 1.4.1 + libyuv 1922 (Pillow's binary), default path", "maxThreads": 1, "ms": round(t * 1e3, 1),
                     "megapixels_per_s": round(mp / t, 1)})
    # cfg1 / cfg3 / cfg4: one thread, maxThreads = 8 (cfg3's 4:4:4 path does use them; RGB->YUV has no threading), libyuv build
    def y2r_rows(cfg, w, h, depth, fmt, rng, mc, rgb_depth, up, alpha=False, premult=False, reps=5):
        im = abi.make_yuv(w, h, depth, fmt, rng, mc, with_alpha=alpha)
        synth.fill_yuv(im, 0x12345678)
        mpx = w * h / 1e6
        for lib_, name_, avoid, thr in ((ref, "reference from source, built-in fp32 path", True, 1), (ref, "reference from source, built-in fp32 path", True, 8),
                                        (pil, "libavif 1.4.1 + libyuv 1922 (Pillow's binary), default path", False, 1)):
            if lib_ is None:
                continue
            out = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=up, avoid_libyuv=avoid, alpha_premultiplied=premult, max_threads=thr)
            t = best_of(lambda: lib_.avifImageYUVToRGB(im.struct, out.struct), reps)
            rows.append({"config": cfg, "implementation": name_, "maxThreads": thr, "ms": round(t * 1e3, 3), "megapixels_per_s": round(mpx / t, 1)})

    y2r_rows("cfg1", 256, 256, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_FULL, 6, 8, abi.AVIF_CHROMA_UPSAMPLING_AUTOMATIC, reps=20)
    y2r_rows("cfg3", 7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, abi.AVIF_CHROMA_UPSAMPLING_AUTOMATIC, alpha=True, premult=True, reps=3)
    for lib_, name_, avoid in ((ref, "reference from source, built-in fp32 path", True), (pil, "libavif 1.4.1 + libyuv 1922 (Pillow's binary), default path (libyuv is BT.601-only: "
                                                                                        "BT.709 runs the built-in path)", False)):
        if lib_ is None:
            continue
        src = abi.make_rgb(3840, 2160, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=avoid)
        synth.fill_rgb(src, 0xCAFEBABE, opaque=True)
        dst = abi.make_yuv(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
        t = best_of(lambda: lib_.avifImageRGBToYUV(dst.struct, src.struct), 5)
        rows.append({"config": "cfg4", "implementation": name_, "maxThreads": 1, "ms": round(t * 1e3, 2), "megapixels_per_s": round(3840 * 2160 / 1e6 / t, 1),
                     "note": "avifImageRGBToYUV has no threading (src/reformat.c:221-571)"})
    # cfg5: 64 tiles over a pool of host threads
    lib, name = (ref, "reference from source, built-in fp32 path") if ref is not None else (pil, "Pillow's libavif")
    if lib is not None:
        tiles = []
        for k in range(64):
            timg = abi.make_yuv(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
            if k < 4:
                synth.fill_yuv(timg, 0x12345678 + k)
            else:
                for p in range(3):
                    timg.planes[p][...] = tiles[k % 4][0].planes[p]
            trgb = abi.make_rgb(1920, 1080, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
            tiles.append((timg, trgb))
        workers = min(64, cores)
        with ThreadPoolExecutor(workers) as pool:
            def run():
                list(pool.map(lambda t: lib.avifImageYUVToRGB(t[0].struct, t[1].struct), tiles))
            t = best_of(run, 3)
        rows.append({"config": "cfg5 (64 tiles)", "implementation": name + ", one single-threaded conversion per tile", "host_threads": workers, "ms": round(t * 1e3, 1),
                     "megapixels_per_s": round(64 * 1920 * 1080 / 1e6 / t, 1)})
    for r in rows:
        r["host_cores"] = cores
        print(json.dumps(r))


if __name__ == "__main__":
    main()
