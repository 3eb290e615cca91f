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
