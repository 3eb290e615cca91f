"""End-to-end (PCIe-inclusive) rate of the synchronous drop-in calls on HOST buffers, the way seam A / seam B callers use the
library: pageable user memory in, pageable user memory out.  Never the bench's `value` (that is HBM-resident); this is the number
an unmodified avifdec / avifenc sees.  One JSON line per row:
    python tests/tools/e2e_bench.py            # BASELINE's five configurations through the C ABI, then seam B (hip-backed libavif)
Rows: cfg1-cfg5 through avifhipImageYUVToRGB / avifhipImageRGBToYUV; cfg5 also as one canvas through avifhipImageYUVToRGBRects;
the reference's own entry point over the hooks (oracle/_ref/libavif_hipbackend.so) with maxThreads = 1 and 8."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from libavif_amd import abi, farm, native, synth  # noqa: E402

lib = native.load()
lib.avifhipSetArithmetic(0)
BIL = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR


def timeit(fn, reps):
    fn()
    fn()
    best, total = 1e9, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best, total = min(best, dt), total + dt
    return best, total / reps


def nbytes_yuv(img):
    return sum(p.nbytes for p in img.planes if p is not None) + (img.alpha.nbytes if img.alpha is not None else 0)


def row(name, call, w, h, best, mean, moved, extra=None):
    d = {"config": name, "call": call, "best_ms": round(best * 1e3, 3), "mean_ms": round(mean * 1e3, 3), "megapixels_per_s": round(w * h / 1e6 / best),
         "host_link_GBps": round(moved / best / 1e9, 1), "kernel": native.last_kernel()}
    d.update(extra or {})
    print(json.dumps(d), flush=True)


def y2r(name, w, h, depth, fmt, rng, mc, rgb_depth, reps=8, alpha=False, premult=False, up=BIL, fn=None, call="avifhipImageYUVToRGB (host buffers)", max_threads=1,
        avoid=False, extra=None):
    img = abi.make_yuv(w, h, depth, fmt, rng, mc, with_alpha=alpha)
    synth.fill_yuv(img)
    rgb = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=up, avoid_libyuv=avoid, alpha_premultiplied=premult, max_threads=max_threads)
    f = fn or lib.avifhipImageYUVToRGB

    def go():
        assert f(img.struct, rgb.struct) == 0

    best, mean = timeit(go, reps)
    row(name, call, w, h, best, mean, nbytes_yuv(img) + rgb.pixels.nbytes, extra)


def r2y(name, w, h, reps=8):
    rgb = abi.make_rgb(w, h, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
    synth.fill_rgb(rgb, 0xCAFEBABE, opaque=True)
    img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)

    def go():
        native.check(lib.avifhipImageRGBToYUV(img.struct, rgb.struct))

    best, mean = timeit(go, reps)
    row(name, "avifhipImageRGBToYUV (host buffers)", w, h, best, mean, nbytes_yuv(img) + rgb.pixels.nbytes)


def premultiply(name, depth, reps=6):
    """avifRGBImagePremultiplyAlpha in place on a host-resident 8K RGBA image (what the seam-B hook and avifenc --premultiply see)."""
    rgb = abi.make_rgb(7680, 4320, depth, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
    synth.fill_rgb(rgb, 0x7171)

    def go():
        native.check(lib.avifhipRGBImagePremultiplyAlpha(rgb.struct))

    best, mean = timeit(go, reps)
    row(name, "avifhipRGBImagePremultiplyAlpha (host buffer, in place)", 7680, 4320, best, mean, 2 * rgb.pixels.nbytes)


def cfg5():
    W, H, TW, TH = 15360, 8640, 1920, 1080
    # (a) 64 separate tile images, one call each
    tiles = []
    for k in range(64):
        t = abi.make_yuv(TW, TH, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        if k < 2:
            synth.fill_yuv(t, 0x12345678 + k)
        else:
            for p in range(3):
                t.planes[p][...] = tiles[k % 2][0].planes[p]
        tiles.append((t, abi.make_rgb(TW, TH, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False)))

    def each():
        for t, o in tiles:
            native.check(lib.avifhipImageYUVToRGB(t.struct, o.struct))

    best, mean = timeit(each, 3)
    row("cfg5 (64 tiles, one call each)", "avifhipImageYUVToRGB x 64 (host buffers)", W, H, best, mean, 64 * (nbytes_yuv(tiles[0][0]) + tiles[0][1].pixels.nbytes))
    del tiles
    # (b) the stitched canvas, its 64 rectangles in one call (what one rank of the farm does with ALL tiles)
    canvas = abi.make_yuv(W, H, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
    synth.fill_yuv(canvas, 0x12345678)
    out = abi.make_rgb(W, H, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False)
    rects = farm.grid_rects(W, H, TW, TH)
    crops = (abi.avifCropRect * len(rects))(*[abi.avifCropRect(*r) for r in rects])

    def rects_call():
        native.check(lib.avifhipImageYUVToRGBRects(canvas.struct, out.struct, crops, len(rects)))

    best, mean = timeit(rects_call, 3)
    up, down = C.c_uint64(0), C.c_uint64(0)
    lib.avifhipLastTransferBytes(C.byref(up), C.byref(down))
    row("cfg5 (canvas, 64 rectangles)", "avifhipImageYUVToRGBRects (host canvas)", W, H, best, mean, up.value + down.value, {"bytes_up": up.value, "bytes_down": down.value})

    def whole():
        native.check(lib.avifhipImageYUVToRGB(canvas.struct, out.struct))

    best, mean = timeit(whole, 3)
    row("cfg5 (canvas, whole-image call)", "avifhipImageYUVToRGB (host buffers)", W, H, best, mean, nbytes_yuv(canvas) + out.pixels.nbytes)


def seam_b():
    import oracle_lib

    so = oracle_lib.ORACLE_DIR / "_ref" / "libavif_hipbackend.so"
    if not so.exists():
        print(json.dumps({"config": "seam B", "skipped": "oracle/_ref/libavif_hipbackend.so not built"}))
        return
    be = oracle_lib._bind_libavif(C.CDLL(os.fspath(so), mode=os.RTLD_LOCAL))
    for threads in (1, 8):
        # nearest upsampling: the reference splits the rows over maxThreads pthreads, each calling the hook for its band
        y2r("cfg2 nearest", 7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, up=abi.AVIF_CHROMA_UPSAMPLING_NEAREST, fn=be.avifImageYUVToRGB,
            call=f"avifImageYUVToRGB of the hip-backed libavif (seam B), maxThreads = {threads}", max_threads=threads, extra={"maxThreads": threads})
        y2r("cfg2", 7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, fn=be.avifImageYUVToRGB,
            call=f"avifImageYUVToRGB of the hip-backed libavif (seam B), maxThreads = {threads}", max_threads=threads, extra={"maxThreads": threads})
        y2r("cfg3", 7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, reps=4, alpha=True, premult=True, fn=be.avifImageYUVToRGB,
            call=f"avifImageYUVToRGB of the hip-backed libavif (seam B), maxThreads = {threads}", max_threads=threads, extra={"maxThreads": threads})


if __name__ == "__main__":
    if sys.argv[1:] == ["premultiply"]:
        premultiply("premultiply 8K RGBA8", 8)
        premultiply("premultiply 8K RGBA16", 16)
        sys.exit(0)
    y2r("cfg1", 256, 256, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_FULL, 6, 8, up=abi.AVIF_CHROMA_UPSAMPLING_AUTOMATIC, reps=50)
    y2r("cfg2", 7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8)
    y2r("cfg2 at 4K", 3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8)
    y2r("cfg2 at 1080p", 1920, 1080, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, reps=20)
    y2r("cfg3", 7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, reps=4, alpha=True, premult=True)
    r2y("cfg4", 3840, 2160)
    r2y("cfg4 at 8K", 7680, 4320, reps=4)
    cfg5()
    seam_b()
