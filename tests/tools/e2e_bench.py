"""End-to-end (PCIe-inclusive) rate of the synchronous drop-in calls on HOST buffers, the way seam A / seam B callers use
the library: pageable user memory in, pageable user memory out.  Never the bench's `value` (that is HBM-resident); this
is the number an unmodified avifdec sees.  Usage: python tests/tools/e2e_bench.py"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, native, synth  # noqa: E402

lib = native.load()


def run(name, w, h, depth, fmt, rng, mc, rgb_depth, reps=8, alpha=False, premult=False):
    img = abi.make_yuv(w, h, depth, fmt, rng, mc, with_alpha=alpha)
    synth.fill_yuv(img)
    rgb = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False, alpha_premultiplied=premult)
    native.check(lib.avifhipImageYUVToRGB(img.struct, rgb.struct))
    best, total = 1e9, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        native.check(lib.avifhipImageYUVToRGB(img.struct, rgb.struct))
        dt = time.perf_counter() - t0
        best, total = min(best, dt), total + dt
    in_bytes = sum(p.nbytes for p in img.planes if p is not None) + (img.alpha.nbytes if img.alpha is not None else 0)
    print(json.dumps({"config": name, "call": "avifhipImageYUVToRGB (host buffers)", "best_ms": round(best * 1e3, 2), "mean_ms": round(total / reps * 1e3, 2),
                      "megapixels_per_s": round(w * h / 1e6 / best), "pcie_GBps": round((in_bytes + rgb.pixels.nbytes) / best / 1e9, 1), "kernel": native.last_kernel()}), flush=True)


if __name__ == "__main__":
    run("cfg2", 7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8)
    run("4K", 3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8)
    run("1080p", 1920, 1080, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8)
    run("cfg3", 7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, reps=4, alpha=True, premult=True)
