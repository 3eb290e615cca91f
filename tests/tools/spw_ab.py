"""Strips per wave of the RGB -> YUV tile kernels (AVIFHIP_R2Y_SPW = 1 / 2 / 4), interleaved in one process over cfg4's shape at 4K with 8 frames
cycled (streaming), 4K with one frame (cache-resident) and 1080p with 8 frames, with the byte-movement ceiling of each job:
`python tests/tools/spw_ab.py` -- what kernels_r2y_tile.hip's size rule was chosen from."""
import ctypes as C, json, os, sys
from statistics import median
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stream_sweep as S
from libavif_amd import abi, device, native, synth
lib = S.lib
lib.avifhipSetArithmetic(0); lib.avifhipSetTuning(1)
def frames(w, h, n):
    enc = []
    for k in range(n):
        rgb = abi.make_rgb(w, h, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
        synth.fill_rgb(rgb, 0x12345678 + k % 2, opaque=True)
        img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
        enc.append((device.DeviceYUV(img, upload=False), device.DeviceRGB(rgb, upload=True)))
    return enc
for (w, h, n) in ((3840, 2160, 8), (3840, 2160, 1), (1920, 1080, 8)):
    enc = frames(w, h, n)
    cnt, imgs, rgbs = S.arr(enc)
    got = {}
    for rnd in range(5):
        for spw in ("1", "2", "4"):
            os.environ["AVIFHIP_R2Y_SPW"] = spw
            lib.avifhipTimeRGBToYUVCycle(cnt, imgs, rgbs, 20, 100, None)
            got.setdefault(spw, []).append(lib.avifhipTimeRGBToYUVCycle(cnt, imgs, rgbs, 20, 400, None) * 1e3)
    os.environ.pop("AVIFHIP_R2Y_SPW", None)
    ceil = lib.avifhipTimeStreamCeilingRGBToYUV(cnt, imgs, rgbs, 20, 400, None) * 1e3 if hasattr(lib, "avifhipTimeStreamCeilingRGBToYUV") else None
    print(json.dumps({"job": f"{w}x{h} x{n}", "us": {k: round(median(v), 2) for k, v in got.items()}, "ceiling_us": ceil and round(ceil, 2)}), flush=True)
    del enc
