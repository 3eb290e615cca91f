"""Launch-geometry sweep of the packed 16-bit integer kernels (tile_pk_impl.h) on the GPU box: strips per wave, waves side by
side, tile order (raster / per-XCD chunks of n tile rows), with 4 frames cycled (inputs fit the 256 MB Infinity Cache) and
with 12 (nothing does).  HIP events around bursts of 48 launches; median of 5 bursts."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, device, native, synth  # noqa: E402

lib = native.load()
lib.avifhipSetArithmetic(0)
lib.avifhipSetTuning.argtypes = [C.c_uint32]
W, H = (7680, 4320) if len(sys.argv) < 2 or sys.argv[1] == "8k" else (3840, 2160)
NF = 12


def frames(n, up=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR):
    out = []
    for f in range(n):
        img = abi.make_yuv(W, H, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        synth.fill_yuv(img, 0x12345678 + f)
        rgb = abi.make_rgb(W, H, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=up, avoid_libyuv=False, allocate=False)
        out.append((device.DeviceYUV(img), device.DeviceRGB(rgb)))
    return out


def cyc(fr):
    n = len(fr)
    return n, (C.POINTER(abi.avifImage) * n)(*[C.pointer(f[0].struct) for f in fr]), (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(f[1].struct) for f in fr])


def med(a, iters=48, reps=5):
    t = sorted(lib.avifhipTimeYUVToRGBCycle(*a, 4, iters, None) for _ in range(reps))
    return t[len(t) // 2] * 1e3


fr = frames(NF)
a4, a12 = cyc(fr[:4]), cyc(fr)
for _ in range(6):  # clock ramp
    lib.avifhipTimeYUVToRGBCycle(*a4, 0, 400, None)
res = []
for strips in (4, 2):
    for wx in (3, 2, 1):  # 1 + log2(waves side by side)
        for banded, chunk in ((0, 0), (1, 1), (1, 2), (1, 4), (1, 8)):
            tune = banded | (strips << 8) | (wx << 16) | (chunk << 20)
            lib.avifhipSetTuning(tune)
            t4, t12 = med(a4), med(a12)
            res.append((t4, t12, strips, 1 << (wx - 1), chunk if banded else "raster"))
            print(f"strips/wave {strips}  waves side by side {1 << (wx - 1)}  order {('xcd chunks of %d tile rows' % chunk) if banded else 'raster':28s}"
                  f" 4 frames {t4:6.2f} us ({182.4768 / t4 / 8:.3f})   {NF} frames {t12:6.2f} us ({182.4768 / t12 / 8:.3f})", flush=True)
print("best by 4 frames:", sorted(res)[:6])
print("best by %d frames:" % NF, sorted(res, key=lambda r: r[1])[:6])
lib.avifhipSetTuning(1)
print("default tuning: 4 frames %.2f us, %d frames %.2f us" % (med(a4), NF, med(a12)))
