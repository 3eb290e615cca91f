"""Why does the cfg2 kernel's duration spread between runs?  Run on the GPU box.
Prints per-burst durations from a cold chip, after idling, for separately allocated frames vs one slab, and clocks."""
import ctypes as C
import subprocess
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, device, native, synth  # noqa: E402

lib = native.load()
lib.avifhipSetArithmetic(0)
W, H = 7680, 4320


def clocks(tag):
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "edge"))]
        print(tag, " | ".join(keep[:8]), flush=True)
    except Exception as e:  # noqa: BLE001
        print(tag, "rocm-smi failed", e)


def make_frames(n):
    frames = []
    for f in range(n):
        img = abi.make_yuv(W, H, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        synth.fill_yuv(img, 0x12345678 + f)
        rgb = abi.make_rgb(W, H, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False, allocate=False)
        frames.append((device.DeviceYUV(img), device.DeviceRGB(rgb)))
    return frames


def cyc(frames):
    n = len(frames)
    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(f[0].struct) for f in frames])
    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(f[1].struct) for f in frames])
    return n, imgs, rgbs


t_start = time.time()
clocks("idle before anything:")
frames = make_frames(4)
n, imgs, rgbs = cyc(frames)
print("setup s", round(time.time() - t_start, 2))
cold = [round(lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 0, 40, None) * 1e3, 2) for _ in range(40)]
print("cold bursts of 40 (us/launch):", cold, flush=True)
clocks("after 40 bursts:")
long = [round(lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 0, 2000, None) * 1e3, 2) for _ in range(5)]
print("bursts of 2000:", long, flush=True)
clocks("after long bursts:")
same = [round(lib.avifhipTimeYUVToRGB(frames[0][0].struct, frames[0][1].struct, 0, 40, None) * 1e3, 2) for _ in range(10)]
print("same frame bursts of 40:", same, flush=True)
one = [round(lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 0, 1, None) * 1e3, 2) for _ in range(10)]
print("single launches (event pair around one launch):", one, flush=True)
time.sleep(3.0)
clocks("after 3 s idle:")
again = [round(lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 0, 40, None) * 1e3, 2) for _ in range(12)]
print("after idle, bursts of 40:", again, flush=True)
for nf in (2, 8):
    fr = make_frames(nf)
    a = cyc(fr)
    lib.avifhipTimeYUVToRGBCycle(*a, 0, 200, None)
    print(nf, "frames cycled:", [round(lib.avifhipTimeYUVToRGBCycle(*a, 0, 40, None) * 1e3, 2) for _ in range(6)], flush=True)
    del fr, a
# fp32 twin on the same frames
for _, drgb in frames:
    drgb.struct.avoidLibYUV = 1
print("fp32 twin:", [round(lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 0, 40, None) * 1e3, 2) for _ in range(8)], flush=True)
