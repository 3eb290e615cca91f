"""How the kernel-alone timings of bench.py behave under sustained load (run on the GPU box): 40 consecutive event-timed bursts of 40
launches of one arithmetic, printed one by one -- an fp32 kernel that starts at its best-case duration and settles above it shows clock /
power management, not noise.  python tests/tools/sustain_probe.py [integer|fp32] ..."""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, device, native, synth  # noqa: E402

lib = native.load()
lib.avifhipSetArithmetic(0)
frames = []
for f in range(4):
    img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
    synth.fill_yuv(img, 0x12345678 + f)
    rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, allocate=False)
    frames.append((device.DeviceYUV(img), device.DeviceRGB(rgb)))
imgs = (C.POINTER(abi.avifImage) * 4)(*[C.pointer(f[0].struct) for f in frames])
rgbs = (C.POINTER(abi.avifRGBImage) * 4)(*[C.pointer(f[1].struct) for f in frames])
for fam in (sys.argv[1:] or ["integer", "fp32", "integer", "fp32"]):
    for _, r in frames:
        r.struct.avoidLibYUV = 0 if fam == "integer" else 1
    t0 = time.perf_counter()
    xs = [lib.avifhipTimeYUVToRGBCycle(4, imgs, rgbs, 4, 40, None) * 1e3 for _ in range(40)]
    dt = time.perf_counter() - t0
    print(f"{fam:8s} {native.last_kernel()}  min {min(xs):.2f} median {sorted(xs)[20]:.2f} max {max(xs):.2f} us  ({dt*1e3:.0f} ms wall for 40 bursts)")
    print("   " + " ".join(f"{x:.1f}" for x in xs))
    xs = [lib.avifhipTimeYUVToRGBCycle(4, imgs, rgbs, 4, 400, None) * 1e3 for _ in range(8)]
    print("   bursts of 400: " + " ".join(f"{x:.1f}" for x in xs))
