"""A/B kernel timings on the GPU box (HIP events on the launch stream, same process, interleaved variants).
Usage: python tests/tools/ab_bench.py [cfg2|cfg2n|rgb|cfg3|cfg5] ..."""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from libavif_amd import abi, device, native, synth  # noqa: E402

lib = native.load()
lib.avifhipSetTuning.argtypes = [ctypes.c_uint32]


def setup(name):
    bil = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR
    if name == "cfg2":
        img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=bil, allocate=False)
        bpp = 5.5
    elif name == "cfg2n":
        img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_NEAREST, allocate=False)
        bpp = 5.5
    elif name == "rgb":
        img = abi.make_yuv(7680, 4320, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        rgb = abi.make_rgb(7680, 4320, 8, abi.AVIF_RGB_FORMAT_RGB, upsampling=bil, allocate=False)
        bpp = 4.5
    elif name == "cfg3":
        img = abi.make_yuv(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, with_alpha=True)
        rgb = abi.make_rgb(7680, 4320, 16, abi.AVIF_RGB_FORMAT_RGBA, alpha_premultiplied=True, allocate=False)
        bpp = 16.0
    elif name == "cfg5":
        img = abi.make_yuv(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        rgb = abi.make_rgb(1920, 1080, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=bil, allocate=False)
        bpp = 11.0
    else:
        raise SystemExit(name)
    synth.fill_yuv(img)
    return device.DeviceYUV(img), device.DeviceRGB(rgb), bpp, img.struct.width * img.struct.height


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg2"]
    for name in names:
        dimg, drgb, bpp, px = setup(name)
        variants = [("default", 1, 1), ("no-bands", 0, 1), ("ns1", 1 | (1 << 8), 1), ("ns2", 1 | (2 << 8), 1), ("ns2-run2", 1 | (2 << 8) | (2 << 12), 1),
                    ("ns1-run4", 1 | (1 << 8) | (4 << 12), 1)]
        if "--generic" in sys.argv:
            variants.append(("generic", 1, 0))
        best = {}
        for rep in range(3):
            for label, tune, tiled in variants:
                lib.avifhipSetTuning(tune)
                lib.avifhipSetTiledKernels(tiled)
                ms = lib.avifhipTimeYUVToRGB(dimg.struct, drgb.struct, 3, 30, None)
                best[label] = min(best.get(label, 1e9), ms)
                k = native.last_kernel()
                if rep == 2:
                    gbps = bpp * px / (best[label] * 1e-3) / 1e9
                    print(f"{name:6s} {label:9s} {best[label]*1000:8.1f} us  {px/1e6/(best[label]*1e-3):10.0f} MP/s  {gbps:7.0f} GB/s  {gbps/80:5.1f}% of 8 TB/s   [{k}]")
        lib.avifhipSetTuning(1)
        lib.avifhipSetTiledKernels(1)


if __name__ == "__main__":
    main()
