"""avifRGBImageApplyGainMap on bench.py's 4K job, the gain map's planes read by the apply kernel (round 5) against the conversion launch + RGBA copy of
rounds 2-4 (AVIFHIP_GAINMAP_PLANES=0), interleaved in one process: the apply kernel alone (HIP events), the whole call with light levels, the
asynchronous call without.  One JSON line per (gain-map format, arithmetic).

    python tests/tools/gm_call_bench.py [passes]"""
import ctypes as C
import json
import os
import sys
import time
from statistics import median

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from libavif_amd import abi, device, native, synth  # noqa: E402


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    lib = native.load()
    W, H = 3840, 2160
    base = abi.make_rgb(W, H, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
    synth.fill_rgb(base, 0x4242)
    dbase = device.DeviceRGB(base, upload=True)
    dout = device.DeviceRGB(abi.make_rgb(W, H, 10, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False, allocate=False))
    for fmt, fname in ((abi.AVIF_PIXEL_FORMAT_YUV444, "444"), (abi.AVIF_PIXEL_FORMAT_YUV400, "400")):
        gimg = abi.make_yuv(W, H, 8, fmt, abi.AVIF_RANGE_FULL, 6)
        synth.fill_yuv(gimg, 0x99)
        gm = abi.avifGainMap()
        for i in range(3):
            gm.gainMapMin[i].n, gm.gainMapMin[i].d = 0, 1
            gm.gainMapMax[i].n, gm.gainMapMax[i].d = 3, 1
            gm.gainMapGamma[i].n, gm.gainMapGamma[i].d = 1, 1
            gm.baseOffset[i].n, gm.baseOffset[i].d = 1, 64
            gm.alternateOffset[i].n, gm.alternateOffset[i].d = 1, 64
        gm.baseHdrHeadroom.n, gm.baseHdrHeadroom.d, gm.alternateHdrHeadroom.n, gm.alternateHdrHeadroom.d = 0, 1, 3, 1
        gm.useBaseColorSpace = 1
        dgimg = device.DeviceYUV(gimg)
        gm.image = C.pointer(dgimg.struct)
        clli, diag = abi.avifContentLightLevelInformationBox(), abi.avifDiagnostics()
        for arithmetic, aname in ((0, "auto"), (1, "float")):
            lib.avifhipSetArithmetic(arithmetic)
            rows = {"planes": {"kernel": [], "call": [], "async": []}, "copy": {"kernel": [], "call": [], "async": []}}
            names = {}
            for _ in range(passes):
                for mode in ("planes", "copy"):
                    if mode == "copy":
                        os.environ["AVIFHIP_GAINMAP_PLANES"] = "0"
                    else:
                        os.environ.pop("AVIFHIP_GAINMAP_PLANES", None)
                    ms = lib.avifhipTimeRGBImageApplyGainMap(dbase.struct, 1, 13, C.byref(gm), 3.0, 9, 16, dout.struct, 20, 200, None)
                    names[mode] = native.last_kernel()
                    rows[mode]["kernel"].append(ms)
                    t0 = time.perf_counter()
                    for _ in range(20):
                        native.check(lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, 1, 13, C.byref(gm), 3.0, 9, 16, dout.struct, C.byref(clli), C.byref(diag), None), "apply")
                    rows[mode]["call"].append((time.perf_counter() - t0) / 20 * 1e3)
                    native.check(lib.avifhipSynchronize(None), "sync")
                    t0 = time.perf_counter()
                    for _ in range(50):
                        native.check(lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, 1, 13, C.byref(gm), 3.0, 9, 16, dout.struct, None, C.byref(diag), None), "apply")
                    native.check(lib.avifhipSynchronize(None), "sync")
                    rows[mode]["async"].append((time.perf_counter() - t0) / 50 * 1e3)
            os.environ.pop("AVIFHIP_GAINMAP_PLANES", None)
            out = {"gain_map": f"8-bit 4:{fname[1]}:{fname[2]} {W}x{H}", "arithmetic": aname, "passes": passes}
            for mode in rows:
                out[mode] = {"kernel_name": names[mode], **{k: round(median(v) * 1e3, 2) for k, v in rows[mode].items()}}
            out["unit"] = "us, medians: the apply kernel alone / the whole call with light levels / per call of 50 back to back without"
            print(json.dumps(out), flush=True)
    # host-resident images, the reference's signatures: the RGB base image, and the YUV one (avifImageApplyGainMap: its RGB form lives in HBM only)
    lib.avifhipSetArithmetic(0)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    gimg = abi.make_yuv(W, H, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 6)
    synth.fill_yuv(gimg, 0x99)
    gm.image = C.pointer(gimg.struct)
    ybase = abi.make_yuv(W, H, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_FULL, 1)
    synth.fill_yuv(ybase, 0x77)
    ybase.struct.colorPrimaries, ybase.struct.transferCharacteristics = 1, 13
    out = abi.make_rgb(W, H, 10, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False, allocate=False)
    clli, diag = abi.avifContentLightLevelInformationBox(), abi.avifDiagnostics()
    t_rgb, t_yuv = [], []
    for _ in range(passes + 1):
        t0 = time.perf_counter()
        native.check(lib.avifhipRGBImageApplyGainMap(base.struct, 1, 13, C.byref(gm), 3.0, 9, 16, out.struct, C.byref(clli), C.byref(diag)), "host rgb")
        t_rgb.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        native.check(lib.avifhipImageApplyGainMap(ybase.struct, C.byref(gm), 3.0, 9, 16, out.struct, C.byref(clli), C.byref(diag)), "host yuv")
        t_yuv.append((time.perf_counter() - t0) * 1e3)
    libc.free(C.cast(out.struct.pixels, C.c_void_p))
    out.struct.pixels = None
    print(json.dumps({"host_resident_ms": {"avifhipRGBImageApplyGainMap (RGBA8 base, 4:4:4 map, RGBA10 out: 4 + 3 up, 8 down B/pixel)": round(median(t_rgb[1:]), 3),
                                           "avifhipImageApplyGainMap (8-bit 4:2:0 base: 1.5 + 3 up, 8 down B/pixel)": round(median(t_yuv[1:]), 3)}}), flush=True)
    lib.avifhipSetArithmetic(1)


if __name__ == "__main__":
    main()
